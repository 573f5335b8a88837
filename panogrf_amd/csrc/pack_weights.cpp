// Host-side packing of one pass's per-ray network weights (reference state_dict layout) into
// the MFMA-operand-ordered blob described in pgrf_layout.h.  Pure host code (no HIP calls).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

#include "pgrf_internal.h"
#include "pgrf_layout.h"

namespace pgrf {

namespace {

// in_feat(step, half) -> column of W (or -1); out_row(mt, rho) -> row of W (or -1)
using InMap = std::function<int(int, int)>;
using OutMap = std::function<int(int, int)>;

void pack_layer(float* blob, int id, const float* W, int in_dim, const float* bias, const InMap& in_feat,
                const OutMap& out_row, int col_base = 0) {
  const int ks = kKS[id], mt = kMT[id];
  const int ks4 = (ks + 3) / 4;
  float* A = blob + layer_a_off(id);
  for (int m = 0; m < mt; ++m)
    for (int s = 0; s < ks4 * 4; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        float v = 0.f;
        if (s < ks && W) {
          const int row = out_row(m, lane & 31);
          const int col = in_feat(s, lane >> 5);
          if (row >= 0 && col >= 0) v = W[(size_t)row * in_dim + col_base + col];
        }
        A[((size_t)(m * ks4 + s / 4) * 64 + lane) * 4 + (s & 3)] = v;
      }
  if (kHasBias[id]) {
    float* B = blob + layer_b_off(id);
    for (int m = 0; m < mt; ++m)
      for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
          const int row = out_row(m, rho(r, h));
          B[m * 32 + h * 16 + r] = (row >= 0 && bias) ? bias[row] : 0.f;
        }
  }
}

// "small" layer block for the 4x4x1 MFMA path (pgrf_layout.h): rows [0, nout) of W, features in_feat(r, half)
void pack_small(float* A, float* B, const float* W, int in_dim, const float* bias, int ks, int ng, int nout, int row0,
                const InMap& in_feat) {
  const int ks4 = (ks + 3) / 4;
  for (int g = 0; g < ng; ++g) {
    for (int r = 0; r < ks4 * 4; ++r)
      for (int lane = 0; lane < 64; ++lane) {
        const int o = 4 * g + (lane & 3);
        const int col = r < ks ? in_feat(r, lane >> 5) : -1;
        A[((size_t)(g * ks4 + r / 4) * 64 + lane) * 4 + (r & 3)] = (o < nout && col >= 0) ? W[(size_t)(row0 + o) * in_dim + col] : 0.f;
      }
    for (int i = 0; i < 4; ++i) {
      B[(g * 2 + 0) * 4 + i] = (4 * g + i < nout && bias) ? bias[row0 + 4 * g + i] : 0.f;
      B[(g * 2 + 1) * 4 + i] = 0.f;
    }
  }
}

// "w16" block for the 16x16x4 MFMA path (pgrf_layout.h): in_feat / out_row as for the layer's 32-row tile (tile `m`)
void pack_w16(float* A, float* B, const float* W, int in_dim, const float* bias, int ks, int m, const InMap& in_feat, const OutMap& out_row) {
  const int k4 = w16_k4(ks), c4 = (k4 + 3) / 4;
  for (int j = 0; j < c4 * 4; ++j)
    for (int lane = 0; lane < 64; ++lane) {
      const int kk = lane >> 4, s = 2 * j + (kk & 1);
      float v = 0.f;
      if (j < k4 && s < ks && W) {
        const int row = out_row(m, w16_row(lane & 15));
        const int col = in_feat(s, kk >> 1);
        if (row >= 0 && col >= 0) v = W[(size_t)row * in_dim + col];
      }
      A[((size_t)(j / 4) * 64 + lane) * 4 + (j & 3)] = v;
    }
  if (B)
    for (int q = 0; q < 4; ++q)
      for (int r = 0; r < 4; ++r) {
        const int row = out_row(m, w16_row(4 * q + r));
        B[q * 4 + r] = (row >= 0 && bias) ? bias[row] : 0.f;
      }
}

// head block [half][R]: value = w[col(r, half)]
void pack_head(float* dst, const float* w, int R, const std::function<int(int, int)>& col) {
  for (int h = 0; h < 2; ++h)
    for (int r = 0; r < R; ++r) {
      const int c = col(r, h);
      dst[h * R + r] = (c >= 0 && w) ? w[c] : 0.f;
    }
}

int ident_in(int s, int h) { return rho(s, h); }                 // k-step s reads register s of a 32-feature tile
int ident_out(int, int r) { return r; }                          // rows 0..31 -> outputs 0..31

}  // namespace

// ---- prob_embed.2 folded into its two consumers.  Its output ("neuray feature", aggregate_net.py:61) carries no
// activation and only enters neuray_fc.0 and columns 175..206 of base_fc.0 (ibrnet.py:338,344), both Linear:
//   W_c (W_pe2 r + b_pe2) + b_c = (W_c W_pe2) r + (W_c b_pe2 + b_c),   r = relu(prob_embed.0(.))
// so the per-point 32x32 layer disappears (16 of the 441 MFMAs per view and wave); products accumulated in double (every product
// of two floats is exact in double and the sum runs in index order: pack_device.hip reproduces it bit for bit).
static void fold_host(const float* const* raw, std::vector<float>& base_fused, std::vector<float>& base_bias, std::vector<float>& nf_fused,
                      std::vector<float>& nf_bias) {
  base_fused.assign(raw[R_BASE_0_W], raw[R_BASE_0_W] + 64 * 207);
  base_bias.assign(raw[R_BASE_0_B], raw[R_BASE_0_B] + 64);
  nf_fused.assign(8 * 32, 0.f);
  nf_bias.assign(8, 0.f);
  const float* Wp = raw[R_PE_2_W];   // [32 out][32 in]
  const float* bp = raw[R_PE_2_B];
  for (int o = 0; o < 64; ++o) {
    const float* Wc = raw[R_BASE_0_W] + (size_t)o * 207 + 175;
    for (int j = 0; j < 32; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 32; ++k) acc += (double)Wc[k] * (double)Wp[k * 32 + j];
      base_fused[(size_t)o * 207 + 175 + j] = (float)acc;
    }
    double accb = raw[R_BASE_0_B][o];
    for (int k = 0; k < 32; ++k) accb += (double)Wc[k] * (double)bp[k];
    base_bias[o] = (float)accb;
  }
  for (int o = 0; o < 8; ++o) {
    const float* Wc = raw[R_NF_0_W] + (size_t)o * 32;
    for (int j = 0; j < 32; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 32; ++k) acc += (double)Wc[k] * (double)Wp[k * 32 + j];
      nf_fused[o * 32 + j] = (float)acc;
    }
    double accb = raw[R_NF_0_B][o];
    for (int k = 0; k < 32; ++k) accb += (double)Wc[k] * (double)bp[k];
    nf_bias[o] = (float)accb;
  }
}

// the blob as a function of the EXTENDED table (raw tensors + the four folded ones, pgrf_layout.h X_*): copies, zero padding, the
// positional table and the split-f16 conversion only -- no arithmetic between weights, which is what lets build_pack_maps() below
// recover the source of every blob word by probing this function
static void pack_permute(const float* const* raw, int use_vis, int max_dn, float* blob) {
  std::memset(blob, 0, sizeof(float) * kBlobFloats);
  const float* base_fused = raw[X_BASE_FUSED];
  const float* base_bias = raw[X_BASE_BIAS];
  const float* nf_fused = raw[X_NF_FUSED];
  const float* nf_bias = raw[X_NF_BIAS];
  // ---- dist decoder -----------------------------------------------------------------
  const int dec0[4] = {R_DD_MEAN_0_W, R_DD_VAR_0_W, R_DD_AW_0_W, R_DD_VIS_0_W};
  {  // fused layer 0: tile m = MLP m
    const int ks = kKS[L_DD1], ks4 = (ks + 3) / 4;
    float* A = blob + layer_a_off(L_DD1);
    float* B = blob + layer_b_off(L_DD1);
    for (int m = 0; m < 4; ++m) {
      const float* W = (m == 3 && !use_vis) ? nullptr : raw[dec0[m]];
      const float* b = (m == 3 && !use_vis) ? nullptr : raw[dec0[m] + 1];
      for (int s = 0; s < ks; ++s)
        for (int lane = 0; lane < 64; ++lane)
          A[((size_t)(m * ks4 + s / 4) * 64 + lane) * 4 + (s & 3)] = W ? W[(lane & 31) * 32 + rho(s, lane >> 5)] : 0.f;
      for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) B[m * 32 + h * 16 + r] = b ? b[rho(r, h)] : 0.f;
    }
  }
  for (int m = 0; m < 4; ++m) {
    const bool off = (m == 3 && !use_vis);
    pack_layer(blob, L_DD2_0 + m, off ? nullptr : raw[dec0[m] + 2], 32, off ? nullptr : raw[dec0[m] + 3], ident_in, ident_out);
  }
  {  // heads: mean0, mean1, var0, var1, aw, vis
    const int head_src[6][2] = {{R_DD_MEAN_4_W, 0}, {R_DD_MEAN_4_W, 1}, {R_DD_VAR_4_W, 0}, {R_DD_VAR_4_W, 1}, {R_DD_AW_4_W, 0}, {R_DD_VIS_4_W, 0}};
    for (int j = 0; j < 6; ++j) {
      const bool off = (j == 5 && !use_vis);
      const float* W = off ? nullptr : raw[head_src[j][0]];
      const float* b = off ? nullptr : raw[head_src[j][0] + 1];
      const int row = head_src[j][1];
      pack_head(blob + OFF_DDH_W + j * 32, W ? W + row * 32 : nullptr, 16, [](int r, int h) { return rho(r, h); });
      blob[OFF_DDH_B + j] = b ? b[row] : 0.f;
    }
  }
  // ---- prob_embed -------------------------------------------------------------------
  pack_layer(blob, L_PE1, raw[R_PE_0_W], 34, raw[R_PE_0_B],
             [](int s, int h) { return s < 16 ? rho(s, h) : 32 + h; }, ident_out);
  pack_layer(blob, L_PE2, raw[R_PE_2_W], 32, raw[R_PE_2_B], ident_in, ident_out);
  // ---- neuray_fc --------------------------------------------------------------------
  pack_layer(blob, L_NF1, nf_fused, 32, nf_bias, ident_in, [](int, int r) { return r < 8 ? r : -1; });   // on relu(prob_embed.0)
  pack_head(blob + OFF_NFH_W, raw[R_NF_2_W], 4, [](int r, int h) { return rho(r, h); });
  blob[OFF_NFH_B] = raw[R_NF_2_B][0];
  // ---- ray_dir_fc -------------------------------------------------------------------
  pack_layer(blob, L_RD1, raw[R_RD_0_W], 4, raw[R_RD_0_B], [](int s, int h) { return 2 * s + h; },
             [](int, int r) { return r < 16 ? r : -1; });
  // out feature order of the reference: [rgb(3), img_feats(32)]; tile0 = img part, tile1 rows 0..2 = rgb part
  pack_layer(blob, L_RD2, raw[R_RD_2_W], 16, raw[R_RD_2_B], ident_in,
             [](int m, int r) { return m == 0 ? 3 + r : (r < 3 ? r : -1); });
  // ---- base_fc ----------------------------------------------------------------------
  // columns of base_fc.0: [mean0(35) var0(35) mean1(35) var1(35) | rgb_feat(35) neuray(32)]
  // k-steps of the view-shared block: the 32 image-feature columns of the four statistics (4 x 16 steps), then the
  // 3 colour columns of each, packed two per step with no empty slot (6 steps, 70 in all instead of 4 x 19 = 76):
  //   64: (mean0.r | mean0.g)  65: (var0.r | var0.g)  66: (mean1.r | mean1.g)  67: (var1.r | var1.g)
  //   68: (mean0.b | mean1.b)  69: (var0.b  | var1.b)         (lane half 0 | lane half 1)
  pack_layer(blob, L_BG, base_fused, 207, base_bias,
             [](int s, int h) {
               if (s < 64) return 35 * (s / 16) + 3 + rho(s % 16, h);
               const int t = s - 64;
               if (t < 4) return 35 * t + h;
               return 35 * ((t - 4) + 2 * h) + 2;
             },
             [](int m, int r) { return 32 * m + r; });
  // per-view block: image features (16 steps), colour (r | g), (b | -), prob_embed features (16 steps): 34 steps
  pack_layer(blob, L_BV, base_fused, 207, nullptr,
             [](int s, int h) {
               if (s < 16) return 140 + 3 + rho(s, h);
               if (s == 16) return 140 + h;
               if (s == 17) return h == 0 ? 140 + 2 : -1;
               return 175 + rho(s - 18, h);
             },
             [](int m, int r) { return 32 * m + r; });
  pack_layer(blob, L_B2, raw[R_BASE_2_W], 64, raw[R_BASE_2_B],
             [](int s, int h) { return s < 16 ? rho(s, h) : 32 + rho(s - 16, h); }, ident_out);
  // ---- vis_fc / vis_fc2 -------------------------------------------------------------
  pack_layer(blob, L_VF1, raw[R_VF_0_W], 32, raw[R_VF_0_B], ident_in, ident_out);
  pack_layer(blob, L_VF2, raw[R_VF_2_W], 32, raw[R_VF_2_B], ident_in, ident_out);   // rows 0..31 = x_res
  pack_head(blob + OFF_VFH_W, raw[R_VF_2_W] + 32 * 32, 16, [](int r, int h) { return rho(r, h); });
  blob[OFF_VFH_B] = raw[R_VF_2_B][32];
  pack_layer(blob, L_V21, raw[R_V2_0_W], 32, raw[R_V2_0_B], ident_in, ident_out);
  pack_head(blob + OFF_V2H_W, raw[R_V2_2_W], 16, [](int r, int h) { return rho(r, h); });
  blob[OFF_V2H_B] = raw[R_V2_2_B][0];
  // ---- geometry_fc ------------------------------------------------------------------
  pack_layer(blob, L_G1, raw[R_GEO_0_W], 65, raw[R_GEO_0_B],
             [](int s, int h) {
               if (s < 16) return rho(s, h);
               if (s < 32) return 32 + rho(s - 16, h);
               return h == 0 ? 64 : -1;
             },
             [](int m, int r) { return 32 * m + r; });
  pack_layer(blob, L_G2, raw[R_GEO_2_W], 64, raw[R_GEO_2_B],
             [](int s, int h) { return s < 16 ? rho(s, h) : 32 + rho(s - 16, h); },
             [](int, int r) { return r < 16 ? r : -1; });
  // ---- ray attention ----------------------------------------------------------------
  {  // q|k in tile 0, v in tile 1 — three source matrices, so fill by hand
    const int ks = kKS[L_QKV], ks4 = (ks + 3) / 4;
    float* A = blob + layer_a_off(L_QKV);
    for (int m = 0; m < 2; ++m)
      for (int s = 0; s < ks; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int r = lane & 31, col = rho(s, lane >> 5);
          float v = 0.f;
          if (m == 0) v = (r < 16) ? raw[R_ATT_Q_W][attn_feat(r) * 16 + col] : raw[R_ATT_K_W][attn_feat(r - 16) * 16 + col];
          else if (r < 16) v = raw[R_ATT_V_W][attn_feat(r) * 16 + col];
          A[((size_t)(m * ks4 + s / 4) * 64 + lane) * 4 + (s & 3)] = v;
        }
  }
  pack_layer(blob, L_FC, raw[R_ATT_FC_W], 16, nullptr,
             [](int s, int h) { return 4 * (s >> 1) + 2 * (s & 1) + h; }, [](int, int r) { return r < 16 ? r : -1; });
  pack_head(blob + OFF_LN_G, raw[R_ATT_LN_W], 8, [](int r, int h) { return rho(r, h); });
  pack_head(blob + OFF_LN_B, raw[R_ATT_LN_B], 8, [](int r, int h) { return rho(r, h); });
  // ---- out_geometry_fc --------------------------------------------------------------
  pack_layer(blob, L_OG1, raw[R_OG_0_W], 16, raw[R_OG_0_B], ident_in, [](int, int r) { return r < 16 ? r : -1; });
  pack_head(blob + OFF_OGH_W, raw[R_OG_2_W], 8, [](int r, int h) { return rho(r, h); });
  blob[OFF_OGH_B] = raw[R_OG_2_B][0];
  // ---- rgb_fc -----------------------------------------------------------------------
  // columns of rgb_fc.0: [x(32), vis(1), ray_diff(4)]
  pack_layer(blob, L_RGB1, raw[R_RGB_0_W], 37, raw[R_RGB_0_B],
             [](int s, int h) {
               if (s < 16) return rho(s, h);
               const int c = 32 + 2 * (s - 16) + h;
               return c < 37 ? c : -1;
             },
             [](int, int r) { return r < 16 ? r : -1; });
  pack_layer(blob, L_RGB2, raw[R_RGB_2_W], 16, raw[R_RGB_2_B], ident_in, [](int, int r) { return r < 8 ? r : -1; });
  pack_head(blob + OFF_RGBH_W, raw[R_RGB_4_W], 4, [](int r, int h) { return rho(r, h); });
  blob[OFF_RGBH_B] = raw[R_RGB_4_B][0];
  // ---- positional encoding (ibrnet.py:305-313: float64 table cast to float32) -------
  for (int p = 0; p < max_dn; ++p)
    for (int h = 0; h < 2; ++h)
      for (int r = 0; r < 8; ++r) {
        const int j = rho(r, h);
        const double ang = (double)p / std::pow(10000.0, 2.0 * (j / 2) / 16.0);
        blob[OFF_POS + p * 16 + h * 8 + r] = (float)((j & 1) ? std::cos(ang) : std::sin(ang));
      }
  // ---- per-pixel projection block (pgrf_layout.h: kPrjCh rows)
  {
    const int src_w[5] = {R_DD_MEAN_0_W, R_DD_VAR_0_W, R_DD_AW_0_W, R_PE_0_W, R_DD_VIS_0_W};
    const int in_dim[5] = {32, 32, 32, 34, 32};
    for (int g = 0; g < 5; ++g) {
      const bool off = (g == 4 && !use_vis);
      const float* Wg = off ? nullptr : raw[src_w[g]];
      const float* bg = off ? nullptr : raw[src_w[g] + 1];
      for (int r = 0; r < 32; ++r) {
        for (int k = 0; k < 32; ++k) blob[OFF_PRJ_W + (g * 32 + r) * 32 + k] = Wg ? Wg[r * in_dim[g] + k] : 0.f;
        blob[OFF_PRJ_B + g * 32 + r] = bg ? bg[r] : 0.f;
      }
    }
  }
  // ---- small layers for the 4x4x1 MFMA path (pgrf_layout.h: OFF_S_*)
  pack_small(blob + OFF_S_NF1_A, blob + OFF_S_NF1_B, nf_fused, 32, nf_bias, 16, 2, 8, 0, ident_in);
  for (int o = 0; o < 8; ++o) blob[OFF_S_NF1_H + o] = raw[R_NF_2_W][o];
  pack_small(blob + OFF_S_RDC_A, blob + OFF_S_RDC_B, raw[R_RD_2_W], 16, raw[R_RD_2_B], 8, 1, 3, 0, ident_in);
  pack_small(blob + OFF_S_RGB2_A, blob + OFF_S_RGB2_B, raw[R_RGB_2_W], 16, raw[R_RGB_2_B], 8, 2, 8, 0, ident_in);
  for (int o = 0; o < 8; ++o) blob[OFF_S_RGB2_H + o] = raw[R_RGB_4_W][o];
  // ---- 16-row layers for the 16x16x4 MFMA path (pgrf_layout.h: OFF_W16_*): the same maps as their 32-row tiles above
  {
    const OutMap rows16 = [](int, int r) { return r < 16 ? r : -1; };
    pack_w16(blob + OFF_W16_RGB1, blob + OFF_W16B_RGB1, raw[R_RGB_0_W], 37, raw[R_RGB_0_B], 19, 0,
             [](int s, int h) {
               if (s < 16) return rho(s, h);
               const int c = 32 + 2 * (s - 16) + h;
               return c < 37 ? c : -1;
             }, rows16);
    pack_w16(blob + OFF_W16_G2, blob + OFF_W16B_G2, raw[R_GEO_2_W], 64, raw[R_GEO_2_B], 32, 0,
             [](int s, int h) { return s < 16 ? rho(s, h) : 32 + rho(s - 16, h); }, rows16);
    pack_w16(blob + OFF_W16_V, nullptr, raw[R_ATT_V_W], 16, nullptr, 8, 0, ident_in,
             [](int, int r) { return r < 16 ? attn_feat(r) : -1; });
    pack_w16(blob + OFF_W16_FC, nullptr, raw[R_ATT_FC_W], 16, nullptr, 8, 0,
             [](int s, int h) { return 4 * (s >> 1) + 2 * (s & 1) + h; }, rows16);
    pack_w16(blob + OFF_W16_OG1, blob + OFF_W16B_OG1, raw[R_OG_0_W], 16, raw[R_OG_0_B], 8, 0, ident_in, rows16);
  }
  // ---- split-f16 copies of all A blocks (pgrf_layout.h: layer_h_off) ----------------------------
  for (int id = 0; id < L_COUNT; ++id) {
    const int ks = kKS[id], mt = kMT[id], ks4 = (ks + 3) / 4, G = h_groups(ks);
    const float* A = blob + layer_a_off(id);
    _Float16* H = reinterpret_cast<_Float16*>(blob + layer_h_off(id));
    for (int m = 0; m < mt; ++m)
      for (int g = 0; g < G; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int s = 8 * g + e;
            const float x = (s < ks) ? A[((size_t)(m * ks4 + s / 4) * 64 + lane) * 4 + (s & 3)] : 0.f;
            const _Float16 h1 = (_Float16)x;
            const _Float16 h2 = (_Float16)(x - (float)h1);
            H[((((size_t)m * G + g) * 2 + 0) * 64 + lane) * 8 + e] = h1;
            H[((((size_t)m * G + g) * 2 + 1) * 64 + lane) * 8 + e] = h2;
          }
  }
  // ---- LDS images (pgrf_layout.h: img_layer_off) ----------------------------------------------------
  for (int mode = 0; mode < kImgModes; ++mode) {
    float* img = blob + img_blob_off(mode);
    std::memcpy(img, blob + OFF_DDH_W, sizeof(float) * kHeadFloats);
    for (int id = 0; id < L_COUNT; ++id) {
      if (img_chunk(id) < 0) continue;
      float* dst = img + img_layer_off(id, mode);
      if (mode == 1 && id != L_PE1)
        std::memcpy(dst, blob + layer_h_off(id), sizeof(float) * 512);     // (m=0, g=0): [piece][lane][8 halves]
      else if (mode == 2 && img_full(id))
        std::memcpy(dst, blob + layer_a_off(id), sizeof(float) * a_size(kKS[id], kMT[id]));
      else if (img_a_floats(id, mode) > 0)
        std::memcpy(dst, blob + layer_a_off(id) + (size_t)img_chunk(id) * 256, sizeof(float) * 256);
      if (img_has_bias(id))
        std::memcpy(dst + img_a_floats(id, mode), blob + layer_b_off(id), sizeof(float) * b_size(kMT[id]));
    }
  }
}

int pack_weights_host(const float* const* raw, int n_raw, int use_vis, int max_dn, float* blob) {
  if (!raw || !blob) return set_error(PGRF_ERR_ARG, "pgrf_pack_weights: null argument");
  if (n_raw != R_COUNT) return set_error(PGRF_ERR_ARG, "pgrf_pack_weights: expected %d raw tensors, got %d", R_COUNT, n_raw);
  if (max_dn < 1 || max_dn > kMaxDn) return set_error(PGRF_ERR_ARG, "pgrf_pack_weights: max_dn %d out of range [1,%d]", max_dn, kMaxDn);
  for (int i = 0; i < R_COUNT; ++i) {
    const bool is_vis = (i >= R_DD_VIS_0_W && i <= R_DD_VIS_4_B);
    if (!raw[i] && !(is_vis && !use_vis)) return set_error(PGRF_ERR_ARG, "pgrf_pack_weights: raw tensor %d is null", i);
  }
  std::vector<float> base_fused, base_bias, nf_fused, nf_bias;
  fold_host(raw, base_fused, base_bias, nf_fused, nf_bias);
  const float* ext[X_COUNT];
  for (int i = 0; i < R_COUNT; ++i) ext[i] = raw[i];
  ext[X_BASE_FUSED] = base_fused.data(); ext[X_BASE_BIAS] = base_bias.data(); ext[X_NF_FUSED] = nf_fused.data(); ext[X_NF_BIAS] = nf_bias.data();
  pack_permute(ext, use_vis, max_dn, blob);
  return PGRF_OK;
}

// ---- source map of the blob, for the device-side packer (pack_device.hip) ------------------------------------------------
// pack_permute() is probed with coded inputs instead of restating its layout rules a second time:
//   zeros                                  -> the constant words (positional table)
//   element (t, e) = its id t * 2^14 + e + 1 -> the f32 words (ids < 2^24 are exact floats)
//   every element = 1 + 2^-12              -> which 16-bit halves are a first piece (0x3C00), a second piece (0x0C00) or padding
//   id low / high 11 bits as x and as 4096 + x / 2048   -> the source of first pieces (f16(x) = x) and of second pieces
//                                                          (f16(4096 + x/2048) = 4096, remainder x / 2048 exact in f16)
int build_pack_maps(int use_vis, int max_dn, std::vector<uint32_t>& d0, std::vector<uint32_t>& d1, std::vector<float>& c0) {
  if (max_dn < 1 || max_dn > kMaxDn) return set_error(PGRF_ERR_ARG, "pgrf_pack_weights_device: max_dn %d out of range [1,%d]", max_dn, kMaxDn);
  std::vector<std::vector<float>> buf(X_COUNT);
  const float* ext[X_COUNT];
  for (int t = 0; t < X_COUNT; ++t) { buf[t].assign(kExtFloats[t], 0.f); ext[t] = buf[t].data(); }
  auto run = [&](const std::function<float(uint32_t)>& f, std::vector<float>& out) {
    for (int t = 0; t < X_COUNT; ++t)
      for (int e = 0; e < kExtFloats[t]; ++e) buf[t][e] = f(((uint32_t)t << 14) | (uint32_t)e);
    out.assign(kBlobFloats, 0.f);
    pack_permute(ext, use_vis, max_dn, out.data());
  };
  std::vector<float> ids, piece, lo1, hi1, lo2, hi2;
  run([](uint32_t) { return 0.f; }, c0);
  run([](uint32_t id) { return (float)(id + 1); }, ids);
  run([](uint32_t) { return 1.f + 1.f / 4096.f; }, piece);
  run([](uint32_t id) { return (float)(id & 2047u); }, lo1);
  run([](uint32_t id) { return (float)(id >> 11); }, hi1);
  run([](uint32_t id) { return 4096.f + (float)(id & 2047u) / 2048.f; }, lo2);
  run([](uint32_t id) { return 4096.f + (float)(id >> 11) / 2048.f; }, hi2);
  d0.assign(kBlobFloats, 0u);
  d1.assign(kBlobFloats, 0u);
  auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  auto half_val = [](uint32_t h) { _Float16 x; uint16_t u = (uint16_t)h; std::memcpy(&x, &u, 2); return (float)x; };
  const uint32_t probe_bits = bits(1.f + 1.f / 4096.f);
  auto check_id = [&](uint32_t id) { const uint32_t t = id >> 14, e = id & 16383u; return t < (uint32_t)X_COUNT && (int)e < kExtFloats[t]; };
  for (int i = 0; i < kBlobFloats; ++i) {
    if (c0[i] != 0.f) continue;                                    // constant word
    const uint32_t pb = bits(piece[i]);
    if (pb == 0u) continue;                                        // padding (constant 0)
    if (pb == probe_bits) {                                        // f32 copy
      const uint32_t id = (uint32_t)ids[i] - 1u;
      if (ids[i] < 1.f || !check_id(id)) return set_error(PGRF_ERR_UNSUPPORTED, "pack map: f32 word %d has no source", i);
      d0[i] = 1u | (id << 3);
      continue;
    }
    uint32_t d[2] = {0u, 0u};
    for (int h = 0; h < 2; ++h) {                                  // a pair of split-f16 halves
      const uint32_t ph = (pb >> (16 * h)) & 0xffffu;
      if (ph == 0u) { d[h] = 4u; continue; }
      if (ph != 0x3C00u && ph != 0x0C00u) return set_error(PGRF_ERR_UNSUPPORTED, "pack map: word %d is neither f32 nor split-f16", i);
      const bool second = ph == 0x0C00u;
      const float lo = half_val((bits(second ? lo2[i] : lo1[i]) >> (16 * h)) & 0xffffu) * (second ? 2048.f : 1.f);
      const float hi = half_val((bits(second ? hi2[i] : hi1[i]) >> (16 * h)) & 0xffffu) * (second ? 2048.f : 1.f);
      const uint32_t id = ((uint32_t)hi << 11) | (uint32_t)lo;
      if (!check_id(id)) return set_error(PGRF_ERR_UNSUPPORTED, "pack map: half %d of word %d has no source", h, i);
      d[h] = 1u | (second ? 2u : 0u) | 4u | (id << 3);
    }
    d0[i] = d[0] | 4u;
    d1[i] = d[1];
  }
  return PGRF_OK;
}

// host evaluation of the source map against the packer itself on pseudo-random weights (what pack_device.hip's kernel does, word
// by word): 0 when every word of the blob is reproduced bit for bit
int pack_map_selftest(int use_vis, int max_dn) {
  std::vector<uint32_t> d0, d1;
  std::vector<float> c0;
  if (int rc = build_pack_maps(use_vis, max_dn, d0, d1, c0)) return rc;
  std::vector<std::vector<float>> buf(R_COUNT);
  const float* raw[R_COUNT];
  uint32_t seed = 12345u;
  for (int t = 0; t < R_COUNT; ++t) {
    buf[t].resize(kExtFloats[t]);
    for (auto& v : buf[t]) { seed = seed * 1664525u + 1013904223u; v = ((float)(seed >> 8) / 8388608.f - 1.f) * 0.37f; }
    const bool is_vis = (t >= R_DD_VIS_0_W && t <= R_DD_VIS_4_B);
    raw[t] = (is_vis && !use_vis) ? nullptr : buf[t].data();
  }
  std::vector<float> want(kBlobFloats);
  if (int rc = pack_weights_host(raw, R_COUNT, use_vis, max_dn, want.data())) return rc;
  std::vector<float> base_fused, base_bias, nf_fused, nf_bias;
  fold_host(raw, base_fused, base_bias, nf_fused, nf_bias);
  const float* ext[X_COUNT];
  for (int i = 0; i < R_COUNT; ++i) ext[i] = raw[i];
  ext[X_BASE_FUSED] = base_fused.data(); ext[X_BASE_BIAS] = base_bias.data(); ext[X_NF_FUSED] = nf_fused.data(); ext[X_NF_BIAS] = nf_bias.data();
  auto src = [&](uint32_t d) { const float* p = ext[d >> 17]; return p ? p[(d >> 3) & 16383u] : 0.f; };
  auto half = [&](uint32_t d) -> uint32_t {
    if (!(d & 1u)) return 0u;
    const float x = src(d);
    const _Float16 h1 = (_Float16)x;
    const _Float16 h = (d & 2u) ? (_Float16)(x - (float)h1) : h1;
    uint16_t u; std::memcpy(&u, &h, 2);
    return u;
  };
  for (int i = 0; i < kBlobFloats; ++i) {
    uint32_t got, exp;
    std::memcpy(&exp, &want[i], 4);
    if (!(d0[i] & 4u)) { const float v = (d0[i] & 1u) ? src(d0[i]) : c0[i]; std::memcpy(&got, &v, 4); }
    else got = half(d0[i]) | (half(d1[i]) << 16);
    if (got != exp) return set_error(PGRF_ERR_UNSUPPORTED, "pack map selftest: word %d is %08x, the packer wrote %08x", i, got, exp);
  }
  return PGRF_OK;
}

}  // namespace pgrf
