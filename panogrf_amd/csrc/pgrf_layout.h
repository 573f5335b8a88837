// Packed-weight layout of one network pass (coarse or fine) of the per-ray renderer.
// Shared by the host packer (pack_weights.cpp) and the HIP kernel (render_pass.h).
//
// Every Linear layer y = W x + b of the per-point MLPs is evaluated TRANSPOSED on the
// f32 matrix cores:  Y^T[32 out-rows x 32 points] += W-tile[32 x 2] * X^T[2 x 32 points]
// with v_mfma_f32_32x32x2_f32.  A wave owns 32 sample points; lane l holds point (l & 31)
// and, in accumulator register r, output row  rho(r, l>>5) = (r&3) + 8*(r>>2) + 4*(l>>5).
// Because the B operand of step s wants "feature k=0 in lanes 0-31, feature k=1 in lanes
// 32-63", accumulator register r of one layer IS the B operand of k-step r of the next
// layer: activations never leave registers.  The price is paid once, on the host: the
// A operands (weights) are stored pre-permuted in exactly the order the lanes consume them:
//
//   A block of a layer:  [mt][ks4][lane 0..63][4]   float   (one dwordx4 per lane = 4 k-steps)
//        value(mt, s, lane) = W[out_row(mt, lane & 31)][in_feat(s, lane >> 5)]   (0 if unmapped)
//   bias block:          [mt][half][16]             float   value = b[out_row(mt, rho(r, half))]
//
// The tiny heads (N <= 2 outputs) run on the VALU from "head" blocks  [out][half][R] .
#pragma once

namespace pgrf {

constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct LayerDesc {
  int ks;      // k-steps (each consumes 2 input features)
  int mt;      // 32-row output tiles
  int a_off;   // float offset of the A block inside the blob
  int b_off;   // float offset of the bias block (-1: none)
};

constexpr int a_size(int ks, int mt) { return mt * ((ks + 3) / 4) * 256; }
constexpr int b_size(int mt) { return mt * 32; }

// ---- layer ids ------------------------------------------------------------------------
enum LayerId {
  L_DD1 = 0,                       // dist decoder layer 0 of the 4 MLPs (mean,var,aw,vis), fused: 32 -> 4x32
  L_DD2_0, L_DD2_1, L_DD2_2, L_DD2_3,  // dist decoder layer 2 of each MLP: 32 -> 32
  L_PE1,                           // prob_embed.0  34 -> 32
  L_PE2,                           // prob_embed.2  32 -> 32
  L_NF1,                           // neuray_fc.0   32 -> 8
  L_RD1,                           // ray_dir_fc.0   4 -> 16
  L_RD2,                           // ray_dir_fc.2  16 -> 35   (tile0: img-feat part, tile1: rgb part)
  L_BG,                            // base_fc.0, view-shared 140 columns  -> 64
  L_BV,                            // base_fc.0, per-view 67 columns      -> 64 (accumulates onto L_BG)
  L_B2,                            // base_fc.2     64 -> 32
  L_VF1,                           // vis_fc.0      32 -> 32
  L_VF2,                           // vis_fc.2      32 -> 32 (+ row 32 as a head)
  L_V21,                           // vis_fc2.0     32 -> 32
  L_G1,                            // geometry_fc.0 65 -> 64
  L_G2,                            // geometry_fc.2 64 -> 16
  L_QKV,                           // w_qs|w_ks (tile0), w_vs (tile1)  16 -> 48
  L_FC,                            // ray_attention.fc 16 -> 16
  L_OG1,                           // out_geometry_fc.0 16 -> 16
  L_RGB1,                          // rgb_fc.0      37 -> 16
  L_RGB2,                          // rgb_fc.2      16 -> 8
  L_COUNT
};

constexpr int kKS[L_COUNT] = {16, 16, 16, 16, 16, 17, 16, 16, 2, 8, 70, 34, 32, 16, 16, 16, 33, 32, 8, 8, 8, 19, 8};
constexpr int kMT[L_COUNT] = {4, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 2, 1, 2, 1, 1, 1, 1};
constexpr bool kHasBias[L_COUNT] = {true, true, true, true, true, true, true, true, true, true, true, false, true,
                                    true, true, true, true, true, false, false, true, true, true};

constexpr int layer_a_off(int id) {
  int off = 0;
  for (int i = 0; i < id; ++i) off += a_size(kKS[i], kMT[i]) + (kHasBias[i] ? b_size(kMT[i]) : 0);
  return off;
}
constexpr int layer_b_off(int id) { return kHasBias[id] ? layer_a_off(id) + a_size(kKS[id], kMT[id]) : -1; }
constexpr int kLayersEnd = layer_a_off(L_COUNT);

// ---- VALU head / misc region (float offsets) -------------------------------------------
// dist-decoder heads: 6 outputs (mean0, mean1, var0, var1, aw, vis) x [half][16], then 6 biases (+2 pad)
constexpr int OFF_DDH_W = kLayersEnd;
constexpr int OFF_DDH_B = OFF_DDH_W + 6 * 32;
// neuray_fc.2 head: [half][4] + bias (+3 pad)
constexpr int OFF_NFH_W = OFF_DDH_B + 8;
constexpr int OFF_NFH_B = OFF_NFH_W + 8;
// vis_fc.2 row 32: [half][16] + bias
constexpr int OFF_VFH_W = OFF_NFH_B + 4;
constexpr int OFF_VFH_B = OFF_VFH_W + 32;
// vis_fc2.2: [half][16] + bias
constexpr int OFF_V2H_W = OFF_VFH_B + 4;
constexpr int OFF_V2H_B = OFF_V2H_W + 32;
// out_geometry_fc.2: [half][8] + bias
constexpr int OFF_OGH_W = OFF_V2H_B + 4;
constexpr int OFF_OGH_B = OFF_OGH_W + 16;
// rgb_fc.4: [half][4] + bias
constexpr int OFF_RGBH_W = OFF_OGH_B + 4;
constexpr int OFF_RGBH_B = OFF_RGBH_W + 8;
// LayerNorm gamma, beta: [half][8] each
constexpr int OFF_LN_G = OFF_RGBH_B + 4;
constexpr int OFF_LN_B = OFF_LN_G + 16;
// bias blocks of the 16-row layers on v_mfma_f32_16x16x4_f32 ("w16" blocks below): [lane >> 4][4], kept inside the head region so
// that they ride in the LDS image
constexpr int OFF_W16B_RGB1 = OFF_LN_B + 16;
constexpr int OFF_W16B_G2 = OFF_W16B_RGB1 + 16;
constexpr int OFF_W16B_OG1 = OFF_W16B_G2 + 16;
// positional encoding [PGRF_MAX_DN][half][8]
constexpr int kMaxDn = 128;
constexpr int OFF_POS = OFF_W16B_OG1 + 16;
// Per-pixel projection of the ray_feats maps (done once per pass and frame, see project_ray_feats_kernel):
// the first layers that consume the gathered ray feature — dist-decoder layer 0 of every MLP and the 32
// ray-feature columns of prob_embed.0 — are linear, and bilinear interpolation is linear with weights summing
// to 1, so W*interp(f)+b == interp(W*f+b).  Rows: [mean(32) | var(32) | aw(32) | prob_embed.0(32) | vis(32)].
constexpr int kPrjCh = 160;
constexpr int OFF_PRJ_W = OFF_POS + kMaxDn * 16;          // [kPrjCh][32] row-major
constexpr int OFF_PRJ_B = OFF_PRJ_W + kPrjCh * 32;        // [kPrjCh]
// ---- optional split-f16 copy of every A block (precision mode "f16x3") ------------------------------
// w = h1 + h2 (two IEEE halves, 22 significant bits); the kernel splits the activations the same way and
// issues three v_mfma_f32_32x32x16_f16 per 16-deep k-group: h1*h1' + h1*h2' + h2*h1' (fp32 accumulate).
//   H block of a layer: [mt][g][piece 0|1][lane][8 halves]   with g = group of 8 consecutive k-steps,
//   value(mt, g, lane, e) = A-block value of k-step 8g+e (0 beyond KS).  Sizes below are in floats.
constexpr int h_groups(int ks) { return (ks + 7) / 8; }
constexpr int h_size(int ks, int mt) { return mt * h_groups(ks) * 2 * 64 * 4; }
// ---- "small" layers (f32 mode): layers with <= 8 outputs waste 3/4 of a 32-row MFMA tile.  They are evaluated with
// v_mfma_f32_4x4x1_16b_f32 instead: 16 independent 4x4 blocks, block = 4 consecutive lanes = 4 points, the B operand
// is the lane's own activation register (k = 1 feature pair per instruction, one feature per lane half), the A operand
// holds 4 output rows, and D leaves output row i of the lane's own point in register i.  The two lane halves hold
// partial sums over their half of the features (combined with one v_permlane32_swap + add per output).
//   S block:  [g][r4][lane 0..63][4]  value(g, r, lane) = W[4g + (lane & 3)][in_feat(r, lane >> 5)]   (r = k-register)
//   S bias:   [g][half][4]            bias[4g + i] in half 0, zeros in half 1 (the halves are summed)
// followed by the layer's output head in natural row order where it has one.
constexpr int s_a_size(int ks, int ng) { return ng * ((ks + 3) / 4) * 256; }
constexpr int s_b_size(int ng) { return ng * 8; }
constexpr int OFF_S_BASE = OFF_PRJ_B + kPrjCh;
// neuray_fc.0 32 -> 8 (+ neuray_fc.2 head 8 -> 1)
constexpr int OFF_S_NF1_A = OFF_S_BASE;
constexpr int OFF_S_NF1_B = OFF_S_NF1_A + s_a_size(16, 2);
constexpr int OFF_S_NF1_H = OFF_S_NF1_B + s_b_size(2);        // [8] head weights
// ray_dir_fc.2 rows 0..2 (the rgb part of its 35 outputs) 16 -> 3
constexpr int OFF_S_RDC_A = OFF_S_NF1_H + 8;
constexpr int OFF_S_RDC_B = OFF_S_RDC_A + s_a_size(8, 1);
// rgb_fc.2 16 -> 8 (+ rgb_fc.4 head 8 -> 1)
constexpr int OFF_S_RGB2_A = OFF_S_RDC_B + s_b_size(1);
constexpr int OFF_S_RGB2_B = OFF_S_RGB2_A + s_a_size(8, 2);
constexpr int OFF_S_RGB2_H = OFF_S_RGB2_B + s_b_size(2);     // [8] head weights
constexpr int OFF_S_END = OFF_S_RGB2_H + 8;
// ---- "w16" layers (f32 mode): layers with 16 outputs waste half of a 32-row tile.  They run on v_mfma_f32_16x16x4_f32 instead
// (16 rows x 16 points x 4 features in 32 cycles: twice the useful work per cycle).  Two B-operand registers X = b[2j], Y = b[2j+1] of
// the 32-point layout (feature in_feat(s, 0) in lanes 0-31, in_feat(s, 1) in lanes 32-63) become, through ONE v_permlane16_swap,
// the B operands of k4-step j for points 0-15 (X') and 16-31 (Y'): lane group kk = lane >> 4 carries feature
//   f16(j, kk) = in_feat(2j + (kk & 1), kk >> 1).
// D (4 registers, rows 4 (lane >> 4) + r of point lane & 15) of the two point groups goes back to the 32-point layout by one
// v_permlane16_swap per register pair; MFMA row i is given the logical output row  w16_row(i)  (bits 2 and 3 exchanged) so that the
// 8 result registers are exactly registers 0..7 of the padded 32-row tile: nothing downstream changes.
//   w16 A block: [c4][lane 0..63][4]   value(j = 4 c4 + e, lane) = W[out_row(w16_row(lane & 15))][f16(j, lane >> 4)]
//   w16 bias   : [lane >> 4][4]        value(q, r) = b[out_row(w16_row(4 q + r))]      (OFF_W16B_*, head region)
constexpr int w16_row(int i) { return (i & 3) + 4 * ((i >> 3) & 1) + 8 * ((i >> 2) & 1); }
constexpr int w16_k4(int ks) { return (ks + 1) / 2; }
constexpr int w16_a_size(int ks) { return ((w16_k4(ks) + 3) / 4) * 256; }
constexpr int OFF_W16_RGB1 = OFF_S_END;                              // rgb_fc.0          37 -> 16
constexpr int OFF_W16_G2 = OFF_W16_RGB1 + w16_a_size(19);            // geometry_fc.2     64 -> 16
constexpr int OFF_W16_V = OFF_W16_G2 + w16_a_size(32);               // w_vs              16 -> 16
constexpr int OFF_W16_FC = OFF_W16_V + w16_a_size(8);                // ray_attention.fc  16 -> 16
constexpr int OFF_W16_OG1 = OFF_W16_FC + w16_a_size(8);              // out_geometry_fc.0 16 -> 16
constexpr int OFF_W16_END = OFF_W16_OG1 + w16_a_size(8);
constexpr int OFF_H_BASE = OFF_W16_END;
constexpr int layer_h_off(int id) {
  int off = OFF_H_BASE;
  for (int i = 0; i < id; ++i) off += h_size(kKS[i], kMT[i]);
  return off;
}
constexpr int kSplitEnd = layer_h_off(L_COUNT);
// ---- LDS image: what every workgroup copies into LDS once, so that the loads a layer cannot start without come from
// LDS (~100 cycles) instead of L2 (several hundred, exposed once per layer with two waves per SIMD):
//   [head/LayerNorm region OFF_DDH_W..OFF_POS, verbatim]
//   per layer: [first A chunk of tile 0 (f32: [lane][4] floats; split-f16: [piece][lane][8] halves)] [bias block]
// prob_embed.0 keeps only its k-step-16 chunk (the rest lives in the projection map); the fused dist-decoder layer 0
// is never evaluated per point and has no entry.  One image per precision mode (index 0: f32, 1: split-f16).
constexpr int kHeadFloats = OFF_POS - OFF_DDH_W;
constexpr int img_chunk(int id) { return id == L_DD1 ? -1 : (id == L_PE1 ? 4 : 0); }
constexpr bool img_has_bias(int id) { return kHasBias[id] && id != L_PE1 && id != L_DD1; }
// image mode 3 = f32, heads and biases only (+ prob_embed.0's chunk): every workgroup copies its image once per 128 samples,
// and for the 64-sample kernels with 2 and 3 views the 20 KB of first chunks cost more at workgroup start than they save
// per layer (measured, per instance: render_pass.h kImgMode).
// image modes: 0 = f32, first chunk per layer; 1 = split-f16, first k-group per layer; 2 = f32, the WHOLE A block of
// every layer except the three largest (base_fc.0 shared/per-view, geometry_fc.0) — used by the kernels for more than
// two views, which run one workgroup per CU (so have the LDS) and one wave per SIMD (so cannot hide an L2 round trip).
constexpr int kImgModes = 4;
constexpr bool img_full(int id) { return id != L_BG && id != L_BV && id != L_G1 && id != L_PE1 && id != L_DD1; }
constexpr int img_a_floats(int id, int mode) {
  if (mode == 3 && id != L_PE1) return 0;
  return mode == 1 ? 512 : ((mode == 2 && img_full(id)) ? a_size(kKS[id], kMT[id]) : 256);
}
constexpr int img_layer_size(int id, int mode) {
  return img_chunk(id) < 0 ? 0 : img_a_floats(id, mode) + (img_has_bias(id) ? b_size(kMT[id]) : 0);
}
constexpr int img_layer_off(int id, int mode) {
  int off = kHeadFloats;
  for (int i = 0; i < id; ++i) off += img_layer_size(i, mode);
  return off;
}
constexpr int img_floats(int mode) { return img_layer_off(L_COUNT, mode); }
constexpr int img_blob_off(int mode) {
  int off = kSplitEnd;
  for (int i = 0; i < mode; ++i) off += img_floats(i);
  return off;
}
constexpr int kBlobFloats = img_blob_off(kImgModes);
static_assert(kHeadFloats % 4 == 0 && img_floats(0) % 4 == 0 && img_floats(1) % 4 == 0 && img_floats(2) % 4 == 0 && img_floats(3) % 4 == 0 && kSplitEnd % 4 == 0,
              "LDS image is copied as float4");

// attention feature permutation: accumulator row rho (< 16) of the q / k / v projections holds
// head-feature  attn_feat(rho)  so that register 2*head+s carries dims (2s, 2s+1) in the two lane halves.
constexpr int attn_feat(int rho_) {
  int h = (rho_ >> 2) & 1;
  int r = (rho_ & 3) + 4 * (rho_ >> 3);
  return 4 * (r >> 1) + 2 * (r & 1) + h;
}

// ---- raw (reference-layout) weight table handed to pgrf_pack_weights --------------------
// Order of the `const float*` array; every Linear is [out,in] row-major + [out] bias, exactly as in
// the reference state_dict.  vis_decoder entries may be NULL (use_vis=false).
enum RawId {
  R_DD_MEAN_0_W = 0, R_DD_MEAN_0_B, R_DD_MEAN_2_W, R_DD_MEAN_2_B, R_DD_MEAN_4_W, R_DD_MEAN_4_B,
  R_DD_VAR_0_W, R_DD_VAR_0_B, R_DD_VAR_2_W, R_DD_VAR_2_B, R_DD_VAR_4_W, R_DD_VAR_4_B,
  R_DD_AW_0_W, R_DD_AW_0_B, R_DD_AW_2_W, R_DD_AW_2_B, R_DD_AW_4_W, R_DD_AW_4_B,
  R_DD_VIS_0_W, R_DD_VIS_0_B, R_DD_VIS_2_W, R_DD_VIS_2_B, R_DD_VIS_4_W, R_DD_VIS_4_B,
  R_PE_0_W, R_PE_0_B, R_PE_2_W, R_PE_2_B,
  R_RD_0_W, R_RD_0_B, R_RD_2_W, R_RD_2_B,
  R_BASE_0_W, R_BASE_0_B, R_BASE_2_W, R_BASE_2_B,
  R_VF_0_W, R_VF_0_B, R_VF_2_W, R_VF_2_B,
  R_V2_0_W, R_V2_0_B, R_V2_2_W, R_V2_2_B,
  R_GEO_0_W, R_GEO_0_B, R_GEO_2_W, R_GEO_2_B,
  R_ATT_Q_W, R_ATT_K_W, R_ATT_V_W, R_ATT_FC_W, R_ATT_LN_W, R_ATT_LN_B,
  R_OG_0_W, R_OG_0_B, R_OG_2_W, R_OG_2_B,
  R_RGB_0_W, R_RGB_0_B, R_RGB_2_W, R_RGB_2_B, R_RGB_4_W, R_RGB_4_B,
  R_NF_0_W, R_NF_0_B, R_NF_2_W, R_NF_2_B,
  R_COUNT
};
// the packer's extended table: the raw tensors, then prob_embed.2 folded into base_fc.0 (weight [64, 207], bias [64]) and into
// neuray_fc.0 (weight [8, 32], bias [8]) (pack_weights.cpp fold_host)
enum ExtId { X_BASE_FUSED = R_COUNT, X_BASE_BIAS, X_NF_FUSED, X_NF_BIAS, X_COUNT };
// floats per entry (nn.Linear [out, in] + [out])
constexpr int kExtFloats[X_COUNT] = {
  1024, 32, 1024, 32, 64, 2,   1024, 32, 1024, 32, 64, 2,   1024, 32, 1024, 32, 32, 1,   1024, 32, 1024, 32, 32, 1,     // dist decoder: mean, var, aw, vis
  32 * 34, 32, 1024, 32,       // prob_embed
  16 * 4, 16, 35 * 16, 35,     // ray_dir_fc
  64 * 207, 64, 32 * 64, 32,   // base_fc
  1024, 32, 33 * 32, 33,       // vis_fc
  1024, 32, 32, 1,             // vis_fc2
  64 * 65, 64, 16 * 64, 16,    // geometry_fc
  256, 256, 256, 256, 16, 16,  // ray attention q, k, v, fc, LayerNorm
  256, 16, 16, 1,              // out_geometry_fc
  16 * 37, 16, 8 * 16, 8, 8, 1,   // rgb_fc
  8 * 32, 8, 8, 1,             // neuray_fc
  64 * 207, 64, 8 * 32, 8      // folded
};

}  // namespace pgrf
