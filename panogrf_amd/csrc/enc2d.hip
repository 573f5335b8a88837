// Per-frame 2-D encoder primitives for gfx950: what ResUNetLight (network/ops.py:235-455)
// and DefaultVisEncoder (network/vis_encoder.py:6-32) need besides the 3x3 / 1x1 convolutions, which run on the
// implicit-GEMM MFMA kernel of conv3d.hip (pgrf_conv2d):
//   * the stem: 7x7 stride-2 convolution of the NCHW input image (3 channels) with ERP wrap padding -> channel-last map
//   * InstanceNorm2d(affine) in two launches: per-chunk partial sums, then normalise + activation (+ residual)
//   * bilinear resize (align_corners on/off) between channel-last maps, writing into a channel slice (U-Net concats)
// All maps are channel-last [N,H,W,Cs] with (cs, coff) slices like the 3-D volumes.  These layers run once per frame
// on a few MB: they are launch/latency bound, so the point is few, fused launches (MIOpen + torch elementwise needs
// ~250 launches per frame for the same work), not roofline.
#include "pgrf_experiments.h"
#include <hip/hip_runtime.h>

#include "pgrf_internal.h"

namespace pgrf {

#define PGRF_CHECK_LAUNCH(what)                                                                       \
  do {                                                                                                \
    const hipError_t e__ = hipGetLastError();                                                         \
    if (e__ != hipSuccess) return set_error(PGRF_ERR_LAUNCH, what " launch failed: %s", hipGetErrorString(e__)); \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------
// Direct KxK convolution of a narrow NCHW input (Cin <= 4): the stem conv1 = WrapPadding(3) + Conv2d(3, 16, 7, 2)
// (network/ops.py:262-268, models/common_blocks.py:258-292: zeros along H, circular along W).  One thread = PX
// consecutive output pixels of a row x COUT channels: the weights sit in LDS as [cin][ky][kx][COUT] and every
// broadcast float4 read of them feeds PX pixels.  Measured for 2 x 512 x 1024 -> 16 channels: PX = 1 58 us, PX = 2 62 us,
// PX = 4 82 us (fewer waves left to hide the strided input loads), so the launcher uses PX = 1.
template <int COUT, int PX>
__global__ void __launch_bounds__(256) conv2d_direct_kernel(const float* __restrict__ in, int N, int Cin, int H, int W,
                                                            const float* __restrict__ wgt /* [COUT][Cin][K][K] */, int K,
                                                            int stride, int wrap_w, float* __restrict__ out, int out_cs,
                                                            int out_coff, int Ho, int Wo, const float* __restrict__ bias,
                                                            int relu) {
  extern __shared__ float s_w[];          // [Cin*K*K][COUT]
  const int taps = Cin * K * K;
  for (int i = threadIdx.x; i < taps * COUT; i += blockDim.x) {
    const int co = i % COUT, t = i / COUT;
    s_w[i] = wgt[(size_t)co * taps + t];
  }
  __syncthreads();
  const int Wg = (Wo + PX - 1) / PX;                      // pixel groups per row
  const long long total = (long long)N * Ho * Wg;
  const int pad = K / 2;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
    const int xg = (int)(p % Wg);
    const int yo = (int)((p / Wg) % Ho);
    const int n = (int)(p / ((long long)Wg * Ho));
    const int xo0 = xg * PX;
    float acc[PX][COUT];
#pragma unroll
    for (int j = 0; j < PX; ++j)
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[j][c] = 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
      const float* ip = in + ((size_t)n * Cin + ci) * H * W;
      for (int ky = 0; ky < K; ++ky) {
        const int yi = yo * stride + ky - pad;
        if (yi < 0 || yi >= H) continue;                       // zeros along H
        const float* rowp = ip + (size_t)yi * W;
        for (int kx = 0; kx < K; ++kx) {
          float v[PX];
#pragma unroll
          for (int j = 0; j < PX; ++j) {
            int xi = (xo0 + j) * stride + kx - pad;
            bool ok = true;
            if (xi < 0 || xi >= W) {
              if (wrap_w) xi = xi < 0 ? xi + W : (xi >= W ? xi - W : xi);     // circular along W
              else ok = false;
              if (xi >= W) { xi = W - 1; ok = false; }          // ragged last group: pixel beyond the row
            }
            v[j] = ok ? rowp[xi] : 0.f;
          }
          const float4* wp = reinterpret_cast<const float4*>(s_w + ((ci * K + ky) * K + kx) * COUT);
#pragma unroll
          for (int q = 0; q < COUT / 4; ++q) {
            const float4 w4 = wp[q];
#pragma unroll
            for (int j = 0; j < PX; ++j) {
              acc[j][4 * q] = fmaf(v[j], w4.x, acc[j][4 * q]); acc[j][4 * q + 1] = fmaf(v[j], w4.y, acc[j][4 * q + 1]);
              acc[j][4 * q + 2] = fmaf(v[j], w4.z, acc[j][4 * q + 2]); acc[j][4 * q + 3] = fmaf(v[j], w4.w, acc[j][4 * q + 3]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      if (xo0 + j >= Wo) continue;
      float* op = out + ((((size_t)n * Ho + yo) * Wo) + xo0 + j) * out_cs + out_coff;
#pragma unroll
      for (int q = 0; q < COUT / 4; ++q) {
        float4 v = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
        if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * q); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(op + 4 * q) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// InstanceNorm2d statistics: block (n, chunk) reduces its pixel range to (mean, M2 = sum of squared deviations) per channel.
// Sums are taken of x - shift with shift = the chunk's first pixel (the "shifted data" algorithm): E[x^2] - mean^2 on raw
// fp32 sums loses log2(mean^2/var) bits when a channel carries a large offset (biased conv outputs ahead of the norm).
// Thread layout: C/4 channel quads x 256/(C/4) pixel lanes; pixel lanes are combined through LDS in double.
__global__ void __launch_bounds__(256) instnorm_stats_kernel(const float* __restrict__ x, int cs, int coff, int C, int HW,
                                                             int nchunk, float* __restrict__ partial /* [N][nchunk][C][4]: mean hi, mean lo, M2, - */) {
  __shared__ float s_red[256 * 8];
  const int quads = C / 4, plan = 256 / quads;             // C <= 256, a multiple of 4 and dividing 1024
  const int q = threadIdx.x % quads, pl = threadIdx.x / quads;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(p0 + per, HW);
  float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
  if (pl < plan && p0 < p1) {
    const float* xp = x + (size_t)n * HW * cs + coff + q * 4;
    const float4 sh = *reinterpret_cast<const float4*>(xp + (size_t)p0 * cs);
    // eight loads in flight per thread (one at a time the loop is a chain of memory latencies: 19 us for 8 MB at 64x128x128)
    constexpr int UN = 8;
    int p = p0 + pl;
    for (; p + (UN - 1) * plan < p1; p += UN * plan) {
      float4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const float4*>(xp + (size_t)(p + u * plan) * cs);
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        v[u].x -= sh.x; v[u].y -= sh.y; v[u].z -= sh.z; v[u].w -= sh.w;
        s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
        ss[0] = fmaf(v[u].x, v[u].x, ss[0]); ss[1] = fmaf(v[u].y, v[u].y, ss[1]); ss[2] = fmaf(v[u].z, v[u].z, ss[2]); ss[3] = fmaf(v[u].w, v[u].w, ss[3]);
      }
    }
    for (; p < p1; p += plan) {
      float4 v = *reinterpret_cast<const float4*>(xp + (size_t)p * cs);
      v.x -= sh.x; v.y -= sh.y; v.z -= sh.z; v.w -= sh.w;
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      ss[0] = fmaf(v.x, v.x, ss[0]); ss[1] = fmaf(v.y, v.y, ss[1]); ss[2] = fmaf(v.z, v.z, ss[2]); ss[3] = fmaf(v.w, v.w, ss[3]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { s_red[threadIdx.x * 8 + i] = s[i]; s_red[threadIdx.x * 8 + 4 + i] = ss[i]; }
  __syncthreads();
  if (threadIdx.x < quads) {
    double a[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    for (int l = 0; l < plan; ++l)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += (double)s_red[(l * quads + threadIdx.x) * 8 + i];
    float4* pp = reinterpret_cast<float4*>(partial) + ((size_t)n * nchunk + chunk) * C + threadIdx.x * 4;
    const int cnt = p1 - p0;
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (cnt > 0) {
      const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)n * HW + p0) * cs + coff + threadIdx.x * 4);
      sh[0] = v.x; sh[1] = v.y; sh[2] = v.z; sh[3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double m = cnt > 0 ? a[i] / (double)cnt : 0.0;
      const double mean = (double)sh[i] + m;                           // chunk mean, kept as a float pair
      const float hi = (float)mean;
      pp[i] = make_float4(hi, (float)(mean - (double)hi), (float)fmax(a[4 + i] - a[i] * m, 0.0) /* chunk M2 */, 0.f);
    }
  }
}

// y = act((x - mean) * rstd * gamma + beta [+ residual]); act: 0 none, 1 ReLU, 2 ELU.  Biased variance, as
// nn.InstanceNorm2d(track_running_stats=False) computes it; the chunk statistics are combined in double.
__global__ void __launch_bounds__(256) instnorm_apply_kernel(const float* __restrict__ x, int cs, int coff, int C, int N, int HW,
                                                             const float* __restrict__ partial, int nchunk,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int act, const float* __restrict__ residual, int res_cs,
                                                             int res_coff, float* __restrict__ out, int out_cs, int out_coff) {
  __shared__ float s_scale[256], s_mean[256], s_mlo[256], s_beta[256];
  const int n = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    // chunk (mean, M2) pairs -> instance mean and variance (Chan et al. pairwise update, in double)
    const float4* pp = reinterpret_cast<const float4*>(partial) + (size_t)n * nchunk * C + c;
    const int per = (HW + nchunk - 1) / nchunk;
    double s = 0.0, m2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < nchunk; ++k) {           // independent loads, issued together
      const float4 v = pp[(size_t)k * C];
      const int cnt = min(per, HW - k * per);
      if (cnt > 0) { s += (double)cnt * ((double)v.x + (double)v.y); m2 += (double)v.z; }
    }
    const double mean = s / (double)HW;
#pragma unroll 8
    for (int k = 0; k < nchunk; ++k) {
      const int cnt = min(per, HW - k * per);
      if (cnt > 0) { const float4 v = pp[(size_t)k * C]; const double d = ((double)v.x + (double)v.y) - mean; m2 += (double)cnt * d * d; }
    }
    double var = m2 / (double)HW;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    s_scale[c] = rstd * g;
    // (x - mean) first, as PyTorch does (x*scale - mean*scale cancels for |mean| >> std); the mean as a float pair: x - hi
    // is exact for x near the mean (Sterbenz), so the result carries the double mean's accuracy
    s_mean[c] = (float)mean;
    s_mlo[c] = (float)(mean - (double)s_mean[c]);
    s_beta[c] = b;
  }
  __syncthreads();
  const int quads = C / 4;
  const long long total = (long long)HW * quads;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % quads);
    const size_t p = (size_t)(i / quads) + (size_t)n * HW;
    const float4 v = *reinterpret_cast<const float4*>(x + p * cs + coff + q * 4);
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = fmaf((r[k] - s_mean[q * 4 + k]) - s_mlo[q * 4 + k], s_scale[q * 4 + k], s_beta[q * 4 + k]);
    if (residual) {
      const float4 rv = *reinterpret_cast<const float4*>(residual + p * res_cs + res_coff + q * 4);
      r[0] += rv.x; r[1] += rv.y; r[2] += rv.z; r[3] += rv.w;
    }
    if (act == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = fmaxf(r[k], 0.f);
    } else if (act == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = r[k] > 0.f ? r[k] : expm1f(r[k]);
    }
    *reinterpret_cast<float4*>(out + p * out_cs + out_coff + q * 4) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// F.interpolate(mode='bilinear') between channel-last maps, align_corners on or off (PyTorch's source-index rules:
// aten/src/ATen/native/UpSample.h area_pixel_compute_source_index); thread per (output pixel, 4-channel group).
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* __restrict__ in, int in_cs, int in_coff, int C, int N,
                                                              int H, int W, float* __restrict__ out, int out_cs, int out_coff,
                                                              int Ho, int Wo, int align) {
  const int quads = C / 4;
  const long long total = (long long)N * Ho * Wo * quads;
  const float sy = align ? (Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f) : (float)H / (float)Ho;
  const float sx = align ? (Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f) : (float)W / (float)Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % quads);
    long long p = i / quads;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float fy = align ? sy * (float)yo : fmaxf(sy * ((float)yo + 0.5f) - 0.5f, 0.f);
    float fx = align ? sx * (float)xo : fmaxf(sx * ((float)xo + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* bp = in + (size_t)n * H * W * in_cs + in_coff + q * 4;
    const float4 a = *reinterpret_cast<const float4*>(bp + ((size_t)y0 * W + x0) * in_cs);
    const float4 b = *reinterpret_cast<const float4*>(bp + ((size_t)y0 * W + x1) * in_cs);
    const float4 c = *reinterpret_cast<const float4*>(bp + ((size_t)y1 * W + x0) * in_cs);
    const float4 d = *reinterpret_cast<const float4*>(bp + ((size_t)y1 * W + x1) * in_cs);
    float4 r;
    r.x = hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
    r.y = hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
    r.z = hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z);
    r.w = hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w);
    *reinterpret_cast<float4*>(out + (((size_t)n * Ho + yo) * Wo + xo) * out_cs + out_coff + q * 4) = r;
  }
}

// MaxPool2d(3, 2, 1) as models/convert_module.py:12-27 pads it: zeros above/below (the zeros take part in the max),
// circular left/right; channel-last, thread per (output pixel, 4-channel group).
__global__ void __launch_bounds__(256) maxpool3s2_wrap_kernel(const float* __restrict__ in, int in_cs, int in_coff, int C, int N, int H,
                                                              int W, float* __restrict__ out, int out_cs, int out_coff, int Ho, int Wo) {
  const int quads = C / 4;
  const long long total = (long long)N * Ho * Wo * quads;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % quads);
    long long p = i / quads;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float4 m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yi = 2 * yo + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int xi = 2 * xo + kx - 1;
        xi = xi < 0 ? xi + W : (xi >= W ? xi - W : xi);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (yi >= 0 && yi < H) v = *reinterpret_cast<const float4*>(in + (((size_t)n * H + yi) * W + xi) * in_cs + in_coff + q * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(out + (((size_t)n * Ho + yo) * Wo + xo) * out_cs + out_coff + q * 4) = m;
  }
}

// ------------------------------------------------------------------------------------------------ ERP -> cube map
// Equirec2Cube.run (UniFuse/datasets/util.py:72-111) = scipy.ndimage.map_coordinates(order, mode='wrap') at a fixed
// coordinate table, on the image padded BELOW with its last and first rows rolled by W/2.  The padding is virtual here
// (rows H and H+1 index rows H-1 and 0 at column x - W/2); the legacy 'wrap' has period len-1; arithmetic in double, in
// scipy's tap order and without contraction, so the result is the same float.
__device__ __forceinline__ double legacy_wrap(double c, int len) {
  const double sz = (double)(len - 1);
  if (c < 0.0) c = c + sz * trunc(-c / sz) + sz;
  else if (c > sz) c = c - sz * trunc(c / sz);
  return c;
}

__global__ void __launch_bounds__(256) equirec_to_cube_kernel(const float* __restrict__ erp, int N, int H, int W, int C, long long in_sn,
                                                              long long in_sh, long long in_sw, long long in_sc,
                                                              const float* __restrict__ coor_y, const float* __restrict__ coor_x,
                                                              int face_w, int order, const float* __restrict__ pix_scale,
                                                              const float* __restrict__ mean, const float* __restrict__ stdv,
                                                              float* __restrict__ out, long long out_sn, long long out_sh,
                                                              long long out_sw, long long out_sc) {
  const int Wo = 6 * face_w;
  const long long total = (long long)N * face_w * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % ((long long)face_w * Wo));
    const int n = (int)(i / ((long long)face_w * Wo));
    const double y = legacy_wrap((double)coor_y[p], H + 2), x = legacy_wrap((double)coor_x[p], W);
    // padded-row index -> (source row, column shift)
    auto src = [&](int r, int c) -> long long {
      if (r >= H) { c -= W / 2; if (c < 0) c += W; r = (r == H) ? H - 1 : 0; }
      return (long long)r * in_sh + (long long)c * in_sw;
    };
    const float* ip = erp + n * in_sn;
    float* op = out + n * out_sn + (long long)(p / Wo) * out_sh + (long long)(p % Wo) * out_sw;
    const float ps = pix_scale ? pix_scale[p] : 1.f;
    if (order == 0) {
      const int r = min(max((int)floor(y + 0.5), 0), H + 1), c = min(max((int)floor(x + 0.5), 0), W - 1);
      const long long o = src(r, c);
      for (int ch = 0; ch < C; ++ch) {
        float v = ip[o + ch * in_sc];
        if (pix_scale) v = __fmul_rn(v, ps);
        if (mean) v = __fdiv_rn(__fsub_rn(v, mean[ch]), stdv[ch]);
        op[ch * out_sc] = v;
      }
    } else {
      const double y0 = floor(y), x0 = floor(x), fy = y - y0, fx = x - x0, gy = 1.0 - fy, gx = 1.0 - fx;
      const int r0 = (int)y0, c0 = (int)x0, r1 = min(r0 + 1, H + 1), c1 = min(c0 + 1, W - 1);
      const long long o00 = src(r0, c0), o01 = src(r0, c1), o10 = src(r1, c0), o11 = src(r1, c1);
      for (int ch = 0; ch < C; ++ch) {
        const float* q = ip + ch * in_sc;
        double t = __dmul_rn(__dmul_rn((double)q[o00], gy), gx);
        t = __dadd_rn(t, __dmul_rn(__dmul_rn((double)q[o01], gy), fx));
        t = __dadd_rn(t, __dmul_rn(__dmul_rn((double)q[o10], fy), gx));
        t = __dadd_rn(t, __dmul_rn(__dmul_rn((double)q[o11], fy), fx));
        float v = (float)t;
        if (pix_scale) v = __fmul_rn(v, ps);
        if (mean) v = __fdiv_rn(__fsub_rn(v, mean[ch]), stdv[ch]);
        op[ch * out_sc] = v;
      }
    }
  }
}

static int grid_for(long long total, int cap = 8192) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace pgrf

using namespace pgrf;

extern "C" {

int pgrf_conv2d_direct(const float* in_nchw, int N, int Cin, int H, int W, const float* w_dev, const float* bias_dev, int Cout,
                       int ksize, int stride, int wrap_w, int relu, float* out, int out_cs, int out_coff, void* stream) {
  if (!in_nchw || !w_dev || !out) return set_error(PGRF_ERR_ARG, "pgrf_conv2d_direct: null pointer");
  if (Cin < 1 || Cin > 4 || (ksize & 1) == 0 || ksize < 1 || ksize > 7 || (stride != 1 && stride != 2))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv2d_direct: Cin %d (1..4), ksize %d (odd <= 7), stride %d (1|2)", Cin, ksize, stride);
  if (!(Cout == 16 || Cout == 32 || Cout == 64) || (out_cs & 3) || (out_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv2d_direct: Cout %d (16|32|64), channel stride/offset multiples of 4", Cout);
  if (N < 1 || H < 1 || W < 1) return set_error(PGRF_ERR_ARG, "pgrf_conv2d_direct: bad dims");
  const int pad = ksize / 2;
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  const size_t shmem = (size_t)Cin * ksize * ksize * Cout * sizeof(float);
  if (Cout == 16) {
    const long long total = (long long)N * Ho * Wo;
    hipLaunchKernelGGL((conv2d_direct_kernel<16, 1>), dim3(grid_for(total)), dim3(256), shmem, (hipStream_t)stream, in_nchw, N, Cin, H, W,
                       w_dev, ksize, stride, wrap_w, out, out_cs, out_coff, Ho, Wo, bias_dev, relu);
  } else if (Cout == 64) {
    const long long total = (long long)N * Ho * Wo;
    hipLaunchKernelGGL((conv2d_direct_kernel<64, 1>), dim3(grid_for(total)), dim3(256), shmem, (hipStream_t)stream, in_nchw, N, Cin, H, W,
                       w_dev, ksize, stride, wrap_w, out, out_cs, out_coff, Ho, Wo, bias_dev, relu);
  } else {
    const long long total = (long long)N * Ho * Wo;
    hipLaunchKernelGGL((conv2d_direct_kernel<32, 1>), dim3(grid_for(total)), dim3(256), shmem, (hipStream_t)stream, in_nchw, N, Cin, H, W,
                       w_dev, ksize, stride, wrap_w, out, out_cs, out_coff, Ho, Wo, bias_dev, relu);
  }
  PGRF_CHECK_LAUNCH("conv2d_direct");
  return PGRF_OK;
}

int pgrf_instnorm_chunks(int HW) {               // pixel chunks (= workgroups) per image: ~512 pixels each, at most 32
  int n = (HW + 511) / 512;                      // (few partials: the apply kernel sums them per channel before it starts)
  return n < 1 ? 1 : (n > 32 ? 32 : n);
}

int pgrf_instnorm(const float* x, int cs, int coff, int C, int N, int HW, const float* gamma, const float* beta, float eps,
                  int act, const float* residual, int res_cs, int res_coff, float* out, int out_cs, int out_coff,
                  float* partial, void* stream) {
  if (!x || !out || !partial) return set_error(PGRF_ERR_ARG, "pgrf_instnorm: null pointer");
  if (C < 4 || C > 256 || (C & 3) || 1024 % C != 0 || (cs & 3) || (coff & 3) || (out_cs & 3) || (out_coff & 3) || (res_cs & 3) || (res_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_instnorm: C %d (power of two in [4,256]); strides/offsets multiples of 4", C);
  if (N < 1 || HW < 1 || act < 0 || act > 2) return set_error(PGRF_ERR_ARG, "pgrf_instnorm: bad arguments");
  const int nchunk = pgrf_instnorm_chunks(HW);
  hipLaunchKernelGGL(instnorm_stats_kernel, dim3(nchunk, N), dim3(256), 0, (hipStream_t)stream, x, cs, coff, C, HW, nchunk, partial);
  PGRF_CHECK_LAUNCH("instnorm_stats");
  const long long total = (long long)HW * (C / 4);
  hipLaunchKernelGGL(instnorm_apply_kernel, dim3(grid_for(total, 1024), N), dim3(256), 0, (hipStream_t)stream, x, cs, coff, C, N, HW,
                     partial, nchunk, gamma, beta, eps, act, residual, res_cs, res_coff, out, out_cs, out_coff);
  PGRF_CHECK_LAUNCH("instnorm_apply");
  return PGRF_OK;
}

int pgrf_maxpool3s2_wrap(const float* in, int in_cs, int in_coff, int C, int N, int H, int W, float* out, int out_cs, int out_coff,
                         void* stream) {
  if (!in || !out) return set_error(PGRF_ERR_ARG, "pgrf_maxpool3s2_wrap: null pointer");
  if ((C & 3) || C < 4 || (in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_maxpool3s2_wrap: C and strides/offsets must be multiples of 4");
  if (N < 1 || H < 1 || W < 2) return set_error(PGRF_ERR_ARG, "pgrf_maxpool3s2_wrap: bad dims");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3s2_wrap_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, in_cs, in_coff, C, N, H, W, out,
                     out_cs, out_coff, Ho, Wo);
  PGRF_CHECK_LAUNCH("maxpool3s2_wrap");
  return PGRF_OK;
}

int pgrf_resize_bilinear(const float* in, int in_cs, int in_coff, int C, int N, int H, int W, float* out, int out_cs, int out_coff,
                         int Ho, int Wo, int align_corners, void* stream) {
  if (!in || !out) return set_error(PGRF_ERR_ARG, "pgrf_resize_bilinear: null pointer");
  if ((C & 3) || C < 4 || (in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_resize_bilinear: C and strides/offsets must be multiples of 4");
  if (N < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1) return set_error(PGRF_ERR_ARG, "pgrf_resize_bilinear: bad dims");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, in_cs, in_coff, C, N, H, W,
                     out, out_cs, out_coff, Ho, Wo, align_corners ? 1 : 0);
  PGRF_CHECK_LAUNCH("resize_bilinear");
  return PGRF_OK;
}

int pgrf_equirec_to_cube(const float* erp, int N, int H, int W, int C, long long in_sn, long long in_sh, long long in_sw, long long in_sc,
                         const float* coor_y, const float* coor_x, int face_w, int order, const float* pix_scale, const float* mean,
                         const float* stdv, float* out, long long out_sn, long long out_sh, long long out_sw, long long out_sc,
                         void* stream) {
  if (!erp || !out || !coor_y || !coor_x) return set_error(PGRF_ERR_ARG, "pgrf_equirec_to_cube: null pointer");
  if (N < 1 || H < 2 || W < 2 || C < 1 || face_w < 1) return set_error(PGRF_ERR_ARG, "pgrf_equirec_to_cube: bad dims");
  if (order != 0 && order != 1) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_equirec_to_cube: order must be 0 (nearest) or 1 (linear)");
  if ((mean == nullptr) != (stdv == nullptr)) return set_error(PGRF_ERR_ARG, "pgrf_equirec_to_cube: mean and std go together");
  const long long total = (long long)N * face_w * 6 * face_w;
  hipLaunchKernelGGL(equirec_to_cube_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, erp, N, H, W, C, in_sn, in_sh,
                     in_sw, in_sc, coor_y, coor_x, face_w, order, pix_scale, mean, stdv, out, out_sn, out_sh, out_sw, out_sc);
  PGRF_CHECK_LAUNCH("equirec_to_cube");
  return PGRF_OK;
}

}  // extern "C"
