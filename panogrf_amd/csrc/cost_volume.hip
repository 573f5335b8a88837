// 360-degree spherical cost volume for gfx950.
//
// HBM-write bound: the volume [B,D,h,w,C] is written exactly once, channel-last, 16 B per lane, a
// whole wave writing 1 KiB of consecutive bytes; the source feature maps are tiny (h*w*C*4 B per
// view, L2 resident) and every bilinear tap is one aligned 16-B load per lane.
// Geometry (back-projection, two 3x3 transforms, atan2/acos, tap setup) is computed ONCE per voxel
// by one lane and handed to the C/4 lanes that move that voxel's channels through wave shuffles,
// instead of being recomputed per channel group.
#include "pgrf_experiments.h"
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "pgrf_internal.h"

namespace pgrf {

#define PGRF_CHECK_LAUNCH(what)                                                                       \
  do {                                                                                                \
    const hipError_t e__ = hipGetLastError();                                                         \
    if (e__ != hipSuccess) return set_error(PGRF_ERR_LAUNCH, what " launch failed: %s", hipGetErrorString(e__)); \
  } while (0)

constexpr int kMaxGuided = 16;
struct HypArgs {
  const float* ref_mu;      // [B,hw]
  const float* uniform;     // [n_uniform] ascending
  float* out;               // [B,D,hw]
  float offs[kMaxGuided];
  int n_guided, n_uniform, B, hw;
  float lo, hi;
};

// pipeline3_model.py:723-733 (clamp(ref_mu + k*fixed_sigma)), :774-815 (linspace planes, concat, sort).
// The sorted sequence of a pixel is the merge of its sorted guided planes g[0..n_g) with the ascending uniform planes
// u[0..n_u) shared by all pixels: guided plane k lands in slot  k + #{u < g_k}  (ties: guided first; equal values are
// interchangeable), every other slot d holds the uniform plane  d - #{guided slots < d}.
// One thread = one pixel x kHypSlice consecutive slots (consecutive threads = consecutive pixels: coalesced loads and
// stores); the pixel's guided slots are found once per thread by branch-free binary search over the uniform planes in LDS.
// NG = compile-time bound on n_guided (the sort and the searches are unrolled over it).
constexpr int kHypSlice = 16;
template <int NG>
__global__ void __launch_bounds__(256) depth_hypotheses_kernel(const HypArgs a, int nslice) {
  __shared__ float s_uni[256];
  const int D = a.n_guided + a.n_uniform;
  for (int i = threadIdx.x; i < a.n_uniform && i < 256; i += 256) s_uni[i] = a.uniform[i];
  __syncthreads();
  // 32-bit index arithmetic (the launcher checks the sizes): 64-bit div/mod costs ~100 instructions each
  const unsigned total = (unsigned)a.B * nslice * a.hw;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned row = i / (unsigned)a.hw;
    const unsigned pix = i - row * a.hw;
    const unsigned b = row / (unsigned)nslice;
    const int d0 = (int)(row - b * nslice) * kHypSlice;
    const float mu = a.ref_mu[b * a.hw + pix];
    float gv[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) gv[k] = (k < a.n_guided) ? fminf(fmaxf(mu + a.offs[k], a.lo), a.hi) : 3.0e38f;
#pragma unroll
    for (int k = 1; k < NG; ++k)         // insertion sort of the (few) guided planes
#pragma unroll
      for (int j = k; j > 0; --j)
        if (gv[j] < gv[j - 1]) { const float t = gv[j]; gv[j] = gv[j - 1]; gv[j - 1] = t; }
    int pos[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      int lo = 0;
#pragma unroll
      for (int step = 256; step >= 1; step >>= 1)
        if (lo + step <= a.n_uniform && s_uni[lo + step - 1] < gv[k]) lo += step;
      pos[k] = (k < a.n_guided) ? k + lo : 0x7fffffff;
    }
    int cnt = 0;                         // guided slots before d0
#pragma unroll
    for (int k = 0; k < NG; ++k) cnt += (pos[k] < d0) ? 1 : 0;
    float* o = a.out + ((size_t)b * D + d0) * a.hw + pix;
    for (int d = d0; d < d0 + kHypSlice && d < D; ++d) {
      float val = 0.f;
      bool hit = false;
#pragma unroll
      for (int k = 0; k < NG; ++k)
        if (pos[k] == d) { val = gv[k]; hit = true; }
      if (!hit) val = s_uni[d - cnt];
      cnt += hit ? 1 : 0;
      *o = val;
      o += a.hw;
    }
  }
}

int launch_depth_hypotheses(const float* ref_mu, int B, int hw, const float* offs_host, int n_guided, float lo, float hi,
                            const float* uniform_dev, int n_uniform, float* out, void* stream) {
  if (n_uniform > 256) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_depth_hypotheses: n_uniform=%d > 256", n_uniform);
  const int D = n_guided + n_uniform;
  const int nslice = (D + kHypSlice - 1) / kHypSlice;
  if ((double)B * hw * nslice >= 2147483648.0) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_depth_hypotheses: >= 2^31 work items");
  HypArgs a;
  a.ref_mu = ref_mu; a.uniform = uniform_dev; a.out = out; a.n_guided = n_guided; a.n_uniform = n_uniform;
  a.B = B; a.hw = hw; a.lo = lo; a.hi = hi;
  for (int k = 0; k < kMaxGuided; ++k) a.offs[k] = (k < n_guided) ? offs_host[k] : 0.f;
  const size_t total = (size_t)B * hw * nslice;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (n_guided <= 8)
    hipLaunchKernelGGL(depth_hypotheses_kernel<8>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, a, nslice);
  else
    hipLaunchKernelGGL(depth_hypotheses_kernel<kMaxGuided>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, a, nslice);
  PGRF_CHECK_LAUNCH("depth_hypotheses");
  return PGRF_OK;
}

// ---------------------------------------------------------------------------------------------
constexpr int kMaxSrc = 4;
struct CvArgs {
  const float* feats;     // [B,S,h,w,C]
  const float* depth;     // [B,D,h,w]
  const float* rots;      // [B,S,3,3]
  const float* trans;     // [B,S,3]
  float* out;             // [B,D,h,w,C]
  int B, S, h, w, C, D;
  int ref_idx, n_src;
  int src_idx[kMaxSrc];
  float div;              // each view's cost is divided by this before accumulation (mv: S-2)
  int cost_type;          // 0 abs_diff, 1 dot, 2 none
  int dataset;            // PGRF_DATASET_*: spherical convention (spherical_cost_volume.py:153-190,281-296)
  int stream_stores;      // write the volume with non-temporal stores
};

struct Taps { int o00, o01, o10, o11; float w00, w01, w10, w11; };

// The warp geometry (inv3, pixel_dir, warp_taps) is compiled with `fp contract(on)`: a * b + c is fused where it is WRITTEN as one
// expression, by the front end, and nowhere else.  hipcc's default lets the back end fuse across statements, and what it fuses depends
// on the code the function is inlined into -- the single-stage and the pipelined kernel then disagree by an ulp in a pixel coordinate
// (1.5e-5 in a cost) for general poses.  With the pragma every instance computes the same bits.
__device__ __forceinline__ void inv3(const float* m, float* o) {   // adjugate / determinant
#pragma clang fp contract(on)
  const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const float A = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
  const float det = a * A + b * Bc + c * Cc;
  const float r = 1.f / det;
  o[0] = A * r; o[1] = -(b * i - c * h) * r; o[2] = (b * f - c * e) * r;
  o[3] = Bc * r; o[4] = (a * i - c * g) * r; o[5] = -(a * f - c * d) * r;
  o[6] = Cc * r; o[7] = -(a * h - b * g) * r; o[8] = (a * e - b * d) * r;
}

// Per-wave constants of one batch element: inverse reference rotation and the poses involved
// (spherical_cost_volume.py:143-150: P_w = R_ref^-1 (z*dir - t_ref), P_s = R_src P_w + t_src)
struct BatchPose {
  float Ri[9], tr[3];
};

// unit direction of a reference pixel in the dataset's convention (spherical_cost_volume.py:281-296,
// my_torch_helpers.py:34-56); Python folds the scalar factors in double before they meet the float tensors.
// Depends on the pixel only: computed once per wave and reused for every depth plane of the wave's chunk.
template <bool GEN>
__device__ __forceinline__ void pixel_dir(const CvArgs& a, int y, int x, float& ux, float& uy, float& uz) {
#pragma clang fp contract(on)
  const float PI_F = 3.14159265358979323846f, HALF_PI_F = 1.57079632679489661923f, TWO_PI_F = 6.28318530717958647692f;
  const double PI_D = 3.14159265358979323846;
  const float fx = (float)x, fy = (float)y;
  if (!GEN || a.dataset == PGRF_DATASET_M3D) {
    const float phi = (fy + 0.5f) * (float)(PI_D / (double)a.h);
    const float theta = (fx + 0.5f) * (float)(2.0 * PI_D / (double)a.w) - HALF_PI_F;
    const float sp = sinf(phi);
    ux = sp * cosf(theta); uy = cosf(phi); uz = sp * sinf(theta);
  } else if (a.dataset == PGRF_DATASET_REPLICA_TEST) {
    const float theta = (float)(2.0 * PI_D / (double)a.w) * (fx + 0.5f) - PI_F;
    const float phi = -(fy + 0.5f) * PI_F / (float)a.h + HALF_PI_F;
    const float cp = cosf(phi);
    ux = sinf(theta) * cp; uy = -sinf(phi); uz = cosf(theta) * cp;
  } else if (a.dataset == PGRF_DATASET_RESIDENTIAL) {
    const float theta = PI_F * (2.f * fx / (float)(a.w - 1) - 1.5f);
    const float phi = PI_F * (0.5f - fy / (float)(a.h - 1));
    const float cp = cosf(phi);
    ux = cosf(theta) * cp; uz = sinf(theta) * cp; uy = sinf(phi);
  } else {   // CoffeeArea
    const float theta = (float)(-2.0 * PI_D / (double)(a.w - 1)) * fx + TWO_PI_F;
    const float phi = (float)(PI_D / (double)(a.h - 1)) * fy;
    const float sp = sinf(phi);
    ux = sp * cosf(theta); uy = sp * sinf(theta); uz = cosf(phi);
  }
}

// spherical_cost_volume.py:135-198 + my_torch_helpers.py:62-96 for one (voxel, source view): back-project, move to the
// source frame, cartesian -> sphere -> grid, bilinear tap setup (zeros padding, align_corners=True)
template <bool GEN>
__device__ __forceinline__ Taps warp_taps(const CvArgs& a, const BatchPose& bp, const float* __restrict__ Rs,
                                          const float* __restrict__ ts, float ux, float uy, float uz, float depth) {
#pragma clang fp contract(on)
  const float PI_F = 3.14159265358979323846f, HALF_PI_F = 1.57079632679489661923f, TWO_PI_F = 6.28318530717958647692f;
  const float mx = depth * ux, my = depth * uy, mz = depth * uz;
  const float dx = mx - bp.tr[0], dy = my - bp.tr[1], dz = mz - bp.tr[2];
  const float wx = bp.Ri[0] * dx + bp.Ri[1] * dy + bp.Ri[2] * dz;
  const float wy = bp.Ri[3] * dx + bp.Ri[4] * dy + bp.Ri[5] * dz;
  const float wz = bp.Ri[6] * dx + bp.Ri[7] * dy + bp.Ri[8] * dz;
  const float cx = Rs[0] * wx + Rs[1] * wy + Rs[2] * wz + ts[0];
  const float cy = Rs[3] * wx + Rs[4] * wy + Rs[5] * wz + ts[1];
  const float cz = Rs[6] * wx + Rs[7] * wy + Rs[8] * wz + ts[2];
  const float radius = sqrtf(cx * cx + cy * cy + cz * cz);
  // x mod 2pi for x in [0, 4pi): fmodf is exact, and so is this subtraction (Sterbenz) -> bit-identical, no libm call
  auto wrap2pi = [&](float x) { return x >= TWO_PI_F ? x - TWO_PI_F : x; };
  float u, v;
  if (!GEN || a.dataset == PGRF_DATASET_M3D) {
    const float th = atan2f(cz, cx);
    const float yor = cy / radius;
    const float LIN = 0.17453292519943295f, COSD = 0.984807753012208f, ONE_M_COSD = 0.01519224698779198f;
    float ph;
    if (fabsf(yor) < COSD) ph = acosf(yor);
    else if (cy >= 0.f) ph = LIN * (1.f - yor) / ONE_M_COSD;
    else ph = PI_F - LIN * (yor + 1.f) / ONE_M_COSD;
    u = wrap2pi(th + HALF_PI_F + TWO_PI_F) / PI_F - 1.f;
    v = 2.f * ph / PI_F - 1.f;
  } else if (a.dataset == PGRF_DATASET_REPLICA_TEST) {
    const float th = atan2f(cx, cz);
    const float ph = -asinf(cz / radius);          // the reference takes z here (my_torch_helpers.py:99)
    u = wrap2pi(th + PI_F + TWO_PI_F) / PI_F - 1.f;
    v = 2.f * (-ph + HALF_PI_F) / PI_F - 1.f;
  } else if (a.dataset == PGRF_DATASET_RESIDENTIAL) {
    float th = -atan2f(-cz, cx);
    const float ph = asinf(cy / radius);
    if (th > HALF_PI_F && th <= TWO_PI_F) th -= TWO_PI_F;
    u = fmodf(th + 4.71238898038469f, TWO_PI_F) / PI_F - 1.f;
    v = 2.f * (HALF_PI_F - ph) / PI_F - 1.f;
  } else {   // CoffeeArea
    float th = atan2f(cy, cx);
    const float ph = acosf(cz / radius);
    if (th < 0.f) th += TWO_PI_F;
    u = (TWO_PI_F - th) / PI_F - 1.f;
    v = 2.f * ph / PI_F - 1.f;
  }
  // grid_sample(bilinear, zeros, align_corners=True)
  const float ix = (u + 1.f) / 2.f * (float)(a.w - 1);
  const float iy = (v + 1.f) / 2.f * (float)(a.h - 1);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  const float ax = ix - fx0, ay = iy - fy0;
  const bool vx0 = x0 >= 0 && x0 < a.w, vx1 = x1 >= 0 && x1 < a.w, vy0 = y0 >= 0 && y0 < a.h, vy1 = y1 >= 0 && y1 < a.h;
  const int cx0 = min(max(x0, 0), a.w - 1), cx1 = min(max(x1, 0), a.w - 1);
  const int cy0 = min(max(y0, 0), a.h - 1), cy1 = min(max(y1, 0), a.h - 1);
  Taps t;
  // element offsets of the 4 taps inside one view's [h,w,C] map
  t.o00 = (cy0 * a.w + cx0) * a.C; t.o01 = (cy0 * a.w + cx1) * a.C; t.o10 = (cy1 * a.w + cx0) * a.C; t.o11 = (cy1 * a.w + cx1) * a.C;
  t.w00 = (vx0 && vy0) ? (1.f - ax) * (1.f - ay) : 0.f;
  t.w01 = (vx1 && vy0) ? ax * (1.f - ay) : 0.f;
  t.w10 = (vx0 && vy1) ? (1.f - ax) * ay : 0.f;
  t.w11 = (vx1 && vy1) ? ax * ay : 0.f;
  return t;
}

// One wave = 64 consecutive pixels of one batch element x DCH consecutive depth planes.
//   * per wave, once: the pixels' unit directions (4 libm sin/cos), the inverse reference rotation, the reference
//     features of the pixels each lane will write (they do not depend on the plane);
//   * per plane: lane i computes the warp of voxel (plane, pixel i) into every source view (atan2/acos + tap setup, the
//     part that cannot be shared) and parks the 8 tap fields in LDS; then the wave streams the 64 voxels out, C/CPL
//     lanes per voxel, each lane blending CPL channels from 4 taps (16-byte loads) and writing CPL*4 contiguous
//     bytes: a wave instruction writes 1 KiB of consecutive addresses.
// Template switches keep run-time branches out of the streaming loop:
//   GEN   false = m3d convention only;  C32 true = C == 32 with 8 channels per lane and the 4 inner iterations
//   unrolled (reference features hoisted out of the plane loop), false = any power-of-two C, 4 channels per lane;
//   ABS   true = abs_diff cost (the only type the reference uses), false = run-time cost type;
//   DIVIDE false = scale by 1/div, exact because div is a power of two (1 for two views, 2 for S = 4), true = IEEE division.
#ifndef PGRF_CV_WAVES
#define PGRF_CV_WAVES 1          // minimum waves per SIMD the register allocation must allow (tuning knob)
#endif
#ifndef PGRF_CV_TILE_W_DEFAULT
#define PGRF_CV_TILE_W_DEFAULT 16   // columns of a wave's pixel patch (64 = a 1-pixel-high strip; PGRF_CV_TILE_W overrides at run time;
                                    // measured at 128x256x64, single-stage kernel: 64 -> 80.6, 32 -> 83-85, 16 -> 71-74, 8 -> 73-75 us)
#endif
#ifndef PGRF_CV_PIPE_DEFAULT
#define PGRF_CV_PIPE_DEFAULT 2      // voxel groups in flight per wave of the pipelined instance (0 = single-stage kernel);
                                    // 128x256x64, one view: 1 -> 64-66 us (4 waves per SIMD), 2 -> 64-65 (3), 4 -> 69-70 (2); two views
                                    // per batch (streamed stores): 1 -> 54-55, 2 -> 52-55, 4 -> 55-56 us per view
#endif
#ifndef PGRF_CV_PIPE_ORDER
#define PGRF_CV_PIPE_ORDER 0
#endif
#ifndef PGRF_CV_PIPE_DCH
#define PGRF_CV_PIPE_DCH 4          // depth planes per wave of the pipelined instance (128x256x64, one view: 2 -> 72-76, 4 -> 64-65,
                                    // 8 -> 72-73, 16 -> 75 us), halved while the launch has fewer than 2 048 waves
#endif
#ifndef PGRF_CV_MINITEMS
#define PGRF_CV_MINITEMS 4096    // planes per wave are halved until the launch has this many waves (tuning knob;
                                 // measured at 128x256x64: 4096 -> 74 us, 8192 -> 85, 16384 -> 76, 32768 -> 82)
#endif
template <bool GEN, int NSRC, bool C32, bool ABS, bool DIVIDE>
__global__ void __launch_bounds__(256, PGRF_CV_WAVES) cost_volume_kernel(const CvArgs a, int dch, int nchunk, int tiles_per_img, int tws) {
  // A lane always moves 4 channels (16 B) of a voxel and C/4 consecutive lanes cover the voxel's contiguous C*4 bytes, so
  // every wave load instruction reads whole 128-byte lines and every wave store instruction writes 1 KiB of
  // consecutive addresses.  C32: 8 lanes per voxel, and a lane handles G = 2 voxels (8 apart) per inner iteration for
  // twice the loads in flight; generic C: one voxel per lane and iteration.
  constexpr int G = C32 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds_taps[4][NSRC][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = C32 ? 32 : a.C;
  const int lpv = C / 4;                           // lanes per voxel (power of two, <= 64)
  const int vpg = 64 / lpv;                        // voxels per group (one wave instruction)
  constexpr int NIT_STATIC = C32 ? 4 : 1;
  const int nit = C32 ? 4 : lpv;                   // inner iterations per 64 voxels (G * vpg voxels each)
  const int hw = a.h * a.w;
  const long item = (long)blockIdx.x * 4 + wave;
  const long nitems = (long)a.B * tiles_per_img * nchunk;
  if (item >= nitems) return;
  const int chunk = __builtin_amdgcn_readfirstlane((int)(item % nchunk));
  const int tile = __builtin_amdgcn_readfirstlane((int)(item / nchunk));
  const int b = tile / tiles_per_img;
  // The wave's 64 pixels: a (64 >> tws)-row x (1 << tws)-column patch of the reference image (tws = 6: 64 consecutive pixels in
  // row-major order, any h * w).  A 2-D patch warps into a compact patch of the source map: the four taps of its 64 voxels touch
  // ~80-100 distinct 128-byte texels instead of ~130-200 along a 1-pixel-high strip, so more of the gathers hit the L1 and fewer
  // requests go to the L2.  Voxel `o` (the lane that owns its geometry) is pixel pix_of(o); a store instruction still covers >= 8
  // consecutive pixels of one row (1 KiB of consecutive bytes at C = 32).
  const int twm = (1 << tws) - 1;
  int pix0;
  {
    const int t = tile % tiles_per_img;
    if (tws == 6) pix0 = t * 64;
    else { const int tiles_x = a.w >> tws; pix0 = (t / tiles_x) * (64 >> tws) * a.w + ((t % tiles_x) << tws); }
  }
  auto pix_of = [&](int o) { return pix0 + (o >> tws) * a.w + (o & twm); };
  const int pix = min(pix_of(lane), hw - 1);
  float ux, uy, uz;
  pixel_dir<GEN>(a, pix / a.w, pix % a.w, ux, uy, uz);
  BatchPose bp;
  {
    const float* Rr = a.rots + ((size_t)b * a.S + a.ref_idx) * 9;
    const float* tr = a.trans + ((size_t)b * a.S + a.ref_idx) * 3;
    inv3(Rr, bp.Ri);
    bp.tr[0] = tr[0]; bp.tr[1] = tr[1]; bp.tr[2] = tr[2];
  }
  const int sub = lane / lpv, cg = lane % lpv;
  const float* __restrict__ fb = a.feats + (size_t)b * a.S * hw * C;      // this batch element's views
  const float rdiv = 1.f / a.div;

  // reference features of the voxels this lane writes: plane-independent, so loaded once when the iteration count is static
  float4 rvs[NIT_STATIC][G];
  if constexpr (C32) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int rpix = min(pix_of((it * G + g) * vpg + sub), hw - 1);
        rvs[it][g] = *reinterpret_cast<const float4*>(fb + ((size_t)a.ref_idx * hw + rpix) * C + cg * 4);
      }
  }

  auto cost_of = [&](const float4& wv, const float4& rv) {
    float4 c;
    if (ABS || a.cost_type == 0) { c.x = fabsf(wv.x - rv.x); c.y = fabsf(wv.y - rv.y); c.z = fabsf(wv.z - rv.z); c.w = fabsf(wv.w - rv.w); }
    else if (a.cost_type == 1) { c.x = wv.x * rv.x; c.y = wv.y * rv.y; c.z = wv.z * rv.z; c.w = wv.w * rv.w; }
    else c = wv;
    if (DIVIDE) { c.x /= a.div; c.y /= a.div; c.z /= a.div; c.w /= a.div; }
    else { c.x *= rdiv; c.y *= rdiv; c.z *= rdiv; c.w *= rdiv; }
    return c;
  };
  auto stream_voxels = [&](int it, const float4 (&rv)[G], float* __restrict__ op) {
    float4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      const float* sb = fb + (size_t)a.src_idx[s] * hw * C + cg * 4;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int owner = (it * G + g) * vpg + sub;          // lane that computed this voxel's taps
        const float4* tpv = reinterpret_cast<const float4*>(&lds_taps[wave][s][owner][0]);
        const float4 to = tpv[0], tw = tpv[1];
#ifdef PGRF_CV_ABL_GATHER   // timing experiment: every tap reads element 0 of the source map
        const unsigned o00 = 0, o01 = 0, o10 = 0, o11 = 0;
#else
        const unsigned o00 = (unsigned)__float_as_int(to.x), o01 = (unsigned)__float_as_int(to.y);
        const unsigned o10 = (unsigned)__float_as_int(to.z), o11 = (unsigned)__float_as_int(to.w);
#endif
        const float4 p00 = *reinterpret_cast<const float4*>(sb + o00);
        const float4 p01 = *reinterpret_cast<const float4*>(sb + o01);
        const float4 p10 = *reinterpret_cast<const float4*>(sb + o10);
        const float4 p11 = *reinterpret_cast<const float4*>(sb + o11);
        float4 wv;
        wv.x = p00.x * tw.x + p01.x * tw.y + p10.x * tw.z + p11.x * tw.w;
        wv.y = p00.y * tw.x + p01.y * tw.y + p10.y * tw.z + p11.y * tw.w;
        wv.z = p00.z * tw.x + p01.z * tw.y + p10.z * tw.z + p11.z * tw.w;
        wv.w = p00.w * tw.x + p01.w * tw.y + p10.w * tw.z + p11.w * tw.w;
        const float4 c = cost_of(wv, rv[g]);
        acc[g].x += c.x; acc[g].y += c.y; acc[g].z += c.z; acc[g].w += c.w;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int opix = pix_of((it * G + g) * vpg + sub);
#ifdef PGRF_CV_ABL_STORE    // timing experiment: only values that can never occur are stored
      if (opix < hw && acc[g].x == -12345.f)
#else
      if (opix < hw)
#endif
      {
        if (a.stream_stores) {                             // batches beyond the last-level cache (or PGRF_CV_STREAM_STORES), see pgrf_cost_volume_erp
          typedef float f32x4nt __attribute__((ext_vector_type(4)));
          const f32x4nt vv = {acc[g].x, acc[g].y, acc[g].z, acc[g].w};
          __builtin_nontemporal_store(vv, reinterpret_cast<f32x4nt*>(op + (size_t)opix * C + cg * 4));
        } else {
          *reinterpret_cast<float4*>(op + (size_t)opix * C + cg * 4) = acc[g];
        }
      }
    }
  };

  const int d0 = chunk * dch, d1 = min(d0 + dch, a.D);
  const float* __restrict__ dp = a.depth + (size_t)b * a.D * hw + pix;
  float depth = dp[(size_t)d0 * hw];
  for (int d = d0; d < d1; ++d) {
    const float depth_next = dp[(size_t)min(d + 1, d1 - 1) * hw];     // in flight while this plane is processed
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      const float* Rs = a.rots + ((size_t)b * a.S + a.src_idx[s]) * 9;
      const float* ts = a.trans + ((size_t)b * a.S + a.src_idx[s]) * 3;
#ifdef PGRF_CV_ABL_GEOM     // timing experiment: no warp geometry
      Taps t; t.o00 = lane * 32; t.o01 = lane * 32 + 32; t.o10 = lane * 32 + 8192; t.o11 = lane * 32 + 8224; t.w00 = depth; t.w01 = ux; t.w10 = uy; t.w11 = uz;
#else
      const Taps t = warp_taps<GEN>(a, bp, Rs, ts, ux, uy, uz, depth);
#endif
      float4* dst = reinterpret_cast<float4*>(&lds_taps[wave][s][lane][0]);
      dst[0] = make_float4(__int_as_float(t.o00), __int_as_float(t.o01), __int_as_float(t.o10), __int_as_float(t.o11));
      dst[1] = make_float4(t.w00, t.w01, t.w10, t.w11);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* __restrict__ op = a.out + ((size_t)b * a.D + d) * hw * C;
    if constexpr (C32) {
#pragma unroll
      for (int it = 0; it < 4; ++it) stream_voxels(it, rvs[it], op);
    } else {
      for (int it = 0; it < nit; ++it) {
        const int rpix = min(pix_of(it * vpg + sub), hw - 1);
        float4 rv[G];
        rv[0] = *reinterpret_cast<const float4*>(fb + ((size_t)a.ref_idx * hw + rpix) * C + cg * 4);
        stream_voxels(it, rv, op);
      }
    }
    __builtin_amdgcn_wave_barrier();     // the next plane's tap stores must not pass this plane's tap loads
    depth = depth_next;
  }
}


// ---------------------------------------------------------------------------------------------
// Pipelined instance of the same computation for the shipped case: m3d convention, C = 32, abs_diff, power-of-two divisor,
// images that divide into whole 64-pixel patches, <= 2 source views per voxel.
// The kernel above finishes one group of voxels (loads -> wait for all of them -> blend -> store) before it requests the next,
// and gfx950 retires a wave's loads and stores in issue order on ONE counter: the loads of a group can only be seen complete once
// the stores of the group before them are acknowledged, so a wave alternates between waiting for the read path and for the write
// path and neither stays busy (r05: 63 % of the wave cycles parked in s_waitcnt at 46-49 % of 8 TB/s).  Here a wave keeps P groups
// of 8 voxels in flight across group AND plane boundaries: while group j is blended and stored, the taps of groups j+1 .. j+P-1
// are on their way and group j+P is requested; the warp geometry of plane d+1 is computed (VALU) at the start of plane d's
// streaming, into the other half of a two-plane tap buffer, so the stream never drains at a plane boundary either.  Same
// operations on the same values in the same order as cost_volume_kernel: bit-identical volumes (tests/test_gpu_cost_volume.py).
template <int NSRC, int P, bool NT>
__global__ void __launch_bounds__(256) cost_volume_pipe_kernel(const CvArgs a, int dch, int nchunk, int tiles_per_img, int tws, int order) {
  constexpr int C = 32, NG = 8;                    // 8 lanes per voxel, 8 voxels per group (one wave instruction), 8 groups per plane
  static_assert(NG % P == 0, "the stage index of a group must be static");
  __shared__ __attribute__((aligned(16))) float lds_taps[4][2][NSRC][64][8];   // [wave][plane parity][source][voxel][4 offsets | 4 weights]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hw = a.h * a.w;
  const long item = (long)blockIdx.x * 4 + wave;
  const long nitems = (long)a.B * tiles_per_img * nchunk;
  if (item >= nitems) return;
  // order 0: a tile's depth chunks are consecutive work items (a workgroup = one patch at four depth ranges);
  // order 1: a chunk's tiles are consecutive (a workgroup = four neighbouring patches at the same depths)
  const long ntile_all = (long)a.B * tiles_per_img;
  const int chunk = __builtin_amdgcn_readfirstlane((int)(order ? item / ntile_all : item % nchunk));
  const int tile = __builtin_amdgcn_readfirstlane((int)(order ? item % ntile_all : item / nchunk));
  const int b = tile / tiles_per_img;
  const int twm = (1 << tws) - 1;
  int pix0;
  {
    const int t = tile % tiles_per_img;
    if (tws == 6) pix0 = t * 64;
    else { const int tiles_x = a.w >> tws; pix0 = (t / tiles_x) * (64 >> tws) * a.w + ((t % tiles_x) << tws); }
  }
  auto pix_of = [&](int o) { return pix0 + (o >> tws) * a.w + (o & twm); };
  const int pix = pix_of(lane);
  float ux, uy, uz;
  pixel_dir<false>(a, pix / a.w, pix % a.w, ux, uy, uz);
  BatchPose bp;
  {
    const float* Rr = a.rots + ((size_t)b * a.S + a.ref_idx) * 9;
    const float* tr = a.trans + ((size_t)b * a.S + a.ref_idx) * 3;
    inv3(Rr, bp.Ri);
    bp.tr[0] = tr[0]; bp.tr[1] = tr[1]; bp.tr[2] = tr[2];
  }
  const int sub = lane >> 3, cg = lane & 7;
  const float* __restrict__ fb = a.feats + (size_t)b * a.S * hw * C;
  const float rdiv = 1.f / a.div;
  // group j = voxels j*8 .. j*8+7 of the patch; this lane moves channels cg*4 .. cg*4+3 of voxel j*8 + sub.  (sub < 8 <= patch
  // width: the voxel's pixel is pix_of(j*8) + sub, a wave-uniform offset per group.)
  float4 rvs[NG];
  unsigned vo[NG];                                 // element offset of the lane's 16 bytes inside a plane
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int rpix = pix_of(j * 8) + sub;
    vo[j] = (unsigned)rpix * C + cg * 4;
    rvs[j] = *reinterpret_cast<const float4*>(fb + (size_t)a.ref_idx * hw * C + vo[j]);
  }
  struct Stage { float4 p[NSRC][4]; float4 tw[NSRC]; };
  Stage st[P];
  auto geometry = [&](float depth, int par) {
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      const float* Rs = a.rots + ((size_t)b * a.S + a.src_idx[s]) * 9;
      const float* ts = a.trans + ((size_t)b * a.S + a.src_idx[s]) * 3;
      const Taps t = warp_taps<false>(a, bp, Rs, ts, ux, uy, uz, depth);
      float4* dst = reinterpret_cast<float4*>(&lds_taps[wave][par][s][lane][0]);
      dst[0] = make_float4(__int_as_float(t.o00), __int_as_float(t.o01), __int_as_float(t.o10), __int_as_float(t.o11));
      dst[1] = make_float4(t.w00, t.w01, t.w10, t.w11);
    }
  };
  auto issue = [&](Stage& S, int par, int j) {
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      const float* sb = fb + (size_t)a.src_idx[s] * hw * C + cg * 4;
      const float4* tpv = reinterpret_cast<const float4*>(&lds_taps[wave][par][s][j * 8 + sub][0]);
      const float4 to = tpv[0];
      S.tw[s] = tpv[1];
      S.p[s][0] = *reinterpret_cast<const float4*>(sb + (unsigned)__float_as_int(to.x));
      S.p[s][1] = *reinterpret_cast<const float4*>(sb + (unsigned)__float_as_int(to.y));
      S.p[s][2] = *reinterpret_cast<const float4*>(sb + (unsigned)__float_as_int(to.z));
      S.p[s][3] = *reinterpret_cast<const float4*>(sb + (unsigned)__float_as_int(to.w));
    }
  };
  auto finish = [&](const Stage& S, int j, float* __restrict__ op) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      const float4 tw = S.tw[s];
      const float4 &p00 = S.p[s][0], &p01 = S.p[s][1], &p10 = S.p[s][2], &p11 = S.p[s][3];
      float4 wv;       // ((p00 w00 + p01 w01) + p10 w10) + p11 w11 as the contraction cost_volume_kernel compiles to
      wv.x = fmaf(p11.x, tw.w, fmaf(p10.x, tw.z, fmaf(p00.x, tw.x, p01.x * tw.y)));
      wv.y = fmaf(p11.y, tw.w, fmaf(p10.y, tw.z, fmaf(p00.y, tw.x, p01.y * tw.y)));
      wv.z = fmaf(p11.z, tw.w, fmaf(p10.z, tw.z, fmaf(p00.z, tw.x, p01.z * tw.y)));
      wv.w = fmaf(p11.w, tw.w, fmaf(p10.w, tw.z, fmaf(p00.w, tw.x, p01.w * tw.y)));
      const float4 rv = rvs[j];
      acc.x += fabsf(wv.x - rv.x) * rdiv; acc.y += fabsf(wv.y - rv.y) * rdiv;
      acc.z += fabsf(wv.z - rv.z) * rdiv; acc.w += fabsf(wv.w - rv.w) * rdiv;
    }
    if (NT) {
      typedef float f32x4nt __attribute__((ext_vector_type(4)));
      const f32x4nt vv = {acc.x, acc.y, acc.z, acc.w};
      __builtin_nontemporal_store(vv, reinterpret_cast<f32x4nt*>(op + vo[j]));
    } else {
      *reinterpret_cast<float4*>(op + vo[j]) = acc;
    }
  };

  const int d0 = chunk * dch, d1 = min(d0 + dch, a.D);
  const float* __restrict__ dp = a.depth + (size_t)b * a.D * hw + pix;
  geometry(dp[(size_t)d0 * hw], 0);
  float depth_next = dp[(size_t)min(d0 + 1, d1 - 1) * hw];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < P; ++k) issue(st[k], 0, k);
  for (int d = d0; d < d1; ++d) {
    const int par = (d - d0) & 1;
    // taps of plane d + 1 (the last plane of the chunk computes itself again: the P groups requested from it are never used)
    const float depth_n2 = dp[(size_t)min(d + 2, d1 - 1) * hw];
    geometry(depth_next, par ^ 1);
    depth_next = depth_n2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* __restrict__ op = a.out + ((size_t)b * a.D + d) * hw * C;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      finish(st[j % P], j, op);
      if (j + P < NG) issue(st[j % P], par, j + P);
      else issue(st[j % P], par ^ 1, j + P - NG);
    }
    __builtin_amdgcn_wave_barrier();     // plane d + 2's tap stores (next iteration) must not pass this plane's tap loads
  }
}

int launch_cost_volume(const CvArgs& a, void* stream) {
  const int hw = a.h * a.w;
  // pixel patch of a wave: PGRF_CV_TILE_W columns x 64 / PGRF_CV_TILE_W rows where the image divides into such patches and a store
  // instruction still covers whole patch rows (patch width >= voxels per store instruction); otherwise 64 consecutive pixels
  const char* tw_e = getenv("PGRF_CV_TILE_W");        // read per launch (tests flip it inside one process)
  const int tile_w_env = tw_e ? atoi(tw_e) : PGRF_CV_TILE_W_DEFAULT;
  int tws = 6;
  {
    const int vpg = 64 / (a.C / 4 < 64 ? a.C / 4 : 64);      // voxels per wave instruction (C = 32 streams two such groups per iteration)
    for (int s_ = 3; s_ < 6; ++s_)
      if ((1 << s_) == tile_w_env && a.w % (1 << s_) == 0 && a.h % (64 >> s_) == 0 && (1 << s_) >= vpg) tws = s_;
  }
  const int tiles_per_img = tws == 6 ? (hw + 63) / 64 : hw / 64;
  // planes per wave: amortises the per-pixel prologue; keep >= ~32 waves per CU when the volume allows it
  int dch = 8;
  while (dch > 1 && (long)a.B * tiles_per_img * ((a.D + dch - 1) / dch) < PGRF_CV_MINITEMS) dch >>= 1;
  const int nchunk = (a.D + dch - 1) / dch;
  const long nitems = (long)a.B * tiles_per_img * nchunk;
  const unsigned blocks = (unsigned)((nitems + 3) / 4);
  int ex = 0;
  const bool divide = std::frexp(a.div, &ex) != 0.5f;      // not a power of two: 1/div is not exact
  const bool c32 = (a.C == 32), abs_ = (a.cost_type == 0);
  // the pipelined instance (PGRF_CV_PIPE=0: the single-stage kernel; = 1 / 2 / 4 / 8: groups in flight per wave)
  {
    const char* pe = getenv("PGRF_CV_PIPE");
    const int pipe = pe ? atoi(pe) : PGRF_CV_PIPE_DEFAULT;
    if (pipe > 0 && a.dataset == PGRF_DATASET_M3D && c32 && abs_ && !divide && a.n_src <= 2 && hw % 64 == 0) {
      const char* oe = getenv("PGRF_CV_ORDER");
      int order = oe ? atoi(oe) : PGRF_CV_PIPE_ORDER;
      const char* de = getenv("PGRF_CV_DCH");
      // Work-item order and planes per wave decide WHICH 8 KB pieces of the volume are written at the same time, and in the stage
      // (caches full of the previous kernels' dirty lines) that moves the build by +-15 % in no monotone way
      // (profiles/r06_cost_volume_stage_proxy.txt: tools/cv_stage_proxy.py, [1 GB read-modify-write, hypotheses, BUILD, CostRegNet] per step).
      // Shipped: launches that stream their stores (volumes >= 128 MiB) take chunk-major order with 16 planes per wave, smaller ones
      // tile-major order with 4 -- against the round-5 kernel in the same proxy (profiles/r06_cost_volume_stage_stores.txt): 128x256x64
      // two views 72 -> 68 us per view, one view 95 -> 66, 64x128x64 two views 25.8 -> 20.4, one view 28 -> 25.
      if (!oe) order = a.stream_stores ? 1 : 0;
      int pd = de ? atoi(de) : (a.stream_stores ? 16 : PGRF_CV_PIPE_DCH);
      if (pd < 1) pd = 1;
      if (pd > a.D) pd = a.D;
      if (!de) while (pd > 1 && (long)a.B * tiles_per_img * ((a.D + pd - 1) / pd) < 2048) pd >>= 1;
      const int pn = (a.D + pd - 1) / pd;
      const long pitems = (long)a.B * tiles_per_img * pn;
      const unsigned pblocks = (unsigned)((pitems + 3) / 4);
#define PGRF_CVP(N_, P_) do { \
        if (a.stream_stores) hipLaunchKernelGGL((cost_volume_pipe_kernel<N_, P_, true>), dim3(pblocks), dim3(256), 0, (hipStream_t)stream, a, pd, pn, tiles_per_img, tws, order); \
        else hipLaunchKernelGGL((cost_volume_pipe_kernel<N_, P_, false>), dim3(pblocks), dim3(256), 0, (hipStream_t)stream, a, pd, pn, tiles_per_img, tws, order); } while (0)
      if (a.n_src == 1) { if (pipe >= 8) PGRF_CVP(1, 8); else if (pipe >= 4) PGRF_CVP(1, 4); else if (pipe >= 2) PGRF_CVP(1, 2); else PGRF_CVP(1, 1); }
      else { if (pe && pipe >= 4) PGRF_CVP(2, 4); else if (pe && pipe >= 2) PGRF_CVP(2, 2); else PGRF_CVP(2, 1); }   // (two source views: 40 registers per group in flight -- one group unless asked)
#undef PGRF_CVP
      PGRF_CHECK_LAUNCH("cost_volume (pipelined)");
      return PGRF_OK;
    }
  }
#define PGRF_CV5(GEN_, N_, C32_, ABS_, DIV_) \
  hipLaunchKernelGGL((cost_volume_kernel<GEN_, N_, C32_, ABS_, DIV_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, dch, nchunk, tiles_per_img, tws)
  // the shipped convention gets every specialisation; the other conventions share the run-time-switched kernel
#define PGRF_CV_M3D(N_) do { \
    if (c32 && abs_ && !divide) PGRF_CV5(false, N_, true, true, false); \
    else if (c32 && !divide) PGRF_CV5(false, N_, true, false, false); \
    else if (!divide) PGRF_CV5(false, N_, false, false, false); \
    else PGRF_CV5(false, N_, false, false, true); } while (0)
#define PGRF_CV_GEN(N_) do { if (!divide) PGRF_CV5(true, N_, false, false, false); else PGRF_CV5(true, N_, false, false, true); } while (0)
  if (a.dataset == PGRF_DATASET_M3D) {
    if (a.n_src == 1) PGRF_CV_M3D(1); else if (a.n_src == 2) PGRF_CV_M3D(2); else if (a.n_src == 3) PGRF_CV_M3D(3); else PGRF_CV_M3D(4);
  } else {
    if (a.n_src == 1) PGRF_CV_GEN(1); else if (a.n_src == 2) PGRF_CV_GEN(2); else if (a.n_src == 3) PGRF_CV_GEN(3); else PGRF_CV_GEN(4);
  }
#undef PGRF_CV5
#undef PGRF_CV_M3D
#undef PGRF_CV_GEN
  PGRF_CHECK_LAUNCH("cost_volume");
  return PGRF_OK;
}

}  // namespace pgrf

using namespace pgrf;

extern "C" {

int pgrf_depth_hypotheses(const float* ref_mu, int B, int hw, const float* guided_offsets_host, int n_guided,
                          float min_depth, float max_depth, const float* uniform_centres, int n_uniform,
                          float* depth_volume, void* stream) {
  if (!ref_mu || !depth_volume || (n_guided > 0 && !guided_offsets_host) || (n_uniform > 0 && !uniform_centres))
    return set_error(PGRF_ERR_ARG, "pgrf_depth_hypotheses: null pointer");
  if (n_guided < 0 || n_guided > kMaxGuided || n_uniform < 0 || n_guided + n_uniform < 1)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_depth_hypotheses: n_guided=%d (max %d), n_uniform=%d", n_guided, kMaxGuided, n_uniform);
  if (B < 0 || hw < 1) return set_error(PGRF_ERR_ARG, "pgrf_depth_hypotheses: bad shape B=%d hw=%d", B, hw);
  if (B == 0) return PGRF_OK;
  return launch_depth_hypotheses(ref_mu, B, hw, guided_offsets_host, n_guided, min_depth, max_depth, uniform_centres,
                                 n_uniform, depth_volume, stream);
}

int pgrf_cost_volume_erp(const float* feats, const float* depth_volume, const float* rots, const float* trans, int B,
                         int S, int h, int w, int C, int D, int ref_idx, const int* src_idx_host, int n_src,
                         float src_div, int cost_type, int dataset, float* out, void* stream) {
  if (dataset < PGRF_DATASET_M3D || dataset > PGRF_DATASET_COFFEEAREA)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_cost_volume_erp: dataset convention %d unknown", dataset);
  if (!feats || !depth_volume || !rots || !trans || !out || !src_idx_host)
    return set_error(PGRF_ERR_ARG, "pgrf_cost_volume_erp: null pointer");
  if (B < 0 || S < 2 || h < 2 || w < 2 || D < 1) return set_error(PGRF_ERR_ARG, "pgrf_cost_volume_erp: bad shape");
  if (C < 4 || C > 256 || (C & (C - 1)) != 0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_cost_volume_erp: C=%d unsupported (power of two in [4,256])", C);
  if (n_src < 1 || n_src > kMaxSrc) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_cost_volume_erp: n_src=%d not in [1,%d]", n_src, kMaxSrc);
  if (ref_idx < 0 || ref_idx >= S) return set_error(PGRF_ERR_ARG, "pgrf_cost_volume_erp: ref_idx out of range");
  if (cost_type < 0 || cost_type > 2) return set_error(PGRF_ERR_ARG, "pgrf_cost_volume_erp: unknown cost type %d", cost_type);
  if (!(src_div > 0.f)) return set_error(PGRF_ERR_ARG, "pgrf_cost_volume_erp: src_div must be > 0");
  if ((double)h * w * C >= 2147483648.0) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_cost_volume_erp: a view of >= 2^31 elements");
  if (B == 0) return PGRF_OK;
  CvArgs a;
  a.feats = feats; a.depth = depth_volume; a.rots = rots; a.trans = trans; a.out = out;
  a.B = B; a.S = S; a.h = h; a.w = w; a.C = C; a.D = D; a.ref_idx = ref_idx; a.n_src = n_src; a.div = src_div;
  a.cost_type = cost_type; a.dataset = dataset;
  // Non-temporal stores of the volume.  Round 3 measured them in a clean loop (kernel alone 75-76 -> 62 us at 128x256x64, stage unchanged:
  // the regulariser's first convolution then fetches the 268 MB from HBM) and kept them for batches beyond the 256 MB last-level cache
  // (>= 384 MB).  Round 6 re-measured IN CONTEXT -- every step [1 GB read-modify-write (what a render pass leaves in the caches),
  // hypotheses, build, regulariser], tools/cv_stage_nt.py, profiles/r06_cost_volume_stage_stores.txt: with the caches full of other
  // kernels' dirty lines a plain-store build must evict as it allocates, and ONE 268 MB volume builds in 88 us with plain stores, 66 us
  // streamed, build + CostRegNet 834 -> 819 us (UNet2: equal); two 64x128x64 volumes (128 MiB) 23.7 -> 20.4 us, totals equal; one
  // (64 MiB) 25.1 -> 26.3 us.  So volumes stream from 128 MiB on.  PGRF_CV_STREAM_STORES=0 / 1 forces either way.
  {
    const char* e = getenv("PGRF_CV_STREAM_STORES");
    const double out_bytes = (double)B * D * h * w * C * 4.0;
    a.stream_stores = e ? (atoi(e) != 0) : (out_bytes >= 128.0 * 1048576.0);
  }
  for (int i = 0; i < kMaxSrc; ++i) {
    a.src_idx[i] = i < n_src ? src_idx_host[i] : 0;
    if (i < n_src && (src_idx_host[i] < 0 || src_idx_host[i] >= S)) return set_error(PGRF_ERR_ARG, "pgrf_cost_volume_erp: src_idx out of range");
  }
  return launch_cost_volume(a, stream);
}

}  // extern "C"
