// 360-degree spherical cost volume for gfx950.
//
// HBM-write bound: the volume [B,D,h,w,C] is written exactly once, channel-last, 16 B per lane, a
// whole wave writing 1 KiB of consecutive bytes; the source feature maps are tiny (h*w*C*4 B per
// view, L2 resident) and every bilinear tap is one aligned 16-B load per lane.
// Geometry (back-projection, two 3x3 transforms, atan2/acos, tap setup) is computed ONCE per voxel
// by one lane and handed to the C/4 lanes that move that voxel's channels through wave shuffles,
// instead of being recomputed per channel group.
#include <hip/hip_runtime.h>

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

// pipeline3_model.py:723-733 (clamp(ref_mu + k*fixed_sigma)), :774-815 (linspace planes, concat, sort)
__global__ void depth_hypotheses_kernel(const HypArgs a) {
  const size_t total = (size_t)a.B * a.hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / a.hw, pix = i % a.hw;
    const float mu = a.ref_mu[i];
    float gv[kMaxGuided];
#pragma unroll
    for (int k = 0; k < kMaxGuided; ++k) gv[k] = (k < a.n_guided) ? fminf(fmaxf(mu + a.offs[k], a.lo), a.hi) : 3.0e38f;
    // insertion sort of the (few) guided planes
#pragma unroll
    for (int k = 1; k < kMaxGuided; ++k)
#pragma unroll
      for (int j = k; j > 0; --j)
        if (gv[j] < gv[j - 1]) { const float t = gv[j]; gv[j] = gv[j - 1]; gv[j - 1] = t; }
    // merge with the ascending uniform planes
    float* o = a.out + b * (size_t)(a.n_guided + a.n_uniform) * a.hw + pix;
    int gi = 0, ui = 0;
    const int D = a.n_guided + a.n_uniform;
    for (int d = 0; d < D; ++d) {
      float gcur = 3.0e38f;
#pragma unroll
      for (int k = 0; k < kMaxGuided; ++k) gcur = (k == gi) ? gv[k] : gcur;
      const float ucur = (ui < a.n_uniform) ? a.uniform[ui] : 3.0e38f;
      const bool take_g = (gi < a.n_guided) && (gcur <= ucur);
      o[(size_t)d * a.hw] = take_g ? gcur : ucur;
      gi += take_g ? 1 : 0;
      ui += take_g ? 0 : 1;
    }
  }
}

int launch_depth_hypotheses(const float* ref_mu, int B, int hw, const float* offs_host, int n_guided, float lo, float hi,
                            const float* uniform_dev, int n_uniform, float* out, void* stream) {
  HypArgs a;
  a.ref_mu = ref_mu; a.uniform = uniform_dev; a.out = out; a.n_guided = n_guided; a.n_uniform = n_uniform;
  a.B = B; a.hw = hw; a.lo = lo; a.hi = hi;
  for (int k = 0; k < kMaxGuided; ++k) a.offs[k] = (k < n_guided) ? offs_host[k] : 0.f;
  const size_t total = (size_t)B * hw;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(depth_hypotheses_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, a);
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
};

struct Taps { int o00, o01, o10, o11; float w00, w01, w10, w11; };

__device__ __forceinline__ void inv3(const float* m, float* o) {   // adjugate / determinant
  const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const float A = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
  const float det = a * A + b * Bc + c * Cc;
  const float r = 1.f / det;
  o[0] = A * r; o[1] = -(b * i - c * h) * r; o[2] = (b * f - c * e) * r;
  o[3] = Bc * r; o[4] = (a * i - c * g) * r; o[5] = -(a * f - c * d) * r;
  o[6] = Cc * r; o[7] = -(a * h - b * g) * r; o[8] = (a * e - b * d) * r;
}

// spherical_cost_volume.py:135-198 + my_torch_helpers.py:34-38,62-96 for one (voxel, source view)
template <bool GEN>
__device__ __forceinline__ Taps warp_taps(const CvArgs& a, int b, int src, int y, int x, float depth) {
  const float PI_F = 3.14159265358979323846f, HALF_PI_F = 1.57079632679489661923f, TWO_PI_F = 6.28318530717958647692f;
  const double PI_D = 3.14159265358979323846;
  // pixel centre -> angles -> unit direction, per dataset (spherical_cost_volume.py:281-296, my_torch_helpers.py:34-56);
  // Python folds the scalar factors in double before they meet the float tensors
  float ux, uy, uz;
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
  const float mx = depth * ux, my = depth * uy, mz = depth * uz;
  const float* Rr = a.rots + ((size_t)b * a.S + a.ref_idx) * 9;
  const float* tr = a.trans + ((size_t)b * a.S + a.ref_idx) * 3;
  float Ri[9];
  inv3(Rr, Ri);
  const float dx = mx - tr[0], dy = my - tr[1], dz = mz - tr[2];
  const float wx = Ri[0] * dx + Ri[1] * dy + Ri[2] * dz;
  const float wy = Ri[3] * dx + Ri[4] * dy + Ri[5] * dz;
  const float wz = Ri[6] * dx + Ri[7] * dy + Ri[8] * dz;
  const float* Rs = a.rots + ((size_t)b * a.S + src) * 9;
  const float* ts = a.trans + ((size_t)b * a.S + src) * 3;
  const float cx = Rs[0] * wx + Rs[1] * wy + Rs[2] * wz + ts[0];
  const float cy = Rs[3] * wx + Rs[4] * wy + Rs[5] * wz + ts[1];
  const float cz = Rs[6] * wx + Rs[7] * wy + Rs[8] * wz + ts[2];
  const float radius = sqrtf(cx * cx + cy * cy + cz * cz);
  // cartesian -> (theta, phi) -> grid coordinates u, v in [-1,1] (my_torch_helpers.py:62-120,
  // spherical_cost_volume.py:153-190)
  float u, v;
  if (!GEN || a.dataset == PGRF_DATASET_M3D) {
    const float th = atan2f(cz, cx);
    const float yor = cy / radius;
    const float LIN = 0.17453292519943295f, COSD = 0.984807753012208f, ONE_M_COSD = 0.01519224698779198f;
    float ph;
    if (fabsf(yor) < COSD) ph = acosf(yor);
    else if (cy >= 0.f) ph = LIN * (1.f - yor) / ONE_M_COSD;
    else ph = PI_F - LIN * (yor + 1.f) / ONE_M_COSD;
    u = fmodf(th + HALF_PI_F + TWO_PI_F, TWO_PI_F) / PI_F - 1.f;
    v = 2.f * ph / PI_F - 1.f;
  } else if (a.dataset == PGRF_DATASET_REPLICA_TEST) {
    const float th = atan2f(cx, cz);
    const float ph = -asinf(cz / radius);          // the reference takes z here (my_torch_helpers.py:99)
    u = fmodf(th + PI_F + TWO_PI_F, TWO_PI_F) / PI_F - 1.f;
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
  t.o00 = cy0 * a.w + cx0; t.o01 = cy0 * a.w + cx1; t.o10 = cy1 * a.w + cx0; t.o11 = cy1 * a.w + cx1;
  t.w00 = (vx0 && vy0) ? (1.f - ax) * (1.f - ay) : 0.f;
  t.w01 = (vx1 && vy0) ? ax * (1.f - ay) : 0.f;
  t.w10 = (vx0 && vy1) ? (1.f - ax) * ay : 0.f;
  t.w11 = (vx1 && vy1) ? ax * ay : 0.f;
  return t;
}

// GEN = false: m3d convention only (the other conventions compiled in cost 113 -> 176 us at 128x256x64)
template <bool GEN, int NSRC>      // NSRC: source views accumulated (tap setups kept in registers: 8 per view)
__global__ void __launch_bounds__(256) cost_volume_kernel(const CvArgs a) {
  const int lane = threadIdx.x & 63;
  const int lpv = a.C / 4;                  // lanes per voxel (power of two <= 64)
  const int vpi = 64 / lpv;                 // voxels per inner iteration
  const size_t hw = (size_t)a.h * a.w;
  const size_t nvox = (size_t)a.B * a.D * hw;
  const size_t wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t base = wave_global * 64; base < nvox; base += nwaves * 64) {
    // lane i owns the geometry of voxel base+i
    const size_t vox = base + lane < nvox ? base + lane : nvox - 1;
    const int x = (int)(vox % a.w);
    const int y = (int)((vox / a.w) % a.h);
    const int b = (int)(vox / (hw * a.D));
    const float depth = a.depth[vox];          // [B,D,h,w] has the same linear voxel order as the output
    const int sub = lane / lpv, cg = lane % lpv;
    // tap setup once per (voxel, source view), kept in registers (static indexing only)
    Taps tp[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s) tp[s] = warp_taps<GEN>(a, b, a.src_idx[s], y, x, depth);
    const int mypix = (int)(vox % hw);
    for (int it = 0; it < 64 / vpi; ++it) {
      const int owner = it * vpi + sub;        // lane holding this voxel's geometry
      const size_t myvox = base + owner;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const int ob = __shfl(b, owner, 64);
      const int pix = __shfl(mypix, owner, 64);
      const float4 rv = *reinterpret_cast<const float4*>(a.feats + (((size_t)ob * a.S + a.ref_idx) * hw + pix) * a.C + cg * 4);
#pragma unroll
      for (int s = 0; s < NSRC; ++s) {
        const Taps& t = tp[s];
        const int o00 = __shfl(t.o00, owner, 64), o01 = __shfl(t.o01, owner, 64);
        const int o10 = __shfl(t.o10, owner, 64), o11 = __shfl(t.o11, owner, 64);
        const float w00 = __shfl(t.w00, owner, 64), w01 = __shfl(t.w01, owner, 64);
        const float w10 = __shfl(t.w10, owner, 64), w11 = __shfl(t.w11, owner, 64);
        const float* sb = a.feats + ((size_t)ob * a.S + a.src_idx[s]) * hw * a.C + cg * 4;
        const float4 p00 = *reinterpret_cast<const float4*>(sb + (size_t)o00 * a.C);
        const float4 p01 = *reinterpret_cast<const float4*>(sb + (size_t)o01 * a.C);
        const float4 p10 = *reinterpret_cast<const float4*>(sb + (size_t)o10 * a.C);
        const float4 p11 = *reinterpret_cast<const float4*>(sb + (size_t)o11 * a.C);
        float4 wv;
        wv.x = p00.x * w00 + p01.x * w01 + p10.x * w10 + p11.x * w11;
        wv.y = p00.y * w00 + p01.y * w01 + p10.y * w10 + p11.y * w11;
        wv.z = p00.z * w00 + p01.z * w01 + p10.z * w10 + p11.z * w11;
        wv.w = p00.w * w00 + p01.w * w01 + p10.w * w10 + p11.w * w11;
        float4 c;
        if (a.cost_type == 0) { c.x = fabsf(wv.x - rv.x); c.y = fabsf(wv.y - rv.y); c.z = fabsf(wv.z - rv.z); c.w = fabsf(wv.w - rv.w); }
        else if (a.cost_type == 1) { c.x = wv.x * rv.x; c.y = wv.y * rv.y; c.z = wv.z * rv.z; c.w = wv.w * rv.w; }
        else c = wv;
        acc.x += c.x / a.div; acc.y += c.y / a.div; acc.z += c.z / a.div; acc.w += c.w / a.div;
      }
      if (myvox < nvox) *reinterpret_cast<float4*>(a.out + myvox * a.C + cg * 4) = acc;
    }
  }
}

int launch_cost_volume(const CvArgs& a, void* stream) {
  const size_t nvox = (size_t)a.B * a.D * a.h * a.w;
  const size_t waves = (nvox + 63) / 64;
  size_t blocks = (waves + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
#define PGRF_CV(GEN_, N_) hipLaunchKernelGGL((cost_volume_kernel<GEN_, N_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a)
  if (a.dataset == PGRF_DATASET_M3D) {
    if (a.n_src == 1) PGRF_CV(false, 1); else if (a.n_src == 2) PGRF_CV(false, 2); else if (a.n_src == 3) PGRF_CV(false, 3); else PGRF_CV(false, 4);
  } else {
    if (a.n_src == 1) PGRF_CV(true, 1); else if (a.n_src == 2) PGRF_CV(true, 2); else if (a.n_src == 3) PGRF_CV(true, 3); else PGRF_CV(true, 4);
  }
#undef PGRF_CV
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
  if (B == 0) return PGRF_OK;
  CvArgs a;
  a.feats = feats; a.depth = depth_volume; a.rots = rots; a.trans = trans; a.out = out;
  a.B = B; a.S = S; a.h = h; a.w = w; a.C = C; a.D = D; a.ref_idx = ref_idx; a.n_src = n_src; a.div = src_div;
  a.cost_type = cost_type; a.dataset = dataset;
  for (int i = 0; i < kMaxSrc; ++i) {
    a.src_idx[i] = i < n_src ? src_idx_host[i] : 0;
    if (i < n_src && (src_idx_host[i] < 0 || src_idx_host[i] >= S)) return set_error(PGRF_ERR_ARG, "pgrf_cost_volume_erp: src_idx out of range");
  }
  return launch_cost_volume(a, stream);
}

}  // extern "C"
