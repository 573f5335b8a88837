// Small HBM-bound kernels around the fused render pass: map re-layout, depth samplers, c2f merge.
#include "pgrf_experiments.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

#include "pgrf_internal.h"
#include "ray_dir.h"
#include "pgrf_layout.h"

namespace pgrf {

#define PGRF_CHECK_LAUNCH(what)                                                                       \
  do {                                                                                                \
    const hipError_t e__ = hipGetLastError();                                                         \
    if (e__ != hipSuccess) return set_error(PGRF_ERR_LAUNCH, what " launch failed: %s", hipGetErrorString(e__)); \
  } while (0)

// ------------------------------------------------------------------------------------------
// NCHW -> NHWC(Cpad).  One thread per (pixel, 4-channel group); reads are coalesced along x for
// each channel, writes are one float4 per thread.
__global__ void pack_map_kernel(const float* __restrict__ src, int n, int C, int hw, int Cpad, float* __restrict__ dst) {
  const int groups = Cpad / 4;
  const size_t total = (size_t)n * hw * groups;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i % hw;
    const size_t rest = i / hw;
    const int gq = (int)(rest % groups);
    const size_t img = rest / groups;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = gq * 4 + k;
      v[k] = (c < C) ? src[(img * C + c) * hw + pix] : 0.f;
    }
    *reinterpret_cast<float4*>(dst + (img * hw + pix) * Cpad + gq * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

int launch_pack_map(const float* src, int n, int C, int h, int w, float* dst, void* stream) {
  const int Cpad = (C + 3) / 4 * 4;
  const size_t total = (size_t)n * h * w * (Cpad / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_map_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, src, n, C, h * w, Cpad, dst);
  PGRF_CHECK_LAUNCH("pack_map");
  return PGRF_OK;
}

// ------------------------------------------------------------------------------------------
// Per-pixel projection of the channel-last ray_feats maps through every layer that consumes the gathered
// ray feature linearly (pgrf_layout.h: OFF_PRJ_W).  out[pix][c] = b[c] + <W[c,:], f[pix,:]>, c < nch.
// Tiny (V*hr*wr pixels x 160 x 32 MACs); one thread per (pixel, 4 outputs).
__global__ void project_ray_feats_kernel(const float* __restrict__ rf, const float* __restrict__ blob, int npix, int nch,
                                         float* __restrict__ out) {
  const int groups = nch / 4;
  const long long total = (long long)npix * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int gq = (int)(i % groups);
    const long long pix = i / groups;
    const float4* f = reinterpret_cast<const float4*>(rf + pix * 32);
    float4 fv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) fv[k] = f[k];
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = gq * 4 + j;
      const float4* w = reinterpret_cast<const float4*>(blob + OFF_PRJ_W + c * 32);
      float s = blob[OFF_PRJ_B + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 wv = w[k];
        s = fmaf(wv.x, fv[k].x, s); s = fmaf(wv.y, fv[k].y, s); s = fmaf(wv.z, fv[k].z, s); s = fmaf(wv.w, fv[k].w, s);
      }
      o[j] = s;
    }
    *reinterpret_cast<float4*>(out + pix * kPrjCh + gq * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

int launch_project_ray_feats(const float* rf, const float* blob, int npix, int use_vis, float* out, void* stream) {
  const int nch = use_vis ? kPrjCh : kPrjCh - 32;
  const long long total = (long long)npix * (nch / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(project_ray_feats_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, rf, blob, npix, nch, out);
  PGRF_CHECK_LAUNCH("project_ray_feats");
  return PGRF_OK;
}

// ------------------------------------------------------------------------------------------
// sample_depth (network/render_ops.py:292-339).  Eval: identical for every ray.  Training (jitter != null): the interior
// samples are moved by (r - 0.5) * 0.999 bins, r = the caller's uniform draws [rn, dn-2] (render_ops.py:310-311,329-330);
// the end points stay exact.
__global__ void coarse_depth_kernel(float near, float far, int use_disp, int rn, int dn, const float* __restrict__ jitter,
                                    float* __restrict__ depth) {
  const size_t total = (size_t)rn * dn;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % dn);
    float val = (float)k;
    if (jitter && k > 0 && k < dn - 1) val = val + (jitter[(i / dn) * (size_t)(dn - 2) + (k - 1)] - 0.5f) * 0.999f;
    float d;
    if (use_disp) {
      const float diff = 1.f / far - 1.f / near;
      const float interval = diff / (float)(dn - 1);
      const float tick = (k == 0) ? 0.f : (k == dn - 1 ? diff : interval * val);
      d = 1.f / (1.f / near + tick);
    } else {
      const float diff = far - near;
      const float interval = diff / (float)(dn - 1);
      const float tick = (k == 0) ? 0.f : (k == dn - 1 ? diff : interval * val);
      d = near + tick;
    }
    depth[i] = d;
  }
}

int launch_coarse_depth(float near, float far, int use_disp, int rn, int dn, const float* jitter, float* depth, void* stream) {
  const size_t total = (size_t)rn * dn;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(coarse_depth_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, near, far, use_disp, rn, dn, jitter, depth);
  PGRF_CHECK_LAUNCH("coarse_depth");
  return PGRF_OK;
}

// ------------------------------------------------------------------------------------------
// sample_fine_depth (render_ops.py:413-473) + sort (renderer.py:469-472).  One wave per ray,
// 4 rays per workgroup; the cdf is accumulated serially in index order (like torch.cumsum on
// the CPU path the golden vectors come from), searchsorted(right=True) is a count of cdf <= u.
constexpr int kMaxS = 128;   // max dn, max fdn

// Total order for the rank sorts below: "element (o, index k) comes before element (v, index j)".  NaNs sort last
// (as in torch.sort) and ties go to the lower index, so every rank 0..n-1 is written exactly once even when a NaN
// reaches the sampler (a partial order would leave output slots of the caller's uninitialised buffer unwritten).
__device__ __forceinline__ bool sorts_before(float o, int k, float v, int j) {
  const bool on = (o != o), vn = (v != v);
  if (on || vn) return (!on && vn) || (on == vn && k < j);
  return o < v || (o == v && k < j);
}

// Arithmetic shared by the two sampler kernels below (one expression each, so that both compile to the same operations).
__device__ __forceinline__ float fd_norm_depth(float z, float n_, float f_, int use_disp) {
  return use_disp ? (-1.f / z - n_) / (f_ - n_) : z;
}
__device__ __forceinline__ float fd_sample(float u, float c0, float c1, float b0, float b1, float n_, float f_, int use_disp) {
  float denom = c1 - c0;
  if (denom < 1e-5f) denom = 1.f;
  const float t = (u - c0) / denom;
  float fd = b0 + t * (b1 - b0);
  if (use_disp) {
    fd = fd * (f_ - n_) + n_;
    fd = -1.f / fd;
  }
  return fd;
}

__global__ void __launch_bounds__(256) fine_depth_kernel(const float* __restrict__ depth, const float* __restrict__ hit,
                                                         float qnear, float qfar, int use_disp, int use_all, int rn,
                                                         int dn, int fdn, const float* __restrict__ urand,
                                                         float* __restrict__ out) {
  __shared__ float s_pdf[4][kMaxS];
  __shared__ float s_cdf[4][kMaxS + 1];
  __shared__ float s_ctr[4][kMaxS + 1];
  __shared__ float s_d[4][kMaxS];
  __shared__ float s_val[4][2 * kMaxS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  const bool ok = ray < rn;
  const int rayc = ok ? ray : rn - 1;
  const float n_ = -1.f / qnear, f_ = -1.f / qfar;
  float part = 0.f;
  for (int i = lane; i < dn; i += 64) {
    const float z = depth[(size_t)rayc * dn + i];
    const float d = fd_norm_depth(z, n_, f_, use_disp);
    const float hp = hit[(size_t)rayc * dn + i] + 1e-5f;
    s_d[wv][i] = d;
    s_pdf[wv][i] = hp;
    part += hp;
    if (use_all) s_val[wv][i] = z;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
  __syncthreads();
  for (int i = lane; i < dn; i += 64) s_pdf[wv][i] = s_pdf[wv][i] / part;
  __syncthreads();
  for (int k = lane; k <= dn; k += 64) {
    float c = 0.f;
    for (int j = 0; j < k; ++j) c += s_pdf[wv][j];      // cdf[k] = pdf[0] + ... + pdf[k-1], left to right
    s_cdf[wv][k] = c;
    float ctr;
    if (k == 0) ctr = s_d[wv][0];
    else if (k == dn) ctr = s_d[wv][dn - 1];
    else ctr = (s_d[wv][k] + s_d[wv][k - 1]) / 2.f;
    s_ctr[wv][k] = ctr;
  }
  __syncthreads();
  const float interval = (float)(1.0 / (double)fdn);
  const float half_iv = (float)(0.5 * (1.0 / (double)fdn));
  const int base = use_all ? dn : 0;
  for (int j = lane; j < fdn; j += 64) {
    // eval: bin centres of [0,1); training: the caller's uniform draws [rn, fdn] (render_ops.py:443-449)
    const float u = urand ? urand[(size_t)rayc * fdn + j] : half_iv + (float)j * interval;
    int inds = 0;
    for (int k = 0; k <= dn; ++k) inds += (s_cdf[wv][k] <= u) ? 1 : 0;
    const int below = max(inds - 1, 0), above = min(inds, dn);
    const float c0 = s_cdf[wv][below], c1 = s_cdf[wv][above];
    const float b0 = s_ctr[wv][below], b1 = s_ctr[wv][above];
    s_val[wv][base + j] = fd_sample(u, c0, c1, b0, b1, n_, f_, use_disp);
  }
  __syncthreads();
  // stable rank sort of the (nearly sorted) values
  const int total = base + fdn;
  for (int j = lane; j < total; j += 64) {
    const float v = s_val[wv][j];
    int rank = 0;
    for (int k = 0; k < total; ++k) {
      const float o = s_val[wv][k];
      rank += sorts_before(o, k, v, j) ? 1 : 0;
    }
    if (ok) out[(size_t)ray * total + rank] = v;
  }
}

// The same sampler with ONE LANE PER RAY (dn == fdn in {32, 64}: the shipped 64 + 64 and 32 + 32 configurations).  In the
// wave-per-ray kernel above every ray pays three O(dn) loops of LDS round trips in all 64 lanes (872 us for a 512x1024 frame,
// 1 % of a coarse + fine frame; its bytes would take 50 us at the HBM rate).  Here a wave takes 64 rays: the two input tiles
// (64 rays x DN floats, contiguous in memory) are loaded coalesced and transposed through LDS (row stride 2*DN+1: a whole-wave
// access of one column is conflict-free), each lane then runs the reference's serial algorithm on its own row -- the
// normalising sum as the wave kernel's butterfly tree, the cdf left to right, searchsorted(right=True) as a branch-free upper
// bound (the cdf is a running sum of positives, hence monotone: same count as the comparison loop), four searches in flight
// -- sorts its row by insertion (the values arrive almost sorted; NaNs last, every slot written) and the tile is written
// back coalesced.  Bit-identical to the wave kernel (tests/test_gpu_render.py).
// SHARED: every ray has the SAME coarse depths (the eval-mode coarse pass of pgrf_render_rays: no jitter), and they are not
// appended to the output -- the normalised-depth half of the tile then is one 256-byte row per wave instead of 16 KB, and
// a wave's LDS block shrinks from 33 to 17 KB (9 instead of 4 waves per CU).
template <int DN, bool SHARED>
__global__ void __launch_bounds__(64) fine_depth_lane_kernel(const float* __restrict__ depth, const float* __restrict__ hit,
                                                             float qnear, float qfar, int use_disp, int use_all, int rn,
                                                             const float* __restrict__ urand, float* __restrict__ out) {
  constexpr int ROW = (SHARED ? DN : 2 * DN) + 1;   // [0,DN): normalised depth d, later the coarse z (use_all) -- absent when SHARED;
  constexpr int HP0 = SHARED ? 0 : DN;              // [HP0,HP0+DN): hit -> cdf[1..DN] -> fine z
  __shared__ float s[64 * ROW];
  __shared__ float sd[SHARED ? DN : 1];
  const int lane = threadIdx.x;
  const size_t ray0 = (size_t)blockIdx.x * 64;
  const float n_ = -1.f / qnear, f_ = -1.f / qfar;
  constexpr int LPR = DN / 4;         // lanes per ray row in a tile load (float4 each)
  auto load_tile = [&](const float* __restrict__ src, int col0) {
#pragma unroll
    for (int it = 0; it < DN / 4; ++it) {
      const int e = it * 64 + lane, r = e / LPR, i = (e % LPR) * 4;
      const size_t gr = min(ray0 + (size_t)r, (size_t)rn - 1);
      const float4 v = *reinterpret_cast<const float4*>(src + gr * DN + i);
      float* q = s + r * ROW + col0 + i;
      q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    }
  };
  if constexpr (SHARED) {
    if (lane < DN) sd[lane] = fd_norm_depth(depth[lane], n_, f_, use_disp);
  } else {
    load_tile(depth, 0);
  }
  load_tile(hit, HP0);
  __syncthreads();
  float* row = s + lane * ROW;
  float t[DN];
#pragma unroll
  for (int i = 0; i < DN; ++i) {
    if constexpr (!SHARED) row[i] = fd_norm_depth(row[i], n_, f_, use_disp);
    t[i] = row[HP0 + i] + 1e-5f;
  }
  {   // the wave kernel's sum: lanes >= DN contribute 0 (x + 0 = x), then xor-butterfly = pairs (i, i + half)
    float u[DN];
#pragma unroll
    for (int i = 0; i < DN; ++i) u[i] = t[i];
#pragma unroll
    for (int half = DN / 2; half >= 1; half >>= 1)
#pragma unroll
      for (int i = 0; i < half; ++i) u[i] = u[i] + u[i + half];
    const float total = u[0];
    float c = 0.f;
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      c += t[k] / total;
      row[HP0 + k] = c;               // cdf[k + 1]; cdf[0] = 0 is implicit
    }
  }
  const float interval = (float)(1.0 / (double)DN);
  const float half_iv = (float)(0.5 * (1.0 / (double)DN));
  const size_t ray = min(ray0 + (size_t)lane, (size_t)rn - 1);
  const float* cdf = row + HP0;       // cdf[k] for k >= 1 at cdf[k - 1]
  const float* dd = SHARED ? sd : row;
  auto ctr = [&](int k) { return k == 0 ? dd[0] : (k == DN ? dd[DN - 1] : (dd[k] + dd[k - 1]) / 2.f); };
#pragma unroll
  for (int j0 = 0; j0 < DN; j0 += 4) {
    float u[4];
    int cnt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u[q] = urand ? urand[ray * DN + j0 + q] : half_iv + (float)(j0 + q) * interval;
      cnt[q] = 0;
    }
#pragma unroll
    for (int step = DN / 2; step >= 1; step >>= 1)
#pragma unroll
      for (int q = 0; q < 4; ++q) cnt[q] += (cdf[cnt[q] + step - 1] <= u[q]) ? step : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      cnt[q] += (cdf[cnt[q]] <= u[q]) ? 1 : 0;           // number of cdf[1..DN] <= u; with cdf[0] = 0 <= u: inds = cnt + 1
      const int below = cnt[q], above = min(cnt[q] + 1, DN);
      const float c0 = below == 0 ? 0.f : cdf[below - 1], c1 = cdf[above - 1];
      t[j0 + q] = fd_sample(u[q], c0, c1, ctr(below), ctr(above), n_, f_, use_disp);
    }
  }
#pragma unroll
  for (int j = 0; j < DN; ++j) row[HP0 + j] = t[j];
  if constexpr (!SHARED) {
    if (use_all) {
      __syncthreads();
      load_tile(depth, 0);
      __syncthreads();
    }
  }
  const int base = (use_all && !SHARED) ? 0 : HP0, total = (use_all && !SHARED) ? 2 * DN : DN;
  float* a = row + base;
  for (int j = 1; j < total; ++j) {
    const float v = a[j];
    int i = j - 1;
    float o = a[i];
    // o sorts after v: o > v, or o is NaN and v is not
    while (!(o <= v) && v == v) {
      a[i + 1] = o;
      if (--i < 0) break;
      o = a[i];
    }
    a[i + 1] = v;
  }
  __syncthreads();
  const int lpr = total / 4;
  for (int it = 0; it < total / 4; ++it) {
    const int e = it * 64 + lane, r = e / lpr, i = (e % lpr) * 4;
    if (ray0 + r < (size_t)rn) {
      const float* q = s + r * ROW + base + i;
      *reinterpret_cast<float4*>(out + (ray0 + r) * total + i) = make_float4(q[0], q[1], q[2], q[3]);
    }
  }
}

int launch_fine_depth(const float* depth, const float* hit, float qnear, float qfar, int use_disp, int use_all, int rn,
                      int dn, int fdn, const float* urand, float* out, void* stream, int rows_equal) {
  const char* knob = getenv("PGRF_FINE_DEPTH");          // "wave" / "lane" / "noshare": one kernel for every shape (A/B, tests)
  const bool wave_only = knob && !strcmp(knob, "wave");
  const bool aligned = (((uintptr_t)depth | (uintptr_t)hit | (uintptr_t)out) & 15) == 0;
  // The lane-per-ray kernel is a throughput form: 64 rays per wave, each lane a long serial run (45-225 us whatever the batch while its
  // waves fit the chip at once: 225 us with per-ray random draws, the training call).  A training batch of 2 048 rays is 32 such waves;
  // the wave-per-ray kernel (1.9 ns per ray) finishes it in 14 us.  Measured crossover (tools/sampler_ab.py, dn 64): 16 k rays
  // 36 | 46 us, 32 k rays 66 | 49 us (wave | lane); both give the same bits.
  const bool small = rn < 24576 && !(knob && (!strcmp(knob, "lane") || !strcmp(knob, "noshare")));
  if (!wave_only && !small && dn == fdn && (dn == 32 || dn == 64) && aligned && rn > 0) {
    const int blocks = (rn + 63) / 64;
    const bool shared = rows_equal && !use_all && !(knob && !strcmp(knob, "noshare"));
#define PGRF_FD(DN_, SH_) hipLaunchKernelGGL((fine_depth_lane_kernel<DN_, SH_>), dim3(blocks), dim3(64), 0, (hipStream_t)stream, \
                                             depth, hit, qnear, qfar, use_disp, use_all, rn, urand, out)
    if (dn == 64) { if (shared) PGRF_FD(64, true); else PGRF_FD(64, false); }
    else { if (shared) PGRF_FD(32, true); else PGRF_FD(32, false); }
#undef PGRF_FD
    PGRF_CHECK_LAUNCH("fine_depth_lane");
    return PGRF_OK;
  }
  const int blocks = (rn + 3) / 4;
  hipLaunchKernelGGL(fine_depth_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, depth, hit, qnear, qfar, use_disp,
                     use_all, rn, dn, fdn, urand, out);
  PGRF_CHECK_LAUNCH("fine_depth");
  return PGRF_OK;
}

// ------------------------------------------------------------------------------------------
// render_c2f_all (renderer.py:484-518): stable sort of the concatenated depths, gather colours and
// densities, composite over the union.  One wave per ray.
__global__ void __launch_bounds__(256) merge_c2f_kernel(const float* __restrict__ zc, const float* __restrict__ colc,
                                                        const float* __restrict__ denc, int dnc, const float* __restrict__ zf,
                                                        const float* __restrict__ colf, const float* __restrict__ denf,
                                                        int dnf, int rn, PgrfPassOut o) {
  __shared__ float s_z[4][2 * kMaxS];
  __shared__ float s_sz[4][2 * kMaxS];
  __shared__ float s_a[4][2 * kMaxS];
  __shared__ float s_c[4][2 * kMaxS][3];
  __shared__ float s_den[4][2 * kMaxS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  const bool ok = ray < rn;
  const int rayc = ok ? ray : rn - 1;
  const int total = dnc + dnf;
  for (int i = lane; i < total; i += 64)
    s_z[wv][i] = (i < dnc) ? zc[(size_t)rayc * dnc + i] : zf[(size_t)rayc * dnf + (i - dnc)];
  __syncthreads();
  // Both lists normally arrive sorted (the coarse sampler's and the importance sampler's outputs): the stable rank of an
  // element is then its own index plus a binary search in the OTHER list (coarse before fine on ties, NaNs last: the same
  // total order as sorts_before).  Anything else -- a NaN inside the coarse depths, a caller's unsorted list -- takes the
  // all-pairs rank sort, which writes every slot exactly once for any input.  (All-pairs for every ray: 2.5 ms per frame.)
  auto lt = [](float o, float v) { return (o < v) || (v != v && o == o); };      // o strictly before v, index ties aside
  bool mine_sorted = true;
  for (int j = lane; j + 1 < total; j += 64)
    if (j + 1 != dnc && lt(s_z[wv][j + 1], s_z[wv][j])) mine_sorted = false;
  const bool presorted = __all(mine_sorted);
  for (int j = lane; j < total; j += 64) {
    const float v = s_z[wv][j];
    int rank = 0;
    if (presorted) {
      const bool coarse = j < dnc;
      const float* other = &s_z[wv][coarse ? dnc : 0];
      int lo = 0, hi = coarse ? dnf : dnc;
      while (lo < hi) {          // coarse: number of fine values strictly before v; fine: number of coarse values not after v
        const int mid = (lo + hi) >> 1;
        const float o = other[mid];
        if (coarse ? lt(o, v) : !lt(v, o)) lo = mid + 1; else hi = mid;
      }
      rank = (coarse ? j : j - dnc) + lo;
    } else {
      for (int k = 0; k < total; ++k) {
        const float w = s_z[wv][k];
        rank += sorts_before(w, k, v, j) ? 1 : 0;
      }
    }
    const float den = (j < dnc) ? denc[(size_t)rayc * dnc + j] : denf[(size_t)rayc * dnf + (j - dnc)];
    const float* cp = (j < dnc) ? colc + ((size_t)rayc * dnc + j) * 3 : colf + ((size_t)rayc * dnf + (j - dnc)) * 3;
    s_sz[wv][rank] = v;
    s_den[wv][rank] = den;
    s_a[wv][rank] = 1.f - __expf(-fmaxf(den, 0.f));
    s_c[wv][rank][0] = cp[0]; s_c[wv][rank][1] = cp[1]; s_c[wv][rank][2] = cp[2];
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // transmittance T_i = prod_{j<i} (1 - alpha_j + 1e-10): an inclusive product scan over the lanes per block of 64 sorted
  // samples, times the product of the earlier blocks (the serial loop over LDS it replaces was most of this kernel: a frame's
  // merge took 2.5 ms)
  float carry = 1.f;
  for (int i = lane; i < total + lane; i += 64) {       // every lane runs every block: the shuffles need the whole wave
    const bool in = i < total;
    const float f = in ? (1.f - s_a[wv][i] + 1e-10f) : 1.f;
    float incl = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float up = __shfl_up(incl, off, 64);
      if (lane >= off) incl *= up;
    }
    float T = __shfl_up(incl, 1, 64);
    if (lane == 0) T = 1.f;
    T *= carry;
    carry *= __shfl(incl, 63, 64);
    if (!in) continue;
    const float hitp = s_a[wv][i] * T;
    s_a[wv][i] = hitp;                                   // kept for the uncertainty epilogue (only this lane reads slot i)
    const size_t pt = (size_t)ray * total + i;
    if (ok) {
      if (o.hit_prob) o.hit_prob[pt] = hitp;
      if (o.density) o.density[pt] = s_den[wv][i];
      if (o.colors) { o.colors[pt * 3] = s_c[wv][i][0]; o.colors[pt * 3 + 1] = s_c[wv][i][1]; o.colors[pt * 3 + 2] = s_c[wv][i][2]; }
    }
    acc[0] += hitp * s_c[wv][i][0]; acc[1] += hitp * s_c[wv][i][1]; acc[2] += hitp * s_c[wv][i][2];
    acc[3] += hitp * s_sz[wv][i];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], off, 64);
  if (ok && lane == 0) {
    if (o.pixel_colors) { o.pixel_colors[(size_t)ray * 3] = acc[0]; o.pixel_colors[(size_t)ray * 3 + 1] = acc[1]; o.pixel_colors[(size_t)ray * 3 + 2] = acc[2]; }
    if (o.render_depth) o.render_depth[ray] = acc[3];
  }
  if (o.render_uncert) {      // sum_i w_i (z_i - depth)^2 + 1e-5 over the merged samples (renderer.py:507-514)
    float u = 0.f;
    for (int i = lane; i < total; i += 64) { const float d = s_sz[wv][i] - acc[3]; u = fmaf(d * d, s_a[wv][i], u); }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) u += __shfl_xor(u, off, 64);
    if (ok && lane == 0) o.render_uncert[ray] = u + 1e-5f;
  }
}

// cfg render_uncert for a network pass (renderer.py:305-312): sum_i hit_prob_i (z_i - render_depth)^2 + 1e-5, one wave per
// ray over the pass's own outputs (the composite inside the render kernel is not touched: the key is off in every shipped
// configuration, and 0.27 GB of reads per 512x1024x64 frame are ~40 us)
__global__ void __launch_bounds__(256) render_uncert_kernel(const float* __restrict__ depth, const float* __restrict__ hit,
                                                            const float* __restrict__ rdepth, int rn, int dn, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= rn) return;
  const float mu = rdepth[ray];
  float u = 0.f;
  for (int i = lane; i < dn; i += 64) {
    const float d = depth[(size_t)ray * dn + i] - mu;
    u = fmaf(d * d, hit[(size_t)ray * dn + i], u);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) u += __shfl_xor(u, off, 64);
  if (lane == 0) out[ray] = u + 1e-5f;
}

int launch_render_uncert(const float* depth, const float* hit, const float* rdepth, int rn, int dn, float* out, void* stream) {
  hipLaunchKernelGGL(render_uncert_kernel, dim3((rn + 3) / 4), dim3(256), 0, (hipStream_t)stream, depth, hit, rdepth, rn, dn, out);
  PGRF_CHECK_LAUNCH("render_uncert");
  return PGRF_OK;
}

// Backward of the merge (round 4; SURVEY.md 8 f-3 with render_c2f_all): the sort is a permutation of the concatenated samples
// (the depths are constants of the training graph: the importance sampler's inputs are detached, renderer.py:437), so the gradient
// of the merged composite -- w_i = a_i T_i, T_i = prod_{j<i} (1 - a_j + 1e-10), a = 1 - exp(-relu(sigma)) -- is computed on the
// sorted samples and GATHERED back through the same ranks: coarse slot k -> g_colors_c / g_density_c, fine slot k -> g_colors_f /
// g_density_f.  grad: upstream of pixel_colors [rn,3], render_depth [rn], hit_prob / density [rn,total], colors [rn,total,3]
// (any may be null).  One wave per ray; ranks exactly as in merge_c2f_kernel.
__global__ void __launch_bounds__(256) merge_c2f_bwd_kernel(const float* __restrict__ zc, const float* __restrict__ colc,
                                                            const float* __restrict__ denc, int dnc, const float* __restrict__ zf,
                                                            const float* __restrict__ colf, const float* __restrict__ denf, int dnf,
                                                            int rn, PgrfPassGrad g, float* __restrict__ gcolc, float* __restrict__ gdenc,
                                                            float* __restrict__ gcolf, float* __restrict__ gdenf) {
  __shared__ float s_z[4][2 * kMaxS];
  __shared__ float s_sz[4][2 * kMaxS];
  __shared__ float s_a[4][2 * kMaxS];
  __shared__ float s_c[4][2 * kMaxS][3];
  __shared__ float s_den[4][2 * kMaxS];
  __shared__ int s_src[4][2 * kMaxS];
  __shared__ float s_T[4][2 * kMaxS];
  __shared__ float s_q[4][2 * kMaxS];       // g_w_i * w_i
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  const bool ok = ray < rn;
  const int rayc = ok ? ray : rn - 1;
  const int total = dnc + dnf;
  for (int i = lane; i < total; i += 64)
    s_z[wv][i] = (i < dnc) ? zc[(size_t)rayc * dnc + i] : zf[(size_t)rayc * dnf + (i - dnc)];
  __syncthreads();
  auto lt = [](float o, float v) { return (o < v) || (v != v && o == o); };
  bool mine_sorted = true;
  for (int j = lane; j + 1 < total; j += 64)
    if (j + 1 != dnc && lt(s_z[wv][j + 1], s_z[wv][j])) mine_sorted = false;
  const bool presorted = __all(mine_sorted);
  for (int j = lane; j < total; j += 64) {
    const float v = s_z[wv][j];
    int rank = 0;
    if (presorted) {
      const bool coarse = j < dnc;
      const float* other = &s_z[wv][coarse ? dnc : 0];
      int lo = 0, hi = coarse ? dnf : dnc;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float o = other[mid];
        if (coarse ? lt(o, v) : !lt(v, o)) lo = mid + 1; else hi = mid;
      }
      rank = (coarse ? j : j - dnc) + lo;
    } else {
      for (int k = 0; k < total; ++k) rank += sorts_before(s_z[wv][k], k, v, j) ? 1 : 0;
    }
    const float den = (j < dnc) ? denc[(size_t)rayc * dnc + j] : denf[(size_t)rayc * dnf + (j - dnc)];
    const float* cp = (j < dnc) ? colc + ((size_t)rayc * dnc + j) * 3 : colf + ((size_t)rayc * dnf + (j - dnc)) * 3;
    s_sz[wv][rank] = v;
    s_den[wv][rank] = den;
    s_a[wv][rank] = 1.f - __expf(-fmaxf(den, 0.f));
    s_c[wv][rank][0] = cp[0]; s_c[wv][rank][1] = cp[1]; s_c[wv][rank][2] = cp[2];
    s_src[wv][rank] = j;
  }
  __syncthreads();
  const float gp0 = g.pixel_colors ? g.pixel_colors[(size_t)rayc * 3] : 0.f, gp1 = g.pixel_colors ? g.pixel_colors[(size_t)rayc * 3 + 1] : 0.f,
              gp2 = g.pixel_colors ? g.pixel_colors[(size_t)rayc * 3 + 2] : 0.f, gd = g.render_depth ? g.render_depth[rayc] : 0.f;
  // forward transmittance (inclusive product scan per block of 64 sorted samples, as in the forward kernel)
  float carry = 1.f;
  for (int i = lane; i < total + lane; i += 64) {
    const bool in = i < total;
    const float f = in ? (1.f - s_a[wv][i] + 1e-10f) : 1.f;
    float incl = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float up = __shfl_up(incl, off, 64);
      if (lane >= off) incl *= up;
    }
    float T = __shfl_up(incl, 1, 64);
    if (lane == 0) T = 1.f;
    T *= carry;
    carry *= __shfl(incl, 63, 64);
    if (!in) continue;
    const float w = s_a[wv][i] * T;
    const size_t pt = (size_t)rayc * total + i;
    const float gw = (g.hit_prob ? g.hit_prob[pt] : 0.f) + gp0 * s_c[wv][i][0] + gp1 * s_c[wv][i][1] + gp2 * s_c[wv][i][2] + gd * s_sz[wv][i];
    s_T[wv][i] = T;
    s_q[wv][i] = gw * w;
    // colours: direct upstream + through the pixel colour
    const int j = s_src[wv][i];
    float* gc = (j < dnc) ? gcolc + ((size_t)rayc * dnc + j) * 3 : gcolf + ((size_t)rayc * dnf + (j - dnc)) * 3;
    if (ok) {
      gc[0] = (g.colors ? g.colors[pt * 3] : 0.f) + w * gp0;
      gc[1] = (g.colors ? g.colors[pt * 3 + 1] : 0.f) + w * gp1;
      gc[2] = (g.colors ? g.colors[pt * 3 + 2] : 0.f) + w * gp2;
    }
  }
  __syncthreads();
  // S_k = sum_{i>k} g_w_i w_i: reverse scan over the blocks of 64, last block first
  const int nblk = (total + 63) >> 6;
  float tail = 0.f;
  for (int b = nblk - 1; b >= 0; --b) {
    const int i = b * 64 + lane;
    const bool in = i < total;
    const float q = in ? s_q[wv][i] : 0.f;
    float incl = q;                                        // inclusive suffix sum within the block
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float dn_ = __shfl_down(incl, off, 64);
      if (lane + off < 64) incl += dn_;
    }
    const float S = incl - q + tail;                       // strictly after i
    tail += __shfl(incl, 0, 64);
    if (!in || !ok) continue;
    const float a = s_a[wv][i], T = s_T[wv][i];
    const size_t pt = (size_t)rayc * total + i;
    const float gw = (g.hit_prob ? g.hit_prob[pt] : 0.f) + gp0 * s_c[wv][i][0] + gp1 * s_c[wv][i][1] + gp2 * s_c[wv][i][2] + gd * s_sz[wv][i];
    const float ga = gw * T - S / (1.f - a + 1e-10f);
    const float den = s_den[wv][i];
    const float gs = (den > 0.f ? ga * (1.f - a) : 0.f) + (g.density ? g.density[pt] : 0.f);   // d alpha / d sigma = exp(-sigma) = 1 - alpha
    const int j = s_src[wv][i];
    if (j < dnc) gdenc[(size_t)rayc * dnc + j] = gs; else gdenf[(size_t)rayc * dnf + (j - dnc)] = gs;
  }
}

int launch_merge_c2f_bwd(const float* zc, const float* colc, const float* denc, int dnc, const float* zf, const float* colf,
                         const float* denf, int dnf, int rn, const PgrfPassGrad& g, float* gcolc, float* gdenc, float* gcolf,
                         float* gdenf, void* stream) {
  const int blocks = (rn + 3) / 4;
  hipLaunchKernelGGL(merge_c2f_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, zc, colc, denc, dnc, zf, colf, denf, dnf, rn, g,
                     gcolc, gdenc, gcolf, gdenf);
  PGRF_CHECK_LAUNCH("merge_c2f backward");
  return PGRF_OK;
}

int launch_merge_c2f(const float* zc, const float* colc, const float* denc, int dnc, const float* zf, const float* colf,
                     const float* denf, int dnf, int rn, const PgrfPassOut& o, void* stream) {
  const int blocks = (rn + 3) / 4;
  hipLaunchKernelGGL(merge_c2f_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, zc, colc, denc, dnc, zf, colf, denf, dnf, rn, o);
  PGRF_CHECK_LAUNCH("merge_c2f");
  return PGRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Per-ray direction table of pgrf_render_rays: the render pass would otherwise recompute the same ~250 instructions
// (sin/cos of the pixel angles, normalisation, rotation) in all 64 lanes of every wave of every pass.
template <bool GEN>
__global__ void ray_dirs_kernel(const PassArgs p, float* __restrict__ tab) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= p.rn) return;
  float wx, wy, wz;
  ray_direction<GEN>(p, ray, wx, wy, wz);
  reinterpret_cast<float4*>(tab)[ray] = make_float4(wx, wy, wz, 0.f);
}

int launch_ray_dirs(const PassArgs& a, float* tab, void* stream) {
  const int blocks = (a.rn + 255) / 256;
  if (a.dataset != PGRF_DATASET_M3D || a.perspective)
    hipLaunchKernelGGL(ray_dirs_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, tab);
  else
    hipLaunchKernelGGL(ray_dirs_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, tab);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(PGRF_ERR_LAUNCH, "ray_dirs launch failed: %s", hipGetErrorString(e));
  return PGRF_OK;
}

}  // namespace pgrf
