// Fused per-ray renderer pass for gfx950 (CDNA4).  One launch does, for every ray of the batch:
// ERP ray generation -> projection into the V source views -> bilinear gathers (NHWC maps) ->
// mixture-of-logistics visibility decoder -> IBRNet-with-NeuRay aggregation (all Linear layers on
// v_mfma_f32_32x32x2_f32, transposed so activations stay in registers: see pgrf_layout.h) ->
// 4-head attention over the samples of the ray (QK^T on MFMA, K/V staged in LDS) -> density,
// colour blending -> alpha compositing (transmittance = inclusive product scan over the wave's 32 samples by __shfl_up,
// the totals of a ray's earlier waves through LDS; pixel colour / depth by wave-shuffle reduction).
// The 16-output layers run on v_mfma_f32_16x16x4_f32 (mfma_layer16), the <= 8-output layers on v_mfma_f32_4x4x1_16b_f32 (mfma_small).
//
// Work decomposition: workgroup = 4 waves = 128 sample points = 128/DN rays.  A wave owns 32
// consecutive samples of one ray for ALL source views; lane l holds point (l & 31); lanes l and
// l+32 hold the two halves of each 32-feature activation tile (accumulator layout of the 32x32
// MFMA).  Per-point scalars are computed redundantly by both halves.
//
// Reference semantics (file:line into /root/reference) are cited at each stage.
#pragma once
#include "pgrf_experiments.h"
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "pgrf_internal.h"
#include "pgrf_layout.h"
#include "render_pass_args.h"
#include "ray_dir.h"

// timing experiment only: -DPGRF_NO_BARRIER removes the workgroup barriers (results are then wrong)
#ifdef PGRF_NO_BARRIER
#define PGRF_SYNC() do {} while (0)
#else
#define PGRF_SYNC() __syncthreads()
#endif

// Precision of the Linear layers, fixed per translation unit (-DPGRF_PREC=0|1); each value lives in its own
// namespace so that both instantiations of the same template can be linked into one library.
//   0  f32   : v_mfma_f32_32x32x2_f32, bit-wise an fp32 fmaf chain (default, reference-equivalent arithmetic)
//   1  f16x3 : activations and weights split as x = h1 + h2 (two halves, 22 significant bits), three
//              v_mfma_f32_32x32x16_f16 per 16-deep k-group (h1*h1' + h1*h2' + h2*h1'), fp32 accumulation:
//              ~2^-21 relative error per product at 1/5 of the matrix-pipe time.  Opt-in.
#ifndef PGRF_PREC
#define PGRF_PREC 0
#endif
#if PGRF_PREC == 0
#define PGRF_PREC_NS p32
#else
#define PGRF_PREC_NS p16
#endif

#ifndef PGRF_PV_PACKED
#define PGRF_PV_PACKED 1
#endif
namespace pgrf {
namespace PGRF_PREC_NS {


constexpr int kPrec = PGRF_PREC;
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Every f32 instance carries an LDS image (kHasImg below).  The layer functions decide "image or blob" on this CONSTANT, never on the
// pointer: a run-time select between an LDS and a global pointer is a generic pointer, its loads are flat_load_dwordx4 (52 per tile:
// every layer's bias block) and each of those waits for vmcnt(0) AND lgkmcnt(0) -- draining the gathers in flight at every layer start.
#ifdef PGRF_NO_LDS_IMG
constexpr bool kImg = false;
#else
constexpr bool kImg = (kPrec == 0);
#endif
#ifdef PGRF_ABL_WLDS
__shared__ float4 g_abl_w[512];
#endif



// Combine a value with the same value of lane ^ 32 (the other half of the point's features).  v_permlane32_swap (gfx950)
// exchanges the upper half of one register with the lower half of another on the VALU: no LDS round trip as with
// ds_bpermute.  With both operands = v it yields [lo,lo] and [hi,hi]; their sum / max is the cross-half result in all lanes.
__device__ __forceinline__ float xsum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xmax(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// ELU(x) = median(x, exp(x)-1, 0): x > 0 -> exp(x)-1 >= x > 0; x < 0 -> x <= exp(x)-1 < 0.  One v_med3 instead of cmp+cndmask.
// (Tried: min(exp(x),1) - 1 via the v_exp's clamp modifier, then a two-operand v_max with x.  As fmaxf it costs a second,
// canonicalising v_max on the raw accumulator (IEEE mode) and is 0.8 % slower; as an inline-asm v_max it measured +0.3 % but
// gave wrong, run-to-run varying results -- hazard tracking does not see through the asm.)
#if defined(PGRF_ABL_ELU) && PGRF_ABL_ELU == 1      // ablations (timing only, wrong numbers): 1 = identity, 2 = no multiply, 3 = ReLU
__device__ __forceinline__ float elu1(float x) { return x; }
#elif defined(PGRF_ABL_ELU) && PGRF_ABL_ELU == 2
__device__ __forceinline__ float elu1(float x) { return __builtin_amdgcn_fmed3f(x, __builtin_amdgcn_exp2f(x) - 1.f, 0.f); }
#elif defined(PGRF_ABL_ELU) && PGRF_ABL_ELU == 3
__device__ __forceinline__ float elu1(float x) { return fmaxf(x, 0.f); }
#else
__device__ __forceinline__ float elu1(float x) { return __builtin_amdgcn_fmed3f(x, __expf(x) - 1.f, 0.f); }
#endif
__device__ __forceinline__ float sigmoid1(float x) { return rcp1(1.f + __expf(-x)); }
// nn.Softplus (beta 1, threshold 20) as max(x,0) + log(1 + exp(-|x|)) on v_exp/v_log: absolute error ~1e-7, the size of one
// fp32 rounding of the result's integer part; libm's log1pf is a ~140-instruction double-float routine (measured: the
// 8 calls per wave were 4 % of the kernel).  For x > 20 the second term is below half an ulp of x: same as the threshold.
__device__ __forceinline__ float softplus1(float x) {
  const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 1.4426950408889634f);
  return fmaf(__builtin_amdgcn_logf(1.f + t), 0.6931471805599453f, fmaxf(x, 0.f));
}
// logistic CDF 0.5 + 0.5 tanh(d) == 1 / (1 + exp(-2d))  (dist_decoder.py:129-130), 4 instructions instead of libm tanhf
__device__ __forceinline__ float cdf1(float d) {
  return rcp1(1.f + __builtin_amdgcn_exp2f(d * -2.8853900817779268f));
}

__device__ __forceinline__ void elu16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = elu1(a[r]);
}

// LDS image of the blob (pgrf_layout.h): LI is the workgroup's copy; -DPGRF_NO_LDS_IMG reads everything from global memory.
// Measured (512x1024, 2 views): f32 43.34 -> 43.04 ms; split-f16 29.3 -> 32.3 ms (its 51 KB image costs 20 more spilled
// registers), so that mode keeps reading from global memory.
// kHasImg<V> (below, with the other per-instance policies) says which instances carry an image.
#define PGRF_LIMG(id) (kHasImg<V> ? LI + img_layer_off((id), kImgMode<V>) : (const float*)nullptr), \
                      (kHasImg<V> && kImgMode<V> == 2 && img_full(id)), img_a_floats((id), kImgMode<V>)
#define PGRF_LHEAD(off) (kHasImg<V> ? LI + ((off) - OFF_DDH_W) : W + (off))

// layers with <= 8 outputs on the 4x4x1 MFMA (f32 mode only; -DPGRF_NO_SMALL = padded 32-row tiles)
#ifdef PGRF_NO_SMALL
constexpr bool kSmall = false;
#else
constexpr bool kSmall = (kPrec == 0);
#endif
// not with 4 views: the padded tiles cost less there than the registers the 4x4 path adds (260 -> 162 spilled, 79.8 -> 78.3 ms)
#ifdef PGRF_SMALL_V4
template <int V> constexpr bool kSmallV = kSmall;
#else
template <int V> constexpr bool kSmallV = kSmall && V != 4;
#endif

// which LDS image a kernel instance uses (pgrf_layout.h)
#ifdef PGRF_NO_FULL_IMG
template <int V> constexpr int kImgMode = kPrec;
#else
// Kernels for up to PGRF_TWO_WAVE_MAX_V views are built for two waves per SIMD (<= 256 registers): first-chunk LDS
// image, lazily recomputed tap setups, LDS-parked view directions; beyond that one wave per SIMD with the registers
// (and the LDS, for the full weight image) of the whole SIMD.
// Measured with every view count at two waves: 3 views 73.7 -> 63.4 ms, 4 views 92.7 -> 84.9 ms, although those
// kernels spill 170 / 333 registers to scratch (the second wave hides it; -DPGRF_TWO_WAVE_MAX_V=2 restores one wave).
#ifndef PGRF_TWO_WAVE_MAX_V
#define PGRF_TWO_WAVE_MAX_V 4
#endif
template <int V> constexpr bool kTwoWave = (V <= PGRF_TWO_WAVE_MAX_V);
// 4 views at two waves: `nf` (16 registers per view) is not kept from stage A to stage C but rebuilt there from 3
// registers (one more gather of the prob_embed block of the projection map + 1 MFMA per view): 333 -> 260 spilled
// registers, 84.7 -> 83.9 ms.  (3 views: 60.2 -> 63.7 ms, so not there.)  -DPGRF_NO_RECOMPUTE_NF disables.
#ifdef PGRF_NO_RECOMPUTE_NF
template <int V> constexpr bool kRecomputeNf = false;
#else
template <int V> constexpr bool kRecomputeNf = kTwoWave<V> && V > 3;
#endif
// Two-wave f32 kernels: first-chunk image (mode 0) or the slim one (mode 3), per instance as measured (A/B of whole-library
// builds, same box): 2 views x 64 samples 37.45 -> 37.18 ms and 3 views x 64 samples 54.3 -> 53.5 ms with the slim image;
// 4 views 78.3 -> 78.6 and 2 views x 32 samples 19.07 -> 19.33 ms, so those keep the first chunks.  (No image at all:
// 37.18 / 62.7 / 82.9 / 19.2 ms -- the register allocator again.)
#ifndef PGRF_KT
#define PGRF_KT 2
#endif
#ifndef PGRF_V
#define PGRF_V 2
#endif
// Experiment (-DPGRF_LDS_STATE, off by default): 4 views at two waves (f32) with the per-view state of one view PAIR in LDS
// while the other pair is worked on (see "kLdsState" in render_pass_kernel) -- that needs the LDS the first weight chunks
// would take, so those kernels then carry the slim image.  Measured in round 3 (profiles/r03_dead_ends.txt): spilled
// registers 157 -> 57, HBM-side traffic of the launch 90 -> 41 GB, and the frame 78.3 -> 79.2 ms: the scratch traffic was
// never the limiter (0.5 TB/s, absorbed next to a pipe that is issue-bound), the extra LDS instructions cost what it saved.
#ifdef PGRF_LDS_STATE
template <int V> constexpr bool kLdsState = (V == 4) && kTwoWave<V> && (kPrec == 0);
#else
template <int V> constexpr bool kLdsState = false;
#endif
// (in the 4-view translation units the V = 2 stage functions belong to the wave-pair kernel, whose LDS is taken by the exchange
// buffers: slim image for every sample count there)
template <int V> constexpr bool kSlimImg = ((PGRF_KT == 2) && (V == 2 || V == 3)) || kLdsState<V> || (PGRF_V == 4 && V == 2);
template <int V> constexpr int kImgMode = (kPrec == 1) ? 1 : (kTwoWave<V> ? (kSlimImg<V> ? 3 : 0) : 2);
// no image in the split-f16 mode (see above).  (Tried for > 2 views at two waves: image dropped in favour of parking the
// first view group's prob_embed features through stage B — the register allocator answered with MORE spills, 170 -> 289.)
#ifdef PGRF_NO_LDS_IMG
template <int V> constexpr bool kHasImg = false;
#else
template <int V> constexpr bool kHasImg = (kPrec == 0);
#endif
#endif

// One Linear layer on the matrix cores for NT point tiles sharing the A operand.
//   acc[t][m] (+)= A(m) * b[t]   with b[t][s] = B operand of k-step s (see pgrf_layout.h)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
union H8 { f16x8 v; fp16x2 p[4]; uint4 u; unsigned u4[4]; };

// x[0..7] (fp32) -> hi, lo halves with x ~= hi + lo: hi = rtz16(x), lo = rtz16(x - hi) (the difference is exact)
__device__ __forceinline__ void split8(const float (&x)[8], H8& hi, H8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(x[2 * e], x[2 * e + 1]);
    hi.p[e] = h;
    lo.p[e] = __builtin_amdgcn_cvt_pkrtz(x[2 * e] - (float)h[0], x[2 * e + 1] - (float)h[1]);
  }
}

// split-f16 evaluation of one layer; Hblk = this layer's [mt][g][piece][lane][8] block (pgrf_layout.h)
template <int KS, int MT, int NT, bool INIT_BIAS>
__device__ __forceinline__ void mfma_layer_h(const float* __restrict__ Hblk, const float* __restrict__ B, int lane,
                                             const float (&b)[NT][KS], f32x16 (&acc)[NT][MT], const float* Limg, int limg_a) {
  constexpr int G = (KS + 7) / 8;
  if (INIT_BIAS) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4* bp = reinterpret_cast<const float4*>((kImg ? Limg + limg_a : B) + m * 32 + (lane >> 5) * 16);
      const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
      const f32x16 bv = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][m] = bv;
    }
  }
  // k-group outermost: only one group's split operands (8 registers per point tile) are live at a time
#pragma unroll
  for (int g = 0; g < G; ++g) {
    H8 bh[NT], bl[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (8 * g + e < KS) ? b[t][8 * g + e < KS ? 8 * g + e : 0] : 0.f;
      split8(x, bh[t], bl[t]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      H8 a1, a2;
      if (kImg && m == 0 && g == 0) {
        a1.u = reinterpret_cast<const uint4*>(Limg)[lane];
        a2.u = reinterpret_cast<const uint4*>(Limg)[64 + lane];
      } else {
        a1.u = reinterpret_cast<const uint4*>(Hblk)[((m * G + g) * 2 + 0) * 64 + lane];
        a2.u = reinterpret_cast<const uint4*>(Hblk)[((m * G + g) * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2.v, bh[t].v, acc[t][m], 0, 0, 0);
        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.v, bl[t].v, acc[t][m], 0, 0, 0);
        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.v, bh[t].v, acc[t][m], 0, 0, 0);
      }
    }
#ifdef PGRF_GROUP_FENCE
    // (was needed while the kernel spilled: kept the scheduler from hoisting every group's f32->f16 split in front of the
    // first MFMA of a long layer; with the spills gone it costs 3 %: 25.5 vs 24.6 ms)
    if (G >= 4) __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

template <int KS, int MT, int NT, bool INIT_BIAS>
__device__ __forceinline__ void mfma_layer(const float* __restrict__ A, const float* __restrict__ B, int lane,
                                           const float (&b)[NT][KS], f32x16 (&acc)[NT][MT], const float* __restrict__ Hblk,
                                           const float* Limg, bool limg_full, int limg_a) {
  if constexpr (kPrec == 1) {
    mfma_layer_h<KS, MT, NT, INIT_BIAS>(Hblk, B, lane, b, acc, Limg, limg_a);
    return;
  }
  constexpr int KS4 = (KS + 3) / 4;
#ifdef PGRF_A_PREFETCH
#ifndef PGRF_A_FENCE
#define PGRF_A_FENCE 1
#endif
  // experiment: A operands PGRF_A_PREFETCH groups of four k-steps ahead of their MFMAs, order pinned by scheduling fences
  // (bit 8 of PGRF_A_FENCE: sched_group_barrier pattern "one load, then the group's MFMAs" instead of full fences)
  {
    constexpr int Q = MT * KS4, PD = (PGRF_A_PREFETCH) < Q ? (PGRF_A_PREFETCH) : Q;
    auto load_a = [&](int q) {
      const int m = q / KS4, s4 = q % KS4;
      return (kImg && limg_full)          ? reinterpret_cast<const float4*>(Limg)[(m * KS4 + s4) * 64 + lane]
             : (kImg && limg_a > 0 && m == 0 && s4 == 0) ? reinterpret_cast<const float4*>(Limg)[lane]
                                           : reinterpret_cast<const float4*>(A)[(m * KS4 + s4) * 64 + lane];
    };
    float4 ring[PD + 1];
#pragma unroll
    for (int q = 0; q < PD; ++q) ring[q] = load_a(q);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int m = q / KS4, s4 = q % KS4;
      if (q + PD < Q) ring[(q + PD) % (PD + 1)] = load_a(q + PD);
#if PGRF_A_FENCE == 1
      __builtin_amdgcn_sched_barrier(0);
#elif PGRF_A_FENCE == 2
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // one vector-memory read ...
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);  // ... then the group's MFMAs
#endif
      const float4 a = ring[q % (PD + 1)];
      const float av[4] = {a.x, a.y, a.z, a.w};
      f32x16 bv;
      if (INIT_BIAS && s4 == 0) {
        const float4* bp = reinterpret_cast<const float4*>((kImg ? Limg + limg_a : B) + m * 32 + (lane >> 5) * 16);
        const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        bv = (f32x16){b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (s4 * 4 + j < KS) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (INIT_BIAS && s4 == 0 && j == 0)
              acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[t][0], bv, 0, 0, 0);
            else
              acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[t][s4 * 4 + j], acc[t][m], 0, 0, 0);
          }
        }
      }
#if PGRF_A_FENCE == 1
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    return;
  }
#endif
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 bv;
    if (INIT_BIAS) {
      const float4* bp = reinterpret_cast<const float4*>((kImg ? Limg + limg_a : B) + m * 32 + (lane >> 5) * 16);
      const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
      bv = (f32x16){b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    }
#pragma unroll
    for (int s4 = 0; s4 < KS4; ++s4) {
#ifdef PGRF_ABL_WEIGHTS   // ablation (timing only): every weight load hits the same 1 KiB -> L1 resident
      const float4 a = reinterpret_cast<const float4*>(A)[lane + ((m * KS4 + s4) & 1) * 64];
#elif defined(PGRF_ABL_WLDS)   // ablation (timing only): every weight load from 8 KB of LDS (what an LDS-resident blob would cost)
      const float4 a = g_abl_w[lane + ((m * KS4 + s4) & 7) * 64];
#else
      const float4 a = (kImg && limg_full)          ? reinterpret_cast<const float4*>(Limg)[(m * KS4 + s4) * 64 + lane]
                       : (kImg && limg_a > 0 && m == 0 && s4 == 0) ? reinterpret_cast<const float4*>(Limg)[lane]
                                                     : reinterpret_cast<const float4*>(A)[(m * KS4 + s4) * 64 + lane];
#endif
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (s4 * 4 + j < KS) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            // the bias vector is the C operand of the first MFMA of the tile (shared by the NT point tiles)
            if (INIT_BIAS && s4 == 0 && j == 0)
              acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[t][0], bv, 0, 0, 0);
            else
              acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[t][s4 * 4 + j], acc[t][m], 0, 0, 0);
          }
        }
      }
    }
  }
}

// Layer with few outputs on v_mfma_f32_4x4x1_16b_f32 (pgrf_layout.h "small layers"): out[t][4g+i] = row 4g+i of
// W x + bias for the lane's own point, complete in every lane.  10 cycles per (4 rows x 2 features) against 64 cycles
// per (32 rows x 2 features) for the padded 32x32x2 tile.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KS, int NG, int NT>
__device__ __forceinline__ void mfma_small(const float* __restrict__ S, const float* __restrict__ SB, int lane,
                                           const float (&b)[NT][KS], float (&out)[NT][NG * 4]) {
  constexpr int KS4 = (KS + 3) / 4;
  f32x4 acc[NT][NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const float4 bb = *reinterpret_cast<const float4*>(SB + (g * 2 + (lane >> 5)) * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][g] = (f32x4){bb.x, bb.y, bb.z, bb.w};
  }
#pragma unroll
  for (int r4 = 0; r4 < KS4; ++r4) {
    float4 a[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) a[g] = reinterpret_cast<const float4*>(S)[(g * KS4 + r4) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (r4 * 4 + j < KS) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const float av = j == 0 ? a[g].x : j == 1 ? a[g].y : j == 2 ? a[g].z : a[g].w;
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, b[t][r4 * 4 + j], acc[t][g], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[t][4 * g + i] = xsum(acc[t][g][i]);
}

// Layer with 16 outputs on v_mfma_f32_16x16x4_f32 (pgrf_layout.h "w16" layers): the B registers of two k-steps become the operands of
// one k4-step for points 0-15 and 16-31 through one v_permlane16_swap; the results come back to registers 0..7 of the 32-point layout
// the same way.  Half the matrix-pipe time of the padded 32-row tile (2 x 32 cycles per 4 features against 2 x 64), for KS/2 + 4
// cross-lane swaps.  Two accumulators per point tile alternate, so the 40-cycle dependent latency of this MFMA never shows.
template <int KS, int NT>
__device__ __forceinline__ void mfma_layer16(const float* __restrict__ A16, const float* __restrict__ B16, int lane,
                                             const float (&b)[NT][KS], float (&out)[NT][8]) {
  constexpr int K4 = (KS + 1) / 2, C4 = (K4 + 3) / 4;
  f32x4 d[NT][2];
  {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (B16) { const float4 bb = reinterpret_cast<const float4*>(B16)[lane >> 4]; bv = (f32x4){bb.x, bb.y, bb.z, bb.w}; }
#pragma unroll
    for (int t = 0; t < NT; ++t) d[t][0] = d[t][1] = bv;
  }
#pragma unroll
  for (int c = 0; c < C4; ++c) {
    const float4 a = reinterpret_cast<const float4*>(A16)[c * 64 + lane];
    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = c * 4 + e;
      if (j < K4) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float x = b[t][2 * j], y = (2 * j + 1 < KS) ? b[t][2 * j + 1 < KS ? 2 * j + 1 : 0] : 0.f;
          const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
          d[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], __uint_as_float(r[0]), d[t][0], 0, 0, 0);
          d[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], __uint_as_float(r[1]), d[t][1], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[t][0][i]), __float_as_uint(d[t][1][i]), false, false);
      out[t][i] = __uint_as_float(r[0]);
      out[t][4 + i] = __uint_as_float(r[1]);
    }
}
// 16-row layers on the 16x16x4 MFMA: f32 mode (-DPGRF_NO_W16 = the padded 32-row tiles)
#ifdef PGRF_NO_W16
constexpr bool kW16 = false;
#else
constexpr bool kW16 = (kPrec == 0);
#endif

// partial dot over this lane half's R features (no cross-half sum)
template <int R>
__device__ __forceinline__ float head_partial(const float* __restrict__ w, int half, const f32x16& x) {
  const float* wp = w + half * R;
  // (even / odd features on the halves of one v_pk_fma_f32 chain: measured 0.5 % slower)
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) s = fmaf(wp[r], x[r], s);
  return s;
}
// VALU head: out = sum_r w[half][r] * x[r], summed over both lane halves (bias added by caller).
template <int R>
__device__ __forceinline__ float head_dot(const float* __restrict__ w, int half, const f32x16& x) {
  return xsum(head_partial<R>(w, half, x));
}
// s0, s1 = per-half partial sums of view 0 / view 1 -> the full sum of view `half` (one swap + one add for both views)
__device__ __forceinline__ float pair_mine(float s0, float s1) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s0), __float_as_uint(s1), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LDS word that must really be re-read (values parked there to shorten their live range): a volatile access through the LDS address
// space.  (`*(volatile float*)&lds[..]` is a GENERIC volatile access: flat_load_dword sc0 sc1, which waits for vmcnt AND lgkmcnt.)
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ float lds_reload(const float* p) { return *(volatile lds_f32*)p; }
__device__ __forceinline__ void lds_restore(float* p, float v) { *(volatile lds_f32*)p = v; }

struct Bilin {
  int o00, o01, o10, o11;   // pixel offsets (pixels, not floats) of the 4 taps inside one view's map
  float wnw, wne, wsw, wse;
};

// grid_sample(bilinear, padding_mode=border) tap setup for a point given in FULL-image pixels.
// network/render_ops.py:126-143, network/ops.py:32-52 (x/(W-1)*2-1 with the full-image W,H).
// normalised sampling position of a full-image pixel coordinate (ops.py:32-52): computed once per view, shared by the
// three maps sampled there (the two divisions are IEEE: these are pixel coordinates)
__device__ __forceinline__ void norm_pos(float px, float py, int H, int W, float& xn, float& yn) {
  xn = px / (float)(W - 1) * 2.f - 1.f;
  yn = py / (float)(H - 1) * 2.f - 1.f;
}
__device__ __forceinline__ Bilin bilin_setup(float xn, float yn, int fh, int fw, bool align) {
  float ix, iy;
  if (align) {
    ix = (xn + 1.f) / 2.f * (float)(fw - 1);
    iy = (yn + 1.f) / 2.f * (float)(fh - 1);
  } else {
    ix = ((xn + 1.f) * (float)fw - 1.f) / 2.f;
    iy = ((yn + 1.f) * (float)fh - 1.f) / 2.f;
  }
  ix = fminf(fmaxf(ix, 0.f), (float)(fw - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(fh - 1));
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
  Bilin b;
  b.wnw = (fx1 - ix) * (fy1 - iy);
  b.wne = (ix - fx0) * (fy1 - iy);
  b.wsw = (fx1 - ix) * (iy - fy0);
  b.wse = (ix - fx0) * (iy - fy0);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const int x1 = min(x0 + 1, fw - 1), y1 = min(y0 + 1, fh - 1);   // out-of-range taps carry weight 0
  b.o00 = y0 * fw + x0;
  b.o01 = y0 * fw + x1;
  b.o10 = y1 * fw + x0;
  b.o11 = y1 * fw + x1;
  return b;
}

// Gather one 32-channel NHWC map into the accumulator layout: register 4q+i of half h = channel 8q+4h+i.
template <int STRIDE = 32, bool BATCH = false>
__device__ __forceinline__ f32x16 gather32(const float* __restrict__ map, const Bilin& b, int half) {
  f32x16 out;
#ifdef PGRF_ABL_GATHER    // ablation (timing only): all taps read pixel 0
  const float* base = map + half * 4 - (size_t)b.o00 * 32 * 0;
  Bilin bb = b; bb.o00 = bb.o01 = bb.o10 = bb.o11 = 0;
#define b bb
#else
  const float* base = map + half * 4;
#endif
  if constexpr (BATCH) {
    // experiment (PGRF_BATCH_GATHER): all 16 loads of the gather in flight before the first blend, pinned by scheduling fences.
    // In the second view pair of the 4-view kernels the scheduler falls back to its minimum-pressure order (the first pair's
    // state is live across the region) -- four loads, wait, blend, four times per gather.
    float4 t[4][4];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      t[q][0] = *reinterpret_cast<const float4*>(base + (size_t)b.o00 * STRIDE + q * 8);
      t[q][1] = *reinterpret_cast<const float4*>(base + (size_t)b.o01 * STRIDE + q * 8);
      t[q][2] = *reinterpret_cast<const float4*>(base + (size_t)b.o10 * STRIDE + q * 8);
      t[q][3] = *reinterpret_cast<const float4*>(base + (size_t)b.o11 * STRIDE + q * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      out[4 * q + 0] = t[q][0].x * b.wnw + t[q][1].x * b.wne + t[q][2].x * b.wsw + t[q][3].x * b.wse;
      out[4 * q + 1] = t[q][0].y * b.wnw + t[q][1].y * b.wne + t[q][2].y * b.wsw + t[q][3].y * b.wse;
      out[4 * q + 2] = t[q][0].z * b.wnw + t[q][1].z * b.wne + t[q][2].z * b.wsw + t[q][3].z * b.wse;
      out[4 * q + 3] = t[q][0].w * b.wnw + t[q][1].w * b.wne + t[q][2].w * b.wsw + t[q][3].w * b.wse;
    }
  } else {
    // (tried: the blend on v_pk_mul_f32 / v_pk_fma_f32 with the tap weight broadcast through op_sel -- half the VALU instructions,
    // 36.35 against 35.95 ms per frame: two dependent chains per quad instead of four, and hipcc's pre-emit peephole splits the
    // packed ops that land behind an MFMA back into VOP3 pairs; as inline asm 37.07 ms.  profiles/r05_dead_ends.txt)
#ifndef PGRF_GATHER_QUAD_MAJOR
    // tap-major (35.82 -> 35.62 ms against quad-major, -DPGRF_GATHER_QUAD_MAJOR): the four 16-byte pieces a lane reads of ONE texel's 128-byte line are requested back to back (one L1 miss, three
    // hits under it) and folded into the 16 accumulators as they arrive; per channel the same mul, fma, fma, fma in the same order
    {
      const int offs[4] = {b.o00, b.o01, b.o10, b.o11};
      const float ws[4] = {b.wnw, b.wne, b.wsw, b.wse};
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        float4 t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const float4*>(base + (size_t)offs[tap] * STRIDE + q * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (tap == 0) {
            out[4 * q + 0] = t[q].x * ws[0]; out[4 * q + 1] = t[q].y * ws[0]; out[4 * q + 2] = t[q].z * ws[0]; out[4 * q + 3] = t[q].w * ws[0];
          } else {
            out[4 * q + 0] = fmaf(t[q].x, ws[tap], out[4 * q + 0]); out[4 * q + 1] = fmaf(t[q].y, ws[tap], out[4 * q + 1]);
            out[4 * q + 2] = fmaf(t[q].z, ws[tap], out[4 * q + 2]); out[4 * q + 3] = fmaf(t[q].w, ws[tap], out[4 * q + 3]);
          }
        }
      }
    }
#else
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(base + (size_t)b.o00 * STRIDE + q * 8);
      const float4 c = *reinterpret_cast<const float4*>(base + (size_t)b.o01 * STRIDE + q * 8);
      const float4 d = *reinterpret_cast<const float4*>(base + (size_t)b.o10 * STRIDE + q * 8);
      const float4 e = *reinterpret_cast<const float4*>(base + (size_t)b.o11 * STRIDE + q * 8);
      out[4 * q + 0] = a.x * b.wnw + c.x * b.wne + d.x * b.wsw + e.x * b.wse;
      out[4 * q + 1] = a.y * b.wnw + c.y * b.wne + d.y * b.wsw + e.y * b.wse;
      out[4 * q + 2] = a.z * b.wnw + c.z * b.wne + d.z * b.wsw + e.z * b.wse;
      out[4 * q + 3] = a.w * b.wnw + c.w * b.wne + d.w * b.wsw + e.w * b.wse;
    }
#endif
  }
#ifdef PGRF_ABL_GATHER
#undef b
#endif
  return out;
}

__device__ __forceinline__ void dump_tile32(float* dst, const f32x16& x, int half) {   // dst -> 32 floats of this point
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[rho(r, half)] = x[r];
}
__device__ __forceinline__ void dump_tile16(float* dst, const f32x16& x, int half) {   // regs 0..7 hold 16 features
#pragma unroll
  for (int r = 0; r < 8; ++r) dst[rho(r, half)] = x[r];
}

#ifdef PGRF_STAGE_TIMING
// experiment builds only (tools/stage_timing.py): per-stage wave cycles summed into a device-global table
__device__ unsigned long long g_stage_cycles[32];
#define PGRF_TICK(i)                                                                    \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();                      \
    if (lane == 0 && (blockIdx.x & 63) == 5) atomicAdd(&g_stage_cycles[i], t_now - t_prev);                       \
    t_prev = t_now;                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
#elif defined(PGRF_NO_STAGE_FENCE)
#define PGRF_TICK(i) do {} while (0)
#else
// Stage boundaries as scheduling fences, per instance as measured.  Round 1: 47.5 -> 46.3 ms with them (fewer live ranges crossing
// stages, while the kernel still spilled).  Round 5, after the 16-row layers moved to the 16x16x4 MFMA and the flat loads went (225
// registers, nothing spilled): WITHOUT them 2 views x 64 samples 35.45 -> 34.72 ms and the 4-view pair kernel 65.2 -> 64.5 ms -- the
// scheduler slides one stage's loads and VALU work under the previous stage's MFMAs; 3 views the other way round now (fences on:
// 50.57 -> 50.13 ms although 8-15 registers spill), one view as well (21.57 -> 21.36 ms).  32 and 128 samples follow the 64-sample
// kernels (2 views x 32: 17.66 vs 17.83 ms with fences, x 128: 72.2 vs 73.7; 4 views x 32: 33.35 vs 33.8).  (V is in scope wherever
// a stage ends.)
#ifndef PGRF_FENCE_MASK
// (2 views x 64 samples: fences after neuray_fc, stage B and the density head only -- single-fence sweep + combinations, 34.8 -> 34.25 ms)
// (4 views x 64 samples, the pair kernel: fences after the projection / tap set-up and after neuray_fc of each pair -- single-fence sweep,
//  every other boundary within +-0.3 %: 63.97 -> 63.45 ms)
#define PGRF_FENCE_MASK ((V == 3 || V == 1) ? 0xffffffffu : ((PGRF_KT == 2 && PGRF_V == 2) ? 0x100084u : ((PGRF_KT == 2 && PGRF_V == 4) ? 0x110000u : 0u)))      // bit i = stage boundary i is a fence (tuning knob)
#endif
#ifdef PGRF_STAGE_MARK   // ISA census builds (tools/isa_stage_census.py): a comment line per stage boundary in the assembly
#define PGRF_TICK(i) do { if (((PGRF_FENCE_MASK) >> (i)) & 1u) { __builtin_amdgcn_sched_barrier(0); asm volatile("; PGRF_STAGE_END %0" :: "i"(i)); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PGRF_TICK(i) do { if (((PGRF_FENCE_MASK) >> (i)) & 1u) __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
#endif

struct ViewState {
  f32x16 fa;      // rgb_feat tile A (img-feat part, + dir feat); later: x (base_fc output, 32)
  float fb[3];    // rgb_feat tile B (rgb part + dir feat), the same value in both lane halves
  f32x16 nf;      // relu(prob_embed.0): prob_embed.2 ("neuray feat") is folded into its consumers' weights
  float dd[4];    // [prj_dir - que_dir, <prj_dir,que_dir>]
  float rgb[3];   // gathered source colour (blended at the end)
  float w0;       // sigmoid(neuray_fc) * w
  float v2;       // vis_fc2 output
  float hv, xn, yn;   // kRecomputeNf: what stage C needs to rebuild `nf` (hit|vis input of prob_embed.0, sampling position)
};

struct ViewProj {     // projection of one sample point into one source view
  float px, py, r;    // full-image pixel, distance to the camera
  float xn, yn;       // normalised sampling position
  float dd[4];        // [prj_dir - que_dir, <prj_dir, que_dir>]
  Bilin brf;          // tap setup of the ray_feats / projection map
};

struct PointGeom {
  float p[3];       // world position
  float qdir[3];    // -d/|d|
  float near_hw, far_hw;   // half widths of the sample's interval in normalised inverse depth
};

// ---------------------------------------------------------------------------------------------
// Stage A for NT views starting at V0: projection, gathers, dist decoder, prob_embed, neuray_fc,
// ray_dir_fc.  Fills vs[V0..V0+NT).
// V = views held by THIS wave (the per-instance policies are keyed on it), VT = views of the frame, vb = global index of the
// wave's first view: the wave-pair kernel for 4 views (render_pass_pair_kernel) runs these stages with V = 2, VT = 4, vb = 0 | 2.
template <int V, int V0, int NT, bool DBG, bool GEN, int VT = V>
__device__ __forceinline__ void stage_front(const PassArgs& p, const float* LI, const PointGeom& g, ViewState (&vs)[V], int lane,
                                            size_t pt /* ray*DN+sample */, size_t npts, bool store_ok,
                                            float* const* sdbg
#ifdef PGRF_STAGE_TIMING
                                            , unsigned long long& t_prev
#endif
                                            , int vb = 0) {
  const int half = lane >> 5;
  const float* __restrict__ W = p.blob;

  float dv[NT];
  // Two-wave kernels (<= 2 views, f32): the tap setup of the two late gathers (img_feats, rgb) is recomputed where they
  // happen instead of living through the whole stage (16 registers per view; 41.5 -> 41.0 ms).  The one-wave kernels
  // have registers to spare and would only pay the recomputation (4 views: 93.6 -> 101.6 ms).
#ifdef PGRF_NO_LAZY_TAPS
  constexpr bool kLazyTaps = false;
#else
  // (round 5: with the register pressure down the 2-view kernels for 32 and 64 samples are faster when they KEEP the setups, 34.72 ->
  // 34.62 and 17.66 -> 17.46 ms; 128 samples (72.2 vs 72.8 ms) and the pair kernel's V = 2 stages (64.5 vs 65.2 ms) still recompute)
  constexpr bool kLazyTaps = (kTwoWave<V> && kPrec == 0 && V != 3 && !(PGRF_KT <= 2 && PGRF_V == 2));
#endif
  // gathers with all their loads in flight behind scheduling fences (see gather32): measured SLOWER for the second view pair of
  // the 4-view kernel (78.4 -> 79.7 ms, both pairs 80.5 ms; 163 -> 187 spilled registers) although that pair's gathers are
  // scheduled four loads at a time -- the other wave of the SIMD already hides those waits.  Off; experiment builds only.
#ifdef PGRF_BATCH_GATHER
  constexpr bool kBatchGather = ((PGRF_BATCH_GATHER) >> (V0 > 0 ? 1 : 0)) & 1;          // bit 0 = first group, bit 1 = later groups
#else
  constexpr bool kBatchGather = false;
#endif
#ifdef PGRF_NO_DEDUPE
  constexpr bool kDedupe = false;
#else
  constexpr bool kDedupe = (NT == 2);
#endif
  Bilin brf[NT], bfeat_e[NT], bimg_e[NT];
  float pxy[NT][2];
  const float* prj[NT];
  auto project_view = [&](int v) {
    ViewProj o;
    const float* M = p.w2c + v * 12;
    // world -> source camera  (render_ops.py:158-191)
    const float cx = M[0] * g.p[0] + M[1] * g.p[1] + M[2] * g.p[2] + M[3];
    const float cy = M[4] * g.p[0] + M[5] * g.p[1] + M[6] * g.p[2] + M[7];
    const float cz = M[8] * g.p[0] + M[9] * g.p[1] + M[10] * g.p[2] + M[11];
    const float r = sqrtf(cx * cx + cy * cy + cz * cz);
    // cartesian -> sphere -> ERP pixel in the dataset's convention (spt_utils.py:128-156, :172-192)
    float px, py;
    if (!GEN || p.dataset == PGRF_DATASET_M3D) {
      float th = atan2f(cz, cx) + PGRF_HALF_PI_F;
      if (th < 0.f) th += PGRF_TWO_PI_F;                      // python-style remainder, |th| < 2pi
      const float ph = acosf(cy / (r + 1e-5f));
      px = th / PGRF_TWO_PI_F * (float)(p.W - 1);
      py = ph / PGRF_PI_F * (float)(p.H - 1);
    } else if (p.dataset == PGRF_DATASET_REPLICA_TEST) {
      const float th = atan2f(cx, cz);
      const float ph = -asinf(cy / r);
      px = (float)((double)(p.W - 1) / (2.0 * 3.14159265358979323846)) * (th + PGRF_PI_F);
      py = (float)((double)(p.H - 1) / 3.14159265358979323846) * (-ph + PGRF_HALF_PI_F);
    } else if (p.dataset == PGRF_DATASET_RESIDENTIAL) {
      float th = -atan2f(-cz, cx);
      const float ph = asinf(cy / r);
      if (th > PGRF_HALF_PI_F && th <= PGRF_TWO_PI_F) th -= PGRF_TWO_PI_F;
      px = (0.15915494309189535f * th + 0.75f) * (float)(p.W - 1);
      py = (0.5f - ph / PGRF_PI_F) * (float)(p.H - 1);
    } else {   // CoffeeArea
      float th = atan2f(cy, cx);
      const float ph = acosf(cz / r);
      if (th < 0.f) th += PGRF_TWO_PI_F;
      px = (float)(p.W - 1) * (1.f - th / PGRF_TWO_PI_F);
      py = ph * (float)(p.H - 1) / PGRF_PI_F;
    }
    o.px = px; o.py = py; o.r = r;
    // direction towards the source camera centre  (render_ops.py:195-209): c = -R^T t
    const float ccx = -(M[0] * M[3] + M[4] * M[7] + M[8] * M[11]);
    const float ccy = -(M[1] * M[3] + M[5] * M[7] + M[9] * M[11]);
    const float ccz = -(M[2] * M[3] + M[6] * M[7] + M[10] * M[11]);
    float dx = g.p[0] - ccx, dy = g.p[1] - ccy, dz = g.p[2] - ccz;
    const float dn_ = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-5f);
    { const float rd = -rcp1(dn_); dx *= rd; dy *= rd; dz *= rd; }
    o.dd[0] = dx - g.qdir[0];
    o.dd[1] = dy - g.qdir[1];
    o.dd[2] = dz - g.qdir[2];
    o.dd[3] = dx * g.qdir[0] + dy * g.qdir[1] + dz * g.qdir[2];   // aggregate_net.py:8-14
    // bilinear gathers (render_ops.py:243-245, renderer.py:180-188)
    float xn, yn;
    norm_pos(px, py, p.H, p.W, xn, yn);
    o.brf = bilin_setup(xn, yn, p.hr, p.wr, p.hr == p.H && p.wr == p.W);
    o.xn = xn; o.yn = yn;          // normalised position (bit-identical tap setups later)
    return o;
  };
  ViewProj vp[NT];
  if constexpr (kDedupe) {
    // both lane halves hold the same point: half 0 projects it into view V0, half 1 into view V0+1, and one
    // v_permlane32_swap per value hands each result to the other half (instead of both halves computing both views)
    const ViewProj mine = project_view(vb + V0 + half);
    auto both = [&](float x, float& x0, float& x1) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      x0 = __uint_as_float(r[0]); x1 = __uint_as_float(r[1]);
    };
    auto bothi = [&](int x, int& x0, int& x1) {
      const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
      x0 = (int)r[0]; x1 = (int)r[1];
    };
    both(mine.px, vp[0].px, vp[1].px); both(mine.py, vp[0].py, vp[1].py); both(mine.r, vp[0].r, vp[1].r);
    both(mine.xn, vp[0].xn, vp[1].xn); both(mine.yn, vp[0].yn, vp[1].yn);
#pragma unroll
    for (int c = 0; c < 4; ++c) both(mine.dd[c], vp[0].dd[c], vp[1].dd[c]);
    bothi(mine.brf.o00, vp[0].brf.o00, vp[1].brf.o00); bothi(mine.brf.o01, vp[0].brf.o01, vp[1].brf.o01);
    bothi(mine.brf.o10, vp[0].brf.o10, vp[1].brf.o10); bothi(mine.brf.o11, vp[0].brf.o11, vp[1].brf.o11);
    both(mine.brf.wnw, vp[0].brf.wnw, vp[1].brf.wnw); both(mine.brf.wne, vp[0].brf.wne, vp[1].brf.wne);
    both(mine.brf.wsw, vp[0].brf.wsw, vp[1].brf.wsw); both(mine.brf.wse, vp[0].brf.wse, vp[1].brf.wse);
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) vp[t] = project_view(vb + V0 + t);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int v = V0 + t;
    const float px = vp[t].px, py = vp[t].py, r = vp[t].r, xn = vp[t].xn, yn = vp[t].yn;
    dv[t] = r;
#pragma unroll
    for (int c = 0; c < 4; ++c) vs[v].dd[c] = vp[t].dd[c];
    brf[t] = vp[t].brf;
    prj[t] = p.prj + (size_t)(vb + v) * p.hr * p.wr * kPrjCh;
    pxy[t][0] = xn; pxy[t][1] = yn;
    if constexpr (!kLazyTaps) {
      bfeat_e[t] = bilin_setup(xn, yn, p.hf, p.wf, p.hf == p.H && p.wf == p.W);
      bimg_e[t] = bilin_setup(xn, yn, p.H, p.W, true);
    }
    if (DBG && store_ok) {
      const size_t o = (size_t)(vb + v) * npts + pt;
      if (sdbg[0] && half == 0) { sdbg[0][o * 2] = px; sdbg[0][o * 2 + 1] = py; }
      if (sdbg[1] && half == 0) sdbg[1][o] = r;
      if (sdbg[2] && half == 0) for (int c = 0; c < 4; ++c) sdbg[2][o * 4 + c] = vs[v].dd[c];
      if (sdbg[5]) dump_tile32(sdbg[5] + o * 32, gather32(p.ray_feats + (size_t)(vb + v) * p.hr * p.wr * 32, brf[t], half), half);
    }
  }

  PGRF_TICK(16);  // A: projection + tap setup
  // ---- mixture-of-logistics decoder: 3 (4) MLPs 32 -> 32 -> 32 -> head  (dist_decoder.py:53-107)
  // layer 0 of each MLP is read from the per-pixel projection map (pre-activation incl. bias): no MFMA
  float head[NT][6];
  float hm[6];          // kDedupe: the heads of the view this lane half owns
  hm[5] = 1.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) head[t][5] = 1.f;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (m == 3 && !p.use_vis) break;
    f32x16 a2[NT][1];
    float b2[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x16 a1 = gather32<kPrjCh, kBatchGather>(prj[t] + (m < 3 ? m : 4) * 32, brf[t], half);
#pragma unroll
      for (int s = 0; s < 16; ++s) b2[t][s] = elu1(a1[s]);
    }
#ifdef PGRF_DD_FENCE     // experiment: all 16 blended + activated operands of both views first, then the layer's MFMAs back to back
    __builtin_amdgcn_sched_barrier(0);
#endif
    mfma_layer<16, 1, NT, true>(W + layer_a_off(L_DD2_0 + m), W + layer_b_off(L_DD2_0 + m), lane, b2, a2, W + layer_h_off(L_DD2_0 + m), PGRF_LIMG(L_DD2_0 + m));
#if defined(PGRF_DD_FENCE) && PGRF_DD_FENCE >= 2
    __builtin_amdgcn_sched_barrier(0);
#endif
    if constexpr (kDedupe) {
      // heads of the view this lane half owns (half 0: view V0, half 1: view V0+1): activation and everything after it
      // is computed once per view instead of in both halves
      elu16(a2[0][0]); elu16(a2[1][0]);
      auto mine = [&](int k) {
        return pair_mine(head_partial<16>(PGRF_LHEAD(OFF_DDH_W) + k * 32, half, a2[0][0]),
                         head_partial<16>(PGRF_LHEAD(OFF_DDH_W) + k * 32, half, a2[1][0])) + W[OFF_DDH_B + k];
      };
      if (m == 0) { hm[0] = softplus1(mine(0)); hm[1] = softplus1(mine(1)); }
      else if (m == 1) { hm[2] = softplus1(mine(2)) + 0.05f; hm[3] = softplus1(mine(3)) + 0.05f; }
      else if (m == 2) hm[4] = sigmoid1(mine(4));
      else hm[5] = sigmoid1(mine(5));
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        elu16(a2[t][0]);
        if (m == 0) {
          head[t][0] = softplus1(head_dot<16>(PGRF_LHEAD(OFF_DDH_W) + 0 * 32, half, a2[t][0]) + W[OFF_DDH_B + 0]);
          head[t][1] = softplus1(head_dot<16>(PGRF_LHEAD(OFF_DDH_W) + 1 * 32, half, a2[t][0]) + W[OFF_DDH_B + 1]);
        } else if (m == 1) {   // softplus + AddBias(0.05)  (dist_decoder.py:73-81)
          head[t][2] = softplus1(head_dot<16>(PGRF_LHEAD(OFF_DDH_W) + 2 * 32, half, a2[t][0]) + W[OFF_DDH_B + 2]) + 0.05f;
          head[t][3] = softplus1(head_dot<16>(PGRF_LHEAD(OFF_DDH_W) + 3 * 32, half, a2[t][0]) + W[OFF_DDH_B + 3]) + 0.05f;
        } else if (m == 2) {
          head[t][4] = sigmoid1(head_dot<16>(PGRF_LHEAD(OFF_DDH_W) + 4 * 32, half, a2[t][0]) + W[OFF_DDH_B + 4]);
        } else {
          head[t][5] = sigmoid1(head_dot<16>(PGRF_LHEAD(OFF_DDH_W) + 5 * 32, half, a2[t][0]) + W[OFF_DDH_B + 5]);
        }
      }
    }
  }
  PGRF_TICK(17);  // A: dist decoder MLPs
  // ---- compute_prob (dist_decoder.py:6-51 is_ref branch, :109-140)
  float hvs[NT];         // (hit - 0.5) * 2 in lane half 0, (vis - 0.5) * 2 in lane half 1: the k-step-16 input of prob_embed.0
  auto prob_of = [&](int v, float depth_v, const float (&hd)[6], float& visib, float& hitp) {
    const float rn_ = -1.f / p.ref_dr[2 * v], rf_ = -1.f / p.ref_dr[2 * v + 1];
    const float z = (-1.f / fmaxf(depth_v, 1e-5f) - rn_) / (rf_ - rn_);
    const float nearp = z - g.near_hw, farp = z + g.far_hw;
    const float aw = hd[4];
    visib = 0.f; hitp = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float mean = hd[k], var = hd[2 + k];
      float c0 = cdf1((nearp - mean) * var);
      float c1 = cdf1((farp - mean) * var);
      if (p.use_vis) { c0 *= hd[5]; c1 *= hd[5]; }
      const float mix = (k == 0) ? aw : 1.f - aw;
      visib += (1.f - c0) * mix;
      hitp += (c1 - c0) * mix;
    }
  };
  if constexpr (kDedupe) {
    float visib, hitp;
    prob_of(vb + V0 + half, half ? dv[1] : dv[0], hm, visib, hitp);
    // half 0 holds (hit, vis) of view V0, half 1 of view V0+1; one swap puts hit into the lower and vis into the upper
    // half of each view's register, which is exactly the operand layout prob_embed.0 wants
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint((hitp - 0.5f) * 2.f), __float_as_uint((visib - 0.5f) * 2.f), false, false);
    hvs[0] = __uint_as_float(r[0]);
    hvs[1] = __uint_as_float(r[1]);
    if (DBG && store_ok) {
      const size_t o = (size_t)(vb + V0 + half) * npts + pt;
      if (sdbg[3]) sdbg[3][o] = visib;
      if (sdbg[4]) sdbg[4][o] = hitp;
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int v = vb + V0 + t;
      float visib, hitp;
      prob_of(v, dv[t], head[t], visib, hitp);
      hvs[t] = half ? (visib - 0.5f) * 2.f : (hitp - 0.5f) * 2.f;
      if (DBG && store_ok && half == 0) {
        const size_t o = (size_t)v * npts + pt;
        if (sdbg[3]) sdbg[3][o] = visib;
        if (sdbg[4]) sdbg[4][o] = hitp;
      }
    }
  }
  PGRF_TICK(18);  // A: compute_prob
  // ---- prob_embed: [ray_feats(32), 2(hit-.5), 2(vis-.5)] -> 32 relu -> 32  (aggregate_net.py:53-61)
  {
    // the 32 ray-feature columns (+ bias) come from the projection map; only k-step 16 = (hit, vis) is left
    f32x16 a1[NT][1], a2[NT][1];
    const float a16 = kHasImg<V> ? (LI + img_layer_off(L_PE1, kImgMode<V>))[lane * 4]
                                 : reinterpret_cast<const float4*>(W + layer_a_off(L_PE1))[4 * 64 + lane].x;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x16 pe = gather32<kPrjCh, kBatchGather>(prj[t] + 3 * 32, brf[t], half);
      const float hv = hvs[t];
      if constexpr (kRecomputeNf<V>) { vs[V0 + t].hv = hv; vs[V0 + t].xn = pxy[t][0]; vs[V0 + t].yn = pxy[t][1]; }
      a1[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a16, hv, pe, 0, 0, 0);
    }
    float b2[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 16; ++s) b2[t][s] = fmaxf(a1[t][0][s], 0.f);
    // prob_embed.2 is folded into neuray_fc.0 and base_fc.0 on the host (pack_weights.cpp): the state kept per view
    // is r = relu(prob_embed.0(.)).  Debug builds still evaluate the layer, for the "neuray" dump only.
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 16; ++s) vs[V0 + t].nf[s] = b2[t][s];
    if (DBG && sdbg[8]) {
      mfma_layer<16, 1, NT, true>(W + layer_a_off(L_PE2), W + layer_b_off(L_PE2), lane, b2, a2, W + layer_h_off(L_PE2), PGRF_LIMG(L_PE2));
      if (store_ok)
#pragma unroll
        for (int t = 0; t < NT; ++t) dump_tile32(sdbg[8] + (pt * VT + (vb + V0 + t)) * 32, a2[t][0], half);
    }
  }
  PGRF_TICK(19);  // A: prob_embed
  // ---- neuray_fc 32 -> 8 elu -> 1, weight0 = sigmoid(.) * w   (ibrnet.py:337-338)
  {
    float b16[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 16; ++s) b16[t][s] = vs[V0 + t].nf[s];
    const float wgt = 1.f / ((float)VT + 1e-8f);
    if constexpr (kSmallV<V>) {
      float o8[NT][8];
      mfma_small<16, 2, NT>(W + OFF_S_NF1_A, W + OFF_S_NF1_B, lane, b16, o8);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float y = W[OFF_NFH_B];
#pragma unroll
        for (int o = 0; o < 8; ++o) y = fmaf(W[OFF_S_NF1_H + o], elu1(o8[t][o]), y);
        vs[V0 + t].w0 = sigmoid1(y) * wgt;
      }
    } else {
      f32x16 a1[NT][1];
      mfma_layer<16, 1, NT, true>(W + layer_a_off(L_NF1), W + layer_b_off(L_NF1), lane, b16, a1, W + layer_h_off(L_NF1), PGRF_LIMG(L_NF1));
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x16 h = a1[t][0];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = elu1(h[r]);
        const float y = head_dot<4>(PGRF_LHEAD(OFF_NFH_W), half, h) + W[OFF_NFH_B];
        vs[V0 + t].w0 = sigmoid1(y) * wgt;
      }
    }
  }
  PGRF_TICK(20);  // A: neuray_fc
  // ---- ray_dir_fc 4 -> 16 elu -> 35 elu; rgb_feat = [rgb, img_feats] + dir_feat  (ibrnet.py:325-327)
  {
    float b2[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      b2[t][0] = half ? vs[V0 + t].dd[1] : vs[V0 + t].dd[0];
      b2[t][1] = half ? vs[V0 + t].dd[3] : vs[V0 + t].dd[2];
    }
    f32x16 a1[NT][1], a2[NT][2];
    mfma_layer<2, 1, NT, true>(W + layer_a_off(L_RD1), W + layer_b_off(L_RD1), lane, b2, a1, W + layer_h_off(L_RD1), PGRF_LIMG(L_RD1));
    float b8[NT][8];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 8; ++s) b8[t][s] = elu1(a1[t][0][s]);
    float rgbdir[NT][4];      // ray_dir_fc.2 rows 0..2: the direction feature added to the source colour
    if constexpr (kSmallV<V>) {
      f32x16 a2m[NT][1];
      mfma_layer<8, 1, NT, true>(W + layer_a_off(L_RD2), W + layer_b_off(L_RD2), lane, b8, a2m, W + layer_h_off(L_RD2), PGRF_LIMG(L_RD2));
      mfma_small<8, 1, NT>(W + OFF_S_RDC_A, W + OFF_S_RDC_B, lane, b8, rgbdir);
#pragma unroll
      for (int t = 0; t < NT; ++t) a2[t][0] = a2m[t][0];
    } else {
      mfma_layer<8, 2, NT, true>(W + layer_a_off(L_RD2), W + layer_b_off(L_RD2), lane, b8, a2, W + layer_h_off(L_RD2), PGRF_LIMG(L_RD2));
      // rows 0..2 of tile 1 sit in lane half 0 only; the packed base_fc operands want the colour part in both halves
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const unsigned x = __float_as_uint(a2[t][1][c]);
          rgbdir[t][c] = __uint_as_float(__builtin_amdgcn_permlane32_swap(x, x, false, false)[0]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int v = V0 + t;
      Bilin bfeat[NT], bimg[NT];
      if constexpr (kLazyTaps) {
        bfeat[t] = bilin_setup(pxy[t][0], pxy[t][1], p.hf, p.wf, p.hf == p.H && p.wf == p.W);
        bimg[t] = bilin_setup(pxy[t][0], pxy[t][1], p.H, p.W, true);
      } else {
        bfeat[t] = bfeat_e[t];
        bimg[t] = bimg_e[t];
      }
      const f32x16 imf = gather32<32, kBatchGather>(p.img_feats + (size_t)(vb + v) * p.hf * p.wf * 32, bfeat[t], half);
      const float* ib = p.imgs + (size_t)(vb + v) * p.H * p.W * 4;
      const float4 c00 = *reinterpret_cast<const float4*>(ib + (size_t)bimg[t].o00 * 4);
      const float4 c01 = *reinterpret_cast<const float4*>(ib + (size_t)bimg[t].o01 * 4);
      const float4 c10 = *reinterpret_cast<const float4*>(ib + (size_t)bimg[t].o10 * 4);
      const float4 c11 = *reinterpret_cast<const float4*>(ib + (size_t)bimg[t].o11 * 4);
      vs[v].rgb[0] = c00.x * bimg[t].wnw + c01.x * bimg[t].wne + c10.x * bimg[t].wsw + c11.x * bimg[t].wse;
      vs[v].rgb[1] = c00.y * bimg[t].wnw + c01.y * bimg[t].wne + c10.y * bimg[t].wsw + c11.y * bimg[t].wse;
      vs[v].rgb[2] = c00.z * bimg[t].wnw + c01.z * bimg[t].wne + c10.z * bimg[t].wsw + c11.z * bimg[t].wse;
#pragma unroll
      for (int r = 0; r < 16; ++r) vs[v].fa[r] = imf[r] + elu1(a2[t][0][r]);
#pragma unroll
      for (int c = 0; c < 3; ++c) vs[v].fb[c] = vs[v].rgb[c] + elu1(rgbdir[t][c]);
      if (DBG && store_ok) {
        const size_t o = (size_t)(vb + v) * npts + pt;
        if (sdbg[7]) dump_tile32(sdbg[7] + o * 32, imf, half);
        if (sdbg[6] && half == 0) for (int c = 0; c < 3; ++c) sdbg[6][o * 3 + c] = vs[v].rgb[c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Stage C for NT views: base_fc (per-view part, onto the shared partial G), vis_fc, vis_fc2.
template <int V, int V0, int NT, bool DBG, int VT = V>
__device__ __forceinline__ void stage_mid(const PassArgs& p, const float* LI, const f32x16 (&G)[2], ViewState (&vs)[V], int lane,
                                          size_t pt, bool store_ok, float* const* sdbg, int vb = 0) {
  const int half = lane >> 5;
  const float* __restrict__ W = p.blob;
  const float wgt = 1.f / ((float)VT + 1e-8f);
  if constexpr (kRecomputeNf<V>) {
    const float a16 = kHasImg<V> ? (LI + img_layer_off(L_PE1, kImgMode<V>))[lane * 4]
                                 : reinterpret_cast<const float4*>(W + layer_a_off(L_PE1))[4 * 64 + lane].x;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int v = V0 + t;
      const Bilin b = bilin_setup(vs[v].xn, vs[v].yn, p.hr, p.wr, p.hr == p.H && p.wr == p.W);
      const f32x16 pe = gather32<kPrjCh>(p.prj + (size_t)(vb + v) * p.hr * p.wr * kPrjCh + 3 * 32, b, half);
      const f32x16 a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a16, vs[v].hv, pe, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s) vs[v].nf[s] = fmaxf(a1[s], 0.f);
    }
  }
  f32x16 h[NT][2];
  {
    float b34[NT][34];      // [img feats 16 | (r|g) (b|-) | prob_embed feats 16]: pgrf_layout.h / pack_weights.cpp L_BV
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int v = V0 + t;
#pragma unroll
      for (int s = 0; s < 16; ++s) b34[t][s] = vs[v].fa[s];
      b34[t][16] = half ? vs[v].fb[1] : vs[v].fb[0];
      b34[t][17] = half ? 0.f : vs[v].fb[2];
#pragma unroll
      for (int s = 0; s < 16; ++s) b34[t][18 + s] = vs[v].nf[s];
      h[t][0] = G[0];
      h[t][1] = G[1];
    }
    mfma_layer<34, 2, NT, false>(W + layer_a_off(L_BV), nullptr, lane, b34, h, W + layer_h_off(L_BV), PGRF_LIMG(L_BV));
  }
  f32x16 x[NT][1];
  {
    float b32[NT][32];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        b32[t][s] = elu1(h[t][0][s]);
        b32[t][16 + s] = elu1(h[t][1][s]);
      }
    mfma_layer<32, 1, NT, true>(W + layer_a_off(L_B2), W + layer_b_off(L_B2), lane, b32, x, W + layer_h_off(L_B2), PGRF_LIMG(L_B2));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      elu16(x[t][0]);
      if (DBG && store_ok && sdbg[9]) dump_tile32(sdbg[9] + (pt * VT + (vb + V0 + t)) * 32, x[t][0], half);
    }
  }
  // vis_fc(x * weight): 32 -> 32 elu -> 33 elu; x += x_res; vis = sigmoid(.)   (ibrnet.py:347-350)
  float vis1[NT];
  {
    float b16[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 16; ++s) b16[t][s] = x[t][0][s] * wgt;
    f32x16 a1[NT][1], a2[NT][1];
    mfma_layer<16, 1, NT, true>(W + layer_a_off(L_VF1), W + layer_b_off(L_VF1), lane, b16, a1, W + layer_h_off(L_VF1), PGRF_LIMG(L_VF1));
    float c16[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      elu16(a1[t][0]);
#pragma unroll
      for (int s = 0; s < 16; ++s) c16[t][s] = a1[t][0][s];
    }
    mfma_layer<16, 1, NT, true>(W + layer_a_off(L_VF2), W + layer_b_off(L_VF2), lane, c16, a2, W + layer_h_off(L_VF2), PGRF_LIMG(L_VF2));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float raw = head_dot<16>(PGRF_LHEAD(OFF_VFH_W), half, a1[t][0]) + W[OFF_VFH_B];
      vis1[t] = sigmoid1(elu1(raw));
#pragma unroll
      for (int r = 0; r < 16; ++r) x[t][0][r] += elu1(a2[t][0][r]);
    }
  }
  // vis_fc2(x * vis): 32 -> 32 elu -> 1 sigmoid   (ibrnet.py:351)
  {
    float b16[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 16; ++s) b16[t][s] = x[t][0][s] * vis1[t];
    f32x16 a1[NT][1];
    mfma_layer<16, 1, NT, true>(W + layer_a_off(L_V21), W + layer_b_off(L_V21), lane, b16, a1, W + layer_h_off(L_V21), PGRF_LIMG(L_V21));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      elu16(a1[t][0]);
      vs[V0 + t].v2 = sigmoid1(head_dot<16>(PGRF_LHEAD(OFF_V2H_W), half, a1[t][0]) + W[OFF_V2H_B]);
      vs[V0 + t].fa = x[t][0];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Stage E for NT views: rgb_fc on [x, vis, ray_diff] -> blending logit   (ibrnet.py:365-366)
template <int V, int V0, int NT>
__device__ __forceinline__ void stage_rgb(const PassArgs& p, const float* LI, const ViewState (&vs)[V], int lane, float (&logit)[V]) {
  const int half = lane >> 5;
  const float* __restrict__ W = p.blob;
  float b19[NT][19];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int v = V0 + t;
#pragma unroll
    for (int s = 0; s < 16; ++s) b19[t][s] = vs[v].fa[s];
    b19[t][16] = half ? vs[v].dd[0] : vs[v].v2;
    b19[t][17] = half ? vs[v].dd[2] : vs[v].dd[1];
    b19[t][18] = half ? 0.f : vs[v].dd[3];
  }
  f32x16 a1[NT][1], a2[NT][1];
  float b8[NT][8];
  if constexpr (kW16) {
    float o16[NT][8];
    mfma_layer16<19, NT>(W + OFF_W16_RGB1, PGRF_LHEAD(OFF_W16B_RGB1), lane, b19, o16);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 8; ++s) b8[t][s] = elu1(o16[t][s]);
  } else {
    mfma_layer<19, 1, NT, true>(W + layer_a_off(L_RGB1), W + layer_b_off(L_RGB1), lane, b19, a1, W + layer_h_off(L_RGB1), PGRF_LIMG(L_RGB1));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 8; ++s) b8[t][s] = elu1(a1[t][0][s]);
  }
  if constexpr (kSmallV<V>) {
    float o8[NT][8];
    mfma_small<8, 2, NT>(W + OFF_S_RGB2_A, W + OFF_S_RGB2_B, lane, b8, o8);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float y = W[OFF_RGBH_B];
#pragma unroll
      for (int o = 0; o < 8; ++o) y = fmaf(W[OFF_S_RGB2_H + o], elu1(o8[t][o]), y);
      logit[V0 + t] = y;
    }
  } else {
    mfma_layer<8, 1, NT, true>(W + layer_a_off(L_RGB2), W + layer_b_off(L_RGB2), lane, b8, a2, W + layer_h_off(L_RGB2), PGRF_LIMG(L_RGB2));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x16 hh = a2[t][0];
#pragma unroll
      for (int r = 0; r < 4; ++r) hh[r] = elu1(hh[r]);
      logit[V0 + t] = head_dot<4>(PGRF_LHEAD(OFF_RGBH_W), half, hh) + W[OFF_RGBH_B];
    }
  }
}

// views are processed in groups sharing the weight operand: first group kNT0 views, the rest in a second group
// (measured: all views in ONE group — 3 views 80.5 -> 74.0 ms/frame, 4 views 97.3 -> 96.0; -DPGRF_NT_PAIRS = groups of 2)
#ifdef PGRF_NT_PAIRS
template <int V> constexpr int kNT0 = (V >= 2 ? 2 : 1);
#else
// (at two waves: 3 views in one group 60.1 ms vs 63.5 in groups 2+1; 4 views in one group 88.2 ms vs 85.2 in pairs)
#ifdef PGRF_NT_ALL
template <int V> constexpr int kNT0 = V;
#elif defined(PGRF_NT_PAIRS)
template <int V> constexpr int kNT0 = V < 2 ? V : 2;
#else
template <int V> constexpr int kNT0 = (V <= 3 || !kTwoWave<V>) ? V : 2;
#endif
#endif
// the two fences between view groups (after stage A's and stage C's first group), which for <= 2 views are simply two more stage fences
#ifdef PGRF_MID_FENCE
template <int V> constexpr bool kMidFence = (PGRF_MID_FENCE) != 0;
#else
template <int V> constexpr bool kMidFence = true;
#endif
constexpr int kKStride = 17;   // K rows in LDS: odd stride -> conflict-free column reads (A operand)
constexpr int kVStride = 20;   // V rows: 16-byte aligned rows for broadcast float4 reads

// ---------------------------------------------------------------------------------------------
// Per-point tail of stage D: geometry_fc.0 pre-activations `g1` -> geometry_fc.2 -> positional encoding -> q|k|v -> 4-head
// attention over the samples of the ray (K, V of the workgroup's points in LDS) -> fc + LayerNorm -> out_geometry_fc -> density.
// pt_local = the point's row in lds_k / lds_v, key_base = first row of its ray.  Contains one workgroup barrier.
// HSEL >= 0 (wave-pair kernel): this wave evaluates heads 2 HSEL, 2 HSEL + 1 only and fetches the other two from its partner
// (column tid ^ 64 of the exchange rows `xo`; one more barrier); HSEL = 0 writes K, HSEL = 1 writes V.
template <int KT, int V, bool DBG, int VT = V, int HSEL = -1>
__device__ __forceinline__ float point_tail(const PassArgs& p, const float* LI, const f32x16 (&g1)[1][2], int lane, int si, size_t pt,
                                            int pt_local, int key_base, float* lds_k, float* lds_v, bool store_ok, float* const* sdbg
#ifdef PGRF_STAGE_TIMING
                                            , unsigned long long& t_prev
#endif
                                            , float (*xo)[256] = nullptr, int tid = 0) {
  const int half = lane >> 5;
  const int pl = lane & 31;
  const float* __restrict__ W = p.blob;
  float b32[1][32];
#pragma unroll
  for (int s = 0; s < 16; ++s) { b32[0][s] = elu1(g1[0][0][s]); b32[0][16 + s] = elu1(g1[0][1][s]); }
  f32x16 g2[1][1];
  if constexpr (kW16) {
    float o16[1][8];
    mfma_layer16<32, 1>(W + OFF_W16_G2, PGRF_LHEAD(OFF_W16B_G2), lane, b32, o16);
#pragma unroll
    for (int r = 0; r < 8; ++r) g2[0][0][r] = o16[0][r];
  } else {
    mfma_layer<32, 1, 1, true>(W + layer_a_off(L_G2), W + layer_b_off(L_G2), lane, b32, g2, W + layer_h_off(L_G2), PGRF_LIMG(L_G2));
  }
  float gp[1][8];     // geometry feature + positional encoding (ibrnet.py:356-358)
  {
    const float4* pp = reinterpret_cast<const float4*>(W + OFF_POS + si * 16 + half * 8);
    const float4 p0 = pp[0], p1 = pp[1];
    const float pe[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float ge = elu1(g2[0][0][r]);
      g2[0][0][r] = ge;
      gp[0][r] = ge + pe[r];
    }
    if (DBG && store_ok && sdbg[10]) dump_tile16(sdbg[10] + pt * 16, g2[0][0], half);
  }
  // q|k and v projections (ibrnet.py:80-82); rows permuted so register 2*head+s carries dims (2s,2s+1)
  f32x16 qkv[1][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { qkv[0][0][r] = 0.f; qkv[0][1][r] = 0.f; }   // no bias on w_qs/w_ks/w_vs
  if constexpr (kW16) {       // q | k on the 32-row tile, v (16 rows) on the 16x16x4 MFMA
    f32x16 qk[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) qk[0][0][r] = 0.f;
    mfma_layer<8, 1, 1, false>(W + layer_a_off(L_QKV), nullptr, lane, gp, qk, W + layer_h_off(L_QKV), PGRF_LIMG(L_QKV));
    qkv[0][0] = qk[0][0];
    float v8[1][8];
    mfma_layer16<8, 1>(W + OFF_W16_V, nullptr, lane, gp, v8);
#pragma unroll
    for (int r = 0; r < 8; ++r) qkv[0][1][r] = v8[0][r];
  } else {
    mfma_layer<8, 2, 1, false>(W + layer_a_off(L_QKV), nullptr, lane, gp, qkv, W + layer_h_off(L_QKV), PGRF_LIMG(L_QKV));
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (HSEL != 1) lds_k[pt_local * kKStride + 2 * r + half] = qkv[0][0][8 + r];
    if (HSEL != 0) lds_v[pt_local * kVStride + 2 * r + half] = qkv[0][1][r];
  }
  PGRF_TICK(4);   // stage D up to the q/k/v projections
  PGRF_SYNC();
  PGRF_TICK(5);   // barrier wait
  float oatt[16];
#pragma unroll
  for (int hd = (HSEL < 0 ? 0 : 2 * HSEL); hd < (HSEL < 0 ? 4 : 2 * HSEL + 2); ++hd) {
    f32x16 S[KT];
#pragma unroll
    for (int k2 = 0; k2 < KT; ++k2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[k2][r] = 0.f;
      // One source view: the reference masks the attention with (num_valid_obs > 1), which is false everywhere, and
      // masked_fill(mask == 0, -1e9) makes every logit equal (ibrnet.py:20,358-360): uniform weights, the output is
      // the mean of V over the ray's samples.  Equal logits = the zero-initialised S, so the products are skipped.
      if constexpr (VT > 1)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float a = lds_k[(key_base + k2 * 32 + pl) * kKStride + hd * 4 + 2 * s + half];
        // q / temperature (sqrt(d_k) = 2), times log2(e) so that the softmax below is a bare v_exp_f32
        const float q = qkv[0][0][2 * hd + s] * (0.5f * 1.4426950408889634f);
        S[k2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, q, S[k2], 0, 0, 0);
      }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int k2 = 0; k2 < KT; ++k2)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[k2][r]);
    mx = xmax(mx);
    float l = 0.f, o0, o1, o2, o3;
#ifndef PGRF_PV_MFMA
#if PGRF_PV_PACKED
    {
      // the four value dimensions of a head as two v_pk_fma_f32 (5.1 cycles per pair against 2 x 4.4 for v_fmac: tools/ubench/valu_rates)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < KT; ++k2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pe = __builtin_amdgcn_exp2f(S[k2][r] - mx);
          const float4 vv = *reinterpret_cast<const float4*>(&lds_v[(key_base + k2 * 32 + rho(r, half)) * kVStride + hd * 4]);
          l += pe;
          const f32x2 pp = {pe, pe};
          o01 = __builtin_elementwise_fma(pp, (f32x2){vv.x, vv.y}, o01);
          o23 = __builtin_elementwise_fma(pp, (f32x2){vv.z, vv.w}, o23);
        }
      o0 = o01[0]; o1 = o01[1]; o2 = o23[0]; o3 = o23[1];
    }
#else
    o0 = o1 = o2 = o3 = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < KT; ++k2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pe = __builtin_amdgcn_exp2f(S[k2][r] - mx);
        const float4 vv = *reinterpret_cast<const float4*>(&lds_v[(key_base + k2 * 32 + rho(r, half)) * kVStride + hd * 4]);
        l += pe;
        o0 = fmaf(pe, vv.x, o0); o1 = fmaf(pe, vv.y, o1); o2 = fmaf(pe, vv.z, o2); o3 = fmaf(pe, vv.w, o3);
      }
#endif
#else
    // experiment (measured slower: 38.2 -> 39.0 ms): softmax-weighted values on v_mfma_f32_4x4x1_16b_f32: block = 4
    // consecutive lanes = 4 queries with the SAME key (k2, r, half); B = the lane's own weight for that key, A = value
    // dimension (lane & 3) of that key, so D leaves the 4 output dimensions of the lane's own query in registers 0..3.
    {
      f32x4 ov[KT];
#pragma unroll
      for (int k2 = 0; k2 < KT; ++k2) ov[k2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int k2 = 0; k2 < KT; ++k2) {
          const float pe = __builtin_amdgcn_exp2f(S[k2][r] - mx);
          const float vd = lds_v[(key_base + k2 * 32 + rho(r, half)) * kVStride + hd * 4 + (lane & 3)];
          l += pe;
          ov[k2] = __builtin_amdgcn_mfma_f32_4x4x1f32(vd, pe, ov[k2], 0, 0, 0);
        }
#pragma unroll
      for (int k2 = 1; k2 < KT; ++k2) ov[0] += ov[k2];
      o0 = ov[0][0]; o1 = ov[0][1]; o2 = ov[0][2]; o3 = ov[0][3];
    }
#endif
    l = xsum(l);
    o0 = xsum(o0); o1 = xsum(o1); o2 = xsum(o2); o3 = xsum(o3);
    const float rl = rcp1(l);
    oatt[hd * 4 + 0] = o0 * rl; oatt[hd * 4 + 1] = o1 * rl; oatt[hd * 4 + 2] = o2 * rl; oatt[hd * 4 + 3] = o3 * rl;
  }
  if constexpr (HSEL >= 0) {      // (every lane holds its point's outputs: the partner's column carries the same point)
#pragma unroll
    for (int i = 0; i < 8; ++i) xo[i][tid] = oatt[HSEL * 8 + i];
    PGRF_SYNC();
#pragma unroll
    for (int i = 0; i < 8; ++i) oatt[(1 - HSEL) * 8 + i] = xo[i][tid ^ 64];
  }
  PGRF_TICK(6);   // attention
  // fc + residual + LayerNorm(eps 1e-6)  (ibrnet.py:95-102)
  float bo[1][8];
#pragma unroll
  for (int s = 0; s < 8; ++s) bo[0][s] = half ? oatt[4 * (s >> 1) + 2 * (s & 1) + 1] : oatt[4 * (s >> 1) + 2 * (s & 1)];
  f32x16 fo[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) fo[0][0][r] = (r < 8) ? gp[0][r < 8 ? r : 0] : 0.f;     // residual as the initial accumulator
  if constexpr (kW16) {       // (fc(x) first, then the residual: the reference's order, ibrnet.py:95-98)
    float o16[1][8];
    mfma_layer16<8, 1>(W + OFF_W16_FC, nullptr, lane, bo, o16);
#pragma unroll
    for (int r = 0; r < 8; ++r) fo[0][0][r] = o16[0][r] + gp[0][r];
  } else {
    mfma_layer<8, 1, 1, false>(W + layer_a_off(L_FC), nullptr, lane, bo, fo, W + layer_h_off(L_FC), PGRF_LIMG(L_FC));
  }
  float ln[1][8];
  {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += fo[0][0][r];
    sum = xsum(sum);
    const float mean = sum * 0.0625f;
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { const float d = fo[0][0][r] - mean; sq += d * d; }
    sq = xsum(sq);
    const float rstd = __builtin_amdgcn_rsqf(sq * 0.0625f + 1e-6f);
#pragma unroll
    for (int r = 0; r < 8; ++r)
      ln[0][r] = (fo[0][0][r] - mean) * rstd * W[OFF_LN_G + half * 8 + r] + W[OFF_LN_B + half * 8 + r];
    if (DBG && store_ok && sdbg[11]) {
      f32x16 tmp;
#pragma unroll
      for (int r = 0; r < 8; ++r) tmp[r] = ln[0][r];
      dump_tile16(sdbg[11] + pt * 16, tmp, half);
    }
  }
  // out_geometry_fc 16 -> 16 elu -> 1 relu  (ibrnet.py:361)
  f32x16 og[1][1];
  if constexpr (kW16) {
    float o16[1][8];
    mfma_layer16<8, 1>(W + OFF_W16_OG1, PGRF_LHEAD(OFF_W16B_OG1), lane, ln, o16);
#pragma unroll
    for (int r = 0; r < 8; ++r) og[0][0][r] = o16[0][r];
  } else {
    mfma_layer<8, 1, 1, true>(W + layer_a_off(L_OG1), W + layer_b_off(L_OG1), lane, ln, og, W + layer_h_off(L_OG1), PGRF_LIMG(L_OG1));
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) og[0][0][r] = elu1(og[0][0][r]);
  return fmaxf(head_dot<8>(PGRF_LHEAD(OFF_OGH_W), half, og[0][0]) + W[OFF_OGH_B], 0.f);
}

// ---------------------------------------------------------------------------------------------
// Stage F: alpha compositing along the ray.  slot = the 32-sample tile's index inside the workgroup (tiles of one ray are
// consecutive slots, kt = the tile's position on its ray).  Contains two workgroup barriers.
template <int KT, int V>
__device__ __forceinline__ void composite(const PassArgs& p, float sigma, const float (&col)[3], float tdepth, int lane, int slot, int kt,
                                          int ray, size_t pt, bool store_ok, float* lds_alpha, float (*lds_red)[4]
#ifdef PGRF_STAGE_TIMING
                                          , unsigned long long& t_prev
#endif
                                          ) {
  const int half = lane >> 5;
  const int pl = lane & 31;
  // ---- stage F: alpha compositing along the ray (renderer.py:216-218,291-304; render_ops.py:145-153)
  const float alpha = 1.f - __expf(-fmaxf(sigma, 0.f));
  // transmittance T(i) = prod_{j<i} (1 - alpha_j + 1e-10): inclusive product scan over the wave's 32 samples (both lane
  // halves carry the same value), then the totals of the ray's earlier waves through LDS
  float incl = 1.f - alpha + 1e-10f;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const float up = __shfl_up(incl, off, 32);
    if (pl >= off) incl *= up;
  }
  float T = __shfl_up(incl, 1, 32);
  if (pl == 0) T = 1.f;
  if (KT > 1) {
    if (lane == 31) lds_alpha[slot] = incl;
    PGRF_SYNC();
  }
  PGRF_TICK(10);  // composite: barrier
#pragma unroll
  for (int k2 = 0; k2 < KT - 1; ++k2)
    if (k2 < kt) T *= lds_alpha[slot - kt + k2];
  const float hit = alpha * T;
  // NaN report (the reference's ipdb breakpoints): a NaN in density, colour or weight of any sample sets the flag
  if (p.nan_flag && store_ok && (hit != hit || sigma != sigma || col[0] != col[0] || col[1] != col[1] || col[2] != col[2]))
    *p.nan_flag = 1;
  PGRF_TICK(11);  // composite: transmittance scan
  if (store_ok && half == 0) {
    if (p.out_hit) p.out_hit[pt] = hit;
    if (p.out_density) p.out_density[pt] = sigma;
    if (p.out_colors) { p.out_colors[pt * 3] = col[0]; p.out_colors[pt * 3 + 1] = col[1]; p.out_colors[pt * 3 + 2] = col[2]; }
  }
  PGRF_TICK(12);  // composite: per-point stores
  float acc4[4] = {hit * col[0], hit * col[1], hit * col[2], hit * tdepth};
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc4[c] += __shfl_xor(acc4[c], off, 64);
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < 4; ++c) lds_red[slot][c] = acc4[c];
  PGRF_SYNC();
  if (kt == 0 && lane < 4 && store_ok) {
    float s = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < KT; ++k2) s += lds_red[slot + k2][lane];
    if (lane < 3) { if (p.out_pixel) p.out_pixel[(size_t)ray * 3 + lane] = s; }
    else if (p.out_depth) p.out_depth[ray] = s;
  }
  PGRF_TICK(9);   // composite
}

// GEN = false: the m3d ERP convention only (the shipped configuration; the other conventions and the pinhole branch
// compiled in cost 1.2 % through 5 more spilled registers); GEN = true: run-time switch over PgrfFrame.dataset /
// .perspective.
template <int KT, int V, bool DBG, bool GEN>
__global__ void __launch_bounds__(256, (kTwoWave<V> ? 2 : 1)) render_pass_kernel(const PassArgs p) {
  constexpr int DN = 32 * KT;
  constexpr int RPW = 4 / KT;          // rays per workgroup
  __shared__ __attribute__((aligned(16))) float lds_v[128 * kVStride];
  __shared__ float lds_k[128 * kKStride];
  __shared__ float lds_alpha[4];     // per-wave transmittance totals
  __shared__ float lds_red[4][4];
  __shared__ __attribute__((aligned(16))) float lds_img[kHasImg<V> ? img_floats(kImgMode<V>) : 4];
  const float* LI = lds_img;
  if constexpr (kHasImg<V>) {   // every workgroup copies the image once; the barrier sits after ray generation
    const float4* src = reinterpret_cast<const float4*>(p.blob + img_blob_off(kImgMode<V>));
    for (int i = threadIdx.x; i < img_floats(kImgMode<V>) / 4; i += 256) reinterpret_cast<float4*>(lds_img)[i] = src[i];
  }
  // debug builds: the 12 dump pointers live in LDS, not in SGPRs (24 live SGPRs made the compiler spill scalars)
  __shared__ float* lds_dbg[DBG ? 12 : 1];
  float* const* sdbg = lds_dbg;
  if (DBG) {
    if (threadIdx.x < 12) lds_dbg[threadIdx.x] = reinterpret_cast<float* const*>(&p.dbg)[threadIdx.x];
    __syncthreads();
  }

  const int lane = threadIdx.x & 63;
#ifdef PGRF_ABL_WLDS
  g_abl_w[threadIdx.x] = g_abl_w[threadIdx.x + 256] = make_float4(0.01f, -0.02f, 0.03f, 0.005f);
  __syncthreads();
#endif
#ifdef PGRF_ONE_WG_PER_CU   // experiment: 84 KB of LDS ballast -> one workgroup (one wave per SIMD) per CU
  __shared__ float lds_ballast[21000];
  if (p.rn < 0) lds_ballast[threadIdx.x * 80] = 1.f;
#endif
#ifdef PGRF_STAGE_TIMING
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // provably wave-uniform: the ray index arithmetic runs on the SALU
  const int half = lane >> 5;
  const int pl = lane & 31;
  const int ray_local = wave / KT;
  const int kt = wave % KT;
  // XCD-aware block -> ray mapping: consecutive workgroups are dispatched round-robin over the 8 XCDs, so give
  // every XCD one CONTIGUOUS span of rays (a band of image rows): its private L2 then holds one band of the
  // source maps instead of all of them.  Bijective for any grid size; speed only, never correctness.
  int wg = blockIdx.x;
#ifndef PGRF_NO_XCD_REMAP
  // Per view count, as measured at the end of round 2 (it was +0.9 % for every instance when introduced): with 3 and 4 views the
  // bands help (53.2 -> 52.8 ms, 79.0 -> 78.3 ms); with 1 and 2 views the plain round-robin is faster (2 views x 64 samples
  // 37.32 -> 37.16 ms, x 32 samples 18.97 -> 18.85 ms, split-f16 23.41 -> 22.86 ms), as it is under the tiled visiting
  // order of large source maps (334.3 -> 330.0 ms).
  if (V >= 3 && !p.order_w) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
#endif
  int ray_raw = wg * RPW + ray_local;
  if (p.order_w) {          // tiled visiting order (render_pass_args.h); wave-uniform integer arithmetic
    const int per_tile = p.order_tw * p.order_th, tiles_x = p.order_w / p.order_tw;
    const int t = ray_raw / per_tile, in = ray_raw - t * per_tile;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int iy = in / p.order_tw, ix = in - iy * p.order_tw;
    ray_raw = (ty * p.order_th + iy) * p.order_w + tx * p.order_tw + ix;
  }
  const bool ray_ok = ray_raw < p.rn;
  const int ray = ray_ok ? ray_raw : p.rn - 1;
  const int si = kt * 32 + pl;                       // sample index on the ray
  const int pt_local = ray_local * DN + si;          // 0..127
  const size_t pt = (size_t)ray * DN + si;
  const size_t npts = (size_t)p.rn * DN;
  const float* __restrict__ W = p.blob;

  // ---- ray generation (ray_utils.py:4-16,53-71; spt_utils.py:45-58,106-111; render_ops.py:76-106)
  PointGeom g;
  float tdepth;
  {
    float wx, wy, wz;
    if (p.raytab) {      // pgrf_render_rays computed the directions once per ray (one table for the coarse and the fine pass)
      const float4 rt = reinterpret_cast<const float4*>(p.raytab)[ray];
      wx = rt.x; wy = rt.y; wz = rt.z;
    } else {
      ray_direction<GEN>(p, ray, wx, wy, wz);
    }
    const float* drow = p.depth + (size_t)ray * DN;
    tdepth = drow[si];
    g.p[0] = p.c2w[3] + wx * tdepth;
    g.p[1] = p.c2w[7] + wy * tdepth;
    g.p[2] = p.c2w[11] + wz * tdepth;
    const float wn = sqrtf(wx * wx + wy * wy + wz * wz);
    { const float rw = rcp1(wn); g.qdir[0] = -wx * rw; g.qdir[1] = -wy * rw; g.qdir[2] = -wz * rw; }
    // interval half-widths in normalised inverse depth (render_ops.py:110-122, dist_decoder.py:34-38)
    const float qn = -1.f / p.que_near, qf = -1.f / p.que_far;
    const float d_c = (-1.f / tdepth - qn) / (qf - qn);
    const float d_p = (-1.f / drow[max(si - 1, 0)] - qn) / (qf - qn);
    const float d_n = (-1.f / drow[min(si + 1, DN - 1)] - qn) / (qf - qn);
    const float dist_c = (si == DN - 1) ? 1e6f : d_n - d_c;          // que_dists[i]
    const float dist_p = (si == 0) ? dist_c : d_c - d_p;             // que_dists[i-1], first repeated
    g.near_hw = dist_p / 2.f;
    g.far_hw = dist_c / 2.f;
  }

  ViewState vs[V];
  const bool store_ok = ray_ok;
  if constexpr (kHasImg<V>) __syncthreads();
  PGRF_TICK(0);   // ray generation

  // Two-wave kernels: the direction differences and source colours of every view (7 registers per view) are next
  // needed in stage E; parked in LDS in between they free enough registers that NOTHING spills any more
  // (36 -> 0 spilled registers, 12 GB less scratch traffic per frame, 41.1 -> 38.6 ms).  -DPGRF_NO_PARK disables.
#ifdef PGRF_NO_PARK
  constexpr bool kPark = false;
#else
  constexpr bool kPark = kTwoWave<V>;
#endif
  // (kLdsState: both lane halves hold the same values here, so one column per POINT is enough -- half the LDS)
  constexpr int kParkCols = kLdsState<V> ? 128 : 256;
  const int park_col = kLdsState<V> ? (wave * 32 + pl) : (int)threadIdx.x;
  __shared__ float lds_park[kPark ? V * 7 : 1][kParkCols];
  auto park_views = [&](int v0, int v1) {
#pragma unroll
    for (int v = v0; v < v1; ++v) {
#pragma unroll
      for (int c = 0; c < 4; ++c) lds_park[v * 7 + c][park_col] = vs[v].dd[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) lds_park[v * 7 + 4 + c][park_col] = vs[v].rgb[c];
    }
  };
  // ---- stage A: per-view front end, views in groups of <= 2 sharing the weight operand
  stage_front<V, 0, kNT0<V>, DBG, GEN>(p, LI, g, vs, lane, pt, npts, store_ok, sdbg
#ifdef PGRF_STAGE_TIMING
                                           , t_prev
#endif
  );
  // kLdsState (4 views, two waves per SIMD): between its production in stage A and its consumption in stages B / C the
  // 16-register feature tile `fa` of a view pair is kept in LDS, and so is the pair's stage-C output until stages D / E read
  // it -- the kernel then never holds more than ONE pair's tiles in registers, as the 2-view kernel does.  The compiler's
  // own answer to the same pressure was 157-163 spilled registers = 616-640 B of scratch per lane, written and re-read once
  // per wave: ~40 KB x 1 M waves per 512x1024 frame through L2 / the fabric.
  //   stage A(pair 0) -> LDS | stage A(pair 1) | stage B reads pair 0 from LDS | stage C(pair 1) | pair 0 LDS -> registers,
  //   pair 1's outputs -> the same LDS rows | stage C(pair 0) | stages D, E read pair 1 from LDS
  __shared__ float lds_state[kLdsState<V> ? 32 : 1][kLdsState<V> ? 256 : 1];
  if constexpr (kLdsState<V>) {
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int j = 0; j < 16; ++j) lds_state[v * 16 + j][threadIdx.x] = vs[v].fa[j];
    if constexpr (kPark) park_views(0, 2);           // (and its directions / colours: not carried through pair 1's stage A)
  }
  if constexpr (kMidFence<V>) __builtin_amdgcn_sched_barrier(0);
  if constexpr (V > kNT0<V>) stage_front<V, 2, V - 2, DBG, GEN>(p, LI, g, vs, lane, pt, npts, store_ok, sdbg
#ifdef PGRF_STAGE_TIMING
                                           , t_prev
#endif
  );
  PGRF_TICK(1);   // stage A
  if constexpr (kPark) park_views(kLdsState<V> ? 2 : 0, V);
  // feature j of view v's tile: from the registers, or (kLdsState) from the LDS rows of the pair that is parked there
  auto fa_at = [&](int v, int j, int parked_pair) -> float {
    if constexpr (kLdsState<V>) {
      if ((v >> 1) == parked_pair) return lds_reload(&lds_state[(v & 1) * 16 + j][threadIdx.x]);
    }
    return vs[v].fa[j];
  };

  if constexpr (V > 2) __builtin_amdgcn_sched_barrier(0);   // V>2: keep the stages apart (cross-stage interleaving made the compiler spill 342 VGPRs)
  // ---- stage B: view statistics (ibrnet.py:112-116,339-342) and the view-shared part of base_fc.0
  f32x16 G[1][2];
  {
    const float wgt = 1.f / ((float)V + 1e-8f);
    // k-step order of the shared block (pack_weights.cpp L_BG): the four statistics of the 32 image-feature channels
    // (16 registers each), then the colour channels two per register: (r|g) of each statistic and (b of the w0
    // statistics | b of the w statistics) — 70 k-steps, no empty lane-half slots
    float b70[1][70];
    auto stats = [&](const float (&xv)[V], const float (&wa)[V], float& m, float& var) {
      m = 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) m += xv[v] * wa[v];
      var = 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) { const float d = xv[v] - m; var += wa[v] * (d * d); }
    };
    float w0a[V], w1a[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { w0a[v] = vs[v].w0; w1a[v] = wgt; }
#pragma unroll
    for (int j = 0; j < 17; ++j) {
      float xv[V];
#pragma unroll
      for (int v = 0; v < V; ++v) xv[v] = (j < 16) ? fa_at(v, j < 16 ? j : 0, 0) : (half ? vs[v].fb[1] : vs[v].fb[0]);
      float m0, v0, m1, v1;
      stats(xv, w0a, m0, v0);
      stats(xv, w1a, m1, v1);
      if (j < 16) { b70[0][j] = m0; b70[0][16 + j] = v0; b70[0][32 + j] = m1; b70[0][48 + j] = v1; }
      else { b70[0][64] = m0; b70[0][65] = v0; b70[0][66] = m1; b70[0][67] = v1; }
    }
    {
      float xv[V], wsel[V];
#pragma unroll
      for (int v = 0; v < V; ++v) { xv[v] = vs[v].fb[2]; wsel[v] = half ? wgt : vs[v].w0; }
      stats(xv, wsel, b70[0][68], b70[0][69]);
    }
    mfma_layer<70, 2, 1, true>(W + layer_a_off(L_BG), W + layer_b_off(L_BG), lane, b70, G, W + layer_h_off(L_BG), PGRF_LIMG(L_BG));
  }
  PGRF_TICK(2);   // stage B

  if constexpr (V > 2) __builtin_amdgcn_sched_barrier(0);   // V>2: keep the stages apart (cross-stage interleaving made the compiler spill 342 VGPRs)
  // ---- stage C: per-view base_fc / vis_fc / vis_fc2
  if constexpr (kLdsState<V>) {
    stage_mid<V, 2, 2, DBG>(p, LI, G[0], vs, lane, pt, store_ok, sdbg);          // pair 1 first: its inputs are in registers
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        vs[v].fa[j] = lds_reload(&lds_state[v * 16 + j][threadIdx.x]);     // pair 0's inputs back into registers ...
        lds_restore(&lds_state[v * 16 + j][threadIdx.x], vs[2 + v].fa[j]); // ... pair 1's outputs into their rows
      }
    __builtin_amdgcn_sched_barrier(0);
    stage_mid<V, 0, 2, DBG>(p, LI, G[0], vs, lane, pt, store_ok, sdbg);
  } else {
  stage_mid<V, 0, kNT0<V>, DBG>(p, LI, G[0], vs, lane, pt, store_ok, sdbg);
  if constexpr (kMidFence<V>) __builtin_amdgcn_sched_barrier(0);
  if constexpr (V > kNT0<V>) stage_mid<V, 2, V - 2, DBG>(p, LI, G[0], vs, lane, pt, store_ok, sdbg);
  }
  PGRF_TICK(3);   // stage C

  if constexpr (V > 2) __builtin_amdgcn_sched_barrier(0);   // V>2: keep the stages apart (cross-stage interleaving made the compiler spill 342 VGPRs)
  // ---- stage D: pooled geometry feature -> geometry_fc -> ray attention -> density
  float sigma;
  {
    float vsum = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) vsum += vs[v].v2;
    float w2[V];
    float wmean = 0.f;
    const float rvs = rcp1(vsum + 1e-8f);
#pragma unroll
    for (int v = 0; v < V; ++v) { w2[v] = vs[v].v2 * rvs; wmean += w2[v]; }
    wmean = wmean / (float)V;
    float b33[1][33];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float xr[V];
#pragma unroll
      for (int v = 0; v < V; ++v) xr[v] = fa_at(v, r, 1);
      float m = 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) m += xr[v] * w2[v];
      float var = 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) { const float d = xr[v] - m; var += w2[v] * (d * d); }
      b33[0][r] = m;
      b33[0][16 + r] = var;
    }
    b33[0][32] = wmean;
    f32x16 g1[1][2];
    mfma_layer<33, 2, 1, true>(W + layer_a_off(L_G1), W + layer_b_off(L_G1), lane, b33, g1, W + layer_h_off(L_G1), PGRF_LIMG(L_G1));
    sigma = point_tail<KT, V, DBG>(p, LI, g1, lane, si, pt, pt_local, ray_local * DN, lds_k, lds_v, store_ok, sdbg
#ifdef PGRF_STAGE_TIMING
                                  , t_prev
#endif
    );
  }
  PGRF_TICK(7);   // fc + LayerNorm + density head

  if constexpr (V > 2) __builtin_amdgcn_sched_barrier(0);   // V>2: keep the stages apart (cross-stage interleaving made the compiler spill 342 VGPRs)
  // ---- stage E: colour blending over views (ibrnet.py:364-370)
  float col[3];
  {
    float logit[V];
    if constexpr (kPark) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) vs[v].dd[c] = lds_reload(&lds_park[v * 7 + c][park_col]);
#pragma unroll
        for (int c = 0; c < 3; ++c) vs[v].rgb[c] = lds_reload(&lds_park[v * 7 + 4 + c][park_col]);
      }
    }
    stage_rgb<V, 0, kNT0<V>>(p, LI, vs, lane, logit);
    if constexpr (kLdsState<V>) {        // pair 1's stage-C outputs come back from LDS only now
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int j = 0; j < 16; ++j) vs[2 + v].fa[j] = lds_reload(&lds_state[v * 16 + j][threadIdx.x]);
    }
    if constexpr (V > kNT0<V>) stage_rgb<V, 2, V - 2>(p, LI, vs, lane, logit);
    float mx = logit[0];
#pragma unroll
    for (int v = 1; v < V; ++v) mx = fmaxf(mx, logit[v]);
    float e[V], es = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) { e[v] = __expf(logit[v] - mx); es += e[v]; }
    col[0] = col[1] = col[2] = 0.f;
    const float res_ = rcp1(es);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const float bw = e[v] * res_;
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] += vs[v].rgb[c] * bw;
    }
  }
  PGRF_TICK(8);   // stage E

  if constexpr (V > 2) __builtin_amdgcn_sched_barrier(0);   // V>2: keep the stages apart (cross-stage interleaving made the compiler spill 342 VGPRs)
  composite<KT, V>(p, sigma, col, tdepth, lane, wave, kt, ray, pt, store_ok, lds_alpha, lds_red
#ifdef PGRF_STAGE_TIMING
                   , t_prev
#endif
  );
}

// =============================================================================================
// 4 source views on WAVE PAIRS (round 4).  render_pass_kernel gives a wave all V views of its 32 samples; at V = 4 that is two
// view pairs one after the other with the first pair's state live across the second pair's front end: 157 spilled registers and
// 9-12 points of the roofline behind the 2-view kernel, unmoved by two rounds of in-wave rescheduling (profiles/r03_dead_ends.txt).
// Here a 32-sample tile belongs to TWO waves: role 0 holds views 0, 1 and role 1 views 2, 3 -- each with the register picture of
// the 2-view kernel (V = 2 policies of the stage functions, VT = 4, vb = first view) -- and what couples the views goes through LDS:
//   stage B  the view statistics of feature j need x_v[j] of all four views: role 0 takes j in [0, 8), role 1 j in [8, 16) and the
//            colour channels; each wave sends the other the half of its two tiles that the other owns (+ its w0), computes ITS
//            statistics in the reference's view order, multiplies them with ITS k-steps of the shared base_fc.0 block (a K split:
//            32 | 38 of the 70 k-steps) and the two partial [64 x 32] products are exchanged and added;
//   stage D  the same for the pooled geometry feature (vis weights of the four views, weighted mean / variance of x) and
//            geometry_fc.0 (16 | 17 of 33 k-steps);
//   stage E  role 1 hands over its two blending logits and colours.
// The per-point tail (geometry_fc.2 ... density, ~7 % of the multiply-adds at four views) and the composite run in both waves;
// role 0 stores.  Workgroup = 4 waves = 2 tiles = 64 points; K / V of the ray attention share the LDS rows of the x exchange.
constexpr int kPairXRows = 22;
constexpr int kPairGRows = 32;

// k-steps of a K-split layer: the wave's i-th A chunk (four k-steps) is chunk `CH(i)` of the layer's block
template <int NCH, int MT, int KS, class CH>
__device__ __forceinline__ void mfma_chunks(const float* __restrict__ A, int lane, const float (&b)[NCH * 4], f32x16 (&acc)[MT], CH chunk_of) {
  constexpr int KS4 = (KS + 3) / 4;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = chunk_of(i);
      const float4 a = reinterpret_cast<const float4*>(A)[(m * KS4 + c) * 64 + lane];
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c * 4 + j < KS) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[i * 4 + j], acc[m], 0, 0, 0);
    }
  }
}
// accumulator tiles start from the bias (role 0) or from zero (role 1): the partial products are added afterwards
template <int ROLE, int MT>
__device__ __forceinline__ void pair_acc_init(const float* __restrict__ B, int lane, f32x16 (&acc)[MT]) {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if constexpr (ROLE == 0) {
      const float4* bp = reinterpret_cast<const float4*>(B + m * 32 + (lane >> 5) * 16);
      const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
      acc[m] = (f32x16){b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    }
  }
}
// values of the four views in the reference's order from (own pair, partner's pair)
template <int ROLE>
__device__ __forceinline__ void pair_order(float o0, float o1, float r0, float r1, float (&x)[4]) {
  if constexpr (ROLE == 0) { x[0] = o0; x[1] = o1; x[2] = r0; x[3] = r1; }
  else { x[0] = r0; x[1] = r1; x[2] = o0; x[3] = o1; }
}

// stage B, sending half: the features the PARTNER computes statistics of
template <int ROLE>
__device__ __forceinline__ void pair_send_feats(const ViewState (&vs)[2], float (*xb)[256], int tid, int half, bool with_colour) {
#pragma unroll
  for (int v = 0; v < 2; ++v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xb[v * 8 + i][tid] = vs[v].fa[ROLE == 0 ? 8 + i : i];
    if (ROLE == 0 && with_colour) {
      xb[16 + v][tid] = half ? vs[v].fb[1] : vs[v].fb[0];
      xb[18 + v][tid] = vs[v].fb[2];
    }
  }
}

// stage B, receiving half: statistics of this wave's features over the four views and its k-steps of base_fc.0's shared block
template <int ROLE>
__device__ __forceinline__ void pair_shared_base(const PassArgs& p, const ViewState (&vs)[2], float (*xb)[256], int lane, int tid,
                                                 f32x16 (&G)[2]) {
  const int half = lane >> 5, pt = tid ^ 64;      // pt = the partner's column
  const float* __restrict__ W = p.blob;
  const float wgt = 1.f / (4.f + 1e-8f);
  float w0a[4], w1a[4] = {wgt, wgt, wgt, wgt};
  pair_order<ROLE>(vs[0].w0, vs[1].w0, xb[20][pt], xb[21][pt], w0a);
  auto stats = [&](const float (&xv)[4], const float (&wa)[4], float& m, float& var) {
    m = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) m += xv[v] * wa[v];
    var = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) { const float d = xv[v] - m; var += wa[v] * (d * d); }
  };
  constexpr int NCH = ROLE == 0 ? 8 : 10;
  float bop[NCH * 4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float xv[4];
    pair_order<ROLE>(vs[0].fa[ROLE * 8 + i], vs[1].fa[ROLE * 8 + i], xb[i][pt], xb[8 + i][pt], xv);
    stats(xv, w0a, bop[i], bop[8 + i]);
    stats(xv, w1a, bop[16 + i], bop[24 + i]);
  }
  if constexpr (ROLE == 1) {
    float xv[4], wsel[4];
    pair_order<ROLE>(half ? vs[0].fb[1] : vs[0].fb[0], half ? vs[1].fb[1] : vs[1].fb[0], xb[16][pt], xb[17][pt], xv);
    stats(xv, w0a, bop[32], bop[33]);
    stats(xv, w1a, bop[34], bop[35]);
    pair_order<ROLE>(vs[0].fb[2], vs[1].fb[2], xb[18][pt], xb[19][pt], xv);
#pragma unroll
    for (int v = 0; v < 4; ++v) wsel[v] = half ? wgt : w0a[v];
    stats(xv, wsel, bop[36], bop[37]);
    bop[38] = bop[39] = 0.f;
  }
  pair_acc_init<ROLE, 2>(W + layer_b_off(L_BG), lane, G);
  // this wave's k-steps of the block (k-step order of L_BG: mean0 | var0 | mean1 | var1 of the 16 feature registers, then colours)
  mfma_chunks<NCH, 2, 70>(W + layer_a_off(L_BG), lane, bop, G, [](int i) { return i < 8 ? 4 * (i >> 1) + (i & 1) + 2 * ROLE : 8 + i; });
}

// stage D, receiving half: vis weights, pooled statistics of this wave's features of x, its k-steps of geometry_fc.0
template <int ROLE>
__device__ __forceinline__ void pair_geometry(const PassArgs& p, const ViewState (&vs)[2], float (*xb)[256], int lane, int tid,
                                              f32x16 (&g1)[2]) {
  const int pt = tid ^ 64;
  const float* __restrict__ W = p.blob;
  float v2a[4];
  pair_order<ROLE>(vs[0].v2, vs[1].v2, xb[20][pt], xb[21][pt], v2a);
  float vsum = 0.f;
#pragma unroll
  for (int v = 0; v < 4; ++v) vsum += v2a[v];
  const float rvs = rcp1(vsum + 1e-8f);
  float w2[4], wmean = 0.f;
#pragma unroll
  for (int v = 0; v < 4; ++v) { w2[v] = v2a[v] * rvs; wmean += w2[v]; }
  wmean = wmean / 4.f;
  constexpr int NCH = ROLE == 0 ? 4 : 5;
  float bop[NCH * 4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float xr[4];
    pair_order<ROLE>(vs[0].fa[ROLE * 8 + i], vs[1].fa[ROLE * 8 + i], xb[i][pt], xb[8 + i][pt], xr);
    float m = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) m += xr[v] * w2[v];
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) { const float d = xr[v] - m; var += w2[v] * (d * d); }
    bop[i] = m;
    bop[8 + i] = var;
  }
  if constexpr (ROLE == 1) { bop[16] = wmean; bop[17] = bop[18] = bop[19] = 0.f; }
  pair_acc_init<ROLE, 2>(W + layer_b_off(L_G1), lane, g1);
  mfma_chunks<NCH, 2, 33>(W + layer_a_off(L_G1), lane, bop, g1, [](int i) { return i < 4 ? 4 * (i >> 1) + (i & 1) + 2 * ROLE : 8; });
}

template <int KT, bool DBG, bool GEN>
__global__ void __launch_bounds__(256, 2) render_pass_pair_kernel(const PassArgs p) {
  static_assert(KT == 1 || KT == 2, "one workgroup = two 32-sample tiles");
  constexpr int V = 2, VT = 4;         // V keys the per-instance policies of the stage functions (those of the 2-view kernels)
  constexpr int DN = 32 * KT;
  constexpr int RPW = 2 / KT;          // rays per workgroup
  __shared__ __attribute__((aligned(16))) float lds_x[kPairXRows][256];    // x exchange; later K | V of the workgroup's 64 points
  __shared__ __attribute__((aligned(16))) float lds_g[kPairGRows][256];    // partial-product exchange; later logits / colours
  __shared__ float lds_park[V * 7][128];
  __shared__ float lds_alpha[2];
  __shared__ float lds_red[2][4];
  __shared__ __attribute__((aligned(16))) float lds_img[img_floats(kImgMode<V>)];
  static_assert(64 * (kKStride + kVStride) <= kPairXRows * 256, "K / V alias the x-exchange rows");
  float* lds_k = &lds_x[0][0];
  float* lds_v = lds_k + 64 * kKStride + ((64 * kKStride) & 3 ? 4 - ((64 * kKStride) & 3) : 0);
  const float* LI = lds_img;
  {
    const float4* src = reinterpret_cast<const float4*>(p.blob + img_blob_off(kImgMode<V>));
    for (int i = threadIdx.x; i < img_floats(kImgMode<V>) / 4; i += 256) reinterpret_cast<float4*>(lds_img)[i] = src[i];
  }
  __shared__ float* lds_dbg[DBG ? 12 : 1];
  float* const* sdbg = lds_dbg;
  if (DBG) {
    if (threadIdx.x < 12) lds_dbg[threadIdx.x] = reinterpret_cast<float* const*>(&p.dbg)[threadIdx.x];
    __syncthreads();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef PGRF_STAGE_TIMING
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave & 1, tile = wave >> 1;
  const int vb = role * 2;
  const int half = lane >> 5;
  const int pl = lane & 31;
  const int ray_local = tile / KT;
  const int kt = tile % KT;
  int wg = blockIdx.x;
  if (!p.order_w) {     // XCD bands (see render_pass_kernel): kept from the one-wave-per-tile 4-view kernel
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int ray_raw = wg * RPW + ray_local;
  if (p.order_w) {
    const int per_tile = p.order_tw * p.order_th, tiles_x = p.order_w / p.order_tw;
    const int t = ray_raw / per_tile, in = ray_raw - t * per_tile;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int iy = in / p.order_tw, ix = in - iy * p.order_tw;
    ray_raw = (ty * p.order_th + iy) * p.order_w + tx * p.order_tw + ix;
  }
  const bool ray_ok = ray_raw < p.rn;
  const int ray = ray_ok ? ray_raw : p.rn - 1;
  const int si = kt * 32 + pl;
  const int pt_local = tile * 32 + pl;
  const size_t pt = (size_t)ray * DN + si;
  const size_t npts = (size_t)p.rn * DN;

  // ---- ray generation (as in render_pass_kernel)
  PointGeom g;
  float tdepth;
  {
    float wx, wy, wz;
    if (p.raytab) {
      const float4 rt = reinterpret_cast<const float4*>(p.raytab)[ray];
      wx = rt.x; wy = rt.y; wz = rt.z;
    } else {
      ray_direction<GEN>(p, ray, wx, wy, wz);
    }
    const float* drow = p.depth + (size_t)ray * DN;
    tdepth = drow[si];
    g.p[0] = p.c2w[3] + wx * tdepth;
    g.p[1] = p.c2w[7] + wy * tdepth;
    g.p[2] = p.c2w[11] + wz * tdepth;
    const float wn = sqrtf(wx * wx + wy * wy + wz * wz);
    { const float rw = rcp1(wn); g.qdir[0] = -wx * rw; g.qdir[1] = -wy * rw; g.qdir[2] = -wz * rw; }
    const float qn = -1.f / p.que_near, qf = -1.f / p.que_far;
    const float d_c = (-1.f / tdepth - qn) / (qf - qn);
    const float d_p = (-1.f / drow[max(si - 1, 0)] - qn) / (qf - qn);
    const float d_n = (-1.f / drow[min(si + 1, DN - 1)] - qn) / (qf - qn);
    const float dist_c = (si == DN - 1) ? 1e6f : d_n - d_c;
    const float dist_p = (si == 0) ? dist_c : d_c - d_p;
    g.near_hw = dist_p / 2.f;
    g.far_hw = dist_c / 2.f;
  }
  ViewState vs[V];
  const bool store_ok = ray_ok;              // debug dumps of the wave's own views
  const bool store_pt = ray_ok && role == 0; // per-point results: one wave of the pair
  __syncthreads();
  PGRF_TICK(0);

  // ---- stage A: this wave's two views
  stage_front<V, 0, 2, DBG, GEN, VT>(p, LI, g, vs, lane, pt, npts, store_ok, sdbg
#ifdef PGRF_STAGE_TIMING
                                     , t_prev
#endif
                                     , vb);
  PGRF_TICK(1);
  const int park_col = wave * 32 + pl;
#pragma unroll
  for (int v = 0; v < V; ++v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) lds_park[v * 7 + c][park_col] = vs[v].dd[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) lds_park[v * 7 + 4 + c][park_col] = vs[v].rgb[c];
  }
  // ---- stage B: view statistics + the shared block of base_fc.0, split between the waves of the pair
  f32x16 G[1][2];
  if (role == 0) pair_send_feats<0>(vs, lds_x, tid, half, true); else pair_send_feats<1>(vs, lds_x, tid, half, true);
  lds_x[20][tid] = vs[0].w0;
  lds_x[21][tid] = vs[1].w0;
  __syncthreads();
  if (role == 0) pair_shared_base<0>(p, vs, lds_x, lane, tid, G[0]); else pair_shared_base<1>(p, vs, lds_x, lane, tid, G[0]);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_g[m * 16 + r][tid] = G[0][m][r];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) G[0][m][r] += lds_g[m * 16 + r][tid ^ 64];
  PGRF_TICK(2);

  // ---- stage C: per-view base_fc / vis_fc / vis_fc2 of this wave's views
  stage_mid<V, 0, 2, DBG, VT>(p, LI, G[0], vs, lane, pt, store_ok, sdbg, vb);
  PGRF_TICK(3);

  // ---- stage D: pooled geometry feature and geometry_fc.0, split as stage B; then the per-point tail in both waves
  float sigma;
  {
    f32x16 g1[1][2];
    if (role == 0) pair_send_feats<0>(vs, lds_x, tid, half, false); else pair_send_feats<1>(vs, lds_x, tid, half, false);
    lds_x[20][tid] = vs[0].v2;
    lds_x[21][tid] = vs[1].v2;
    __syncthreads();
    if (role == 0) pair_geometry<0>(p, vs, lds_x, lane, tid, g1[0]); else pair_geometry<1>(p, vs, lds_x, lane, tid, g1[0]);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) lds_g[m * 16 + r][tid] = g1[0][m][r];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) g1[0][m][r] += lds_g[m * 16 + r][tid ^ 64];
    // (the partner has read its x rows before the barrier above: K / V may overwrite them now)
#ifdef PGRF_PAIR_NO_HEAD_SPLIT
    sigma = point_tail<KT, V, DBG, VT>(p, LI, g1, lane, si, pt, pt_local, ray_local * DN, lds_k, lds_v, store_pt, sdbg
#ifdef PGRF_STAGE_TIMING
                                       , t_prev
#endif
    );
#else
    // attention heads 0, 1 in role 0 and 2, 3 in role 1; outputs exchanged through rows 8..15 of lds_g (rows 0..7 carry stage E)
    if (role == 0) sigma = point_tail<KT, V, DBG, VT, 0>(p, LI, g1, lane, si, pt, pt_local, ray_local * DN, lds_k, lds_v, store_pt, sdbg
#ifdef PGRF_STAGE_TIMING
                                       , t_prev
#endif
                                       , lds_g + 8, tid);
    else sigma = point_tail<KT, V, DBG, VT, 1>(p, LI, g1, lane, si, pt, pt_local, ray_local * DN, lds_k, lds_v, store_pt, sdbg
#ifdef PGRF_STAGE_TIMING
                                       , t_prev
#endif
                                       , lds_g + 8, tid);
#endif
  }
  PGRF_TICK(7);

  // ---- stage E: blending logits of this wave's views; softmax over the four
  float col[3];
  {
#pragma unroll
    for (int v = 0; v < V; ++v) {
#pragma unroll
      for (int c = 0; c < 4; ++c) vs[v].dd[c] = lds_reload(&lds_park[v * 7 + c][park_col]);
#pragma unroll
      for (int c = 0; c < 3; ++c) vs[v].rgb[c] = lds_reload(&lds_park[v * 7 + 4 + c][park_col]);
    }
    float lg[V];
    stage_rgb<V, 0, 2>(p, LI, vs, lane, lg);
    // (lds_g was last read before the barrier inside point_tail)
#pragma unroll
    for (int v = 0; v < V; ++v) {
      lds_g[v][tid] = lg[v];
#pragma unroll
      for (int c = 0; c < 3; ++c) lds_g[2 + v * 3 + c][tid] = vs[v].rgb[c];
    }
    __syncthreads();
    float logit[4], rgb[4][3];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int vo = role ? 2 + v : v, vp = role ? v : 2 + v;      // own / partner's view in the reference's order
      logit[vo] = lg[v];
      logit[vp] = lds_g[v][tid ^ 64];
#pragma unroll
      for (int c = 0; c < 3; ++c) { rgb[vo][c] = vs[v].rgb[c]; rgb[vp][c] = lds_g[2 + v * 3 + c][tid ^ 64]; }
    }
    float mx = logit[0];
#pragma unroll
    for (int v = 1; v < 4; ++v) mx = fmaxf(mx, logit[v]);
    float e[4], es = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) { e[v] = __expf(logit[v] - mx); es += e[v]; }
    col[0] = col[1] = col[2] = 0.f;
    const float res_ = rcp1(es);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float bw = e[v] * res_;
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] += rgb[v][c] * bw;
    }
  }
  PGRF_TICK(8);
  composite<KT, V>(p, sigma, col, tdepth, lane, tile, kt, ray, pt, store_pt, lds_alpha, lds_red
#ifdef PGRF_STAGE_TIMING
                   , t_prev
#endif
  );
}


template <int KT, int V, bool DBG>
int launch_instance(const PassArgs& a, void* stream) {
  // debug instances exist in the general form only (they serve every convention at test sizes)
  const bool general = DBG || a.dataset != PGRF_DATASET_M3D || a.perspective;
  if constexpr (V == 4 && KT <= 2 && kPrec == 0) {
    // 4 views, f32: the wave-pair kernel (PGRF_V4_KERNEL=wave selects the one-wave-per-tile kernel: A/B and tests)
    const char* knob = getenv("PGRF_V4_KERNEL");
    if (!(knob && knob[0] == 'w')) {
      constexpr int RPWP = 2 / KT;
      const int nb = (a.rn + RPWP - 1) / RPWP;
      if (general) hipLaunchKernelGGL((render_pass_pair_kernel<KT, DBG, true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((render_pass_pair_kernel<KT, DBG, DBG>), dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return set_error(PGRF_ERR_LAUNCH, "render_pass (wave-pair) launch failed: %s", hipGetErrorString(e));
      return PGRF_OK;
    }
  }
  constexpr int RPW = 4 / KT;
  const int blocks = (a.rn + RPW - 1) / RPW;
  if (general) hipLaunchKernelGGL((render_pass_kernel<KT, V, DBG, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((render_pass_kernel<KT, V, DBG, DBG>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(PGRF_ERR_LAUNCH, "render_pass launch failed: %s", hipGetErrorString(e));
  return PGRF_OK;
}

}  // namespace PGRF_PREC_NS
}  // namespace pgrf
