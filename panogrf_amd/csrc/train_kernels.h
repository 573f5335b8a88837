// Device kernels of the training-mode backward pass (train_pass.hip): generic row-major GEMMs for the Linear layers, the
// activation derivatives, and forward / backward kernels of every non-Linear stage of one network pass.  Plain fp32 VALU
// code for the non-Linear stages; the Linear layers run on the f32 matrix cores (lin_mfma_kernel, dw_mfma_kernel below).
// (Training batches are 512-2048 rays: the staged, layer-by-layer form keeps every activation in HBM and is bound by that
// traffic and by launch count, not by the matrix pipe -- see DESIGN.md section 4.7.)
//
// Reference semantics: network/dist_decoder.py:6-51,109-140 (compute_prob), network/ibrnet.py:7-27,52-116,315-373,
// network/aggregate_net.py:41-89, network/renderer.py:210-219,291-304; every kernel cites what it differentiates.
#pragma once
#include <hip/hip_runtime.h>

#include "ray_dir.h"

namespace pgrf {
namespace train {

// ---------------------------------------------------------------------------------------------------------------
// Linear layers of the staged forward / backward on the f32 matrix cores (round 4; was a 64x64x16 LDS-tiled VALU SGEMM).  The
// layers are SKINNY: rows = sample points x views (10^5 .. 10^6), IN and OUT <= 207.  Same transposed orientation as the fused
// forward kernel (render_pass.h): the wave's 32 rows are the MFMA columns.
//
//   lin_mfma_kernel     Y[rows, M] (=|+=) act(X[rows, K] * W^T + bias)      W stored [M, K] (forward)  or [K, M] (TW: dX = delta W)
//     Y^T tile [32 m x 32 rows] += W[32 m x 2] * X^T[2 x 32 rows] with v_mfma_f32_32x32x2_f32.  Lane (n, half) loads the four
//     floats X[row n][8q + 4 half ..] with one dwordx4 (= the B operands of k-steps 4q .. 4q+3 under the k permutation
//     "k-step 4q + j pairs features 8q + j and 8q + 4 + j") and lane (i, half) the same four columns of W's row i, so both operands
//     are plain row-major loads; rows of X need not be 16-byte aligned (ld = 207, 37, 34 ... : dword-aligned dwordx4).
//     Epilogue: accumulator registers 4g .. 4g+3 of lane (n, half) are Y[row n][32 m + 8g + 4 half ..] -- one dwordx4 store.
//   dw_mfma_kernel      gW[M, K] += delta^T[M, rows] * X[rows, K]          (reduction over the rows)
//     A operand = delta with the ROW as k index: lane (i, half) loads delta[row 2s + half][32 m + i] -- a k-step reads two rows of
//     32 consecutive floats each, coalesced.  A wave owns one m tile x NT n tiles over its share of the workgroup's rows; the four
//     waves of a workgroup are reduced through LDS, then ONE hardware float atomic per output element and workgroup.
typedef float f32x16t __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };     // four floats at dword alignment (global_load_dwordx4)

// p[k0 .. k0+3] with zeros from column K on
__device__ __forceinline__ void load4_guard(const float* __restrict__ p, int k0, int K, float (&o)[4]) {
  if (k0 + 3 < K) {
    const F4U t = *reinterpret_cast<const F4U*>(p + k0);
    o[0] = t.v[0]; o[1] = t.v[1]; o[2] = t.v[2]; o[3] = t.v[3];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (k0 + j < K) ? p[k0 + j] : 0.f;
  }
}

// activations of the per-point MLPs; the derivative is written in terms of the OUTPUT y (nothing else is stored)
enum Act { ACT_NONE = 0, ACT_ELU, ACT_RELU, ACT_SIGMOID, ACT_SOFTPLUS, ACT_SOFTPLUS_B };   // _B: softplus + 0.05 (AddBias)
// (the same formulas as the fused forward kernel, render_pass.h elu1 / sigmoid1 / softplus1, on v_exp_f32 / v_log_f32: the staged
//  forward of the backward then reproduces the activations the fused kernel rendered with -- and libm's expm1f / log1pf, ~30-140
//  instructions each with a branch, were as long as the layers' MFMAs: 96 ELUs per row of base_fc)
__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case ACT_ELU: return __builtin_amdgcn_fmed3f(x, __expf(x) - 1.f, 0.f);
    case ACT_RELU: return fmaxf(x, 0.f);
    case ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __expf(-x));
    case ACT_SOFTPLUS:
    case ACT_SOFTPLUS_B: {
      const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 1.4426950408889634f);
      const float sp = fmaf(__builtin_amdgcn_logf(1.f + t), 0.6931471805599453f, fmaxf(x, 0.f));
      return act == ACT_SOFTPLUS_B ? sp + 0.05f : sp;
    }
    default: return x;
  }
}
__device__ __forceinline__ float act_grad(float y, int act) {
  switch (act) {
    case ACT_ELU: return y > 0.f ? 1.f : y + 1.f;                 // d/dx (e^x - 1) = e^x = y + 1
    case ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ACT_SIGMOID: return y * (1.f - y);
    case ACT_SOFTPLUS: return 1.f - expf(-y);                      // sigmoid(x) = 1 - e^{-softplus(x)}
    case ACT_SOFTPLUS_B: return 1.f - expf(-(y - 0.05f));
    default: return 1.f;
  }
}


template <bool TW, int MT>
__global__ void __launch_bounds__(256) lin_mfma_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias, float* __restrict__ Y, int ldy, long rows, int M, int K,
                                                       int act, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
  const long row = ((long)blockIdx.x * 4 + wave) * 32 + n;
  const int m0 = blockIdx.y * (32 * MT);
  const bool row_ok = row < rows;
  const float* xr = X + (row_ok ? row : rows - 1) * (long)ldx;
  f32x16t acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int nq = (K + 7) >> 3;
  for (int q = 0; q < nq; ++q) {
    const int k0 = 8 * q + 4 * half;
    float b4[4];
    load4_guard(xr, k0, K, b4);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mi = m0 + 32 * m + n;          // this lane's A row (output feature)
      float a4[4];
      if (mi < M) {
        if (!TW) load4_guard(W + (long)mi * ldw, k0, K, a4);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a4[j] = (k0 + j < K) ? W[(long)(k0 + j) * ldw + mi] : 0.f;
        }
      } else {
        a4[0] = a4[1] = a4[2] = a4[3] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc[m], 0, 0, 0);
    }
  }
  if (!row_ok) return;
  float* yr = Y + row * (long)ldy;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = m0 + 32 * m + 8 * g + 4 * half;
      if (c0 >= M) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = acc[m][4 * g + j];
        if (c0 + j < M) {
          if (bias) t += bias[c0 + j];
          t = act_fwd(t, act);
          if (accumulate) t += yr[c0 + j];
        }
        v[j] = t;
      }
      if (c0 + 3 < M) {
        F4U o; o.v[0] = v[0]; o.v[1] = v[1]; o.v[2] = v[2]; o.v[3] = v[3];
        *reinterpret_cast<F4U*>(yr + c0) = o;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c0 + j < M) yr[c0 + j] = v[j];
      }
    }
}

// gW[M, K] += D^T X over rows [blockIdx.z * chunk, +chunk).  D = delta [rows, M] (ldd), X [rows, K] (ldx), gW row-major [M, K].
// A tile GROUP = one 32-row m tile x NT n tiles.  The four waves of a workgroup take WPG = 4 / min(4, groups) waves per group: with
// >= 4 groups (base_fc.0: 2 x 2 groups of 4 n tiles) every wave owns its own group over the SAME rows, so that each row of delta and
// X crosses the L2 once (the waves' re-reads hit the L1); small layers (one group) split the rows four ways instead.
constexpr size_t kRepStride = (size_t)68 * 13312;     // R_COUNT * kGradSlot floats between replicas (train_pass.hip checks)
template <int NT>
__global__ void __launch_bounds__(256) dw_mfma_kernel(const float* __restrict__ D, int ldd, const float* __restrict__ X, int ldx,
                                                      float* __restrict__ gW0, float* __restrict__ gWrep, int M, int K, long rows, int chunk,
                                                      int gx, int groups, float* __restrict__ gb0, float* __restrict__ gbrep, int ldg, int gcol, float oscale) {
  // replica of the gradient this workgroup adds into (train_pass.hip kGradReplicas): 0 = the caller's buffer
  const int rep = blockIdx.z & 7;
  float* __restrict__ gW = rep == 0 ? gW0 : gWrep + (size_t)(rep - 1) * kRepStride;
  // bias gradient = column sums of delta: the lanes hold delta[row][m0 + i] anyway (the groups with n0 == 0 add them up)
  float* __restrict__ gb = gb0 ? (rep == 0 ? gb0 : gbrep + (size_t)(rep - 1) * kRepStride) : nullptr;
  float colsum = 0.f;
  __shared__ float red[3][NT][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, half = lane >> 5;
  const int wpg = groups >= 4 ? 1 : (groups == 2 ? 2 : 4);           // waves per group
  const int grp = blockIdx.x * (4 / wpg) + wave / wpg, sub = wave % wpg;
  const bool grp_ok = grp < groups;
  const int m0 = (grp_ok ? grp / gx : 0) * 32, n0 = (grp_ok ? grp % gx : 0) * (32 * NT);
  const long r_beg = (long)blockIdx.z * chunk, r_end = (r_beg + chunk < rows) ? r_beg + chunk : rows;
  const bool a_ok = grp_ok && m0 + i < M;
  bool b_ok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) b_ok[t] = grp_ok && n0 + 32 * t + i < K;
  f32x16t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // sub-wave s of a group takes the k-steps (row pairs) s, s + wpg, ... of the chunk
  // (loads are unconditional on clamped addresses and masked by a select: predicated loads compile to a branch per load)
  const float* dp = D + (a_ok ? m0 + i : 0);
  const float* xp = X;
  int xoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) xoff[t] = b_ok[t] ? n0 + 32 * t + i : 0;
  const int step = 2 * wpg;
  // U k-steps per trip with ALL their loads issued before the first MFMA: at <= 2 waves per SIMD a trip that waits for its own
  // five loads is one memory latency long (measured: 360 us for base_fc.0's 6.9 GFLOP, 6x its MFMA time)
  constexpr int U = NT == 4 ? 6 : 8;
  const float amask = a_ok ? 1.f : 0.f;
  float bmask[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bmask[t] = b_ok[t] ? 1.f : 0.f;
  for (long r0 = r_beg + 2 * sub + half; r0 < r_end + half; r0 += (long)U * step) {      // (+half: both lane halves run the same trips)
    // every load is unconditional on a clamped in-range row / column and MULTIPLIED by its 0 / 1 mask (a select lets the compiler
    // turn the load back into a branch); 64-bit row arithmetic once per trip, 32-bit offsets inside
    const long rb = r0 < r_end ? r0 : r_end - 1;
    const float* dpr = dp + rb * ldd;
    const float* xpr = xp + rb * ldx;
    float a[U], b[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = r0 + u * step < r_end;
      const int du = ok ? u * step : 0;
      a[u] = dpr[du * ldd] * (ok ? amask : 0.f);
      colsum += a[u];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[u][t] = xpr[du * ldx + xoff[t]] * (ok ? bmask[t] : 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
  }
  __shared__ float cs_red[4][32];
  if (gb) {      // both lane halves walked different rows of column i; then the sub-waves of the group
    colsum += __shfl_xor(colsum, 32, 64);
    if (half == 0) cs_red[wave][i] = colsum;
  }
  if (wpg > 1) {
    if (sub > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][t][r][lane] = acc[t][r];
    }
    __syncthreads();
  }
  if (gb && wpg == 1) __syncthreads();       // (cs_red of this wave is its own: only the memory ordering is needed; wpg > 1 synchronised above)
  if (gb && sub == 0 && grp_ok && n0 == 0 && half == 0 && m0 + i < M) {
    float t = cs_red[wave][i];
    for (int k = 1; k < wpg; ++k) t += cs_red[wave + k][i];
    if (t != 0.f) unsafeAtomicAdd(gb + m0 + i, t);
  }
  if (sub == 0 && grp_ok) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[t][r];
        for (int k = 1; k < wpg; ++k) v += red[wave + k - 1][t][r][lane];
        const int mo = m0 + (r & 3) + 8 * (r >> 2) + 4 * half, ko = n0 + 32 * t + i;     // accumulator row rho(r, half), column = lane
        if (mo < M && ko < K) unsafeAtomicAdd(gW + (long)mo * ldg + gcol + ko, v * oscale);     // (ldg, gcol: a column block of a wider weight; oscale: the layer read s * X)
      }
  }
}

// The same product with its operands staged through LDS (round 4, NT = 1).  The kernel above loads one float per lane and k-step: 256
// bytes per load instruction, and with the number of vector-memory instructions a CU keeps in flight that is ~1.7 TB/s however many of
// them are launched (33 products of a pass as ONE grid: 1.34 GB in 0.78 ms).  Here a wave fetches 32 rows x 32 columns of delta and of X
// with four `dwordx4` loads each (8 lanes per row: 1 KB per instruction), parks them in a private LDS tile (row stride 32: the two lane
// halves of a k-step read rows 2s and 2s + 1 = all 64 banks) and reads its MFMA operands from there; the next stage's loads are in flight
// while a stage is multiplied.  Out-of-range columns / rows are AND-masked after an unconditional load (the load may touch up to 124
// bytes past a row's end: workspace buffers, plan() pads).  Epilogue as above (the tiles' memory is reused for the cross-wave sum).
__global__ void __launch_bounds__(256) dw_lds_kernel(const float* __restrict__ D, int ldd, const float* __restrict__ X, int ldx,
                                                     float* __restrict__ gW0, float* __restrict__ gWrep, int M, int K, long rows, int chunk,
                                                     int gx, int groups, float* __restrict__ gb0, float* __restrict__ gbrep, int ldg, int gcol, float oscale) {
  const int rep = blockIdx.z & 7;
  float* __restrict__ gW = rep == 0 ? gW0 : gWrep + (size_t)(rep - 1) * kRepStride;
  float* __restrict__ gb = gb0 ? (rep == 0 ? gb0 : gbrep + (size_t)(rep - 1) * kRepStride) : nullptr;
  __shared__ float tiles[4][2][32 * 32];                     // per wave: [delta | X] tile; later red[3][16][64] (12 KB of the 32)
  __shared__ float cs_red[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, half = lane >> 5;
  const int wpg = groups >= 4 ? 1 : (groups == 2 ? 2 : 4);           // waves per group
  const int grp = blockIdx.x * (4 / wpg) + wave / wpg, sub = wave % wpg;
  const bool grp_ok = grp < groups;
  const int m0 = (grp_ok ? grp / gx : 0) * 32, n0 = (grp_ok ? grp % gx : 0) * 32;
  const long r_beg = (long)blockIdx.z * chunk, r_end = (r_beg + chunk < rows) ? r_beg + chunk : rows;
  // the sub-waves of a group take contiguous shares of the chunk's rows, in stages of 32
  const long per = (((r_end - r_beg) + wpg - 1) / wpg + 31) / 32 * 32;
  const long my_beg = r_beg + (long)sub * per, my_end = (my_beg + per < r_end) ? my_beg + per : r_end;
  // load mapping: 8 lanes per row, 4 columns each; load j covers rows lr + 8j of the stage
  const int lr = lane >> 3, c4 = (lane & 7) * 4;
  unsigned dmask[4], xmask[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    dmask[e] = (grp_ok && m0 + c4 + e < M) ? 0xffffffffu : 0u;
    xmask[e] = (grp_ok && n0 + c4 + e < K) ? 0xffffffffu : 0u;
  }
  const float* dcol = D + ((grp_ok && m0 + c4 < M) ? m0 + c4 : 0);
  const float* xcol = X + ((grp_ok && n0 + c4 < K) ? n0 + c4 : 0);
  float* tD = tiles[wave][0];
  float* tX = tiles[wave][1];
  f32x16t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;
  F4U dreg[4], xreg[4];
  auto fetch = [&](long s0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long row = s0 + lr + 8 * j;
      row = row < rows ? row : rows - 1;                      // (rows past the share are masked at the LDS write)
      dreg[j] = *reinterpret_cast<const F4U*>(dcol + row * ldd);
      xreg[j] = *reinterpret_cast<const F4U*>(xcol + row * ldx);
    }
  };
  if (my_beg < my_end) fetch(my_beg);
  for (long s0 = my_beg; s0 < my_end; s0 += 32) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned rm = (s0 + lr + 8 * j < my_end) ? 0xffffffffu : 0u;
      float4 dv, xv;
      dv.x = __uint_as_float(__float_as_uint(dreg[j].v[0]) & dmask[0] & rm); dv.y = __uint_as_float(__float_as_uint(dreg[j].v[1]) & dmask[1] & rm);
      dv.z = __uint_as_float(__float_as_uint(dreg[j].v[2]) & dmask[2] & rm); dv.w = __uint_as_float(__float_as_uint(dreg[j].v[3]) & dmask[3] & rm);
      xv.x = __uint_as_float(__float_as_uint(xreg[j].v[0]) & xmask[0] & rm); xv.y = __uint_as_float(__float_as_uint(xreg[j].v[1]) & xmask[1] & rm);
      xv.z = __uint_as_float(__float_as_uint(xreg[j].v[2]) & xmask[2] & rm); xv.w = __uint_as_float(__float_as_uint(xreg[j].v[3]) & xmask[3] & rm);
      *reinterpret_cast<float4*>(tD + (lr + 8 * j) * 32 + c4) = dv;
      *reinterpret_cast<float4*>(tX + (lr + 8 * j) * 32 + c4) = xv;
    }
    if (s0 + 32 < my_end) fetch(s0 + 32);                    // in flight under this stage's products
    __builtin_amdgcn_wave_barrier();                         // (one wave, in-order LDS: the reads below see the writes above)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float a = tD[(2 * ks + half) * 32 + i], b = tX[(2 * ks + half) * 32 + i];
      colsum += a;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (gb) {      // both lane halves walked different rows of column i; then the sub-waves of the group
    colsum += __shfl_xor(colsum, 32, 64);
    if (half == 0) cs_red[wave][i] = colsum;
  }
  __syncthreads();                                           // every wave is done with its tiles: their memory becomes the reduction buffer
  float* red = &tiles[0][0][0];                              // [3][16][64]
  if (wpg > 1 && sub > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (gb && sub == 0 && grp_ok && n0 == 0 && half == 0 && m0 + i < M) {
    float t = cs_red[wave][i];
    for (int k = 1; k < wpg; ++k) t += cs_red[wave + k][i];
    if (t != 0.f) unsafeAtomicAdd(gb + m0 + i, t);
  }
  if (sub == 0 && grp_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
      for (int k = 1; k < wpg; ++k) v += red[((wave + k - 1) * 16 + r) * 64 + lane];
      const int mo = m0 + (r & 3) + 8 * (r >> 2) + 4 * half, ko = n0 + i;
      if (mo < M && ko < K) unsafeAtomicAdd(gW + (long)mo * ldg + gcol + ko, v * oscale);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// CHAINS of Linear layers in one launch (round 4).  In lin_mfma_kernel's orientation the accumulator registers of one layer ARE the
// B operands of the next (register 4g + j of lane (n, half) is column 8g + 4 half + j: exactly what k-step 4g + j asks of that lane),
// so a wave carries its 32 rows through up to three layers without the activations leaving registers; every layer's output is
// still WRITTEN (the backward needs it) but no longer re-read -- and in the backward chain the activation derivative and the dX
// product of a whole MLP are one launch (was: one act kernel + one dX GEMM per layer).
//   forward  chain_fwd_kernel:  b <- X (any width, streamed from global for the first layer); per layer  acc = W b (+ bias, act)
//            -> stored to Y_k -> b (layers after the first: IN <= 64, register resident)
//   backward chain_bwd_kernel:  d <- dY of the top layer; per layer (top first)  d *= act'(Y_k), stored as delta_k (what the dW
//            kernel reads), then d <- W_k^T d; the last layer's product may be wider than 64 (streamed tile by tile) and is written
//            or accumulated into dX.  `dx == 0` on the last layer: delta only.
// Both kernels first copy the chain's weights (and biases) into LDS, zero padded to whole MFMA tiles, so that every operand read in
// the layer loops is unconditional (a guarded global load per k-step made the first version latency bound: 99 us per launch for
// ~25 us of HBM traffic); activations are read with unconditional dwordx4 loads whose out-of-range elements are masked with an
// integer AND -- they may touch up to 12 bytes past a row's last column, so X / Y / dY must be WORKSPACE buffers (plan() pads).
struct ChainLayer {
  const float* W;      // [OUT, IN] row-major
  const float* bias;   // forward only (null = none)
  float* Y;            // forward: output [rows, ldy];  backward: the layer's stored output (read for act')
  float* D;            // backward: dY of the layer, overwritten by delta
  int ldy, ldd, IN, OUT, act, dx;
  int S, off;          // LDS row stride of the weight copy (multiple of 4, S / 4 odd: conflict-free b128 reads) and its offset (floats)
  int ldw;             // row stride of W in memory (IN, or the full width when the layer is a column block of a wider weight)
};
struct ChainArgs {
  ChainLayer l[3];
  int n, lds_floats;
  const float* X; int ldx;           // forward input
  const float* add; int ldadd, addV; // forward, first layer: + add[row / addV][0 .. OUT) before the activation (a term shared by the addV
                                     // consecutive rows of a point: base_fc.0's product with the view-independent statistics)
  float* dX; int lddx, accumulate;   // backward output (null = none)
  long rows;
  float xscale;                      // the first layer reads xscale * X (forward); the chain's dX is xscale * (delta W) (backward)
};
__host__ __device__ inline int chain_stride(int IN) { int s = (IN + 3) / 4 * 4; if (((s / 4) & 1) == 0) s += 4; return s; }
__host__ __device__ inline int chain_rows(int OUT) { return OUT > 32 ? 64 : 32; }

__device__ __forceinline__ void chain_stage_weights(const ChainArgs& a, float* lds, bool with_bias) {
#ifdef PGRF_TRAIN_ABL_NO_STAGE   // timing experiment: the weights are not staged (the products then read whatever the LDS holds)
  return;
#endif
  // (round 5: 16-byte zero stores and a (row, column) walk without integer division -- the staging of a chain's few KB was 0.22 ms of a
  //  5.6 ms training step, 52 chain launches of ~1 300 workgroups each)
  for (int i = threadIdx.x * 4; i < a.lds_floats; i += blockDim.x * 4) *reinterpret_cast<float4*>(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);   // (lds_floats % 4 == 0)
  __syncthreads();
  const int r0 = threadIdx.x >> 5, c0 = threadIdx.x & 31, rstep = blockDim.x >> 5;          // rstep rows x 32 columns per sweep: a row's columns are consecutive addresses
  for (int k = 0; k < a.n; ++k) {
    const ChainLayer& L = a.l[k];
    float* w = lds + L.off;
    for (int r = r0; r < L.OUT; r += rstep) {
      const float* src = L.W + (long)r * L.ldw;
      float* dst = w + r * L.S;
      for (int c = c0; c < L.IN; c += 32) dst[c] = src[c];
    }
    if (with_bias && L.bias && threadIdx.x < L.OUT) w[chain_rows(L.OUT) * L.S + threadIdx.x] = L.bias[threadIdx.x];
  }
  __syncthreads();
}
// The lane's four floats of quad q (columns 8q + 4 half ..), elements from column K on zero.  Quads that lie inside the row for both
// lane halves (8q + 8 <= K: a wave-uniform test) are a bare dwordx4; only the row's last quad pays for the AND masks (the load
// itself stays unconditional there too) -- masking every quad cost 18 VALU instructions per load, as many cycles as its MFMAs.
__device__ __forceinline__ void load4_mask(const float* __restrict__ p, int q, int half, int K, float (&o)[4]) {
  const int k0 = 8 * q + 4 * half;
  if (8 * q + 8 <= K) {
    const F4U t = *reinterpret_cast<const F4U*>(p + k0);
    o[0] = t.v[0]; o[1] = t.v[1]; o[2] = t.v[2]; o[3] = t.v[3];
  } else {
    const int kc = k0 < K ? k0 : 0;
    const F4U t = *reinterpret_cast<const F4U*>(p + kc);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __uint_as_float(__float_as_uint(t.v[j]) & (k0 + j < K ? 0xffffffffu : 0u));
  }
}
// the same for stores: v -> p[8q + 4 half ..], columns < C only
__device__ __forceinline__ void store4_guard(float* __restrict__ p, int q, int half, int C, const float (&v)[4]) {
  const int c0 = 8 * q + 4 * half;
  if (8 * q + 8 <= C) {
    F4U o; o.v[0] = v[0]; o.v[1] = v[1]; o.v[2] = v[2]; o.v[3] = v[3];
    *reinterpret_cast<F4U*>(p + c0) = o;
  } else if (8 * q < C) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (c0 + j < C) p[c0 + j] = v[j];
  }
}
// CODE SIZE matters here: with the activation switch evaluated per element and one product variant per (tiles, quads) shape the
// forward kernel was 38 000 instructions (300 KB: every wave missed the 64 KB instruction cache all the way through, 130 us per
// launch).  Now: products in units of four quads (32 columns of k), `act` switched once per layer, the layers after the first in a
// real loop (one copy of the code).
// REGISTERS: the kernels are templated on QD = quads (8 columns) a wave carries per row: 4 for chains whose inner widths are all <= 32
// (64 VGPRs, eight waves per SIMD), 8 otherwise (base_fc, geometry_fc, vis_fc's 33 outputs).  One width-64 kernel for everything ran
// at 196 (forward) and 264 (backward) VGPRs -- one or two waves per SIMD, every load and LDS latency exposed.
// acc[m] += W[32m + n][8q + 4 half + j] * b[q][j]  over the four quads q = Q0 .. Q0+3 (W from LDS)
template <int QD, int MT, int Q0>
__device__ __forceinline__ void chain_mma_fwd(const float* __restrict__ w, int S, int n, int half, const float (&b)[QD][4], f32x16t (&acc)[QD / 4]) {
#pragma unroll
  for (int q = Q0; q < Q0 + 4; ++q)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 a4 = *reinterpret_cast<const float4*>(w + (32 * m + n) * S + 8 * q + 4 * half);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[q][0], acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[q][1], acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[q][2], acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[q][3], acc[m], 0, 0, 0);
    }
}
// acc[m] += W[8q + 4 half + j][col0 + 32 m + n] * d[q][j]  over q = Q0 .. Q0+3 (the transposed product of the backward)
template <int QD, int MT, int Q0>
__device__ __forceinline__ void chain_mma_bwd(const float* __restrict__ w, int S, int col0, int n, int half, const float (&d)[QD][4], f32x16t (&acc)[QD / 4]) {
#pragma unroll
  for (int q = Q0; q < Q0 + 4; ++q) {
    const float* wr = w + (8 * q + 4 * half) * S + col0 + n;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[j * S + 32 * m], d[q][j], acc[m], 0, 0, 0);
  }
}
// the products of one layer, nk quads of k (<= QD), mt tiles of 32 outputs (<= QD / 4): uniform branches over small bodies
template <int QD>
__device__ __forceinline__ void chain_layer_fwd(const float* __restrict__ w, int S, int n, int half, int mt, int nk, const float (&b)[QD][4], f32x16t (&acc)[QD / 4]) {
  if constexpr (QD == 4) {
    chain_mma_fwd<4, 1, 0>(w, S, n, half, b, acc);
  } else {
    if (mt == 1) { chain_mma_fwd<8, 1, 0>(w, S, n, half, b, acc); if (nk > 4) chain_mma_fwd<8, 1, 4>(w, S, n, half, b, acc); }
    else { chain_mma_fwd<8, 2, 0>(w, S, n, half, b, acc); if (nk > 4) chain_mma_fwd<8, 2, 4>(w, S, n, half, b, acc); }
  }
}
template <int QD>
__device__ __forceinline__ void chain_layer_bwd(const float* __restrict__ w, int S, int col0, int n, int half, int mt, int nk, const float (&d)[QD][4], f32x16t (&acc)[QD / 4]) {
  if constexpr (QD == 4) {
    chain_mma_bwd<4, 1, 0>(w, S, col0, n, half, d, acc);
  } else {
    if (mt == 1) { chain_mma_bwd<8, 1, 0>(w, S, col0, n, half, d, acc); if (nk > 4) chain_mma_bwd<8, 1, 4>(w, S, col0, n, half, d, acc); }
    else { chain_mma_bwd<8, 2, 0>(w, S, col0, n, half, d, acc); if (nk > 4) chain_mma_bwd<8, 2, 4>(w, S, col0, n, half, d, acc); }
  }
}
// b <- act(acc + bias) (zeros from column OUT on).  The softplus pair only ever closes the dist decoder (OUT <= 2): first quad only
// (host: chain_check).
template <int QD, int ACT>
__device__ __forceinline__ void chain_act_fwd(const f32x16t (&acc)[QD / 4], const float* __restrict__ bias, int OUT, int half, float (&b)[QD][4]) {
#pragma unroll
  for (int m = 0; m < QD / 4; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = 32 * m + 8 * g + 4 * half;
      const float4 b4 = *reinterpret_cast<const float4*>(bias + c0);
      const float v[4] = {acc[m][4 * g] + b4.x, acc[m][4 * g + 1] + b4.y, acc[m][4 * g + 2] + b4.z, acc[m][4 * g + 3] + b4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t;
        if ((ACT == ACT_SOFTPLUS || ACT == ACT_SOFTPLUS_B) && (m > 0 || g > 0)) t = 0.f;
        else t = act_fwd(v[j], ACT);
        b[4 * m + g][j] = (c0 + j < OUT) ? t : 0.f;
      }
    }
}
template <int QD, int ACT>
__device__ __forceinline__ void chain_act_bwd(const float (&y)[QD][4], float (&d)[QD][4]) {
#pragma unroll
  for (int q = 0; q < QD; ++q) {
    if ((ACT == ACT_SOFTPLUS || ACT == ACT_SOFTPLUS_B) && q > 0) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) d[q][j] *= act_grad(y[q][j], ACT);
  }
}

// WPB: waves per workgroup (4 or 8): 8 halves the number of workgroups and with it the weight stagings of a launch
template <int QD, int WPB = 4>
__global__ void __launch_bounds__(64 * WPB, (QD == 4 ? 6 : 3) * 4 / WPB) chain_fwd_kernel(const ChainArgs a) {
  extern __shared__ float chain_lds[];
  chain_stage_weights(a, chain_lds, true);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
  const long ntiles = (a.rows + 32 * WPB - 1) / (32 * WPB);
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {      // (workgroups are persistent: the weights are staged once)
    const long row = (tile * WPB + wave) * 32 + n;
    const bool row_ok = row < a.rows;
    const long rowc = row_ok ? row : a.rows - 1;
    float b[QD][4];
#pragma unroll 1
    for (int k = 0; k < a.n; ++k) {
      const ChainLayer& L = a.l[k];
      const float* w = chain_lds + L.off;
      const int S = L.S, OUT = L.OUT, IN = L.IN, mt = OUT > 32 ? 2 : 1;
      f32x16t acc[QD / 4];
#pragma unroll
      for (int m = 0; m < QD / 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
      if (k == 0) {
        // first layer: any width, QD quads of the wave's rows in flight at a time
        const float* xr = a.X + rowc * (long)a.ldx;
        const int nq = (IN + 7) >> 3;
        for (int q0 = 0; q0 < nq; q0 += QD) {
#pragma unroll
          for (int u = 0; u < QD; ++u) load4_mask(xr, q0 + u, half, IN, b[u]);
          if (a.xscale != 1.f) {
#pragma unroll
            for (int u = 0; u < QD; ++u)
#pragma unroll
              for (int j = 0; j < 4; ++j) b[u][j] *= a.xscale;
          }
          chain_layer_fwd<QD>(w + 8 * q0, S, n, half, mt, nq - q0, b, acc);
        }
      } else {
        chain_layer_fwd<QD>(w, S, n, half, mt, (IN + 7) >> 3, b, acc);
      }
      if (k == 0 && a.add) {
        const float* ar = a.add + (rowc / a.addV) * (long)a.ldadd;
#pragma unroll
        for (int m = 0; m < QD / 4; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float o[4];
            load4_mask(ar, 4 * m + g, half, OUT, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][4 * g + j] += o[j];
          }
      }
      const float* bias = w + chain_rows(OUT) * S;
      switch (L.act) {
        case ACT_ELU: chain_act_fwd<QD, ACT_ELU>(acc, bias, OUT, half, b); break;
        case ACT_RELU: chain_act_fwd<QD, ACT_RELU>(acc, bias, OUT, half, b); break;
        case ACT_SIGMOID: chain_act_fwd<QD, ACT_SIGMOID>(acc, bias, OUT, half, b); break;
        case ACT_SOFTPLUS: chain_act_fwd<QD, ACT_SOFTPLUS>(acc, bias, OUT, half, b); break;
        case ACT_SOFTPLUS_B: chain_act_fwd<QD, ACT_SOFTPLUS_B>(acc, bias, OUT, half, b); break;
        default: chain_act_fwd<QD, ACT_NONE>(acc, bias, OUT, half, b); break;
      }
      if (row_ok) {
        float* yr = L.Y + row * (long)L.ldy;
#pragma unroll
        for (int q = 0; q < QD; ++q) {
          store4_guard(yr, q, half, OUT, b[q]);
        }
      }
    }
  }
}

template <int QD, int WPB = 4>
__global__ void __launch_bounds__(64 * WPB, QD == 4 ? (WPB == 4 ? 5 : 3) : (WPB == 4 ? 2 : 1)) chain_bwd_kernel(const ChainArgs a) {
  extern __shared__ float chain_lds[];
  chain_stage_weights(a, chain_lds, false);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
  const long ntiles = (a.rows + 32 * WPB - 1) / (32 * WPB);
  constexpr int MTM = QD / 4;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long row = (tile * WPB + wave) * 32 + n;
    const bool row_ok = row < a.rows;
    const long rowc = row_ok ? row : a.rows - 1;
    float d[QD][4];
    {
      const float* dr = a.l[0].D + rowc * (long)a.l[0].ldd;
      const int OUT = a.l[0].OUT;
#pragma unroll
      for (int q = 0; q < QD; ++q) load4_mask(dr, q, half, OUT, d[q]);
    }
#pragma unroll 1
    for (int k = 0; k < a.n; ++k) {
      const ChainLayer& L = a.l[k];
      const float* w = chain_lds + L.off;
      const int S = L.S, OUT = L.OUT, IN = L.IN, nko = (OUT + 7) >> 3;
      // delta = dY * act'(Y), stored for the weight-gradient kernel (a top layer without activation: dY is delta already)
      if (L.act != ACT_NONE) {
        const float* yr = L.Y + rowc * (long)L.ldy;
        float y[QD][4];
#pragma unroll
        for (int q = 0; q < QD; ++q) load4_mask(yr, q, half, OUT, y[q]);
        switch (L.act) {
          case ACT_ELU: chain_act_bwd<QD, ACT_ELU>(y, d); break;
          case ACT_RELU: chain_act_bwd<QD, ACT_RELU>(y, d); break;
          case ACT_SIGMOID: chain_act_bwd<QD, ACT_SIGMOID>(y, d); break;
          case ACT_SOFTPLUS: chain_act_bwd<QD, ACT_SOFTPLUS>(y, d); break;
          default: chain_act_bwd<QD, ACT_SOFTPLUS_B>(y, d); break;
        }
      }
#ifdef PGRF_TRAIN_ABL_NO_DELTA   // timing experiment: delta is not written (what a fusion of the dW products into this kernel would save)
      if (false) {
#else
      if ((L.act != ACT_NONE || k > 0) && row_ok) {
#endif
        float* dr = L.D + row * (long)L.ldd;
#pragma unroll
        for (int q = 0; q < QD; ++q) {
          store4_guard(dr, q, half, OUT, d[q]);
        }
      }
      if (!L.dx) break;
      const int mtiles = (IN + 31) >> 5;
      if (k + 1 == a.n) {
        // the chain's output: dX (=|+=) delta W, any width, QD / 4 32-column tiles at a time
        if (!a.dX) break;
        if (a.xscale != 1.f) {
#pragma unroll
          for (int q = 0; q < QD; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[q][j] *= a.xscale;
        }
        float* xr = a.dX + rowc * (long)a.lddx;
        for (int m0 = 0; m0 < mtiles; m0 += MTM) {
          f32x16t acc[MTM];
          const int mt = mtiles - m0 >= MTM ? MTM : 1;
#pragma unroll
          for (int m = 0; m < MTM; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float o[4] = {0.f, 0.f, 0.f, 0.f};
              if (a.accumulate && m < mt) load4_mask(xr, 4 * (m0 + m) + g, half, IN, o);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[m][4 * g + j] = o[j];
            }
          chain_layer_bwd<QD>(w, S, 32 * m0, n, half, mt, nko, d, acc);
          if (!row_ok) continue;
#pragma unroll
          for (int m = 0; m < MTM; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float v[4] = {acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
              if (m < mt) store4_guard(xr, 4 * (m0 + m) + g, half, IN, v);
            }
        }
      } else {
        // dY of the next layer down stays in registers (IN <= 8 QD)
        f32x16t acc[MTM];
#pragma unroll
        for (int m = 0; m < MTM; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        chain_layer_bwd<QD>(w, S, 0, n, half, mtiles, nko, d, acc);
#pragma unroll
        for (int m = 0; m < MTM; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[4 * m + g][j] = (32 * m + 8 * g + 4 * half + j < IN) ? acc[m][4 * g + j] : 0.f;
      }
    }
  }
}

// dY[r, c] *= act'(Y[r, c]);  gbias[c] += sum_r dY[r, c].  cols <= 64: thread = (column, row slot), the column sum stays in a register
// over the block's rows, then one LDS reduction over the row slots and ONE atomic per column and block (the first version did an
// LDS atomic per element: 2.4 ms of a 16 ms training step).
__global__ void __launch_bounds__(256) act_bwd_kernel(float* __restrict__ dY, int lddy, const float* __restrict__ Y, int ldy, long rows,
                                                      int cols, int cp /* power of two >= cols, <= 64 */, int act, float* __restrict__ gbias) {
  __shared__ float s_b[256];
  const int c = threadIdx.x & (cp - 1), slot = threadIdx.x / cp, nslot = 256 / cp;
  float sum = 0.f;
  if (c < cols) {
    // four rows per trip with their loads issued together (one dependent load pair per trip ran at half the HBM rate)
    const long step = (long)gridDim.x * nslot;
    long r = (long)blockIdx.x * nslot + slot;
    for (; r + 3 * step < rows; r += 4 * step) {
      float dv[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { dv[u] = dY[(r + u * step) * lddy + c]; yv[u] = Y[(r + u * step) * ldy + c]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = dv[u] * act_grad(yv[u], act);
        dY[(r + u * step) * lddy + c] = d;
        sum += d;
      }
    }
    for (; r < rows; r += step) {
      const float d = dY[r * lddy + c] * act_grad(Y[r * ldy + c], act);
      dY[r * lddy + c] = d;
      sum += d;
    }
  }
  if (!gbias) return;
  s_b[threadIdx.x] = sum;
  __syncthreads();
  if (slot == 0 && c < cols) {
    float t = 0.f;
    for (int k = 0; k < nslot; ++k) t += s_b[k * cp + c];
    if (t != 0.f) unsafeAtomicAdd(&gbias[c], t);
  }
}

// dst[r, dc + c] (=|+=) src[r, sc + c] * scale  for c < cols: column-slice copies between concatenated buffers
__global__ void slice_kernel(float* __restrict__ dst, int ldd, int dc, const float* __restrict__ src, int lds, int sc, long rows, int cols,
                             int accumulate) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i; const long r = iu / (unsigned)cols; const int c = (int)(iu % (unsigned)cols);      // (n < 2^32: host guard; a 64-bit divide is ~100 instructions)
    const float v = src[r * lds + sc + c];
    float* d = dst + r * ldd + dc + c;
    *d = accumulate ? *d + v : v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// geometry of every (point, view): projection into the source ERP (render_ops.py:158-209, spt_utils.py:128-192), bilinear tap
// setups (ops.py:32-52, grid_sample border padding), ray-direction difference (aggregate_net.py:8-14), gathered source colour,
// and per point the interval half-widths in normalised inverse depth (render_ops.py:110-122).  Constants w.r.t. every
// trainable quantity: computed once, used by the forward recompute and by the gather backward.
struct Taps { int o[4]; float w[4]; };
__device__ __forceinline__ Taps tap_setup(float xn, float yn, int fh, int fw, bool align) {
  float ix, iy;
  if (align) { ix = (xn + 1.f) / 2.f * (float)(fw - 1); iy = (yn + 1.f) / 2.f * (float)(fh - 1); }
  else { ix = ((xn + 1.f) * (float)fw - 1.f) / 2.f; iy = ((yn + 1.f) * (float)fh - 1.f) / 2.f; }
  ix = fminf(fmaxf(ix, 0.f), (float)(fw - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(fh - 1));
  const float fx0 = floorf(ix), fy0 = floorf(iy), fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
  Taps t;
  t.w[0] = (fx1 - ix) * (fy1 - iy); t.w[1] = (ix - fx0) * (fy1 - iy); t.w[2] = (fx1 - ix) * (iy - fy0); t.w[3] = (ix - fx0) * (iy - fy0);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = min(x0 + 1, fw - 1), y1 = min(y0 + 1, fh - 1);
  t.o[0] = y0 * fw + x0; t.o[1] = y0 * fw + x1; t.o[2] = y1 * fw + x0; t.o[3] = y1 * fw + x1;
  return t;
}

struct GeomOut {
  int* tap_rf_o; float* tap_rf_w;     // [PV,4] each
  int* tap_if_o; float* tap_if_w;
  float* rgb;                          // [PV,3]
  float* dd;                           // [PV,4]
  float* depth_v;                      // [PV]
  float* hw;                           // [P,2] near / far half width
};

__global__ void geometry_kernel(const PassArgs a, int dn, int V, GeomOut o) {
  const long PV = (long)a.rn * dn * V;
  const long pv = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pv >= PV) return;
  const int v = (int)(pv % V);
  const long p = pv / V;
  const int ray = (int)(p / dn), s = (int)(p % dn);
  float wx, wy, wz;
  ray_direction<true>(a, ray, wx, wy, wz);
  const float* drow = a.depth + (size_t)ray * dn;
  const float td = drow[s];
  const float P0 = a.c2w[3] + wx * td, P1 = a.c2w[7] + wy * td, P2 = a.c2w[11] + wz * td;
  const float wn = sqrtf(wx * wx + wy * wy + wz * wz);
  const float q0 = -wx / wn, q1 = -wy / wn, q2 = -wz / wn;
  if (v == 0) {
    const float qn = -1.f / a.que_near, qf = -1.f / a.que_far;
    const float d_c = (-1.f / td - qn) / (qf - qn);
    const float d_p = (-1.f / drow[max(s - 1, 0)] - qn) / (qf - qn);
    const float d_n = (-1.f / drow[min(s + 1, dn - 1)] - qn) / (qf - qn);
    const float dist_c = (s == dn - 1) ? 1e6f : d_n - d_c;
    const float dist_p = (s == 0) ? dist_c : d_c - d_p;
    o.hw[p * 2] = dist_p / 2.f;
    o.hw[p * 2 + 1] = dist_c / 2.f;
  }
  const float* M = a.w2c + v * 12;
  const float cx = M[0] * P0 + M[1] * P1 + M[2] * P2 + M[3];
  const float cy = M[4] * P0 + M[5] * P1 + M[6] * P2 + M[7];
  const float cz = M[8] * P0 + M[9] * P1 + M[10] * P2 + M[11];
  const float r = sqrtf(cx * cx + cy * cy + cz * cz);
  float px, py;
  if (a.dataset == PGRF_DATASET_M3D) {
    float th = atan2f(cz, cx) + PGRF_HALF_PI_F;
    if (th < 0.f) th += PGRF_TWO_PI_F;
    const float ph = acosf(cy / (r + 1e-5f));
    px = th / PGRF_TWO_PI_F * (float)(a.W - 1);
    py = ph / PGRF_PI_F * (float)(a.H - 1);
  } else if (a.dataset == PGRF_DATASET_REPLICA_TEST) {
    const float th = atan2f(cx, cz), ph = -asinf(cy / r);
    px = (float)((double)(a.W - 1) / (2.0 * 3.14159265358979323846)) * (th + PGRF_PI_F);
    py = (float)((double)(a.H - 1) / 3.14159265358979323846) * (-ph + PGRF_HALF_PI_F);
  } else if (a.dataset == PGRF_DATASET_RESIDENTIAL) {
    float th = -atan2f(-cz, cx);
    const float ph = asinf(cy / r);
    if (th > PGRF_HALF_PI_F && th <= PGRF_TWO_PI_F) th -= PGRF_TWO_PI_F;
    px = (0.15915494309189535f * th + 0.75f) * (float)(a.W - 1);
    py = (0.5f - ph / PGRF_PI_F) * (float)(a.H - 1);
  } else {
    float th = atan2f(cy, cx);
    const float ph = acosf(cz / r);
    if (th < 0.f) th += PGRF_TWO_PI_F;
    px = (float)(a.W - 1) * (1.f - th / PGRF_TWO_PI_F);
    py = ph * (float)(a.H - 1) / PGRF_PI_F;
  }
  o.depth_v[pv] = r;
  const float ccx = -(M[0] * M[3] + M[4] * M[7] + M[8] * M[11]);
  const float ccy = -(M[1] * M[3] + M[5] * M[7] + M[9] * M[11]);
  const float ccz = -(M[2] * M[3] + M[6] * M[7] + M[10] * M[11]);
  float dx = P0 - ccx, dy = P1 - ccy, dz = P2 - ccz;
  const float dn_ = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-5f);
  dx = -dx / dn_; dy = -dy / dn_; dz = -dz / dn_;
  o.dd[pv * 4 + 0] = dx - q0; o.dd[pv * 4 + 1] = dy - q1; o.dd[pv * 4 + 2] = dz - q2;
  o.dd[pv * 4 + 3] = dx * q0 + dy * q1 + dz * q2;
  const float xn = px / (float)(a.W - 1) * 2.f - 1.f, yn = py / (float)(a.H - 1) * 2.f - 1.f;
  const Taps trf = tap_setup(xn, yn, a.hr, a.wr, a.hr == a.H && a.wr == a.W);
  const Taps tif = tap_setup(xn, yn, a.hf, a.wf, a.hf == a.H && a.wf == a.W);
  const Taps tim = tap_setup(xn, yn, a.H, a.W, true);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o.tap_rf_o[pv * 4 + i] = trf.o[i]; o.tap_rf_w[pv * 4 + i] = trf.w[i];
    o.tap_if_o[pv * 4 + i] = tif.o[i]; o.tap_if_w[pv * 4 + i] = tif.w[i];
  }
  const float* ib = a.imgs + (size_t)v * a.H * a.W * 4;
  float c[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 t = *reinterpret_cast<const float4*>(ib + (size_t)tim.o[i] * 4);
    c[0] += t.x * tim.w[i]; c[1] += t.y * tim.w[i]; c[2] += t.z * tim.w[i];
  }
  o.rgb[pv * 3] = c[0]; o.rgb[pv * 3 + 1] = c[1]; o.rgb[pv * 3 + 2] = c[2];
}

// out[pv, dc + c] = sum_t w_t * map[v][o_t][c]  for a 32-channel channel-last map (render_ops.py:243-245)
__global__ void gather32_kernel(const float* __restrict__ map, long view_stride, const int* __restrict__ to, const float* __restrict__ tw,
                                long PV, int V, float* __restrict__ out, int ldo, int dc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PV * 8) return;
  const long pv = i / 8; const int q = (int)(i % 8);
  const float* mb = map + (pv % V) * view_stride + q * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 m = *reinterpret_cast<const float4*>(mb + (size_t)to[pv * 4 + t] * 32);
    const float w = tw[pv * 4 + t];
    s.x += m.x * w; s.y += m.y * w; s.z += m.z * w; s.w += m.w * w;
  }
  *reinterpret_cast<float4*>(out + pv * ldo + dc + q * 4) = s;
}
// rgb_feat = [rgb | gathered img_feats] + dir_feat -> xin[:, 140:175]  (ibrnet.py:325-327): the gather of the image features with the direction
// feature added on its way out (round 5: was the gather into a [PV, 32] buffer + an elementwise launch that re-read it)
__global__ void gather32_rgbfeat_kernel(const float* __restrict__ map, long view_stride, const int* __restrict__ to, const float* __restrict__ tw,
                                        long PV, int V, const float* __restrict__ rgb, const float* __restrict__ dfeat, float* __restrict__ xin) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PV * 8) return;
  const long pv = i / 8; const int q = (int)(i % 8);
  const float* mb = map + (pv % V) * view_stride + q * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 m = *reinterpret_cast<const float4*>(mb + (size_t)to[pv * 4 + t] * 32);
    const float w = tw[pv * 4 + t];
    s.x += m.x * w; s.y += m.y * w; s.z += m.z * w; s.w += m.w * w;
  }
  const float* df = dfeat + pv * 35;
  float* xo = xin + pv * 207 + 140;
  xo[3 + q * 4] = s.x + df[3 + q * 4]; xo[4 + q * 4] = s.y + df[4 + q * 4]; xo[5 + q * 4] = s.z + df[5 + q * 4]; xo[6 + q * 4] = s.w + df[6 + q * 4];
  if (q == 0) { xo[0] = rgb[pv * 3] + df[0]; xo[1] = rgb[pv * 3 + 1] + df[1]; xo[2] = rgb[pv * 3 + 2] + df[2]; }
}
// gmap[v][o_t][c] += w_t * g[pv, c]   (transpose of the gather: scatter-add with atomics)
__global__ void gather32_bwd_kernel(float* __restrict__ gmap, long view_stride, const int* __restrict__ to, const float* __restrict__ tw,
                                    long PV, int V, const float* __restrict__ g, int ldg, int gc) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < PV * 32; i += (long)gridDim.x * blockDim.x) {
    const long pv = i / 32; const int c = (int)(i % 32);
    const float gv = g[pv * ldg + gc + c];
    if (gv == 0.f) continue;
    float* mb = gmap + (pv % V) * view_stride + c;
  #pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float w = tw[pv * 4 + t];
      if (w != 0.f) unsafeAtomicAdd(mb + (size_t)to[pv * 4 + t] * 32, gv * w);
    }
  }
}

// The same scatter with the taps of a RAY merged before they reach memory (round 5).  The kernel above issues one 128-byte atomic per
// (sample, view, tap): 33.5 M float atomics for 2 048 rays x 64 samples x 2 views, ~206 us whatever the launch shape (device-scope
// float atomics execute beyond the XCD's L2).  But the samples of a ray project onto an epipolar CURVE of the source map: consecutive
// samples share most of their four texels.  Here one wave = one (ray, view), lane & 31 = channel, and the two lane halves walk the two
// halves of the ray's samples in order, each keeping the last kGE texels it touched as (texel, partial sum) pairs in registers (a FIFO:
// a miss flushes the oldest pair with ONE atomic and shifts); everything is flushed at the end of the walk.  The taps of kGB samples
// are fetched together (a walk of dependent loads would be one memory round trip per sample).  Same products, the sum of a texel's
// contributions merely associates per ray first.
#ifndef PGRF_SCATTER_GE
#define PGRF_SCATTER_GE 6
#endif
#ifndef PGRF_SCATTER_GB
#define PGRF_SCATTER_GB 4
#endif
#ifndef PGRF_SCATTER_GW
#define PGRF_SCATTER_GW 1
#endif
constexpr int kGE = PGRF_SCATTER_GE, kGB = PGRF_SCATTER_GB, kGW = PGRF_SCATTER_GW;     // FIFO depth, samples fetched together, waves per (ray, view)
__global__ void __launch_bounds__(256) gather32_bwd_ray_kernel(float* __restrict__ gmap, long view_stride, const int* __restrict__ to,
                                                               const float* __restrict__ tw, int rn, int dn, int V,
                                                               const float* __restrict__ g, int ldg, int gc) {
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= (long)rn * V * kGW) return;
  const int part = (int)(wv % kGW), v = (int)((wv / kGW) % V), ray = (int)(wv / ((long)kGW * V));
  const int lane = threadIdx.x & 63, c = lane & 31, half = lane >> 5;
  const int seg = (dn + 2 * kGW - 1) / (2 * kGW), s0 = (part * 2 + half) * seg, s1 = (s0 + seg < dn) ? s0 + seg : dn;
  float* __restrict__ mb = gmap + (size_t)v * view_stride + c;
  int key[kGE];
  float val[kGE];
#pragma unroll
  for (int e = 0; e < kGE; ++e) { key[e] = -1; val[e] = 0.f; }
  for (int sb = s0; sb < s1; sb += kGB) {
    float gv[kGB];
    int4 o4[kGB];
    float4 w4[kGB];
#pragma unroll
    for (int u = 0; u < kGB; ++u) {
      const int s = (sb + u < s1) ? sb + u : s1 - 1;
      const long pv = ((long)ray * dn + s) * V + v;
      gv[u] = g[pv * ldg + gc + c];
      o4[u] = *reinterpret_cast<const int4*>(to + pv * 4);
      w4[u] = *reinterpret_cast<const float4*>(tw + pv * 4);
    }
#pragma unroll
    for (int u = 0; u < kGB; ++u) {
      const bool live = sb + u < s1;
      const int o[4] = {o4[u].x, o4[u].y, o4[u].z, o4[u].w};
      const float wt[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (live && wt[t] != 0.f) {                 // (zero weight: a tap outside the map; uniform over the half)
          const float contrib = gv[u] * wt[t];
          bool hit = false;
#pragma unroll
          for (int e = 0; e < kGE; ++e) {
            const bool m = key[e] == o[t];
            val[e] += m ? contrib : 0.f;
            hit = hit || m;
          }
          if (!hit) {
            if (key[kGE - 1] >= 0 && val[kGE - 1] != 0.f) unsafeAtomicAdd(mb + (size_t)key[kGE - 1] * 32, val[kGE - 1]);
#pragma unroll
            for (int e = kGE - 1; e > 0; --e) { key[e] = key[e - 1]; val[e] = val[e - 1]; }
            key[0] = o[t]; val[0] = contrib;
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < kGE; ++e)
    if (key[e] >= 0 && val[e] != 0.f) unsafeAtomicAdd(mb + (size_t)key[e] * 32, val[e]);
}

// ---------------------------------------------------------------------------------------------------------------
// compute_prob (dist_decoder.py:6-51 is_ref branch, :109-140): mixture of two logistics -> visibility, hit probability
struct ProbIO {
  const float* depth_v;  // [PV]
  const float* hw;       // [P,2]
  const float* ref_dr;   // [V,2]
  const float *mean, *var, *aw, *visp;   // [PV,2] each; aw / visp use column 0 only (visp null without use_vis)
  float *vis, *hit;      // [PV]
};
__device__ __forceinline__ float logistic2(float d) { return 1.f / (1.f + expf(-2.f * d)); }      // 0.5 + 0.5 tanh(d)
__global__ void prob_kernel(ProbIO io, long PV, int V, float* __restrict__ pe_in /* [PV,34]: columns 32, 33 written */) {
  const long pv = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pv >= PV) return;
  const int v = (int)(pv % V); const long p = pv / V;
  const float rn_ = -1.f / io.ref_dr[2 * v], rf_ = -1.f / io.ref_dr[2 * v + 1];
  const float z = (-1.f / fmaxf(io.depth_v[pv], 1e-5f) - rn_) / (rf_ - rn_);
  const float nearp = z - io.hw[p * 2], farp = z + io.hw[p * 2 + 1];
  const float aw = io.aw[pv * 2], u = io.visp ? io.visp[pv * 2] : 1.f;
  float vis = 0.f, hit = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float m = io.mean[pv * 2 + k], s = io.var[pv * 2 + k];
    const float c0 = logistic2((nearp - m) * s) * u, c1 = logistic2((farp - m) * s) * u;
    const float mix = k == 0 ? aw : 1.f - aw;
    vis += (1.f - c0) * mix; hit += (c1 - c0) * mix;
  }
  io.vis[pv] = vis; io.hit[pv] = hit;
  pe_in[pv * 34 + 32] = (hit - 0.5f) * 2.f;
  pe_in[pv * 34 + 33] = (vis - 0.5f) * 2.f;
}
struct ProbGrad { const float* g_pe_in; /* [PV,34] */ float *g_mean, *g_var, *g_aw, *g_visp; /* [PV,2] each, column 0 for aw / visp */ };
__global__ void prob_bwd_kernel(ProbIO io, ProbGrad gr, long PV, int V) {
  const long pv = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pv >= PV) return;
  const int v = (int)(pv % V); const long p = pv / V;
  const float ghit = 2.f * gr.g_pe_in[pv * 34 + 32], gvis = 2.f * gr.g_pe_in[pv * 34 + 33];
  const float rn_ = -1.f / io.ref_dr[2 * v], rf_ = -1.f / io.ref_dr[2 * v + 1];
  const float z = (-1.f / fmaxf(io.depth_v[pv], 1e-5f) - rn_) / (rf_ - rn_);
  const float nearp = z - io.hw[p * 2], farp = z + io.hw[p * 2 + 1];
  const float aw = io.aw[pv * 2], u = io.visp ? io.visp[pv * 2] : 1.f;
  float gaw = 0.f, gu = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float m = io.mean[pv * 2 + k], s = io.var[pv * 2 + k];
    const float r0 = logistic2((nearp - m) * s), r1 = logistic2((farp - m) * s);
    const float c0 = r0 * u, c1 = r1 * u;
    const float mix = k == 0 ? aw : 1.f - aw;
    const float gmix = (1.f - c0) * gvis + (c1 - c0) * ghit;
    gaw += k == 0 ? gmix : -gmix;
    const float gc0 = -mix * (gvis + ghit), gc1 = mix * ghit;
    gu += gc0 * r0 + gc1 * r1;
    const float ga0 = gc0 * u * 2.f * r0 * (1.f - r0), ga1 = gc1 * u * 2.f * r1 * (1.f - r1);
    gr.g_mean[pv * 2 + k] = -s * (ga0 + ga1);
    gr.g_var[pv * 2 + k] = (nearp - m) * ga0 + (farp - m) * ga1;
  }
  gr.g_aw[pv * 2] = gaw;
  if (gr.g_visp) gr.g_visp[pv * 2] = gu;
}

// ---------------------------------------------------------------------------------------------------------------
// weighted mean / variance over the views of a point (ibrnet.py:112-116): mean = sum_v w x, var = sum_v w (x - mean)^2
// x [PV, ldx] (columns xc .. xc+F), w [PV] -> out [P, ldo]: mean at column mc, var at column vc
// (round 5: NS weightings of the same rows in one launch -- x is read once)
struct StatFwd { const float* w; int mc, vc; };
template <int NS>
__global__ void stats_kernel(const float* __restrict__ x, int ldx, int xc, int V, int F, long P, float* __restrict__ out, int ldo, StatFwd s0, StatFwd s1) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * F) return;
  const unsigned iu = (unsigned)i; const long p = iu / (unsigned)F; const int j = (int)(iu % (unsigned)F);
  float xs[4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
    if (v < V) xs[v] = x[(p * V + v) * ldx + xc + j];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const StatFwd& S = k == 0 ? s0 : s1;
    float ws[4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (v < V) ws[v] = S.w[p * V + v];
    float m = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (v < V) m += xs[v] * ws[v];
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (v < V) { const float d = xs[v] - m; var += ws[v] * (d * d); }
    out[p * ldo + S.mc + j] = m;
    out[p * ldo + S.vc + j] = var;
  }
}
// gx[PV, .] += ..., gw[PV] += ...   A WAVE per point, lane = feature (F <= 64, V <= 4): every access is a run of F consecutive
// floats (the first version gave a thread one (point, view) row and walked the features at a stride of ld floats: 744 us per call
// for 262 144 rows against ~25 us of traffic); the per-view weight gradient is a wave reduction.
// Round 5: (i) NS = 2 statistics of the SAME rows under two weightings (ibrnet.py:128-137: weight0 from neuray_fc, weight1 uniform) in one
// sweep -- x is read and gx read-modified-written once, the two terms added in the order the two launches added them: 2 x 100 us -> 139 us;
// (ii) PGRF_STAT_PTS points per wave, two in flight (measured per 2 048-ray training step, tools/ab_train_variants.sh: 1 point 5.95 ms,
// 2 points 5.89, 4 points 6.07: the kernel is bound by the round trips of short rows at a 207-float stride, and wants waves, not long ones).
struct StatSet { const float* w; int mc, vc, gmc, gvc; float* gw; float gws = 1.f; };    // weights [PV], stat / g columns of (mean, var), gw [PV] or null; gws: stats_bwd4_kernel WRITES gws * (weight gradient)
#ifndef PGRF_STAT_PTS
#define PGRF_STAT_PTS 2
#endif
constexpr int kStatPts = PGRF_STAT_PTS, kStatFly = kStatPts >= 2 ? 2 : 1;      // points per wave, points in flight
template <int NS>
__global__ void __launch_bounds__(256) stats_bwd_kernel(const float* __restrict__ x, int ldx, int xc, int V, int F, long P,
                                                        const float* __restrict__ stat, int lds, const float* __restrict__ g, int ldg,
                                                        float* __restrict__ gx, int ldgx, int gxc, StatSet s0, StatSet s1) {
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int j = threadIdx.x & 63;
  const bool on = j < F;
  const int jj = on ? j : 0;
  const long p_beg = wv * kStatPts, p_end = (p_beg + kStatPts < P) ? p_beg + kStatPts : P;
  for (long p0 = p_beg; p0 < p_end; p0 += kStatFly) {
    float xs[kStatFly][4], go[kStatFly][4], ws[kStatFly][NS][4], m[kStatFly][NS], gm[kStatFly][NS], gvar[kStatFly][NS];
    bool ok[kStatFly];
    // every load of the two points first
#pragma unroll
    for (int e = 0; e < kStatFly; ++e) {
      ok[e] = p0 + e < p_end;
      const long p = ok[e] ? p0 + e : p0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < V) {
          xs[e][u] = x[(p * V + u) * ldx + xc + jj];
          go[e][u] = gx[(p * V + u) * ldgx + gxc + jj];
        }
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const StatSet& S = k == 0 ? s0 : s1;
        m[e][k] = stat[p * lds + S.mc + jj]; gm[e][k] = g[p * ldg + S.gmc + jj]; gvar[e][k] = g[p * ldg + S.gvc + jj];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < V) ws[e][k][u] = S.w[p * V + u];
      }
    }
#pragma unroll
    for (int e = 0; e < kStatFly; ++e) {
      if (!ok[e]) break;                      // (wave-uniform)
      const long p = p0 + e;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const StatSet& S = k == 0 ? s0 : s1;
        float Sw = 0.f;                       // sum_u w_u (x_u - m): zero only when the weights sum to one
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < V) Sw += ws[e][k][u] * (xs[e][u] - m[e][k]);
        const float gmt = gm[e][k] - 2.f * Sw * gvar[e][k];    // total gradient reaching the mean
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u < V) {
            const float d = xs[e][u] - m[e][k];
            go[e][u] += ws[e][k][u] * gmt + 2.f * ws[e][k][u] * d * gvar[e][k];
            if (S.gw) {
              float part = on ? xs[e][u] * gmt + d * d * gvar[e][k] : 0.f;
#pragma unroll
              for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
              if (j == 0) S.gw[p * V + u] += part;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < V && on) gx[(p * V + u) * ldgx + gxc + j] = go[e][u];
    }
  }
}

// The same with FOUR features per lane (round 5, late): 16 lanes per point (F <= 64: lane li holds columns 4 li .. 4 li + 3), four points per
// wave, every row access one dword-aligned 16-byte load per lane; the last quad of a row is masked after the load and stored column by
// column (the next column belongs to someone else).  A quarter of the memory instructions of the lane-per-feature form for the same bytes.
// W2 (the pooled-geometry statistics, ibrnet.py:352-358): set 0's weights are wgt2 = v2 / (sum_v v2 + 1e-8) with v2 = x[:, w2x], and g[:, w2g]
// carries d / d(mean_v wgt2): the gradient of v2 (was wgt2_bwd_kernel behind a zeroed gwgt2 buffer) is formed here from the per-view weight
// gradients the wave has just reduced, and added to gx[:, w2x]; s0.gw is not written.
template <int NS, bool W2 = false>
__global__ void __launch_bounds__(256) stats_bwd4_kernel(const float* __restrict__ x, int ldx, int xc, int V, int F, long P,
                                                         const float* __restrict__ stat, int lds, const float* __restrict__ g, int ldg,
                                                         float* __restrict__ gx, int ldgx, int gxc, StatSet s0, StatSet s1, int w2x = 0, int w2g = 0) {
  const int lane = threadIdx.x & 63, li = lane & 15, c0 = 4 * li;
  const long p = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool pok = p < P;
  const long pc = pok ? p : P - 1;
  const bool on = c0 < F;                       // this lane has at least one live column
  const int cc = on ? c0 : 0;
  unsigned msk[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) msk[e] = (on && c0 + e < F) ? 0xffffffffu : 0u;
  // 16-byte row accesses only where all four columns exist: the lane that holds a ragged tail (F not a multiple of four) reads its
  // live columns one by one, so nothing is touched past column F - 1 (the last row of a buffer has no padding behind it)
  const bool full = on ? (c0 + 4 <= F) : (F >= 4);
  auto ld4 = [&](const float* q, float (&o)[4]) {
    if (full) {
      const F4U t = *reinterpret_cast<const F4U*>(q);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = __uint_as_float(__float_as_uint(t.v[e]) & msk[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = msk[e] ? q[e] : 0.f;
    }
  };
  float xs[4][4], go[4][4];
  float gw2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (u < V) {
      ld4(x + (pc * V + u) * ldx + xc + cc, xs[u]);
      ld4(gx + (pc * V + u) * ldgx + gxc + cc, go[u]);
    }
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const StatSet& S = k == 0 ? s0 : s1;
    float m[4], gm[4], gvar[4], ws[4];
    ld4(stat + pc * lds + S.mc + cc, m);
    ld4(g + pc * ldg + S.gmc + cc, gm);
    ld4(g + pc * ldg + S.gvc + cc, gvar);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < V) ws[u] = S.w[pc * V + u];
    float gmt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float Sw = 0.f;                           // sum_u w_u (x_u - m): zero only when the weights sum to one
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < V) Sw += ws[u] * (xs[u][e] - m[e]);
      gmt[e] = gm[e] - 2.f * Sw * gvar[e];      // total gradient reaching the mean
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u < V) {
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xs[u][e] - m[e];
          go[u][e] += ws[u] * gmt[e] + 2.f * ws[u] * d * gvar[e];
          part += msk[e] ? xs[u][e] * gmt[e] + d * d * gvar[e] : 0.f;
        }
        if (S.gw || (W2 && k == 0)) {
#pragma unroll
          for (int off = 8; off >= 1; off >>= 1) part += __shfl_xor(part, off, 16);
          if (W2 && k == 0) gw2[u] = part;
          else if (li == 0 && pok) S.gw[p * V + u] = part * S.gws;      // (written: this kernel is the buffer's only producer)
        }
      }
    }
  }
  if (W2 && li == 0 && pok) {
    float v2[4], S2 = 0.f, dot = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < V) { v2[u] = x[(p * V + u) * ldx + w2x]; S2 += v2[u]; }
    const float den = S2 + 1e-8f, gmw = g[p * ldg + w2g] / (float)V;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < V) dot += (gw2[u] + gmw) * v2[u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < V) gx[(p * V + u) * ldgx + w2x] += (gw2[u] + gmw) / den - dot / (den * den);
  }
  if (pok && on) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < V) {
        float* q = gx + (p * V + u) * ldgx + gxc + c0;
        if (c0 + 4 <= F) { F4U o; o.v[0] = go[u][0]; o.v[1] = go[u][1]; o.v[2] = go[u][2]; o.v[3] = go[u][3]; *reinterpret_cast<F4U*>(q) = o; }
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c0 + e < F) q[e] = go[u][e];
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// ray attention (ibrnet.py:7-27,52-102): 4 heads of dimension 4 over the dn samples of a ray, temperature 2; for ONE
// source view the reference's mask makes every logit equal (uniform weights).  q, k, v, o: [P,16] with head h at columns 4h..
// One thread per (ray, head, query sample); the probabilities are recomputed in the backward (never stored).
__global__ void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ vv, int rn, int dn,
                                int uniform, float* __restrict__ o) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rn * 4 * dn) return;
  const int s = (int)(i % dn), h = (int)((i / dn) % 4), ray = (int)(i / (4L * dn));
  const float* kb = k + (size_t)ray * dn * 16 + h * 4;
  const float* vb = vv + (size_t)ray * dn * 16 + h * 4;
  const float* qp = q + ((size_t)ray * dn + s) * 16 + h * 4;
  const float q0 = qp[0] * 0.5f, q1 = qp[1] * 0.5f, q2 = qp[2] * 0.5f, q3 = qp[3] * 0.5f;
  float mx = -3.0e38f;
  if (!uniform)
    for (int t = 0; t < dn; ++t) mx = fmaxf(mx, q0 * kb[t * 16] + q1 * kb[t * 16 + 1] + q2 * kb[t * 16 + 2] + q3 * kb[t * 16 + 3]);
  float l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int t = 0; t < dn; ++t) {
    const float e = uniform ? 1.f : expf(q0 * kb[t * 16] + q1 * kb[t * 16 + 1] + q2 * kb[t * 16 + 2] + q3 * kb[t * 16 + 3] - mx);
    l += e;
    a0 += e * vb[t * 16]; a1 += e * vb[t * 16 + 1]; a2 += e * vb[t * 16 + 2]; a3 += e * vb[t * 16 + 3];
  }
  float* op = o + ((size_t)ray * dn + s) * 16 + h * 4;
  op[0] = a0 / l; op[1] = a1 / l; op[2] = a2 / l; op[3] = a3 / l;
}
// The same for dn <= 64 and more than one source view with the head's keys and values in LDS: one single-wave workgroup per (ray, head), lane =
// query; the kernel above reads k and v through eight broadcast global loads per key and pass (1 024 vector-memory instructions per wave, 61 us).
__global__ void __launch_bounds__(64) attn_fwd64_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ vv, int rn,
                                                        int dn, float* __restrict__ o) {
  __shared__ float s_k[64][4], s_v[64][4];
  const int ray = blockIdx.x >> 2, h = blockIdx.x & 3, s = threadIdx.x;
  const size_t me = ((size_t)ray * dn + s) * 16 + h * 4;
  const bool ok = s < dn;
  float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, v4 = q4;
  if (ok) { q4 = *reinterpret_cast<const float4*>(q + me); k4 = *reinterpret_cast<const float4*>(k + me); v4 = *reinterpret_cast<const float4*>(vv + me); }
  *reinterpret_cast<float4*>(s_k[s]) = k4;
  *reinterpret_cast<float4*>(s_v[s]) = v4;
  __syncthreads();
  const float q0 = q4.x * 0.5f, q1 = q4.y * 0.5f, q2 = q4.z * 0.5f, q3 = q4.w * 0.5f;
  float mx = -3.0e38f;
  for (int t = 0; t < dn; ++t) {
    const float4 kt = *reinterpret_cast<const float4*>(s_k[t]);
    mx = fmaxf(mx, q0 * kt.x + q1 * kt.y + q2 * kt.z + q3 * kt.w);
  }
  float l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int t = 0; t < dn; ++t) {
    const float4 kt = *reinterpret_cast<const float4*>(s_k[t]), vt = *reinterpret_cast<const float4*>(s_v[t]);
    const float e = expf(q0 * kt.x + q1 * kt.y + q2 * kt.z + q3 * kt.w - mx);
    l += e;
    a0 += e * vt.x; a1 += e * vt.y; a2 += e * vt.z; a3 += e * vt.w;
  }
  if (ok) *reinterpret_cast<float4*>(o + me) = make_float4(a0 / l, a1 / l, a2 / l, a3 / l);
}
// Backward of the attention of ONE (ray, head) per workgroup, thread = sample index (dn <= 128), no atomics:
//   P_st = softmax_t(q_s k_t / 2);  gP_st = <go_s, v_t>;  gS_st = P_st (gP_st - <go_s, o_s>)
//   phase 1 (thread = query s): row maximum / sum, <go_s, o_s>, and gq_s = 0.5 sum_t gS_st k_t
//   phase 2 (thread = key t):   gk_t = 0.5 sum_s gS_st q_s,  gv_t = sum_s P_st go_s   (P_st recomputed from the row statistics)
// gq / gk / gv are ACCUMULATED (+=) into zero-initialised buffers.
__global__ void __launch_bounds__(128) attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ vv,
                                                       const float* __restrict__ o, const float* __restrict__ go, int rn, int dn, int uniform,
                                                       float* __restrict__ gq, float* __restrict__ gk, float* __restrict__ gv) {
  __shared__ float s_q[PGRF_MAX_DN][4], s_k[PGRF_MAX_DN][4], s_v[PGRF_MAX_DN][4], s_go[PGRF_MAX_DN][4];
  __shared__ float s_mx[PGRF_MAX_DN], s_l[PGRF_MAX_DN], s_dot[PGRF_MAX_DN];
  const int ray = blockIdx.x >> 2, h = blockIdx.x & 3, t = threadIdx.x;
  const size_t base = (size_t)ray * dn * 16 + h * 4;
  if (t < dn) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s_q[t][c] = q[base + (size_t)t * 16 + c]; s_k[t][c] = k[base + (size_t)t * 16 + c];
      s_v[t][c] = vv[base + (size_t)t * 16 + c]; s_go[t][c] = go[base + (size_t)t * 16 + c];
    }
  }
  __syncthreads();
  if (t < dn) {          // phase 1: query t
    const float q0 = 0.5f * s_q[t][0], q1 = 0.5f * s_q[t][1], q2 = 0.5f * s_q[t][2], q3 = 0.5f * s_q[t][3];
    float mx = -3.0e38f, l = 0.f;
    if (!uniform) {
      for (int u = 0; u < dn; ++u) mx = fmaxf(mx, q0 * s_k[u][0] + q1 * s_k[u][1] + q2 * s_k[u][2] + q3 * s_k[u][3]);
      for (int u = 0; u < dn; ++u) l += expf(q0 * s_k[u][0] + q1 * s_k[u][1] + q2 * s_k[u][2] + q3 * s_k[u][3] - mx);
    } else {
      mx = 0.f; l = (float)dn;
    }
    const size_t me = base + (size_t)t * 16;
    const float dot = s_go[t][0] * o[me] + s_go[t][1] * o[me + 1] + s_go[t][2] * o[me + 2] + s_go[t][3] * o[me + 3];
    s_mx[t] = mx; s_l[t] = l; s_dot[t] = dot;
    if (!uniform) {      // masked logits are constants: no gradient reaches q or k
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int u = 0; u < dn; ++u) {
        const float p = expf(q0 * s_k[u][0] + q1 * s_k[u][1] + q2 * s_k[u][2] + q3 * s_k[u][3] - mx) / l;
        const float gp = s_go[t][0] * s_v[u][0] + s_go[t][1] * s_v[u][1] + s_go[t][2] * s_v[u][2] + s_go[t][3] * s_v[u][3];
        const float gs = 0.5f * p * (gp - dot);
        a0 += gs * s_k[u][0]; a1 += gs * s_k[u][1]; a2 += gs * s_k[u][2]; a3 += gs * s_k[u][3];
      }
      gq[me] += a0; gq[me + 1] += a1; gq[me + 2] += a2; gq[me + 3] += a3;
    }
  }
  __syncthreads();
  if (t < dn) {          // phase 2: key t
    float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    for (int s = 0; s < dn; ++s) {
      const float p = uniform ? 1.f / s_l[s]
                              : expf(0.5f * (s_q[s][0] * s_k[t][0] + s_q[s][1] * s_k[t][1] + s_q[s][2] * s_k[t][2] + s_q[s][3] * s_k[t][3]) - s_mx[s]) / s_l[s];
      v0 += p * s_go[s][0]; v1 += p * s_go[s][1]; v2 += p * s_go[s][2]; v3 += p * s_go[s][3];
      if (!uniform) {
        const float gp = s_go[s][0] * s_v[t][0] + s_go[s][1] * s_v[t][1] + s_go[s][2] * s_v[t][2] + s_go[s][3] * s_v[t][3];
        const float gs = 0.5f * p * (gp - s_dot[s]);
        k0 += gs * s_q[s][0]; k1 += gs * s_q[s][1]; k2 += gs * s_q[s][2]; k3 += gs * s_q[s][3];
      }
    }
    const size_t me = base + (size_t)t * 16;
    gv[me] += v0; gv[me + 1] += v1; gv[me + 2] += v2; gv[me + 3] += v3;
    gk[me] += k0; gk[me + 1] += k1; gk[me + 2] += k2; gk[me + 3] += k3;
  }
}

// The same for dn <= 64 and more than one source view, one single-wave workgroup per (ray, head), lane = sample (21 KB of LDS: seven of them per CU): the kernel above
// evaluates exp(q_s k_t / 2 - max_s) four times per (s, t) (row sum, gq, and twice in the key phase) and was VALU bound (163 us for
// 2 048 rays); here the 64 x 64 probabilities of the head are computed once into LDS (row-major, lane = key in the second phase:
// conflict free).
__global__ void __launch_bounds__(64) attn_bwd64_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ vv,
                                                         const float* __restrict__ o, const float* __restrict__ go, int rn, int dn,
                                                         float* __restrict__ gq, float* __restrict__ gk, float* __restrict__ gv) {
  __shared__ float s_p[64][65];
  __shared__ float s_k[64][4], s_v[64][4], s_q[64][4], s_go[64][4], s_dot[64];
  const int ray = blockIdx.x >> 2, h = blockIdx.x & 3, t = threadIdx.x;
  const size_t me = ((size_t)ray * dn + t) * 16 + h * 4;
  const bool ok = t < dn;
  float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, v4 = q4, g4 = q4, o4 = q4;
  if (ok) {
    q4 = *reinterpret_cast<const float4*>(q + me); k4 = *reinterpret_cast<const float4*>(k + me); v4 = *reinterpret_cast<const float4*>(vv + me);
    g4 = *reinterpret_cast<const float4*>(go + me); o4 = *reinterpret_cast<const float4*>(o + me);
  }
  *reinterpret_cast<float4*>(s_q[t]) = q4; *reinterpret_cast<float4*>(s_k[t]) = k4;
  *reinterpret_cast<float4*>(s_v[t]) = v4; *reinterpret_cast<float4*>(s_go[t]) = g4;
  const float dot = g4.x * o4.x + g4.y * o4.y + g4.z * o4.z + g4.w * o4.w;
  s_dot[t] = dot;
  __syncthreads();
  // phase 1, lane = query s: logits -> probabilities (stored), gq
  {
    const float q0 = 0.5f * q4.x, q1 = 0.5f * q4.y, q2 = 0.5f * q4.z, q3 = 0.5f * q4.w;
    float mx = -3.0e38f;
    for (int u = 0; u < dn; ++u) {
      const float4 ku = *reinterpret_cast<const float4*>(s_k[u]);
      const float lg = q0 * ku.x + q1 * ku.y + q2 * ku.z + q3 * ku.w;
      s_p[t][u] = lg;
      mx = fmaxf(mx, lg);
    }
    float l = 0.f;
    for (int u = 0; u < dn; ++u) { const float e = expf(s_p[t][u] - mx); s_p[t][u] = e; l += e; }
    const float rl = 1.f / l;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int u = 0; u < dn; ++u) {
      const float p = s_p[t][u] * rl;
      s_p[t][u] = p;
      const float4 ku = *reinterpret_cast<const float4*>(s_k[u]), vu = *reinterpret_cast<const float4*>(s_v[u]);
      const float gp = g4.x * vu.x + g4.y * vu.y + g4.z * vu.z + g4.w * vu.w;
      const float gs = 0.5f * p * (gp - dot);
      a0 += gs * ku.x; a1 += gs * ku.y; a2 += gs * ku.z; a3 += gs * ku.w;
    }
    if (ok) *reinterpret_cast<float4*>(gq + me) = make_float4(a0, a1, a2, a3);      // (written, not accumulated: every (sample, head) has one owner)
  }
  __syncthreads();
  // phase 2, lane = key t: gk, gv from the stored column p[s][t]
  {
    float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    for (int s = 0; s < dn; ++s) {
      const float p = s_p[s][t];
      const float4 gs4 = *reinterpret_cast<const float4*>(s_go[s]), qs = *reinterpret_cast<const float4*>(s_q[s]);
      v0 += p * gs4.x; v1 += p * gs4.y; v2 += p * gs4.z; v3 += p * gs4.w;
      const float gp = gs4.x * v4.x + gs4.y * v4.y + gs4.z * v4.z + gs4.w * v4.w;
      const float gs = 0.5f * p * (gp - s_dot[s]);
      k0 += gs * qs.x; k1 += gs * qs.y; k2 += gs * qs.z; k3 += gs * qs.w;
    }
    if (ok) {
      *reinterpret_cast<float4*>(gv + me) = make_float4(v0, v1, v2, v3);
      *reinterpret_cast<float4*>(gk + me) = make_float4(k0, k1, k2, k3);
    }
  }
}

// LayerNorm(16, eps 1e-6) (ibrnet.py:101): y = (x - mean) * rstd * gamma + beta
__global__ void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, long P,
                                 float* __restrict__ y) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float m = 0.f;
  for (int c = 0; c < 16; ++c) m += x[p * 16 + c];
  m *= 0.0625f;
  float sq = 0.f;
  for (int c = 0; c < 16; ++c) { const float d = x[p * 16 + c] - m; sq += d * d; }
  const float rstd = rsqrtf(sq * 0.0625f + 1e-6f);
  for (int c = 0; c < 16; ++c) y[p * 16 + c] = (x[p * 16 + c] - m) * rstd * gamma[c] + beta[c];
}
// (the first version added every element's term to 32 LDS words with atomics: 8 192 same-address LDS atomics per workgroup, 71 us
//  for 8 MB of data; now each thread keeps the 32 column sums of ITS points in registers and the workgroup reduces once)
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ gy,
                                                            long P, float* __restrict__ gx, float* __restrict__ ggamma, float* __restrict__ gbeta) {
  __shared__ float s_red[4][32];
  float sg[16], sb[16], gm[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { sg[c] = 0.f; sb[c] = 0.f; gm[c] = gamma[c]; }
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    float xv[16], gv[16];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float4 a4 = *reinterpret_cast<const float4*>(x + p * 16 + 4 * c4), g4 = *reinterpret_cast<const float4*>(gy + p * 16 + 4 * c4);
      xv[4 * c4] = a4.x; xv[4 * c4 + 1] = a4.y; xv[4 * c4 + 2] = a4.z; xv[4 * c4 + 3] = a4.w;
      gv[4 * c4] = g4.x; gv[4 * c4 + 1] = g4.y; gv[4 * c4 + 2] = g4.z; gv[4 * c4 + 3] = g4.w;
    }
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) m += xv[c];
    m *= 0.0625f;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { const float d = xv[c] - m; sq += d * d; }
    const float rstd = rsqrtf(sq * 0.0625f + 1e-6f);
    float a = 0.f, b = 0.f;            // mean of g*gamma, mean of g*gamma*xhat
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float xh = (xv[c] - m) * rstd, gg = gv[c] * gm[c];
      a += gg; b += gg * xh;
      sg[c] += gv[c] * xh;
      sb[c] += gv[c];
      xv[c] = xh;
    }
    a *= 0.0625f; b *= 0.0625f;
    float o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) o[c] = rstd * (gv[c] * gm[c] - a - xv[c] * b);
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) *reinterpret_cast<float4*>(gx + p * 16 + 4 * c4) = make_float4(o[4 * c4], o[4 * c4 + 1], o[4 * c4 + 2], o[4 * c4 + 3]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { sg[c] += __shfl_xor(sg[c], off, 64); sb[c] += __shfl_xor(sb[c], off, 64); }
    if (lane == 0) { s_red[wave][c] = sg[c]; s_red[wave][16 + c] = sb[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const float t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    atomicAdd(threadIdx.x < 16 ? &ggamma[threadIdx.x] : &gbeta[threadIdx.x - 16], t);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// colour blending over the views (ibrnet.py:364-370): blend = softmax_v(logit), colors = sum_v rgb_v * blend_v
__global__ void blend_kernel(const float* __restrict__ logit, const float* __restrict__ rgb, long P, int V, float* __restrict__ blend,
                             float* __restrict__ colors) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float mx = logit[p * V];
  for (int v = 1; v < V; ++v) mx = fmaxf(mx, logit[p * V + v]);
  float es = 0.f;
  for (int v = 0; v < V; ++v) es += expf(logit[p * V + v] - mx);
  float c[3] = {0.f, 0.f, 0.f};
  for (int v = 0; v < V; ++v) {
    const float b = expf(logit[p * V + v] - mx) / es;
    blend[p * V + v] = b;
    c[0] += rgb[(p * V + v) * 3] * b; c[1] += rgb[(p * V + v) * 3 + 1] * b; c[2] += rgb[(p * V + v) * 3 + 2] * b;
  }
  colors[p * 3] = c[0]; colors[p * 3 + 1] = c[1]; colors[p * 3 + 2] = c[2];
}
__global__ void blend_bwd_kernel(const float* __restrict__ blend, const float* __restrict__ rgb, const float* __restrict__ gcol, long P, int V,
                                 float* __restrict__ glogit) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float dot = 0.f;
  for (int v = 0; v < V; ++v) {
    const float gb = gcol[p * 3] * rgb[(p * V + v) * 3] + gcol[p * 3 + 1] * rgb[(p * V + v) * 3 + 1] + gcol[p * 3 + 2] * rgb[(p * V + v) * 3 + 2];
    dot += blend[p * V + v] * gb;
  }
  for (int v = 0; v < V; ++v) {
    const float gb = gcol[p * 3] * rgb[(p * V + v) * 3] + gcol[p * 3 + 1] * rgb[(p * V + v) * 3 + 1] + gcol[p * 3 + 2] * rgb[(p * V + v) * 3 + 2];
    glogit[p * V + v] = blend[p * V + v] * (gb - dot);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// alpha compositing along a ray (renderer.py:210-219,291-304; render_ops.py:145-153), one thread per ray:
//   alpha = 1 - exp(-relu(sigma)), T_i = prod_{j<i} (1 - alpha_j + 1e-10), hit_i = alpha_i T_i
// backward: with G_i = d/d hit_i (own gradient + through the pixel colour and the depth) and B_i = sum_{k>i} G_k hit_k:
//   d/d alpha_i = T_i G_i - B_i / (1 - alpha_i + 1e-10),  d/d sigma_i = (1 - alpha_i) d/d alpha_i,  d/d colour_i = hit_i g_pixel
// One wave per ray, like the backward below (lane l owns samples [E l, E l + E); the transmittance by a log-step prefix product, the
// pixel / depth sums by a wave reduction): the thread-per-ray loop was 29 us on the path between the coarse pass and the sampler.
__global__ void __launch_bounds__(256) composite_kernel(const float* __restrict__ sigma, const float* __restrict__ colors, const float* __restrict__ depth, int rn,
                                 int dn, float* __restrict__ hit, float* __restrict__ pixel, float* __restrict__ rdepth) {
  const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= rn) return;
  constexpr int EMAX = (PGRF_MAX_DN + 63) / 64;
  const int E = (dn + 63) >> 6;
  float alpha[EMAX], f[EMAX], fl = 1.f;
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    const int i = E * lane + e;
    const bool ok = e < E && i < dn;
    alpha[e] = ok ? 1.f - expf(-fmaxf(sigma[(size_t)ray * dn + i], 0.f)) : 0.f;
    f[e] = ok ? 1.f - alpha[e] + 1e-10f : 1.f;
    fl *= f[e];
  }
  float incl = fl;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const float t = __shfl_up(incl, off, 64); if (lane >= off) incl *= t; }
  float T = __shfl_up(incl, 1, 64);
  if (lane == 0) T = 1.f;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, dd = 0.f;
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    const int i = E * lane + e;
    if (e < E && i < dn) {
      const size_t p = (size_t)ray * dn + i;
      const float h = alpha[e] * T;
      if (hit) hit[p] = h;
      c0 += h * colors[p * 3]; c1 += h * colors[p * 3 + 1]; c2 += h * colors[p * 3 + 2];
      dd += h * depth[p];
    }
    T *= f[e];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    c0 += __shfl_xor(c0, off, 64); c1 += __shfl_xor(c1, off, 64); c2 += __shfl_xor(c2, off, 64); dd += __shfl_xor(dd, off, 64);
  }
  if (lane == 0) {
    if (pixel) { pixel[ray * 3] = c0; pixel[ray * 3 + 1] = c1; pixel[ray * 3 + 2] = c2; }
    if (rdepth) rdepth[ray] = dd;
  }
}
// One WAVE per ray (the first version ran a thread per ray: 32 waves on the whole chip walking 64 samples each with a scratch array,
// 96 us for 1.5 MB of data).  Lane l owns samples [E l, E l + E), E = ceil(dn / 64) <= 2; T (exclusive prefix product of the
// transmittance factors) and B (exclusive suffix sum of G h) by log-step shuffles.
__global__ void __launch_bounds__(256) composite_bwd_kernel(const float* __restrict__ sigma, const float* __restrict__ colors, const float* __restrict__ depth, int rn,
                                     int dn, const float* __restrict__ g_pixel, const float* __restrict__ g_depth, const float* __restrict__ g_hit,
                                     const float* __restrict__ g_colors, const float* __restrict__ g_density, float* __restrict__ gsigma,
                                     float* __restrict__ gcol) {
  const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= rn) return;
  constexpr int EMAX = (PGRF_MAX_DN + 63) / 64;
  const int E = (dn + 63) >> 6;
  const float gp0 = g_pixel ? g_pixel[ray * 3] : 0.f, gp1 = g_pixel ? g_pixel[ray * 3 + 1] : 0.f, gp2 = g_pixel ? g_pixel[ray * 3 + 2] : 0.f;
  const float gd = g_depth ? g_depth[ray] : 0.f;
  float alpha[EMAX], f[EMAX], G[EMAX], sg[EMAX];
  float fl = 1.f;                                        // product of this lane's factors
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    const int i = E * lane + e;
    const bool ok = e < E && i < dn;
    const size_t p = (size_t)ray * dn + (ok ? i : 0);
    sg[e] = ok ? sigma[p] : 0.f;
    alpha[e] = 1.f - expf(-fmaxf(sg[e], 0.f));
    f[e] = ok ? 1.f - alpha[e] + 1e-10f : 1.f;
    G[e] = ok ? (g_hit ? g_hit[p] : 0.f) + gp0 * colors[p * 3] + gp1 * colors[p * 3 + 1] + gp2 * colors[p * 3 + 2] + gd * depth[p] : 0.f;
    fl *= f[e];
  }
  float incl = fl;                                       // inclusive prefix product over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const float t = __shfl_up(incl, off, 64); if (lane >= off) incl *= t; }
  float T = __shfl_up(incl, 1, 64);
  if (lane == 0) T = 1.f;
  float Tt[EMAX], h[EMAX], gl = 0.f;
#pragma unroll
  for (int e = 0; e < EMAX; ++e) { Tt[e] = T; h[e] = alpha[e] * T; gl += G[e] * h[e]; T *= f[e]; }
  float sincl = gl;                                      // inclusive suffix sum over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const float t = __shfl_down(sincl, off, 64); if (lane + off < 64) sincl += t; }
  float B = __shfl_down(sincl, 1, 64);
  if (lane == 63) B = 0.f;
#pragma unroll
  for (int e = EMAX - 1; e >= 0; --e) {
    const int i = E * lane + e;
    if (e < E && i < dn) {
      const size_t p = (size_t)ray * dn + i;
      const float galpha = Tt[e] * G[e] - B / f[e];
      gsigma[p] = (sg[e] > 0.f ? (1.f - alpha[e]) * galpha : 0.f) + (g_density ? g_density[p] : 0.f);
      gcol[p * 3] = h[e] * gp0 + (g_colors ? g_colors[p * 3] : 0.f);
      gcol[p * 3 + 1] = h[e] * gp1 + (g_colors ? g_colors[p * 3 + 1] : 0.f);
      gcol[p * 3 + 2] = h[e] * gp2 + (g_colors ? g_colors[p * 3 + 2] : 0.f);
    }
    B += G[e] * h[e];
  }
}

}  // namespace train
}  // namespace pgrf
