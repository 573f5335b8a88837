// 3-D cost regularisation primitives for gfx950:
//   * 3x3x3 convolution with the ERP "wrap" padding (zeros along D and H, circular along W;
//     models/common_blocks.py:458-503) as an implicit GEMM on v_mfma_f32_32x32x2_f32, bias and
//     LeakyReLU fused, stride 1 or 2, channel-sliced input/output so U-Net concatenations cost nothing
//   * AvgPool3d(2), trilinear x2 upsampling (align_corners=False)
// Activations are channel-last [B,D,H,W,Cs] (Cs = channel stride of the buffer) so that the 32 voxels of
// an MFMA tile x 32 input channels are 32 consecutive 128-byte lines: each lane loads 64 contiguous
// bytes (its voxel, its half of the channel chunk) straight into the B operand registers — no LDS
// staging: the 27 taps re-read the same lines from L1/L2.  Output channels sit on the MFMA rows
// (accumulator registers), so a lane writes 4 consecutive channels per 16-byte store.
#include "pgrf_experiments.h"
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "conv3d_common.h"

namespace pgrf {

// SPLIT: the 4 waves of the workgroup share ONE output tile and split the (tap, chunk) reduction 4 ways
// (partials combined through LDS) — for the deep, low-resolution layers whose plain decomposition yields
// fewer waves than the chip has SIMDs.
#ifndef PGRF_CONV_WAVES
#define PGRF_CONV_WAVES 1        // minimum waves per SIMD the register allocation must allow (tuning knob)
#endif
// LEAN: two waves per SIMD (<= 256 registers): ONE operand buffer of half a step (16 of the chunk's 32 channels),
// refilled in place behind the MFMAs that read it, for interior AND boundary waves (the latter with masked operands).
// Needs Cin % 32 == 0 and the 27-tap mode; everything else takes the LEAN = false instantiations.
// SW (with SPLIT): waves sharing one output tile -- 4 (the whole workgroup, one tile) or 2 (two tiles per workgroup, a wave pair each)
template <int MT, int NT, bool SPLIT, bool LEAN = false, int SW = 4>
__global__ void __launch_bounds__(256, LEAN ? 2 : PGRF_CONV_WAVES) conv3d_wrap_kernel(const ConvArgs a) {
  static_assert(SW == 4 || SW == 2, "split ways");
  __shared__ __attribute__((aligned(16))) float red[SPLIT ? (4 / SW) * (SW - 1) * MT * NT * 16 * 64 : 4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31, kk = lane >> 5;
  const long long nvox = (long long)a.B * a.Do * a.Ho * a.Wo;
  const int sgrp = SPLIT ? wave / SW : 0, ssub = SPLIT ? wave % SW : 0;      // tile of the workgroup, share of its reduction
  const long long tile0 = SPLIT ? ((long long)blockIdx.x * (4 / SW) + sgrp) * NT : ((long long)blockIdx.x * 4 + wave) * NT;
  const int cot0 = blockIdx.y * MT;
  if (!SPLIT && tile0 * 32 >= nvox) return;

  // per-tile voxel decode
  int vb[NT], vz[NT], vy[NT], vx[NT];
  bool vok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    long long v = (tile0 + t) * 32 + j;
    vok[t] = v < nvox;
    if (!vok[t]) v = nvox - 1;
    vx[t] = (int)(v % a.Wo); v /= a.Wo;
    vy[t] = (int)(v % a.Ho); v /= a.Ho;
    vz[t] = (int)(v % a.Do); vb[t] = (int)(v / a.Do);
  }
  // input-side coordinates and the linear index of the centre tap, once per tile: a tap then costs one wave-uniform
  // offset, the W wrap and two unsigned range checks (the per-tap index arithmetic was 1.7 VALU per MFMA)
  int zin[NT], yin[NT], xin[NT], cen[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    zin[t] = vz[t] * a.stride; yin[t] = vy[t] * a.stride; xin[t] = vx[t] * a.stride;
    cen[t] = ((vb[t] * a.D + zin[t]) * a.H + yin[t]) * a.W;          // row start; + x gives the voxel
  }
  f32x16 acc[NT][MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 bv;
    if (a.bias && (!SPLIT || ssub == 0)) {
      const float4* bp = reinterpret_cast<const float4*>(a.bias + (cot0 + m) * 32 + kk * 16);
      const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
      bv = (f32x16){b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[r] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][m] = bv;
  }

  const bool full_chunks = (a.Cin & 31) == 0;
  // Byte offsets (32-bit; the launcher checks the buffer is < 4 GiB) of the three W-neighbours of every tile voxel,
  // circular wrap applied, lane half's channel offset included.  A tap's operand address is then
  //   (a.in + wave-uniform offset of (dz, dy, chunk))  +  voff[t][dx]
  // i.e. a scalar base plus a per-lane offset that does not change over the whole reduction: no per-tap index
  // arithmetic, range checks or selects on the vector ALU (they were 2 VALU instructions per MFMA, and the f32 MFMA
  // shares its pipe with the VALU).
  unsigned voff[NT][3];
  unsigned xbad[NT];          // bit dx: the W neighbour is outside the map and the padding is zeros (wrap_w == 0)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    xbad[t] = 0;
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      int xi = xin[t] + dxi - 1;
      if (!a.wrap_w && (xi < 0 || xi >= a.W)) xbad[t] |= 1u << dxi;
      xi = xi < 0 ? xi + a.W : (xi >= a.W ? xi - a.W : xi);          // circular along W
      voff[t][dxi] = ((unsigned)(cen[t] + xi) * (unsigned)a.in_cs + (unsigned)(a.in_coff + kk * 16)) * 4u;
    }
  }
  // Wave-level classification of the three depth and the three height offsets:
  //   zm/ym  : SOME lane has an in-range source for this offset (otherwise the tap is skipped: whole footprint in the
  //            zero padding; a superset is enough, so the 27 taps are the outer product of 3 + 3 flags);
  //   zam/yam: EVERY lane of every tile has one -> the tap's loads need no mask at all (fast path, > 95 % of the
  //            tiles of a volume); the others (volume faces, ragged last tile) take the masked path.
  unsigned tapmask = 0, zam = 0, yam = 0, zm_ = 0, ym_ = 0;
  {
    bool zany[3] = {false, false, false}, yany[3] = {false, false, false};
    bool zall[3] = {true, true, true}, yall[3] = {true, true, true};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const bool zr = vok[t] && (unsigned)(zin[t] + d - 1) < (unsigned)a.D;
        const bool yr = vok[t] && (unsigned)(yin[t] + d - 1) < (unsigned)a.H;
        zany[d] = zany[d] || zr; yany[d] = yany[d] || yr;
        zall[d] = zall[d] && zr; yall[d] = yall[d] && yr;
      }
    unsigned zm = 0, ym = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (__any(zany[d])) zm |= 1u << d;
      if (__any(yany[d])) ym |= 1u << d;
      if (__all(zall[d])) zam |= 1u << d;
      if (__all(yall[d])) yam |= 1u << d;
    }
    for (int r = 0; r < 9; ++r)
      if (((zm >> (r / 3)) & 1u) && ((ym >> (r % 3)) & 1u)) tapmask |= 1u << r;     // bit r = tap row (dz, dy) = (r/3, r%3)
    zm_ = zm; ym_ = ym;
  }
  if (a.wtaps == 1) tapmask &= 1u << 4;                                 // pointwise: centre tap only
#ifdef PGRF_CONV_NO_FAST   // experiment: every tap through the masked path
  zam = yam = 0;
#endif

  // Operand buffers of one (tap, 32-channel chunk) step.  Two of them ping-pong: the loads of step i+1 are
  // issued before the MFMAs of step i, so the L2/L1 round trip hides under 16*MT*NT MFMAs (the kernel runs at
  // 1-2 waves per SIMD, there is nobody else to hide it).
  struct Step {
    float4 bv[NT][4];
    float4 av[MT][4];
    bool ok[NT];
    bool need_mask;
    int c0;
  };
  // row = tap row index dz*3 + dy; DXI = the W offset + 1 as a compile-time constant (a run-time dx made the compiler
  // index voff[][] through scratch memory)
  auto issue = [&](Step& st, int row, int ch, auto dx_tag) {
    constexpr int dxi = decltype(dx_tag)::value;
#ifdef PGRF_CONV_ABL_NOISSUE   // timing experiment: operands are loaded once, every later step reuses the registers
    if (!(row == __ffs(tapmask) - 1 && ch == 0 && dxi < 2)) { st.need_mask = false; return; }
#endif
    const int dzi = row / 3, dyi = row % 3, tap = row * 3 + dxi;
    st.c0 = ch * 32 + kk * 16;
    // wave-uniform part of the address: the tap's row offset and the channel chunk
    const char* sb = reinterpret_cast<const char*>(a.in) + ((long long)((dzi - 1) * a.H + (dyi - 1)) * a.W * a.in_cs + ch * 32) * 4;
    const bool fast = full_chunks && a.wrap_w && ((zam >> dzi) & (yam >> dyi) & 1u);
    st.need_mask = !fast;
    if (fast) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
#ifdef PGRF_CONV_ABL_B      // timing experiment: every tap reads the centre voxel of chunk 0 (L1 resident)
        for (int q = 0; q < 4; ++q) st.bv[t][q] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.in) + voff[t][1] + q * 16);
#else
        for (int q = 0; q < 4; ++q) st.bv[t][q] = *reinterpret_cast<const float4*>(sb + voff[t][dxi] + q * 16);
#endif
    } else {
      const char* cb = reinterpret_cast<const char*>(a.in) + (long long)ch * 128;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        st.ok[t] = vok[t] && (unsigned)(zin[t] + dzi - 1) < (unsigned)a.D && (unsigned)(yin[t] + dyi - 1) < (unsigned)a.H &&
                   !((xbad[t] >> dxi) & 1u);                                                              // zeros along D, H (and W)
        // always-valid address (the centre voxel when the tap is padding): loads are unconditional and issue back to
        // back; padding, ragged tiles and the channel tail are applied by a select at use
        const char* ip = st.ok[t] ? sb + voff[t][dxi] : cb + voff[t][1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long long qo = (st.c0 + q * 4 + 4 <= a.Cin) ? q * 16 : -(long long)st.c0 * 4;      // tail: channel 0
          st.bv[t][q] = *reinterpret_cast<const float4*>(ip + qo);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#ifdef PGRF_CONV_ABL_A      // timing experiment: every step reads the weights of tap 0, chunk 0
      const float4* ap = reinterpret_cast<const float4*>(a.wpk + wpk_index(cot0 + m, 0, 0, a.chunks, 0, 0, a.wtaps)) + lane;
#else
      const float4* ap = reinterpret_cast<const float4*>(a.wpk + wpk_index(cot0 + m, a.wtaps == 1 ? 0 : tap, ch, a.chunks, 0, 0, a.wtaps)) + lane;
#endif
      st.av[m][0] = ap[0]; st.av[m][1] = ap[64]; st.av[m][2] = ap[128]; st.av[m][3] = ap[192];
    }
  };
  auto apply_mask = [&](Step& st) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = st.ok[t] && (st.c0 + q * 4 + 4 <= a.Cin);
        st.bv[t][q].x = ok ? st.bv[t][q].x : 0.f; st.bv[t][q].y = ok ? st.bv[t][q].y : 0.f;
        st.bv[t][q].z = ok ? st.bv[t][q].z : 0.f; st.bv[t][q].w = ok ? st.bv[t][q].w : 0.f;
      }
  };
  auto mfma_group = [&](const Step& st, int m, int s4) {       // 4 k-steps x NT voxel tiles of one A float4
    const float av[4] = {st.av[m][s4].x, st.av[m][s4].y, st.av[m][s4].z, st.av[m][s4].w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 bq = st.bv[t][s4];
        const float bs = (j == 0) ? bq.x : (j == 1) ? bq.y : (j == 2) ? bq.z : bq.w;
        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bs, acc[t][m], 0, 0, 0);
      }
  };
  auto compute = [&](Step& st) {
    if (st.need_mask) apply_mask(st);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) mfma_group(st, m, s4);
  };
  // One pipeline step: the MFMAs of `cur` while the operands of the next step (row, ch, dx) are requested into `nxt`.
  // With one wave per SIMD nothing else keeps the matrix pipe busy while ~24 loads and their addresses are issued
  // (measured: 13 % of the kernel), so when the next step needs no padding mask its loads are spread between the MFMA
  // groups of this one — a single basic block, pinned with sched_group_barrier (the scheduler otherwise hoists every
  // load in front of the first MFMA).  A masked next step keeps the plain issue-then-compute order.
#ifdef PGRF_CONV_NO_FUSE
  constexpr bool kFuse = false;
#else
  constexpr bool kFuse = true;
#endif
  // In-place pipeline step: the MFMAs of the step held in `st`, and behind each MFMA group the loads that refill exactly
  // the registers that group was the last to read with the operands of the NEXT step (row, ch, dx): the A float4 of
  // group (m, s4) right after it, the B float4s of k-quad s4 after the last cout tile has used them.  One operand
  // buffer instead of two (96 registers fewer than ping-pong), and the ~24 loads of a step are spread between its
  // MFMA groups instead of draining the matrix pipe in front of them (one wave per SIMD: nobody else fills it;
  // measured 13 % of the kernel).  A refilled register is next read >= 4 groups (4096 cycles) later.
  auto step_fast = [&](Step& st, int row, int ch, auto dx_tag) {
    constexpr int dxi = decltype(dx_tag)::value;
    const int dzi = row / 3, dyi = row % 3, tap = row * 3 + dxi;
    const char* sb = reinterpret_cast<const char*>(a.in) + ((long long)((dzi - 1) * a.H + (dyi - 1)) * a.W * a.in_cs + ch * 32) * 4;
    const float4* ap[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
      ap[m] = reinterpret_cast<const float4*>(a.wpk + wpk_index(cot0 + m, tap, ch, a.chunks, 0, 0, 27)) + lane;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        mfma_group(st, m, s4);
#ifndef PGRF_CONV_ABL_NOREFILL_A      // timing experiments: no weight / no activation refills (stale operands)
        st.av[m][s4] = ap[m][s4 * 64];
#endif
#ifndef PGRF_CONV_ABL_NOREFILL_B
        if (m == MT - 1) {
#pragma unroll
          for (int t = 0; t < NT; ++t) st.bv[t][s4] = *reinterpret_cast<const float4*>(sb + voff[t][dxi] + s4 * 16);
        }
#endif
#ifndef PGRF_CONV_SPREAD_REFILL   // default: refills in one batch behind the group (spreading them measured 0.8 % slower)
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);                       // the group's MFMAs ...
#if !defined(PGRF_CONV_ABL_NOREFILL_A) && !defined(PGRF_CONV_ABL_NOREFILL_B)
        if (m == MT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1 + NT, 0);      // ... then the refills
        else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
#else
        if (m == MT - 1) {
          // experiment: tile t's B quad is last read by MFMA 3*NT + t of the group; its refill right behind that MFMA
          __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT + 1, 0);
#pragma unroll
          for (int t = 0; t < NT - 1; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#endif
      }
  };
  auto mfma_all = [&](const Step& st) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) mfma_group(st, m, s4);
  };
  // Wave-uniform iterator over macro steps = (valid tap row, chunk); a macro step is the three W taps dx = -1, 0, +1.
  // Steps alternate between the two operand buffers, so the pattern (buffer, dx) repeats every 6 steps = 2 macro
  // steps: the loop body below is that period, with dx a compile-time constant in every issue.
  // SPLIT: wave w takes the macro steps whose running index is w modulo 4.
  unsigned rem = tapmask;
  int row = rem ? __ffs(rem) - 1 : -1, ch = 0, midx = 0;
  auto next1 = [&]() {
    if (++ch == a.chunks) { ch = 0; rem &= rem - 1; row = rem ? __ffs(rem) - 1 : -1; }
    ++midx;
  };
  auto advance = [&]() {
    next1();
    if (SPLIT) while (row >= 0 && (midx & (SW - 1)) != ssub) next1();       // this wave's share of the macro steps
  };
  if (SPLIT) while (row >= 0 && (midx & (SW - 1)) != ssub) next1();
  typedef std::integral_constant<int, 0> DX0;
  typedef std::integral_constant<int, 1> DX1;
  typedef std::integral_constant<int, 2> DX2;
  // Interior waves (every lane of every tile has all its taps in range; > 90 % of a volume): a loop of fused steps with
  // no padding logic at all.  Everything else (volume faces, ragged last tile, channel tails, pointwise mode): the
  // generic loop with masked loads.
  // (every tap row that is needed at all is in range for every lane: also the depth-1 volumes of the 2-D encoders, whose
  // dz = +-1 rows are skipped as a whole)
  if constexpr (LEAN) {
    struct Half {
      float4 bv[NT][2];
      float4 av[MT][2];
    };
    Half P;
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;
    // masked path: per-tile range bits (bit d: depth offset d-1 in range, bit 3+d: height offset d-1 in range; all zero for
    // the voxels of a ragged last tile) and the validity of what P currently holds (bit t: tile t)
    unsigned tb[NT], pm = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tb[t] = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (vok[t] && (unsigned)(zin[t] + d - 1) < (unsigned)a.D) tb[t] |= 1u << d;
        if (vok[t] && (unsigned)(yin[t] + d - 1) < (unsigned)a.H) tb[t] |= 8u << d;
      }
    }
    constexpr unsigned kAllTiles = (1u << NT) - 1u;
    auto group = [&](int m, int s) {                      // 4 k-steps x NT voxel tiles of one A float4
      const float av[4] = {P.av[m][s].x, P.av[m][s].y, P.av[m][s].z, P.av[m][s].w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 bq = P.bv[t][s];
          const float bs = (jj == 0) ? bq.x : (jj == 1) ? bq.y : (jj == 2) ? bq.z : bq.w;
          acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bs, acc[t][m], 0, 0, 0);
        }
    };
    // One half step: (mask and) multiply what P holds while P is refilled in place with the operands of
    // (row, ch, dx, h): A quad (m, s) right behind its MFMA group, the B quads s behind the last cout tile's group.
    // COMPUTE = false is the prologue (loads only).
    auto half_step = [&](int row, int ch, auto dx_tag, auto h_tag, auto masked_tag, auto compute_tag) {
      constexpr int dxi = decltype(dx_tag)::value, h = decltype(h_tag)::value;
      constexpr bool MASKED = decltype(masked_tag)::value, COMPUTE = decltype(compute_tag)::value;
      const int dzi = row / 3, dyi = row % 3, tap = row * 3 + dxi;
      const long long rowoff = (long long)((dzi - 1) * a.H + (dyi - 1)) * a.W * a.in_cs * 4;
      const char* cb = reinterpret_cast<const char*>(a.in) + ch * 128 + h * 32;       // wave-uniform
      const char* sb = cb + rowoff;
      unsigned off[NT], npm = 0;
      if constexpr (MASKED) {
        if (COMPUTE && !__all(pm == kAllTiles)) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const bool ok = (pm >> t) & 1u;
              P.bv[t][q].x = ok ? P.bv[t][q].x : 0.f; P.bv[t][q].y = ok ? P.bv[t][q].y : 0.f;
              P.bv[t][q].z = ok ? P.bv[t][q].z : 0.f; P.bv[t][q].w = ok ? P.bv[t][q].w : 0.f;
            }
        }
        // always-valid addresses: the tap's voxel, or the centre voxel where the tap is padding
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bool ok = ((tb[t] >> dzi) & (tb[t] >> (3 + dyi)) & 1u) && !((xbad[t] >> dxi) & 1u);
          off[t] = ok ? voff[t][dxi] + (unsigned)rowoff : voff[t][1];
          npm |= (ok ? 1u : 0u) << t;
        }
      }
      const float4* ap[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        ap[m] = reinterpret_cast<const float4*>(a.wpk + wpk_index(cot0 + m, tap, ch, a.chunks, 0, 0, 27)) + lane + h * 128;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if constexpr (COMPUTE) group(m, q);
#ifdef PGRF_CONV_ABL_NOREFILL_A
          if (!COMPUTE)
#endif
          P.av[m][q] = ap[m][q * 64];
#ifdef PGRF_CONV_ABL_NOREFILL_B
          if (!COMPUTE)
#endif
          if (m == MT - 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              if constexpr (MASKED) P.bv[t][q] = *reinterpret_cast<const float4*>(cb + off[t] + q * 16);
              else P.bv[t][q] = *reinterpret_cast<const float4*>(sb + voff[t][dxi] + q * 16);
            }
          }
#if !defined(PGRF_CONV_ABL_NOREFILL_A) && !defined(PGRF_CONV_ABL_NOREFILL_B)
          if constexpr (COMPUTE) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);                       // the group's MFMAs ...
            if (m == MT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1 + NT, 0);      // ... then the refills
            else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
#endif
        }
      if constexpr (MASKED) pm = npm;
    };
    auto run = [&](auto masked_tag) {
      typedef std::true_type Y;
      half_step(row, ch, DX0(), H0(), masked_tag, std::false_type());
      bool more;
      do {
        half_step(row, ch, DX0(), H1(), masked_tag, Y());
        half_step(row, ch, DX1(), H0(), masked_tag, Y());
        half_step(row, ch, DX1(), H1(), masked_tag, Y());
        half_step(row, ch, DX2(), H0(), masked_tag, Y());
        half_step(row, ch, DX2(), H1(), masked_tag, Y());
        const int prow = row, pch = ch;
        advance();
        more = row >= 0;
        half_step(more ? row : prow, more ? ch : pch, DX0(), H0(), masked_tag, Y());     // (last: a reload nobody reads)
      } while (more);
    };
    const bool interior = a.wrap_w && zam == zm_ && yam == ym_;
    if (row >= 0) {
      if (interior) run(std::false_type());
      else run(std::true_type());
    }
  } else {
  const bool interior = kFuse && full_chunks && a.wrap_w && zam == zm_ && yam == ym_ && a.wtaps == 27 && tapmask != 0u;
  if (interior && row >= 0) {              // (row < 0: a split-K wave that got no macro step)
    Step P;
    P.need_mask = false;
    issue(P, row, ch, DX0());                  // fast path of issue(): plain loads
    // bottom-tested loop of whole macro steps (a mid-loop exit made the compiler rotate the loop and copy half of the
    // accumulator AGPRs twice per iteration); the last iteration's third step reloads its own row instead of a next one
    bool more;
    do {
      step_fast(P, row, ch, DX1());
      step_fast(P, row, ch, DX2());
      const int prow = row, pch = ch;
      advance();
      more = row >= 0;
      step_fast(P, more ? row : prow, more ? ch : pch, DX0());
    } while (more);
  } else if (row >= 0 && a.wtaps == 1) {          // pointwise product: the centre tap of every chunk
    Step P, Q;
    issue(P, row, ch, DX1()); advance();
    while (true) {
      const bool hasQ = row >= 0;
      if (hasQ) { issue(Q, row, ch, DX1()); advance(); }
      compute(P);
      if (!hasQ) break;
      const bool hasP = row >= 0;
      if (hasP) { issue(P, row, ch, DX1()); advance(); }
      compute(Q);
      if (!hasP) break;
    }
  } else if (row >= 0) {
    Step P, Q;
    issue(P, row, ch, DX0());
    while (true) {
      issue(Q, row, ch, DX1()); compute(P);
      issue(P, row, ch, DX2()); compute(Q);
      advance();
      const bool more1 = row >= 0;
      if (more1) issue(Q, row, ch, DX0());
      compute(P);
      if (!more1) break;
      issue(P, row, ch, DX1()); compute(Q);
      issue(Q, row, ch, DX2()); compute(P);
      advance();
      const bool more2 = row >= 0;
      if (more2) issue(P, row, ch, DX0());
      compute(Q);
      if (!more2) break;
    }
  }
  }
  if (SPLIT) {
    float* rg = red + sgrp * (SW - 1) * MT * NT * 16 * 64;           // this tile's partial buffers
    if (ssub > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(&rg[((((ssub - 1) * NT + t) * MT + m) * 4 + q) * 256 + lane * 4]) =
                make_float4(acc[t][m][q * 4], acc[t][m][q * 4 + 1], acc[t][m][q * 4 + 2], acc[t][m][q * 4 + 3]);
    }
    __syncthreads();
    if (ssub > 0) return;
#pragma unroll
    for (int w = 0; w < SW - 1; ++w)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(&rg[(((w * NT + t) * MT + m) * 4 + q) * 256 + lane * 4]);
            acc[t][m][q * 4] += v.x; acc[t][m][q * 4 + 1] += v.y; acc[t][m][q * 4 + 2] += v.z; acc[t][m][q * 4 + 3] += v.w;
          }
  }
  if (a.tapmajor) {      // [27][voxels]: 32 consecutive voxels of one row per store instruction and lane half
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!vok[t]) continue;
      const long long ov = (((long long)vb[t] * a.Do + vz[t]) * a.Ho + vy[t]) * a.Wo + vx[t];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rho_(r, kk);
        if (row < 27) a.out[(long long)row * nvox + ov] = acc[t][0][r];
      }
    }
    return;
  }
  // epilogue: LeakyReLU, optional residual, 16-byte channel stores
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (!vok[t]) continue;
    const long long ov = (((long long)vb[t] * a.Do + vz[t]) * a.Ho + vy[t]) * a.Wo + vx[t];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = (cot0 + m) * 32 + q * 8 + kk * 4;
        if (c + 4 > a.Cout) continue;
        float4 v = make_float4(acc[t][m][q * 4], acc[t][m][q * 4 + 1], acc[t][m][q * 4 + 2], acc[t][m][q * 4 + 3]);
        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.residual) rr = *reinterpret_cast<const float4*>(a.residual + ov * a.res_cs + a.res_coff + c);
        if (a.res_first) { v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
        v.x = conv_act(a, v.x); v.y = conv_act(a, v.y); v.z = conv_act(a, v.z); v.w = conv_act(a, v.w);
        if (!a.res_first) { v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
        *reinterpret_cast<float4*>(a.out + ov * a.out_cs + a.out_coff + c) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Narrow layers (CostRegNet: 8/16/32 channels).  Two kinds of padding waste in the general kernel above:
//  * K side: a (tap, chunk) step is 32 channels deep, so Cin = 16 (8) fills half (a quarter) of it.  Here the
//    reduction index is k = tap * CIN + c, cut into 32-deep steps: a step holds 2 (4) taps when CIN = 16 (8).
//  * M side: Cout = 8 (16) fills 8 (16) of the tile's 32 rows.  Here the rows carry NS = 4 (2) copies of the output
//    channels for NS consecutive output depths z = NS*zq + delta: row = delta * (32/NS) + cout.  The input taps then
//    span NS + 2 depth planes and the weight of row (delta, c) at depth offset dz' is W[c][dz' - delta] (zero outside
//    the 3-tap support — a block-Toeplitz operand): (NS+2)/3 times the steps for NS times the outputs.
// NS > 1 needs stride 1 and Do % NS == 0.  One 32-row tile (MT = 1), NT voxel tiles per wave, ping-pong prefetch.
struct SmallArgs {
  ConvArgs c;
  int ns;        // depth shifts packed on the rows
  int nsteps;    // ceil((ns + 2) * 9 * Cin / 32)
  unsigned in_bytes;          // conv3d_brick16_kernel: extent of the input behind in + in_coff (buffer descriptor)
  int nbx;                    //   bricks along x (blockIdx.x = output-channel group * nbx + brick)
  int nch;                    //   16-channel chunks of the input (instances with CIN = 0)
  int grp_floats;             //   packed floats per group of 16 output channels
};

template <int CIN, int NS, int NT>
__global__ void __launch_bounds__(256) conv3d_small_kernel(const SmallArgs sa) {
  const ConvArgs& a = sa.c;
  constexpr int NTAP = (NS + 2) * 9;
  constexpr int TPH = CIN >= 16 ? 1 : 16 / CIN;        // taps per lane half per step
  constexpr int QPT = CIN >= 16 ? 4 : CIN / 4;         // float4 loads per tap and lane
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31, kk = lane >> 5;
  const int Dq = a.Do / NS;
  const long long nbase = (long long)a.B * Dq * a.Ho * a.Wo;
  const long long tile0 = ((long long)blockIdx.x * 4 + wave) * NT;
  if (tile0 * 32 >= nbase) return;
  int vb[NT], vz[NT], vy[NT], vx[NT];
  bool vok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    long long v = (tile0 + t) * 32 + j;
    vok[t] = v < nbase;
    if (!vok[t]) v = nbase - 1;
    vx[t] = (int)(v % a.Wo); v /= a.Wo;
    vy[t] = (int)(v % a.Ho); v /= a.Ho;
    vz[t] = (int)(v % Dq) * NS; vb[t] = (int)(v / Dq);      // first output depth of the group
  }
  int zin[NT], yin[NT], xin[NT], cen[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    zin[t] = vz[t] * a.stride; yin[t] = vy[t] * a.stride; xin[t] = vx[t] * a.stride;
    cen[t] = ((vb[t] * a.D + zin[t]) * a.H + yin[t]) * a.W;
  }
  f32x16 acc[NT];
  {
    f32x16 bv;
    if (a.bias) {
      const float4* bp = reinterpret_cast<const float4*>(a.bias + kk * 16);
      const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
      bv = (f32x16){b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[r] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = bv;
  }
  struct Step {
    float4 bv[NT][4];
    float4 av[4];
    bool ok[NT][TPH];
    bool need_mask;
  };
  auto issue = [&](Step& st, int step) {
    bool all = true;
#pragma unroll
    for (int u = 0; u < TPH; ++u) {
      // CIN 32: step = tap, the half takes channels kk*16..; CIN 16: tap = 2*step + kk; CIN 8: tap = 4*step + 2*kk + u
      const int tap = CIN == 32 ? step : (CIN == 16 ? 2 * step + kk : 4 * step + 2 * kk + u);
      const bool tap_ok = tap < NTAP;
      const int tp = tap_ok ? tap : 0;
      const int dz = tp / 9 - 1, dy = (tp / 3) % 3 - 1, dx = tp % 3 - 1;
      const int c0 = CIN == 32 ? kk * 16 : 0;
      const int toff = (dz * a.H + dy) * a.W;        // per lane half when taps are packed along K
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        int xi = xin[t] + dx;
        xi = xi < 0 ? xi + a.W : (xi >= a.W ? xi - a.W : xi);
        st.ok[t][u] = vok[t] && tap_ok && (unsigned)(zin[t] + dz) < (unsigned)a.D && (unsigned)(yin[t] + dy) < (unsigned)a.H;
        all = all && st.ok[t][u];
        const int lin = st.ok[t][u] ? cen[t] + toff + xi : cen[t] + xin[t];
        const float* ip = a.in + (long long)lin * a.in_cs + a.in_coff + c0;
#pragma unroll
        for (int q = 0; q < QPT; ++q) st.bv[t][u * QPT + q] = *reinterpret_cast<const float4*>(ip + q * 4);
      }
    }
    st.need_mask = !__all(all);
    const float4* ap = reinterpret_cast<const float4*>(a.wpk) + (size_t)step * 256 + lane;
    st.av[0] = ap[0]; st.av[1] = ap[64]; st.av[2] = ap[128]; st.av[3] = ap[192];
  };
  auto compute = [&](Step& st) {
    if (st.need_mask)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < TPH; ++u)
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
          float4& v = st.bv[t][u * QPT + q];
          const bool ok = st.ok[t][u];
          v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float av[4] = {st.av[s4].x, st.av[s4].y, st.av[s4].z, st.av[s4].w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 bq = st.bv[t][s4];
          const float bs = (jj == 0) ? bq.x : (jj == 1) ? bq.y : (jj == 2) ? bq.z : bq.w;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bs, acc[t], 0, 0, 0);
        }
    }
  };
  Step P, Q;
  int step = 0;
  issue(P, step++);
  while (true) {
    const bool hasQ = step < sa.nsteps;
    if (hasQ) issue(Q, step++);
    compute(P);
    if (!hasQ) break;
    const bool hasP = step < sa.nsteps;
    if (hasP) issue(P, step++);
    compute(Q);
    if (!hasP) break;
  }
  // epilogue: row = q*8 + kk*4 + i  ->  depth shift = row / (32/NS), channel = row % (32/NS)
  constexpr int RPS = 32 / NS;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (!vok[t]) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = q * 8 + kk * 4;
      const int delta = row / RPS, c = row % RPS;
      if (c + 4 > a.Cout) continue;
      const long long ov = (((long long)vb[t] * a.Do + vz[t] + delta) * a.Ho + vy[t]) * a.Wo + vx[t];
      float4 v = make_float4(acc[t][q * 4], acc[t][q * 4 + 1], acc[t][q * 4 + 2], acc[t][q * 4 + 3]);
      v.x = v.x > 0.f ? v.x : v.x * a.slope; v.y = v.y > 0.f ? v.y : v.y * a.slope;
      v.z = v.z > 0.f ? v.z : v.z * a.slope; v.w = v.w > 0.f ? v.w : v.w * a.slope;
      if (a.residual) {
        const float4 rr = *reinterpret_cast<const float4*>(a.residual + ov * a.res_cs + a.res_coff + c);
        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
      }
      *reinterpret_cast<float4*>(a.out + ov * a.out_cs + a.out_coff + c) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Narrow layers on the 16-row tile (v_mfma_f32_16x16x4_f32: same FLOP rate as the 32x32x2 form, half the rows).  With 32 rows
// conv3d_small_kernel needs NS = 32 / Cout depth shifts to fill the tile and pays (NS + 2) / (3 NS) useful products: 50 % at
// Cout = 8.  Sixteen rows are filled by NS = 2 shifts of 8 channels (4 input planes for 2 outputs: 75 % useful; 288 instead of
// 432 MFMA cycles per output voxel) or by 16 channels with no shift at all (100 %).  Columns = 16 voxels per tile, NT tiles per
// wave share the A operand; the reduction index k = tap * CIN + c is cut into groups of 16 (4 MFMAs of depth 4): lane group
// g = lane / 16 supplies k = 16 G + 4 g + m to MFMA m, i.e. one 16-byte load per lane feeds four MFMAs.  Accumulators are 4
// registers per tile, so eight tiles fit next to a double-buffered operand set at three waves per SIMD.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
// 16-byte load through a buffer descriptor (offsets past num_records read zeros); the clang builtin of this toolchain lowers its
// b128 form to a one-dword load, so the LLVM intrinsic is bound directly
__device__ f32x4v pgrf_buffer_load_x4(i32x4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
template <int CIN, int NS, int NT>
__global__ void __launch_bounds__(256) conv3d_small16_kernel(const SmallArgs sa) {
  const ConvArgs& a = sa.c;
  constexpr int NTAP = (NS + 2) * 9;
  constexpr int NG = ((NTAP * CIN + 15) / 16 + 1) & ~1;    // groups of 16 reduction indices, padded to a pair (zero A rows)
  constexpr int RPS = 16 / NS;                             // rows per shift = channels carried
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Dq = a.Do / NS;
  const unsigned nbase = (unsigned)(a.B * Dq * a.Ho * a.Wo);
  const unsigned tile0 = (unsigned)((blockIdx.x * 4 + wave) * NT);
  if (tile0 * 16u >= nbase) return;
  int zin[NT], yin[NT], xin[NT], cen[NT];
  unsigned vflat[NT];
  bool vok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    unsigned v = (tile0 + t) * 16u + j;
    vok[t] = v < nbase;
    if (!vok[t]) v = nbase - 1u;
    vflat[t] = v;
    const int vx = (int)(v % (unsigned)a.Wo); v /= (unsigned)a.Wo;
    const int vy = (int)(v % (unsigned)a.Ho); v /= (unsigned)a.Ho;
    const int vz = (int)(v % (unsigned)Dq) * NS, vb = (int)(v / (unsigned)Dq);
    zin[t] = vz * a.stride; yin[t] = vy * a.stride; xin[t] = vx * a.stride;
    cen[t] = (((vb * a.D + zin[t]) * a.H + yin[t]) * a.W) * a.in_cs;          // element offset of the row start (x = 0)
  }
  const int zstep = a.H * a.W * a.in_cs, ystep = a.W * a.in_cs;
  const float* inb = a.in + a.in_coff;
  f32x4v acc[NT];
  {
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + g * 4);        // rows 4g..4g+3, plain order: bias[row % RPS]
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4v){b4.x, b4.y, b4.z, b4.w};
  }
  struct Grp {
    float4 bv[NT];
    float4 av;
    unsigned okmask;          // bit t: this lane's operand of tile t lies inside the volume
    bool need_mask;
  };
  auto issue = [&](Grp& gr, int G) {
    // reduction order: CIN >= 16 -> [16-channel chunk][tap][16 channels] (shared with the LDS-brick kernel below), CIN = 8 ->
    // [tap][8 channels]; groups that pad the walk to a pair re-read the last tap against zero weights
    const int k0 = G * 16 + 4 * g;
    const int tap = CIN >= 16 ? min(G, NTAP * (CIN / 16) - 1) % NTAP : min(k0 / CIN, NTAP - 1);
    const int c0 = CIN >= 16 ? 16 * (min(G, NTAP * (CIN / 16) - 1) / NTAP) + 4 * g : k0 % CIN;
    const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
    unsigned okm = 0;
    bool all = true;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int xi = xin[t] + dx;
      xi = xi < 0 ? xi + a.W : (xi >= a.W ? xi - a.W : xi);
      const bool ok = vok[t] && (unsigned)(zin[t] + dz) < (unsigned)a.D && (unsigned)(yin[t] + dy) < (unsigned)a.H;
      okm |= ok ? (1u << t) : 0u;
      all = all && ok;
      const unsigned off = ok ? (unsigned)(cen[t] + __mul24(dz, zstep) + __mul24(dy, ystep) + __mul24(xi, a.in_cs) + c0)
                              : (unsigned)(cen[t] + __mul24(xin[t], a.in_cs) + c0);
      gr.bv[t] = *reinterpret_cast<const float4*>(inb + off);
    }
    gr.okmask = okm;
    gr.need_mask = !__all(all);
    gr.av = reinterpret_cast<const float4*>(a.wpk)[(unsigned)G * 64u + (unsigned)lane];
  };
  auto compute = [&](Grp& gr) {
    if (gr.need_mask)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (!((gr.okmask >> t) & 1u)) gr.bv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float av[4] = {gr.av.x, gr.av.y, gr.av.z, gr.av.w};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 bq = gr.bv[t];
        const float bs = (m == 0) ? bq.x : (m == 1) ? bq.y : (m == 2) ? bq.z : bq.w;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bs, acc[t], 0, 0, 0);
      }
  };
  Grp P, Q;
  issue(P, 0);
  int G = 0;
  do {
    issue(Q, G + 1);
    compute(P);
    issue(P, min(G + 2, NG - 1));
    compute(Q);
    G += 2;
  } while (G < NG);
  // epilogue: rows 4g..4g+3 -> shift delta = row / RPS, channels row % RPS .. +3
  const int row = 4 * g, delta = row / RPS, c = row % RPS;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (!vok[t] || c + 4 > a.Cout) continue;
    unsigned v = vflat[t];
    const int vx = (int)(v % (unsigned)a.Wo); v /= (unsigned)a.Wo;
    const int vy = (int)(v % (unsigned)a.Ho); v /= (unsigned)a.Ho;
    const int vz = (int)(v % (unsigned)Dq) * NS, vb = (int)(v / (unsigned)Dq);
    const unsigned ov = (unsigned)(((vb * a.Do + vz + delta) * a.Ho + vy) * a.Wo + vx);
    float4 o = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    o.x = o.x > 0.f ? o.x : o.x * a.slope; o.y = o.y > 0.f ? o.y : o.y * a.slope;
    o.z = o.z > 0.f ? o.z : o.z * a.slope; o.w = o.w > 0.f ? o.w : o.w * a.slope;
    if (a.residual) {
      const float4 rr = *reinterpret_cast<const float4*>(a.residual + (ov * (unsigned)a.res_cs + (unsigned)(a.res_coff + c)));
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    }
    *reinterpret_cast<float4*>(a.out + (ov * (unsigned)a.out_cs + (unsigned)(a.out_coff + c))) = o;
  }
}

// The 8-channel x 2-shift case with the B operand staged through LDS.  From registers the kernel above is bound by the vector
// memory path, not by the MFMA: a 16-row tile consumes 16 bytes per lane per 128 MFMA cycles, 32 B/clk/CU through the L1 for
// four SIMDs (641 us for conv0 of CostRegNet whatever the tiles-per-wave, against 246 us of MFMA time).  Here a workgroup owns a
// brick of 2 x 4 x 32 output voxels; the 4 x 6 x 34 input voxels it touches (zero padding in z / y and the wrap in x resolved
// while filling) are copied once per 16-channel chunk into LDS and every tap is a ds_read_b128 at a compile-time offset -- no
// masks, no address arithmetic, 3.2x instead of 36x input traffic through the L1.  A voxel is 64 bytes = four 16-byte slots; slot
// s of the voxel in column px is stored at s ^ ((px >> 1) & 3), which puts the 16 lanes of every ds_read_b128 lane group
// ({0-3, 12-15, 20-27}, ...: two channel quads x eight columns) on 16 different slots for every tap shift.  The next chunk travels global -> registers while the current one is multiplied.
//   NS = 2 (Cout = 8): rows = 2 depth shifts x 8 channels, one tile pair (x halves) per wave, 36 plane taps.
//   NS = 1 (Cout = 16): rows = 16 channels; the wave owns both output planes of its row (4 tiles).  Walking the 36 PLANE taps
//   (pz, dy, dx) instead of the 27 kernel taps per output plane, one B operand serves output plane 0 (kernel tap dz = pz - 1) and
//   output plane 1 (dz = pz - 2): the depth-shift reuse without a single structural zero (100 % useful products).
//   NS = 1, YB = 2 (small volumes): bricks of 2 x 2 x 32, so that a launch has twice the workgroups, four of them fit a CU (35 KB of
//   LDS) and a wave's MFMA chain halves: wave = (row, output plane), the 27 kernel taps of its plane, one A operand per tap.
struct Brick16 {
  // A-operand stream of one chunk: which kernel tap the n-th A load fetches, and where step s finds its operands.  `ns` = 1 selects
  // the plane walk (two output planes per wave); everything else is one operand per step.
  static constexpr int a_count(int ns, int s) { return ns != 1 ? 1 : (s / 9 <= 2) + (s / 9 >= 1); }
  static constexpr int a_pos(int ns, int s) {
    if (ns != 1) return s;
    const int pz = s / 9, r = s % 9;
    return pz == 0 ? r : pz == 1 ? 9 + 2 * r : pz == 2 ? 27 + 2 * r : 45 + r;
  }
  static constexpr int a_tap(int ns, int n) {
    if (ns != 1) return n;
    if (n < 9) return n;
    if (n < 27) return (n - 9) % 2 == 0 ? 9 + (n - 9) / 2 : (n - 9) / 2;
    if (n < 45) return (n - 27) % 2 == 0 ? 18 + (n - 27) / 2 : 9 + (n - 27) / 2;
    return 18 + (n - 45);
  }
};
template <int CIN, int NS, int YB = 4>
__global__ void __launch_bounds__(256, YB == 2 ? 4 : 3) conv3d_brick16_kernel(const SmallArgs sa) {
  const ConvArgs& a = sa.c;
  static_assert(YB == 4 || (YB == 2 && NS == 1), "bricks of two rows: the unshifted (16-channel) layers only");
  constexpr bool kPlaneWave = YB == 2;                     // wave = (row, output plane) instead of (row, both planes)
  constexpr int WALK = kPlaneWave ? 0 : NS;                // Brick16 stream selector: 1 = plane walk
  constexpr int NTAP = (NS + 2) * 9, NSTEP = kPlaneWave ? 27 : 36, NSTREAM = NS == 2 ? 36 : (kPlaneWave ? 27 : 54);
  constexpr int NACC = (NS == 2 || kPlaneWave) ? 2 : 4;
  const int NCH = CIN ? CIN / 16 : sa.nch;                // CIN = 0: any multiple of 16, chunk count at run time
  const int NGRP = NS == 2 ? NTAP * NCH : ((NTAP * NCH + 1) & ~1);              // groups in the packed block (padded to a pair)
  constexpr int PX = 34, PY = YB + 2, PZ = 4, VS = 16;     // brick extent in voxels, voxel pitch in floats
  constexpr int NV = PX * PY * PZ, NROW = PY * PZ, HROW = NROW / 2, NPRE = HROW + 1;
  __shared__ __attribute__((aligned(16))) float brick[NV * VS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  // brick of this workgroup: grid = (bricks along x, along y, z pairs x batch).  Workgroups go round-robin over the 8 XCDs in x-fastest
  // order, so with W = 256 (8 bricks along x) an XCD owns one 32-column band of the volume and the y / z halos stay in its L2.
  const int nbz = (a.D + 1) >> 1;
  const int vb = (int)blockIdx.z / nbz, bz = (int)blockIdx.z - vb * nbz;
  const int cg = (int)blockIdx.x / sa.nbx;                 // group of 16 output channels (NS = 1 layers with Cout = 32, 64, ...)
  const int x0 = ((int)blockIdx.x - cg * sa.nbx) * 32, y0 = (int)blockIdx.y * YB, z0 = bz * 2;
  const float* inb = a.in + a.in_coff;
  // fill plan (NPRE = 13 (9) loads and ds_writes per thread and chunk, ~no address arithmetic): the brick is 24 (16) rows (pz, py) of
  // 34 voxels x 4 channel quads.  Columns 0..31: thread -> (column, quad) = tid % 128, the two halves of the workgroup take half of the
  // rows (two planes) each; the row term of the address is wave-uniform (soffset of the buffer load, a compile-time ds offset) and
  // rows in the zero padding are skipped by a uniform branch.  Columns 32, 33 (8 quads per row): one more load per thread, whose
  // lanes in the padding carry an offset past the end of the buffer descriptor and read zeros.
  const unsigned long long inb_bits = (unsigned long long)inb;
  const i32x4v rsrc = {(int)(unsigned)inb_bits, (int)(unsigned)(inb_bits >> 32), (int)sa.in_bytes, 0x00020000};          // raw buffer, 32-bit data
  const int half = __builtin_amdgcn_readfirstlane(wave >> 1);
  auto wrap_x = [&](int x) { x %= a.W; return x < 0 ? x + a.W : x; };
  const int ystep = a.W * a.in_cs, zstep = a.H * ystep;
  const int row0 = ((vb * a.D + z0 - 1 + 2 * half) * a.H + y0 - 1) * ystep;              // element offset of row (pz = 2 half, py = 0), uniform
  unsigned src_col, dst_col, src_x, dst_x;
  {
    const int cq = tid & 3, px = (tid & 127) >> 2;
    src_col = (unsigned)(wrap_x(x0 - 1 + px) * a.in_cs + 4 * cq) * 4u;
    dst_col = (unsigned)(((half * HROW * PX + px) * VS + 4 * (cq ^ ((px >> 1) & 3))) * 4);
    const int row = tid >> 3, pxx = 32 + ((tid >> 2) & 1), pz = row / PY, py = row - pz * PY;
    const int z = z0 - 1 + pz, y = y0 - 1 + py;
    const bool ok = tid < NROW * 8 && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H;
    src_x = ok ? (unsigned)((((vb * a.D + z) * a.H + y) * a.W + wrap_x(x0 - 1 + pxx)) * a.in_cs + 4 * cq) * 4u : 0x80000000u;
    dst_x = (unsigned)(((row * PX + pxx) * VS + 4 * (cq ^ ((pxx >> 1) & 3))) * 4);
  }
  float4 pre[NPRE];
  auto fetch_one = [&](int chunk, int i) {
    if (i < HROW) {
      const int z = z0 - 1 + 2 * half + i / PY, y = y0 - 1 + i % PY;              // uniform
      pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H) {
        const f32x4v r = pgrf_buffer_load_x4(rsrc, (int)src_col, (row0 + (i / PY) * zstep + (i % PY) * ystep + 16 * chunk) * 4, 0);
        pre[i] = make_float4(r.x, r.y, r.z, r.w);
      }
    } else {
      const f32x4v r = pgrf_buffer_load_x4(rsrc, (int)src_x, 64 * chunk, 0);
      pre[i] = make_float4(r.x, r.y, r.z, r.w);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < HROW; ++i) *reinterpret_cast<float4*>(reinterpret_cast<char*>(brick) + dst_col + i * (PX * VS * 4)) = pre[i];
    if (tid < NROW * 8) *reinterpret_cast<float4*>(reinterpret_cast<char*>(brick) + dst_x) = pre[HROW];
  };
  static_assert(NROW * 8 <= 256, "one load per thread covers the two last columns");
#pragma unroll
  for (int i = 0; i < NPRE; ++i) fetch_one(0, i);
  f32x4v acc[NACC];                                       // NS = 2: [x half]; NS = 1: [output plane][x half]
  {
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + cg * 16 + g * 4);
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = (f32x4v){b4.x, b4.y, b4.z, b4.w};
  }
  // wave w multiplies row y0 + w (YB = 2: row y0 + w % 2 of output plane w / 2): tiles x0 + j and x0 + 16 + j.  Per dx one LDS base at
  // the (dz, dy) = (-1, -1) corner, so that every tap is that base + a non-negative immediate: voxel (pz = 0 (or the wave's plane),
  // py = its row, px = j + dx + 1), this lane's (swizzled) channel quad
  const int wrow = kPlaneWave ? (wave & 1) : wave, wzo = kPlaneWave ? (wave >> 1) : 0;
  const float* bl[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) bl[d] = &brick[((wzo * PY + wrow) * PX + j + d) * VS + 4 * (g ^ (((j + d) >> 1) & 3))];
  const unsigned long long w_bits = (unsigned long long)(a.wpk + (size_t)cg * sa.grp_floats);
  const i32x4v wrsrc = {(int)(unsigned)w_bits, (int)(unsigned)(w_bits >> 32), NGRP * 1024, 0x00020000};
  park();
  __syncthreads();
#pragma unroll 1
  for (int chunk = 0; chunk < NCH; ++chunk) {
    // A operands (global, L2-resident: the 36 KB of a chunk cycle through the L1) run AD loads ahead of their MFMAs, B operands (LDS)
    // one step; the next chunk's fill is spread over the steps, one load each, so that the in-order vector-memory queue never puts a
    // burst of slow loads in front of an A operand.  The scheduling barriers pin that order (left alone, the scheduler sinks every
    // load to its use and chains four MFMAs on one accumulator).
    constexpr int AD = 6, RING = 8;
    float4 av[RING], bv[2][2];
    auto issue_a = [&](int n) {                            // n = position in the A stream
      const f32x4v r = pgrf_buffer_load_x4(wrsrc, lane * 16, (chunk * NTAP + Brick16::a_tap(WALK, n)) * 1024, 0);
      av[n % RING] = make_float4(r.x, r.y, r.z, r.w);
    };
    auto issue_a_upto = [&](int from, int to) {
#pragma unroll
      for (int n = from; n < to; ++n)
        if (n < NSTREAM) issue_a(n);
    };
    auto a_limit = [](int s) {                             // A loads issued once step s has issued: everything up to AD - 1 beyond its own
      return s < 0 ? AD - 1 : (s >= NSTEP ? NSTREAM : Brick16::a_pos(WALK, s) + Brick16::a_count(WALK, s) + AD - 1);
    };
    auto issue_b = [&](int slot, int s) {
      const int pz = s / 9, dy = (s / 3) % 3, dx = s % 3;
      const int off = (pz * PY + dy) * PX * VS;
      bv[slot][0] = *reinterpret_cast<const float4*>(bl[dx] + off);
      bv[slot][1] = *reinterpret_cast<const float4*>(bl[dx] + off + 16 * VS);          // 16 columns on: same (px >> 1) & 3
    };
    auto pick = [](const float4& q, int m) { return (m == 0) ? q.x : (m == 1) ? q.y : (m == 2) ? q.z : q.w; };
    auto compute = [&](int slot, int s) {
      const int n0 = Brick16::a_pos(WALK, s);
      if (WALK != 1) {
        const float4 a4 = av[n0 % RING];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(a4, m), pick(bv[slot][t], m), acc[t], 0, 0, 0);
      } else {
        const int pz = s / 9;
        const bool lo = pz <= 2, hi = pz >= 1;             // output plane 0 takes kernel tap dz = pz - 1, plane 1 takes dz = pz - 2
        const float4 a_lo = av[n0 % RING], a_hi = av[(n0 + (lo ? 1 : 0)) % RING];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (lo) {
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(a_lo, m), pick(bv[slot][t], m), acc[t], 0, 0, 0);
          }
          if (hi) {
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(a_hi, m), pick(bv[slot][t], m), acc[2 + t], 0, 0, 0);
          }
        }
      }
    };
    const bool more = chunk + 1 < NCH;
    issue_a_upto(0, a_limit(-1));
    issue_b(0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      issue_a_upto(a_limit(st - 1), a_limit(st));
      if (st + 1 < NSTEP) issue_b((st + 1) & 1, st + 1);
      if (more && (st & 1) == 0 && st / 2 < NPRE) fetch_one(chunk + 1, st / 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(st & 1, st);
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(NPRE <= (NSTEP + 1) / 2, "one fill load per pair of steps");
    if (more) {
      __syncthreads();
      park();
      __syncthreads();
    }
  }
  // epilogue: rows 4g .. 4g+3 = (NS = 2) shift g / 2, channels 4 (g % 2) .. + 3 | (NS = 1) channels 4g .. 4g + 3 of the tile's plane
  const int y = y0 + wrow;
  if (y < a.H) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      const int x = x0 + 16 * (t & 1) + j;
      const int z = z0 + (NS == 2 ? (g >> 1) : (kPlaneWave ? wzo : (t >> 1))), c = NS == 2 ? (g & 1) * 4 : cg * 16 + 4 * g;
      if (x >= a.W || z >= a.D) continue;
      const unsigned ov = (unsigned)(((vb * a.D + z) * a.H + y) * a.W + x);
      float4 o = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      o.x = o.x > 0.f ? o.x : o.x * a.slope; o.y = o.y > 0.f ? o.y : o.y * a.slope;
      o.z = o.z > 0.f ? o.z : o.z * a.slope; o.w = o.w > 0.f ? o.w : o.w * a.slope;
      if (a.residual) {
        const float4 rr = *reinterpret_cast<const float4*>(a.residual + (ov * (unsigned)a.res_cs + (unsigned)(a.res_coff + c)));
        o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
      }
      *reinterpret_cast<float4*>(a.out + (ov * (unsigned)a.out_cs + (unsigned)(a.out_coff + c))) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Trilinear x2 up-sampling FUSED into the convolution that follows it (CostRegNet's decoder: models/cost_reg.py:58-94,
// 136-158: `conv(interpolate(x, scale_factor=2, mode='trilinear'))` for conv7 / conv9 / conv11).  The composition is linear in
// the LOW-resolution tensor t: the output voxel (2i + p) -- p in {0,1}^3 its parity -- depends on the 3x3x3 neighbourhood of low-res
// voxel i only,   out[2i + p] = sum_{n in {-1,0,1}^3} K_p[n] t[i + n],   K_p = the 3x3x3 kernel folded with the interpolation
// weights of the up to three up-sampled taps per axis.  So the 8 parities x Cout output channels of one low-res voxel are 8*Cout
// ROWS of one GEMM that share a single B operand: Cout = 8 fills two whole 32-row MFMA tiles (the depth-shift packing of
// conv3d_small_kernel wastes half of its rows), the up-sampled volume is never written or read, and the operand loads come
// from a tensor an eighth of the size.
// Borders: the interpolation clamps at the volume faces and the convolution pads the UP-SAMPLED volume (zeros along D and H,
// circular along W), so the folded kernel of a voxel depends on its position class per axis -- first / interior / last --
// 27 combinations, each with its own packed weight block (pgrf_pack_conv3d_upfused).  Work items enumerate the voxels of one
// class box at a time, so the A operand stays wave-uniform.
struct UpArgs {
  ConvArgs c;                 // in = low-res tensor (D, H, W = its dims), out / residual = high-res (Do = 2D, ...)
  const float* bias4;         // [Cout] plain
  int nsteps, combo_stride;   // 32-deep reduction steps; floats per class block
  int item_start[28];         // prefix sums of wave items per class box
  int lo[27][3], n[27][3];    // box origin / extent per class (z, y, x)
};

template <int CIN, int MT, int NT>
__global__ void __launch_bounds__(256) conv3d_upfused_kernel(const UpArgs ua) {
  const ConvArgs& a = ua.c;
  constexpr int TPH = CIN >= 16 ? 1 : 16 / CIN;
  constexpr int QPT = CIN >= 16 ? 4 : CIN / 4;
  constexpr int COUT = MT * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, kk = lane >> 5;
  const int item = blockIdx.x * 4 + wave;
  if (item >= ua.item_start[27]) return;
  int cls = 0;
#pragma unroll 1
  for (int c = 1; c < 27; ++c) cls = (item >= ua.item_start[c]) ? c : cls;
  cls = __builtin_amdgcn_readfirstlane(cls);
  const int nz = ua.n[cls][0], ny = ua.n[cls][1], nx = ua.n[cls][2];
  const int nbox = a.B * nz * ny * nx;                         // 32-bit index arithmetic throughout (sizes checked by the launcher):
  const int tile0 = (item - ua.item_start[cls]) * NT;          // 64-bit div / mod is ~100 instructions each
  int vb[NT], vz[NT], vy[NT], vx[NT], cen[NT];
  bool vok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    unsigned v = (unsigned)((tile0 + t) * 32 + j);
    vok[t] = v < (unsigned)nbox;
    if (!vok[t]) v = (unsigned)nbox - 1u;
    vx[t] = ua.lo[cls][2] + (int)(v % (unsigned)nx); v /= (unsigned)nx;
    vy[t] = ua.lo[cls][1] + (int)(v % (unsigned)ny); v /= (unsigned)ny;
    vz[t] = ua.lo[cls][0] + (int)(v % (unsigned)nz); vb[t] = (int)(v / (unsigned)nz);
    cen[t] = (((vb[t] * a.D + vz[t]) * a.H + vy[t]) * a.W) * a.in_cs;       // element offset of the voxel's ROW start (x = 0)
  }
  const int zstep = a.H * a.W * a.in_cs, ystep = a.W * a.in_cs;          // < 2^23 (launcher): 24-bit multiplies below
  f32x16 acc[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.f;
  const float* wcls = ua.c.wpk + (size_t)cls * ua.combo_stride;
  const float* inb = a.in + a.in_coff + (CIN == 32 ? kk * 16 : 0);       // uniform base + 32-bit element offsets (size checked by the launcher)
  struct Step {
    float4 bv[NT][4];
    float4 av[MT][4];
  };
  // NO masking of out-of-volume neighbours: their FOLDED weights are exactly zero (the zero padding of the up-sampled volume and
  // the interpolation clamp are inside the class's kernel), so any finite value may stand in -- the address is clamped into the
  // volume.  (conv3d_small_kernel needs its selects because there the padding is not in the weights.)
  auto issue = [&](Step& st, int step) {
#pragma unroll
    for (int u = 0; u < TPH; ++u) {
      int tap = CIN == 32 ? step : (CIN == 16 ? 2 * step + kk : 4 * step + 2 * kk + u);
      tap = tap < 27 ? tap : 13;                       // padding steps carry zero weights
      const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        int xi = vx[t] + dx;
        xi = xi < 0 ? xi + a.W : (xi >= a.W ? xi - a.W : xi);
        const int dzc = min(max(vz[t] + dz, 0), a.D - 1) - vz[t], dyc = min(max(vy[t] + dy, 0), a.H - 1) - vy[t];   // -1 | 0 | 1
#if defined(PGRF_UPF_ABL) && (PGRF_UPF_ABL & 2)      // timing experiment: every B operand from one resident line
        const unsigned off = (unsigned)(lane * 4 + 0 * (dzc + dyc + xi + cen[t] + zstep + ystep));
#else
        const unsigned off = (unsigned)(cen[t] + __mul24(dzc, zstep) + __mul24(dyc, ystep) + __mul24(xi, a.in_cs));
#endif
        const float* ip = inb + off;
#pragma unroll
        for (int q = 0; q < QPT; ++q) st.bv[t][u * QPT + q] = *reinterpret_cast<const float4*>(ip + q * 4);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#if defined(PGRF_UPF_ABL) && (PGRF_UPF_ABL & 1)      // timing experiment: every A operand from the same 4 KB
      const float4* ap = reinterpret_cast<const float4*>(wcls) + ((unsigned)(m * 0 + (step & 1)) * 256u + (unsigned)lane);
#else
      const float4* ap = reinterpret_cast<const float4*>(wcls) + ((unsigned)(m * ua.nsteps + step) * 256u + (unsigned)lane);
#endif
      st.av[m][0] = ap[0]; st.av[m][1] = ap[64]; st.av[m][2] = ap[128]; st.av[m][3] = ap[192];
    }
  };
  auto compute = [&](Step& st) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float4 aq = st.av[m][s4];
          const float as = (jj == 0) ? aq.x : (jj == 1) ? aq.y : (jj == 2) ? aq.z : aq.w;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float4 bq = st.bv[t][s4];
            const float bs = (jj == 0) ? bq.x : (jj == 1) ? bq.y : (jj == 2) ? bq.z : bq.w;
            acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bs, acc[t][m], 0, 0, 0);
          }
        }
  };
  // bottom-tested loop over PAIRS of steps (nsteps is even: the packer pads with a zero-weight step): no mid-loop exit, so the
  // compiler does not rotate the loop and copy the accumulators (63 v_accvgpr_mov per iteration in the first version)
  // (Measured, 64x128x256 out, conv11: 198 us = 48 % MFMA-pipe duty in every variant tried -- 1, 2 or 4 voxel tiles per wave,
  // masks at issue or at use or none, 64- or 32-bit addressing, residual loads batched, scheduling fences that pin the next
  // step's loads in front of this step's MFMAs (209-224 us): the compiler sinks the loads to their first use either way.)
  Step P, Q;
  issue(P, 0);
  int step = 0;
  do {
    issue(Q, step + 1);
    compute(P);
    issue(P, min(step + 2, ua.nsteps - 1));          // (the last one re-loads its own step; nobody reads it)
    compute(Q);
    step += 2;
  } while (step < ua.nsteps);
  // epilogue: row = m*32 + q*8 + kk*4 + i = parity * COUT + channel -> high-res voxel 2i + parity.  All residual loads are
  // issued BEFORE the first store (the compiler cannot prove that `out` and `residual` do not alias and serialised one HBM
  // round trip per 16-byte piece: 16 of them per wave, a third of the kernel)
  unsigned ovs[NT][MT][4];
  int cs_[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) cs_[m][q] = (m * 32 + q * 8 + kk * 4) % COUT;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int par = (m * 32 + q * 8 + kk * 4) / COUT;
        const int oz = 2 * vz[t] + (par >> 2), oy = 2 * vy[t] + ((par >> 1) & 1), ox = 2 * vx[t] + (par & 1);
        ovs[t][m][q] = (unsigned)(((vb[t] * a.Do + oz) * a.Ho + oy) * a.Wo + ox);
      }
  float4 rr[NT][MT][4];
  if (a.residual) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rr[t][m][q] = *reinterpret_cast<const float4*>(a.residual + (ovs[t][m][q] * (unsigned)a.res_cs + (unsigned)(a.res_coff + cs_[m][q])));
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = cs_[m][q];
        const float4 b4 = ua.bias4 ? *reinterpret_cast<const float4*>(ua.bias4 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v = make_float4(acc[t][m][q * 4] + b4.x, acc[t][m][q * 4 + 1] + b4.y, acc[t][m][q * 4 + 2] + b4.z, acc[t][m][q * 4 + 3] + b4.w);
        v.x = v.x > 0.f ? v.x : v.x * a.slope; v.y = v.y > 0.f ? v.y : v.y * a.slope;
        v.z = v.z > 0.f ? v.z : v.z * a.slope; v.w = v.w > 0.f ? v.w : v.w * a.slope;
        if (a.residual) { v.x += rr[t][m][q].x; v.y += rr[t][m][q].y; v.z += rr[t][m][q].z; v.w += rr[t][m][q].w; }
#if defined(PGRF_UPF_ABL) && (PGRF_UPF_ABL & 4)      // timing experiment: no output stores
        if (vok[t] && v.x == -12345.f)
#else
        if (vok[t])
#endif
          *reinterpret_cast<float4*>(a.out + (ovs[t][m][q] * (unsigned)a.out_cs + (unsigned)(a.out_coff + c))) = v;
      }
  }
}

// Cout == 1 (UNet2's last block, CostRegNet's prob head): VALU dot products, LPV lanes per voxel.
// wlin: [27][Cin] (tap-major), bias scalar.
__global__ void __launch_bounds__(256) conv3d_cout1_kernel(const ConvArgs a, const float* __restrict__ wlin, int has_bias, int lpv) {
  const long long nvox = (long long)a.B * a.Do * a.Ho * a.Wo;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long v = gid / lpv;
  const int sub = (int)(gid % lpv);
  const bool ok = v < nvox;
  if (!ok) v = nvox - 1;
  const long long ov = v;
  const int x = (int)(v % a.Wo); v /= a.Wo;
  const int y = (int)(v % a.Ho); v /= a.Ho;
  const int z = (int)(v % a.Do); const int b = (int)(v / a.Do);
  float s = 0.f;
  for (int tap = 0; tap < 27; ++tap) {
    const int zi = z * a.stride + tap / 9 - 1, yi = y * a.stride + (tap / 3) % 3 - 1;
    int xi = x * a.stride + tap % 3 - 1;
    if (!a.wrap_w && (xi < 0 || xi >= a.W)) continue;
    xi = xi < 0 ? xi + a.W : (xi >= a.W ? xi - a.W : xi);
    if (zi < 0 || zi >= a.D || yi < 0 || yi >= a.H) continue;
    const float* ip = a.in + ((((long long)b * a.D + zi) * a.H + yi) * a.W + xi) * a.in_cs + a.in_coff;
    const float* wp = wlin + tap * a.Cin;
    if ((a.Cin & 3) == 0) {
      for (int c = sub * 4; c < a.Cin; c += lpv * 4) {
        const float4 xv = *reinterpret_cast<const float4*>(ip + c);
        const float4 wv = *reinterpret_cast<const float4*>(wp + c);
        s = fmaf(xv.x, wv.x, s); s = fmaf(xv.y, wv.y, s); s = fmaf(xv.z, wv.z, s); s = fmaf(xv.w, wv.w, s);
      }
    } else {
      for (int c = sub; c < a.Cin; c += lpv) s = fmaf(ip[c], wp[c], s);
    }
  }
  for (int off = lpv >> 1; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (ok && sub == 0) {
    if (has_bias) s += wlin[27 * a.Cin];
    const float rr = a.residual ? a.residual[ov * a.res_cs + a.res_coff] : 0.f;
    if (a.res_first) s += rr;
    s = conv_act(a, s);
    if (!a.res_first) s += rr;
    a.out[ov * a.out_cs + a.out_coff] = s;
  }
}

// Cout == 1 from ONE input channel (the second convolution of UNet2's last block, models/common_blocks.py:366-455 with one channel):
// the general kernel above spends ~40 instructions of 64-bit index arithmetic and branches per tap (78 us at 64x128x256 for 8 MB in,
// 8 MB out).  One thread per output, 32-bit indices, the 27 weights in scalar registers, row bases per (dz, dy): 11 us.
__global__ void __launch_bounds__(256) conv3d_cout1_c1_kernel(const ConvArgs a, const float* __restrict__ wlin, int has_bias) {
  const unsigned nvox = (unsigned)(a.B * a.D * a.H * a.W);
  const unsigned v = blockIdx.x * 256u + threadIdx.x;
  if (v >= nvox) return;
  const int x = (int)(v % (unsigned)a.W);
  unsigned t = v / (unsigned)a.W;
  const int y = (int)(t % (unsigned)a.H); t /= (unsigned)a.H;
  const int z = (int)(t % (unsigned)a.D), b = (int)(t / (unsigned)a.D);
  const int xl = x == 0 ? a.W - 1 : x - 1, xr = x == a.W - 1 ? 0 : x + 1;
  const float* base = a.in + a.in_coff;
  float s = 0.f;
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
    const int zi = z + dz - 1;
    if (zi < 0 || zi >= a.D) continue;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yi = y + dy - 1;
      if (yi < 0 || yi >= a.H) continue;
      const unsigned row = (unsigned)(((b * a.D + zi) * a.H + yi) * a.W);
      const float* w = wlin + (dz * 3 + dy) * 3;
      s = fmaf(base[(row + (unsigned)xl) * (unsigned)a.in_cs], w[0], s);
      s = fmaf(base[(row + (unsigned)x) * (unsigned)a.in_cs], w[1], s);
      s = fmaf(base[(row + (unsigned)xr) * (unsigned)a.in_cs], w[2], s);
    }
  }
  if (has_bias) s += wlin[27];
  const float rr = a.residual ? a.residual[(size_t)v * a.res_cs + a.res_coff] : 0.f;
  if (a.res_first) s += rr;
  s = conv_act(a, s);
  if (!a.res_first) s += rr;
  a.out[(size_t)v * a.out_cs + a.out_coff] = s;
}

// Cout == 1 from EIGHT input channels (CostRegNet's prob head, models/cost_reg.py:165-172, at full resolution): the general
// kernel above issues 54 16-byte loads per output voxel (27 taps x 2) and ran at 143 us for 64x128x256 -- 67 MB in, 8 MB out.
// Here a thread owns YB = 4 consecutive outputs along y of one (b, z, x): an input row (dz, y0 - 1 + r, dx) is loaded once
// and feeds the up to three outputs whose 3-tap window covers it, 27 loads per output instead of 54; consecutive lanes are
// consecutive x, so a wave load covers 64 voxels x 16 B of one row.  Weights [27][8] (+ bias) sit in LDS.
constexpr int kC8YB = 4;
__global__ void __launch_bounds__(256) conv3d_cout1_c8_kernel(const ConvArgs a, const float* __restrict__ wlin, int has_bias) {
  __shared__ __attribute__((aligned(16))) float s_w[27 * 8 + 8];
  for (int i = threadIdx.x; i < 27 * 8 + 1; i += 256) s_w[i] = (i < 216 || has_bias) ? wlin[i] : 0.f;
  __syncthreads();
  const int ygroups = (a.Ho + kC8YB - 1) / kC8YB;
  const long long nthreads = (long long)a.B * a.Do * ygroups * a.Wo;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nthreads) return;
  const int x = (int)(t % a.Wo); t /= a.Wo;
  const int yg = (int)(t % ygroups); t /= ygroups;
  const int z = (int)(t % a.Do); const int b = (int)(t / a.Do);
  const int y0 = yg * kC8YB;
  int xs[3];
  bool xok[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    int xi = x + dx - 1;
    xok[dx] = a.wrap_w || (xi >= 0 && xi < a.W);
    xs[dx] = xi < 0 ? xi + a.W : (xi >= a.W ? xi - a.W : xi);
  }
  float s[kC8YB];
#pragma unroll
  for (int o = 0; o < kC8YB; ++o) s[o] = 0.f;
  for (int dz = 0; dz < 3; ++dz) {
    const int zi = z + dz - 1;
    if (zi < 0 || zi >= a.D) continue;
#pragma unroll
    for (int r = 0; r < kC8YB + 2; ++r) {
      const int yi = y0 + r - 1;
      if (yi < 0 || yi >= a.H) continue;
      const float* rowp = a.in + (((long long)b * a.D + zi) * a.H + yi) * a.W * a.in_cs + a.in_coff;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (!xok[dx]) continue;
        const float4 v0 = *reinterpret_cast<const float4*>(rowp + (long long)xs[dx] * a.in_cs);
        const float4 v1 = *reinterpret_cast<const float4*>(rowp + (long long)xs[dx] * a.in_cs + 4);
#pragma unroll
        for (int o = 0; o < kC8YB; ++o) {
          const int ky = r - o;                          // this row is tap ky of output o
          if (ky < 0 || ky > 2) continue;
          const float4 w0 = *reinterpret_cast<const float4*>(&s_w[((dz * 3 + ky) * 3 + dx) * 8]);
          const float4 w1 = *reinterpret_cast<const float4*>(&s_w[((dz * 3 + ky) * 3 + dx) * 8 + 4]);
          float acc = s[o];
          acc = fmaf(v0.x, w0.x, acc); acc = fmaf(v0.y, w0.y, acc); acc = fmaf(v0.z, w0.z, acc); acc = fmaf(v0.w, w0.w, acc);
          acc = fmaf(v1.x, w1.x, acc); acc = fmaf(v1.y, w1.y, acc); acc = fmaf(v1.z, w1.z, acc); acc = fmaf(v1.w, w1.w, acc);
          s[o] = acc;
        }
      }
    }
  }
  const float bias = s_w[216];
#pragma unroll
  for (int o = 0; o < kC8YB; ++o) {
    const int y = y0 + o;
    if (y >= a.Ho) break;
    const long long ov = (((long long)b * a.Do + z) * a.Ho + y) * a.Wo + x;
    float v = s[o] + bias;
    const float rr = a.residual ? a.residual[ov * a.res_cs + a.res_coff] : 0.f;
    if (a.res_first) v += rr;
    v = conv_act(a, v);
    if (!a.res_first) v += rr;
    a.out[ov * a.out_cs + a.out_coff] = v;
  }
}

// The same head with the input brick staged in LDS.  The kernel above is bound by the L1: 27 x 32 bytes per output (halved by the
// y reuse) at the 64 B/clk/CU of the vector-memory path is the 57 us it takes at 64x128x256.  Here a workgroup owns 2 x 16 x 32
// outputs; the 4 x 18 x 34 input voxels go through the L1 once (quad-planar in LDS: consecutive lanes = consecutive 16-byte
// slots, conflict-free), a thread owns 2 x 2 outputs (z, y) of one column and walks the 4 x 4 input rows it needs: the weights
// it needs come from scalar loads (FMA with a scalar operand), and every input row of 3 voxels is read ONCE and feeds all of the
// thread's outputs whose window covers it: 24 ds_read_b128 per output instead of 54 loads through the L1.
struct C8Sub {                                               // sub-step t of the head kernel below: the t-th valid (pz, r, zo, yo), pz-major
  int pz, r, zo, yo;
  static constexpr C8Sub of(int t) {
    int n = 0;
    for (int pz = 0; pz < 4; ++pz)
      for (int r = 0; r < 4; ++r)
        for (int zo = 0; zo < 2; ++zo)
          for (int yo = 0; yo < 2; ++yo) {
            if (pz - zo < 0 || pz - zo > 2 || r - yo < 0 || r - yo > 2) continue;
            if (n == t) return C8Sub{pz, r, zo, yo};
            ++n;
          }
    return C8Sub{0, 0, 0, 0};
  }
};
template <int... I, class F>
__device__ __forceinline__ void c8_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void c8_static_for(F&& f) { c8_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
__global__ void __launch_bounds__(256, 2) conv3d_cout1_c8_brick_kernel(const ConvArgs a, const float* __restrict__ wlin, int has_bias, unsigned in_bytes) {
  constexpr int BX = 32, BY = 16, PX = BX + 2, PY = BY + 2, PZ = 4, NVOX = PX * PY * PZ, NROW = PY * PZ;
  __shared__ float4 bq[2][NVOX];                           // [channel quad][voxel]: 78 336 B, two workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbz = (a.D + 1) >> 1;
  const int vb = (int)blockIdx.z / nbz, bz = (int)blockIdx.z - vb * nbz;
  const int x0 = (int)blockIdx.x * BX, y0 = (int)blockIdx.y * BY, z0 = bz * 2;
  {
    // fill: 72 rows (pz, py) of 34 voxels x 2 quads; thread slot u = tid % 128 < 68 -> (column u / 2, quad u % 2), the two halves of the
    // workgroup take 36 rows each (row term wave-uniform, rows in the zero padding skipped)
    const unsigned long long bits = (unsigned long long)(a.in + a.in_coff);
    const i32x4v rsrc = {(int)(unsigned)bits, (int)(unsigned)(bits >> 32), (int)in_bytes, 0x00020000};
    const int half = wave >> 1, u = tid & 127, px = u >> 1, q = u & 1;
    int x = (x0 - 1 + px) % a.W;
    x = x < 0 ? x + a.W : x;
    const unsigned voff = u < 2 * PX ? (unsigned)(x * a.in_cs + 4 * q) * 4u : 0x80000000u;
    const int ystep = a.W * a.in_cs;
    // (all 36 loads of a thread in flight at once: the compute phase needs a quarter of the 256 registers two workgroups per CU allow)
    float4 v[NROW / 2];
#pragma unroll
    for (int i = 0; i < NROW / 2; ++i) {
      const int r = half * (NROW / 2) + i, pz = r / PY, py = r - pz * PY;          // uniform
      const int z = z0 - 1 + pz, y = y0 - 1 + py;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H) {
        const f32x4v t = pgrf_buffer_load_x4(rsrc, (int)voff, (((vb * a.D + z) * a.H + y) * ystep) * 4, 0);
        v[i] = make_float4(t.x, t.y, t.z, t.w);
      }
    }
    if (u < 2 * PX) {
#pragma unroll
      for (int i = 0; i < NROW / 2; ++i) bq[q][(half * (NROW / 2) + i) * PX + px] = v[i];
    }
  }
  __syncthreads();
  const int xi = lane & 31, yp = wave * 2 + (lane >> 5);   // column, pair of output rows 2 yp, 2 yp + 1
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};              // [output plane][output row]
  // walk the 4 x 4 input (plane, row) pairs of the thread's column: each is read once (6 ds_read_b128) and feeds every (output plane,
  // output row) whose window covers it: 36 sub-steps (pz, r, zo, yo) of 24 FMAs against the 24 weights of kernel (plane, row) =
  // (pz - zo, r - yo), which come from scalar loads issued ONE sub-step ahead.
  constexpr int NSUB = 36;
  typedef float f32x8s __attribute__((ext_vector_type(8)));
  f32x8s wn[3];                                            // weights of the next sub-step: [dx][8 channels]
  auto load_w = [&](auto t_tag) __attribute__((always_inline)) {
    constexpr C8Sub u = C8Sub::of(decltype(t_tag)::value);
    // uniform address: scalar loads.  Made opaque per sub-step and chained on the accumulators, so that the loads of sub-step
    // t + 1 are issued behind the FMAs of sub-step t - 1 and not all at the kernel entry (216 weights do not fit the scalar
    // registers: the compiler spilled them lane by lane), then declared wave-uniform again behind the asm
    unsigned long long wbits = (unsigned long long)(wlin + ((u.pz - u.zo) * 3 + (u.r - u.yo)) * 24);
    unsigned wlo = (unsigned)wbits, whi = (unsigned)(wbits >> 32);
    asm volatile("" : "+s"(wlo), "+s"(whi), "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
    wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
    const f32x8s __attribute__((address_space(4)))* wp = (const f32x8s __attribute__((address_space(4)))*)(((unsigned long long)whi << 32) | wlo);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) wn[dx] = wp[dx];
  };
  float4 in0[3], in1[3];
  load_w(std::integral_constant<int, 0>{});
  c8_static_for<NSUB>([&](auto t_tag) __attribute__((always_inline)) {
    constexpr int t = decltype(t_tag)::value;
    constexpr C8Sub u = C8Sub::of(t), up = C8Sub::of(t > 0 ? t - 1 : 0);
    const f32x8s w0 = wn[0], w1 = wn[1], w2 = wn[2];
    if constexpr (t + 1 < NSUB) load_w(std::integral_constant<int, t + 1>{});
    if constexpr (t == 0 || up.pz != u.pz || up.r != u.r) {          // a new input row: row 2 yp - 1 + r of input plane z0 - 1 + pz
      const float4* p0 = &bq[0][(u.pz * PY + 2 * yp + u.r) * PX + xi];
      const float4* p1 = &bq[1][(u.pz * PY + 2 * yp + u.r) * PX + xi];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) { in0[dx] = p0[dx]; in1[dx] = p1[dx]; }
    }
    float sacc = acc[u.zo][u.yo];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const f32x8s w = dx == 0 ? w0 : dx == 1 ? w1 : w2;
      sacc = fmaf(in0[dx].x, w[0], sacc); sacc = fmaf(in0[dx].y, w[1], sacc); sacc = fmaf(in0[dx].z, w[2], sacc); sacc = fmaf(in0[dx].w, w[3], sacc);
      sacc = fmaf(in1[dx].x, w[4], sacc); sacc = fmaf(in1[dx].y, w[5], sacc); sacc = fmaf(in1[dx].z, w[6], sacc); sacc = fmaf(in1[dx].w, w[7], sacc);
    }
    acc[u.zo][u.yo] = sacc;
  });
  const float bias = has_bias ? wlin[216] : 0.f;
  const int x = x0 + xi;
  if (x < a.W) {
#pragma unroll
    for (int zo = 0; zo < 2; ++zo)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) {
        const int z = z0 + zo, y = y0 + 2 * yp + yo;
        if (z >= a.D || y >= a.H) continue;
        const long long ov = (((long long)vb * a.D + z) * a.H + y) * a.W + x;
        float v = acc[zo][yo] + bias;
        const float rr = a.residual ? a.residual[ov * a.res_cs + a.res_coff] : 0.f;
        if (a.res_first) v += rr;
        v = conv_act(a, v);
        if (!a.res_first) v += rr;
        a.out[ov * a.out_cs + a.out_coff] = v;
      }
  }
}

// Second half of the Cout == 1 convolution: out[p] = act(bias + sum_tap y[tap][p + tap]) where y[tap][q] =
// <w[:, tap], x[q, :]> was produced by the pointwise MFMA pass in tap-major order (so a wave reads 64 consecutive
// voxels of one row per tap; the channel-last form touched one 128-byte line per lane and tap: 975 -> ~100 us).
__global__ void __launch_bounds__(256) cout1_gather_kernel(const float* __restrict__ y, const float* __restrict__ bias0, float* __restrict__ out,
                                                           int out_cs, int out_coff, int B, int D, int H, int W, float slope,
                                                           const float* __restrict__ residual, int res_cs, int res_coff) {
  const long long nvox = (long long)B * D * H * W;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (long long)gridDim.x * blockDim.x) {
    long long t = v;
    const int x = (int)(t % W); t /= W;
    const int yy = (int)(t % H); t /= H;
    const int z = (int)(t % D); const int b = (int)(t / D);
    float s = 0.f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int zi = z + tap / 9 - 1, yi = yy + (tap / 3) % 3 - 1;
      int xi = x + tap % 3 - 1;
      xi = xi < 0 ? xi + W : (xi >= W ? xi - W : xi);
      if (zi >= 0 && zi < D && yi >= 0 && yi < H) s += y[(long long)tap * nvox + (((long long)b * D + zi) * H + yi) * W + xi];
    }
    if (bias0) s += bias0[0];
    s = s > 0.f ? s : s * slope;
    if (residual) s += residual[v * res_cs + res_coff];
    out[v * out_cs + out_coff] = s;
  }
}

// F.interpolate(scale_factor=2, mode='trilinear', align_corners=False) on channel-last data
__device__ __forceinline__ void up_idx(int o, int n, int& i0, int& i1, float& l0, float& l1) {
  float src = 0.5f * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
// The same gather for a convolution over torch.cat((up(x), skip)) -- the last decoder step of UNet2 (models/common_blocks.py:225-241):
// the projection onto the 27 tap vectors is linear and channel-wise, so it commutes with the trilinear up-sampling,
//     <w_tap, up(x)[q]> = up(<w_tap, x>)[q].
// x is projected at LOW resolution (y_lo, 1/8 of the voxels), the skip half at full resolution (y_hi), and this kernel interpolates
// the 27 low-resolution planes at q = p + tap on the fly (same per-axis weights as upsample3d_kernel).  The up-sampled tensor is never
// written (64 channels at full resolution: 537 MB at 64x128x256) and the full-resolution projection reads half the channels.
__global__ void __launch_bounds__(256) cout1_gather_up_kernel(const float* __restrict__ y_hi, const float* __restrict__ y_lo,
                                                              const float* __restrict__ bias0, float* __restrict__ out, int out_cs, int out_coff,
                                                              int B, int D, int H, int W, float slope, const float* __restrict__ residual,
                                                              int res_cs, int res_coff) {
  const long long nvox = (long long)B * D * H * W;
  const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
  const long long nlow = (long long)B * Dl * Hl * Wl;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (long long)gridDim.x * blockDim.x) {
    long long t = v;
    const int x = (int)(t % W); t /= W;
    const int yy = (int)(t % H); t /= H;
    const int z = (int)(t % D); const int b = (int)(t / D);
    // per axis: the three high-res neighbours, their low-res support and weights
    int zi[3], yi[3], xi[3], z0[3], z1[3], y0[3], y1[3], x0[3], x1[3];
    float lz0[3], lz1[3], ly0[3], ly1[3], lx0[3], lx1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      zi[k] = z + k - 1; yi[k] = yy + k - 1;
      int xw = x + k - 1;
      xi[k] = xw < 0 ? xw + W : (xw >= W ? xw - W : xw);
      up_idx(min(max(zi[k], 0), D - 1), Dl, z0[k], z1[k], lz0[k], lz1[k]);
      up_idx(min(max(yi[k], 0), H - 1), Hl, y0[k], y1[k], ly0[k], ly1[k]);
      up_idx(xi[k], Wl, x0[k], x1[k], lx0[k], lx1[k]);
    }
    float s = 0.f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int a = tap / 9, bb = (tap / 3) % 3, c = tap % 3;
      // taps outside the volume: every load still happens (clamped address) and the term is multiplied by 0 -- a `continue` here made
      // each tap its own branch region whose nine loads were waited for before the next tap's were issued (27 serial round trips)
      const bool ok = zi[a] >= 0 && zi[a] < D && yi[bb] >= 0 && yi[bb] < H;
      const int zc = min(max(zi[a], 0), D - 1), yc = min(max(yi[bb], 0), H - 1);
      const float* lo = y_lo + (long long)tap * nlow + (long long)b * Dl * Hl * Wl;
      const int r00 = (z0[a] * Hl + y0[bb]) * Wl, r01 = (z0[a] * Hl + y1[bb]) * Wl;
      const int r10 = (z1[a] * Hl + y0[bb]) * Wl, r11 = (z1[a] * Hl + y1[bb]) * Wl;
      const float up = lz0[a] * (ly0[bb] * (lx0[c] * lo[r00 + x0[c]] + lx1[c] * lo[r00 + x1[c]]) + ly1[bb] * (lx0[c] * lo[r01 + x0[c]] + lx1[c] * lo[r01 + x1[c]])) +
                       lz1[a] * (ly0[bb] * (lx0[c] * lo[r10 + x0[c]] + lx1[c] * lo[r10 + x1[c]]) + ly1[bb] * (lx0[c] * lo[r11 + x0[c]] + lx1[c] * lo[r11 + x1[c]]));
      const float term = up + y_hi[(long long)tap * nvox + (((long long)b * D + zc) * H + yc) * W + xi[c]];
      s += ok ? term : 0.f;
    }
    if (bias0) s += bias0[0];
    s = s > 0.f ? s : s * slope;
    if (residual) s += residual[v * res_cs + res_coff];
    out[v * out_cs + out_coff] = s;
  }
}

// AvgPool3d(2) on channel-last data; thread per (output voxel, 4-channel group)
__global__ void avgpool3d_kernel(const float* __restrict__ in, int in_cs, int in_coff, float* __restrict__ out, int out_cs,
                                 int out_coff, int B, int D, int H, int W, int C, int stream_loads) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2, g = C / 4;
  const long long total = (long long)B * Do * Ho * Wo * g;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % g);
    long long v = i / g;
    const int x = (int)(v % Wo); v /= Wo;
    const int y = (int)(v % Ho); v /= Ho;
    const int z = (int)(v % Do); const int b = (int)(v / Do);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zi = 2 * z + (k >> 2), yi = 2 * y + ((k >> 1) & 1), xi = 2 * x + (k & 1);
      typedef float f32x4nt __attribute__((ext_vector_type(4)));
      const f32x4nt* tp = reinterpret_cast<const f32x4nt*>(in + ((((long long)b * D + zi) * H + yi) * W + xi) * in_cs + in_coff + cg * 4);
      const f32x4nt t = stream_loads ? __builtin_nontemporal_load(tp) : *tp;          // large inputs: non-temporal
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    s.x /= 8.f; s.y /= 8.f; s.z /= 8.f; s.w /= 8.f;
    *reinterpret_cast<float4*>(out + (i / g) * out_cs + out_coff + cg * 4) = s;
  }
}

// One thread = one INPUT voxel and channel quad = the 2x2x2 output block around it: the 27 neighbours are loaded once
// (3.4 loads per output instead of 8; the one-thread-per-output version ran at 2.4 TB/s of output because of its load
// count).  Every output is the same expression as torch's trilinear kernel (align_corners=False): output 2i+a reads
// inputs (i-1, i) for a = 0 and (i, i+1) for a = 1, clamped, with up_idx's weights.
__global__ void __launch_bounds__(256) upsample3d_kernel(const float* __restrict__ in, int in_cs, int in_coff, float* __restrict__ out,
                                                         int out_cs, int out_coff, int B, int D, int H, int W, int C, int stream_stores) {
  const int g = C / 4;
  const long long total = (long long)B * D * H * W * g;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % g);
    long long v = i / g;
    const int x = (int)(v % W); v /= W;
    const int y = (int)(v % H); v /= H;
    const int z = (int)(v % D); const int b = (int)(v / D);
    const int zs[3] = {max(z - 1, 0), z, min(z + 1, D - 1)}, ys[3] = {max(y - 1, 0), y, min(y + 1, H - 1)};
    const int xs[3] = {max(x - 1, 0), x, min(x + 1, W - 1)};
    float4 n[3][3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int bb = 0; bb < 3; ++bb)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          n[a][bb][c] = *reinterpret_cast<const float4*>(in + ((((long long)b * D + zs[a]) * H + ys[bb]) * W + xs[c]) * in_cs + in_coff + cg * 4);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int i0, i1; float dz0, dz1;
      up_idx(2 * z + a, D, i0, i1, dz0, dz1);
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        float dy0, dy1;
        up_idx(2 * y + bb, H, i0, i1, dy0, dy1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float dx0, dx1;
          up_idx(2 * x + c, W, i0, i1, dx0, dx1);
          const float4 &v000 = n[a][bb][c], &v001 = n[a][bb][c + 1], &v010 = n[a][bb + 1][c], &v011 = n[a][bb + 1][c + 1];
          const float4 &v100 = n[a + 1][bb][c], &v101 = n[a + 1][bb][c + 1], &v110 = n[a + 1][bb + 1][c], &v111 = n[a + 1][bb + 1][c + 1];
          float4 r;
#define PGRF_TRI(q) r.q = dz0 * (dy0 * (dx0 * v000.q + dx1 * v001.q) + dy1 * (dx0 * v010.q + dx1 * v011.q)) + \
                          dz1 * (dy0 * (dx0 * v100.q + dx1 * v101.q) + dy1 * (dx0 * v110.q + dx1 * v111.q));
          PGRF_TRI(x) PGRF_TRI(y) PGRF_TRI(z) PGRF_TRI(w)
#undef PGRF_TRI
          const long long ov = (((long long)b * 2 * D + 2 * z + a) * 2 * H + 2 * y + bb) * 2 * W + 2 * x + c;
          if (stream_stores) {     // large outputs: non-temporal (the consumer cannot find > 64 MB in the last-level cache anyway)
            typedef float f32x4nt __attribute__((ext_vector_type(4)));
            const f32x4nt vv = {r.x, r.y, r.z, r.w};
            __builtin_nontemporal_store(vv, reinterpret_cast<f32x4nt*>(out + ov * out_cs + out_coff + cg * 4));
          } else {
            *reinterpret_cast<float4*>(out + ov * out_cs + out_coff + cg * 4) = r;
          }
        }
      }
    }
  }
}

static int grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---------------------------------------------------------------------------------------------------------------
// The same fused step (up-sampling x2 + 3x3x3 convolution, Cin = 16 -> Cout = 8: conv11 of CostRegNet) on 16-row MFMA tiles with
// the LOW-resolution input brick in LDS (the scheme of conv3d_brick16_kernel).  Rows = 2 z parities x 8 channels; the four (y, x)
// parities of a low-res voxel are four accumulators fed by the SAME B operand and their own folded weights, so every product is
// useful (3.4 MFMAs per high-res voxel against 4.5 for the direct form on the up-sampled volume and a 64-row tile pair per 32
// voxels for conv3d_upfused_kernel, whose waves quantise badly: 4112 items on 2048 slots).
// Position classes: a brick is ONE low-res z plane (class of the plane = block-uniform) and a wave owns ONE low-res row (y class =
// wave-uniform); the class only selects the weight block (scalar offset).  Along x every column is multiplied with the INTERIOR
// weights.  The two seam columns (x = W - 1 and 0: clamped interpolation next to the circular wrap) get a corrected B operand
// instead: with r0, r1, r2 the three x neighbours a seam lane reads, E the interior operator of the axis and
// D = (t[W-1] - t[0]) / 2, the values
//     x = 0:      x parity 0: (r0 + D, r1 - D, r2 + 3D)    parity 1: (r0 - 2D, r1, r2)
//     x = W - 1:  x parity 0: (r0, r1, r2 + 2D)            parity 1: (r0 - 3D, r1 + D, r2 - D)
// satisfy E^T b = E_seam^T r for every filter tap, i.e. the interior weights applied to them give exactly the seam voxel's
// result.  Bricks start at column -1 (= W - 1), so both seam columns are lanes 0 and 1 of the first tile of the FIRST brick of
// a row of bricks: that brick forms D once per brick row in LDS, reads it per tap and adds c * D with a per-lane coefficient
// that is zero off the seam (8 FMAs per tap in one brick out of W / 32, nothing elsewhere).
struct UpBrickArgs {
  ConvArgs c;                 // in = low-res tensor (D, H, W = its dims), out / residual = high-res (2D, 2H, 2W)
  unsigned in_bytes;
};

__global__ void __launch_bounds__(256, 4) conv3d_upbrick_kernel(const UpBrickArgs ua) {
  const ConvArgs& a = ua.c;
  constexpr int NTAP = 27, NSET = 4, NSTREAM = NTAP * NSET;
  constexpr int PX = 34, PY = 6, PZ = 3, VS = 16;
  constexpr int NV = PX * PY * PZ, NROW = PY * PZ;
  __shared__ __attribute__((aligned(16))) float brick[NV * VS + NROW * VS];     // 39 168 B + 1 152 B seam deltas: four workgroups per CU
  float* seamv = brick + NV * VS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int vb = (int)blockIdx.z / a.D, iz = (int)blockIdx.z - vb * a.D;
  const int x0 = (int)blockIdx.x * 32 - 1, y0 = (int)blockIdx.y * 4;          // columns x0 .. x0 + 31 (mod W): brick 0 starts at the seam
  const float* inb = a.in + a.in_coff;
  const unsigned long long inb_bits = (unsigned long long)inb;
  const i32x4v rsrc = {(int)(unsigned)inb_bits, (int)(unsigned)(inb_bits >> 32), (int)ua.in_bytes, 0x00020000};
  // fill: 18 rows (pz, py) of 34 voxels x 4 channel quads.  Columns 0..31: thread -> (column, quad) = tid % 128, the two halves of
  // the workgroup take 9 rows each (row term of the address wave-uniform); columns 32, 33 (144 quads): one more load per thread.
  const int half = wave >> 1;
  auto wrap_x = [&](int x) { x %= a.W; return x < 0 ? x + a.W : x; };
  const int ystep = a.W * a.in_cs;
  {
    float4 pre[10];
    const int cq = tid & 3, px = (tid & 127) >> 2;
    const unsigned src_col = (unsigned)(wrap_x(x0 - 1 + px) * a.in_cs + 4 * cq) * 4u;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int r = half * 9 + i, pz = r / PY, py = r - pz * PY;          // uniform
      const int z = iz - 1 + pz, y = y0 - 1 + py;
      pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H) {
        const f32x4v v = pgrf_buffer_load_x4(rsrc, (int)src_col, (((vb * a.D + z) * a.H + y) * ystep) * 4, 0);
        pre[i] = make_float4(v.x, v.y, v.z, v.w);
      }
    }
    const int row = tid >> 3, pxx = 32 + ((tid >> 2) & 1), pz = row / PY, py = row - pz * PY;
    const int z = iz - 1 + pz, y = y0 - 1 + py;
    const bool ok = tid < 144 && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H;
    const unsigned src_x = ok ? (unsigned)((((vb * a.D + z) * a.H + y) * a.W + wrap_x(x0 - 1 + pxx)) * a.in_cs + 4 * cq) * 4u : 0x80000000u;
    const f32x4v v = pgrf_buffer_load_x4(rsrc, (int)src_x, 0, 0);
    pre[9] = make_float4(v.x, v.y, v.z, v.w);
    char* dst_col = reinterpret_cast<char*>(brick) + ((half * 9 * PX + px) * VS + 4 * (cq ^ ((px >> 1) & 3))) * 4;
#pragma unroll
    for (int i = 0; i < 9; ++i) *reinterpret_cast<float4*>(dst_col + i * (PX * VS * 4)) = pre[i];
    if (tid < 144) *reinterpret_cast<float4*>(reinterpret_cast<char*>(brick) + ((row * PX + pxx) * VS + 4 * (cq ^ ((pxx >> 1) & 3))) * 4) = pre[9];
  }
  // wave w multiplies low-res row y0 + w: tiles x0 + j and x0 + 16 + j, four (py, px) parity sets each
  const int iy = y0 + wave;
  const int zc = iz == 0 ? 0 : (iz == a.D - 1 ? 2 : 1), yc = iy == 0 ? 0 : (iy >= a.H - 1 ? 2 : 1);
  const int cls_off = (zc * 3 + yc) * (NSTREAM * 1024);              // bytes, wave-uniform
  const float* bl[3];                                      // per nx: voxel (pz = 0, py = w, px = j + nx), this lane's (swizzled) channel quad
#pragma unroll
  for (int d = 0; d < 3; ++d) bl[d] = &brick[(wave * PX + j + d) * VS + 4 * (g ^ (((j + d) >> 1) & 3))];
  const unsigned long long w_bits = (unsigned long long)a.wpk;
  const i32x4v wrsrc = {(int)(unsigned)w_bits, (int)(unsigned)(w_bits >> 32), 9 * NSTREAM * 1024, 0x00020000};
  f32x4v acc[NSET][2];
  {
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + g * 4);
#pragma unroll
    for (int s = 0; s < NSET; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[s][t] = (f32x4v){b4.x, b4.y, b4.z, b4.w};
  }
  constexpr int AD = 6, RING = 10;                         // A loads in flight ahead of their MFMAs / ring size (>= AD + NSET)
  float4 av[RING], bv[2][2];
  auto issue_a = [&](int n) {                              // n = tap * 4 + parity set
    const f32x4v r = pgrf_buffer_load_x4(wrsrc, lane * 16, cls_off + ((n & 3) * NTAP + (n >> 2)) * 1024, 0);          // packed [class][set][tap]
    av[n % RING] = make_float4(r.x, r.y, r.z, r.w);
  };
  auto issue_b = [&](int slot, int tap) {
    const int nz = tap / 9, ny = (tap / 3) % 3, nx = tap % 3;
    const int off = (nz * PY + ny) * PX * VS;
    bv[slot][0] = *reinterpret_cast<const float4*>(bl[nx] + off);
    bv[slot][1] = *reinterpret_cast<const float4*>(bl[nx] + off + 16 * VS);
  };
  auto pick = [](const float4& q, int m) { return (m == 0) ? q.x : (m == 1) ? q.y : (m == 2) ? q.z : q.w; };
#pragma unroll
  for (int n = 0; n < AD; ++n) issue_a(n);
  __syncthreads();
  // the seam brick (first of its row of bricks; lanes 0 = column W - 1 and 1 = column 0 of tile 0): per-row deltas D, per-lane coefficients
  const bool seam = blockIdx.x == 0;
  const float* sv = seamv + wave * VS + 4 * g;             // row (pz = 0, py = w), this lane's channel quad (not swizzled: a broadcast read)
  if (seam) {
    if (tid < NROW * 4) {
      const int row = tid >> 2, q = tid & 3;               // brick columns 1, 2 = x = W - 1, 0
      const float4 va = *reinterpret_cast<const float4*>(&brick[(row * PX + 1) * VS + 4 * (q ^ 0)]);
      const float4 vb4 = *reinterpret_cast<const float4*>(&brick[(row * PX + 2) * VS + 4 * (q ^ 1)]);
      *reinterpret_cast<float4*>(&seamv[row * VS + 4 * q]) =
          make_float4(0.5f * (va.x - vb4.x), 0.5f * (va.y - vb4.y), 0.5f * (va.z - vb4.z), 0.5f * (va.w - vb4.w));
    }
    __syncthreads();
  }
  // SEAM: tile 0 takes corrected operands
  auto run = [&](auto seam_tag) __attribute__((always_inline)) {
    constexpr bool SEAM = decltype(seam_tag)::value;
    float cf[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // coefficient of D per (x parity, nx), zero off the seam
    float4 dd = make_float4(0.f, 0.f, 0.f, 0.f);           // D of the current (nz, ny) row
    if constexpr (SEAM) {
      if (j == 1) { cf[0][0] = 1.f; cf[0][1] = -1.f; cf[0][2] = 3.f; cf[1][0] = -2.f; }               // x = 0
      if (j == 0) { cf[0][2] = 2.f; cf[1][0] = -3.f; cf[1][1] = 1.f; cf[1][2] = -1.f; }               // x = W - 1
    }
    auto load_d = [&](int tap) { dd = *reinterpret_cast<const float4*>(sv + ((tap / 9) * PY + (tap / 3) % 3) * VS); };
    issue_b(0, 0);
    if constexpr (SEAM) load_d(0);
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
#pragma unroll
      for (int n = AD + 4 * tap; n < AD + 4 * tap + 4; ++n)
        if (n < NSTREAM) issue_a(n);
      if (tap + 1 < NTAP) issue_b((tap + 1) & 1, tap + 1);
      __builtin_amdgcn_sched_barrier(0);
      float4 bs[2];                                        // tile 0 of the seam brick: corrected operand per x parity
      if constexpr (SEAM) {
        const float4 r = bv[tap & 1][0];
        const int nx = tap % 3;
#pragma unroll
        for (int p = 0; p < 2; ++p)
          bs[p] = make_float4(fmaf(cf[p][nx], dd.x, r.x), fmaf(cf[p][nx], dd.y, r.y), fmaf(cf[p][nx], dd.z, r.z), fmaf(cf[p][nx], dd.w, r.w));
        if (nx == 2 && tap + 1 < NTAP) load_d(tap + 1);     // the next row's D: in place, a whole tap ahead of its first use
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int s = 0; s < NSET; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float b = (SEAM && t == 0) ? pick(bs[s & 1], m) : pick(bv[tap & 1][t], m);
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(av[(4 * tap + s) % RING], m), b, acc[s][t], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (seam) run(std::true_type{});
  else run(std::false_type{});
  // epilogue: rows 4g .. 4g+3 = z parity g / 2, channels 4 (g % 2) .. + 3 of high-res voxel (2 iz + pz, 2 iy + py, 2 ix + px)
  if (iy < a.H) {
    const int pz = g >> 1, c = (g & 1) * 4;
#pragma unroll
    for (int s = 0; s < NSET; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int ix = x0 + 16 * t + j;                          // -1 = column W - 1 (first brick); the last brick ends at column W - 2
        if (ix > a.W - 2) continue;
        ix = ix < 0 ? ix + a.W : ix;
        const unsigned ov = (unsigned)(((vb * a.Do + 2 * iz + pz) * a.Ho + 2 * iy + (s >> 1)) * a.Wo + 2 * ix + (s & 1));
        float4 o = make_float4(acc[s][t][0], acc[s][t][1], acc[s][t][2], acc[s][t][3]);
        o.x = o.x > 0.f ? o.x : o.x * a.slope; o.y = o.y > 0.f ? o.y : o.y * a.slope;
        o.z = o.z > 0.f ? o.z : o.z * a.slope; o.w = o.w > 0.f ? o.w : o.w * a.slope;
        if (a.residual) {
          const float4 rr = *reinterpret_cast<const float4*>(a.residual + (ov * (unsigned)a.res_cs + (unsigned)(a.res_coff + c)));
          o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        *reinterpret_cast<float4*>(a.out + (ov * (unsigned)a.out_cs + (unsigned)(a.out_coff + c))) = o;
      }
  }
}

}  // namespace pgrf

using namespace pgrf;

extern "C" {

size_t pgrf_conv3d_packed_floats(int Cout, int Cin) {
  const size_t cot = (Cout + 31) / 32, chunks = (Cin + 31) / 32;
  size_t n = cot * 27 * chunks * 1024 + cot * 32 + (((size_t)27 * Cin + 1 + 3) & ~(size_t)3);
  if (Cout == 1) n += chunks * 1024;          // pointwise block: 27 tap-vectors as the rows of one 32 x Cin matrix
  return n;
}

// layout of the packed block: [A operands | bias in accumulator order | linear copy [27][Cin] of output channel 0 | bias0]
int pgrf_pack_conv3d(const float* w_host, const float* bias_host, int Cout, int Cin, float* packed_host) {
  if (!w_host || !packed_host) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d: null pointer");
  if (Cout < 1 || Cin < 1) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d: bad channels");
  const int cot = (Cout + 31) / 32, chunks = (Cin + 31) / 32;
  const size_t a_floats = (size_t)cot * 27 * chunks * 1024;
  std::memset(packed_host, 0, sizeof(float) * pgrf_conv3d_packed_floats(Cout, Cin));
  for (int ct = 0; ct < cot; ++ct)
    for (int tap = 0; tap < 27; ++tap)
      for (int ch = 0; ch < chunks; ++ch)
        for (int s = 0; s < 16; ++s)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = 32 * ct + (lane & 31), ci = 32 * ch + 16 * (lane >> 5) + s;
            if (co < Cout && ci < Cin) packed_host[wpk_index(ct, tap, ch, chunks, s, lane)] = w_host[((size_t)co * Cin + ci) * 27 + tap];
          }
  float* B = packed_host + a_floats;
  for (int ct = 0; ct < cot; ++ct)
    for (int h = 0; h < 2; ++h)
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * ct + rho_(r, h);
        B[ct * 32 + h * 16 + r] = (bias_host && co < Cout) ? bias_host[co] : 0.f;
      }
  float* L = B + cot * 32;
  for (int tap = 0; tap < 27; ++tap)
    for (int ci = 0; ci < Cin; ++ci) L[tap * Cin + ci] = w_host[((size_t)0 * Cin + ci) * 27 + tap];
  L[27 * Cin] = bias_host ? bias_host[0] : 0.f;
  if (Cout == 1) {
    float* PW = L + (((size_t)27 * Cin + 1 + 3) & ~(size_t)3);
    for (int ch = 0; ch < chunks; ++ch)
      for (int s = 0; s < 16; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int tap = lane & 31, ci = 32 * ch + 16 * (lane >> 5) + s;
          if (tap < 27 && ci < Cin) PW[wpk_index(0, 0, ch, chunks, s, lane, 1)] = w_host[(size_t)ci * 27 + tap];
        }
  }
  return PGRF_OK;
}

static int conv3d_wrap_impl(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                            const float* packed, int has_bias, int B, int D, int H, int W, int stride, float lrelu_slope,
                            const float* residual, int res_cs, int res_coff, int wrap_w, int elu, int res_first, void* stream) {
  if (!in || !out || !packed) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_wrap: null pointer");
  if (stride != 1 && stride != 2) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wrap: stride %d", stride);
  if (B < 1 || D < 1 || H < 1 || W < 2) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_wrap: bad dims");
  if ((double)B * D * H * W * in_cs * 4.0 >= 4294967296.0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wrap: input buffer of >= 4 GiB (32-bit byte offsets)");
  if (((Cin & 3) == 0 || Cout != 1) && ((in_cs & 3) || (in_coff & 3)))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wrap: input channel stride/offset must be multiples of 4");
  if (Cout != 1 && ((Cin & 3) || (Cout & 3) || (out_cs & 3) || (out_coff & 3) || (res_cs & 3) || (res_coff & 3)))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wrap: channels/strides must be multiples of 4 (or Cout == 1)");
  ConvArgs a;
  a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = Cin; a.out = out; a.out_cs = out_cs; a.out_coff = out_coff;
  a.Cout = Cout; a.B = B; a.D = D; a.H = H; a.W = W; a.stride = stride; a.slope = lrelu_slope;
  a.Do = (D - 1) / stride + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
  a.chunks = (Cin + 31) / 32; a.cot_total = (Cout + 31) / 32;
  a.residual = residual; a.res_cs = res_cs; a.res_coff = res_coff; a.wtaps = 27; a.tapmajor = 0; a.wrap_w = wrap_w; a.elu = elu; a.res_first = res_first;
  const size_t a_floats = (size_t)a.cot_total * 27 * a.chunks * 1024;
  a.wpk = packed;
  a.bias = has_bias ? packed + a_floats : nullptr;
  const long long nvox = (long long)B * a.Do * a.Ho * a.Wo;
  if (Cout == 1) {
    const int lpv = ((Cin & 3) == 0 && Cin >= 32) ? 8 : 1;
    const float* wlin = packed + a_floats + a.cot_total * 32;
    if (Cin == 8 && stride == 1 && wrap_w && !getenv("PGRF_CONV_NO_C8") && !getenv("PGRF_CONV_NO_C8_BRICK") &&
        (double)B * D * H * W * in_cs * 4.0 < 2147483648.0 && (long long)B * ((D + 1) / 2) <= 65535 && (H + 15) / 16 <= 65535) {
      const unsigned in_bytes = (unsigned)(((size_t)B * D * H * W * in_cs - (size_t)in_coff) * sizeof(float));
      const dim3 grid((unsigned)((W + 31) / 32), (unsigned)((H + 15) / 16), (unsigned)(B * ((D + 1) / 2)));
      hipLaunchKernelGGL(conv3d_cout1_c8_brick_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, wlin, has_bias ? 1 : 0, in_bytes);
      PGRF_CHECK_LAUNCH("conv3d_cout1_c8_brick");
      return PGRF_OK;
    }
    if (Cin == 8 && stride == 1 && !getenv("PGRF_CONV_NO_C8")) {
      const long long threads = (long long)B * a.Do * ((a.Ho + kC8YB - 1) / kC8YB) * a.Wo;
      hipLaunchKernelGGL(conv3d_cout1_c8_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, wlin,
                         has_bias ? 1 : 0);
      PGRF_CHECK_LAUNCH("conv3d_cout1_c8");
      return PGRF_OK;
    }
    if (Cin == 1 && stride == 1 && wrap_w && (double)nvox * (in_cs > 1 ? in_cs : 1) < 4294967296.0 && !getenv("PGRF_CONV_NO_C1")) {
      hipLaunchKernelGGL(conv3d_cout1_c1_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, wlin, has_bias ? 1 : 0);
      PGRF_CHECK_LAUNCH("conv3d_cout1_c1");
      return PGRF_OK;
    }
    const long long threads = nvox * lpv;
    hipLaunchKernelGGL(conv3d_cout1_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, wlin,
                       has_bias ? 1 : 0, lpv);
    PGRF_CHECK_LAUNCH("conv3d_cout1");
    return PGRF_OK;
  }
  const long long tiles = (nvox + 31) / 32;
  // tile shape per wave: more voxel tiles per wave when the layer is large enough to still fill 256 CUs
  const int mt = (a.cot_total % 2 == 0) ? 2 : 1;
  const long long waves_nt4 = (tiles + 3) / 4 * (a.cot_total / mt);
  // one wave per SIMD with 4 voxel tiles beats two waves with 2 (each weight operand then feeds twice the MFMAs:
  // measured 256->256 at 32x64x16: 908 vs 1312 us), so the thresholds are "every SIMD gets a wave", not two
  int thr = 1024;
  if (const char* e = getenv("PGRF_CONV_THR")) thr = atoi(e);                                              // tuning knob
  int nt = waves_nt4 >= thr ? 4 : ((tiles + 1) / 2 * (a.cot_total / mt) >= thr ? 2 : 1);
  if (const char* e = getenv("PGRF_CONV_NT")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) nt = v; }   // tuning knob
  const long long waves_plain = (tiles + nt - 1) / nt * (a.cot_total / mt);
  bool split = (nt == 1) && waves_plain < 1536 && (long long)27 * a.chunks >= 16;
  if (const char* e = getenv("PGRF_CONV_SPLIT")) split = (nt == 1) && atoi(e) != 0;                       // tuning knob
  if (split) {
    // tuning knobs: one cout tile per workgroup (twice the workgroups, half the chain per wave) / the two-wave LEAN instantiation
    // 2-D layers (depth-1 volumes: the per-frame encoders and the MVS feature nets, a few thousand output pixels): ONE cout tile per workgroup --
    // twice the workgroups (two waves per SIMD instead of one) and half the dependent MFMA chain per wave: encoders 1.44 -> 1.22 ms,
    // init net 5.51 -> 5.10 ms (round 5).  3-D layers keep two tiles per workgroup (UNet2's low-resolution levels: 587 vs 675 us, round 3).
    int smt = (a.D == 1) ? 1 : mt;
    if (const char* e = getenv("PGRF_CONV_SPLIT_MT")) { const int v = atoi(e); if (v == 1) smt = 1; else if (v == 2) smt = mt; }
    bool slean = false;
    if (const char* e = getenv("PGRF_CONV_SPLIT_LEAN")) slean = atoi(e) != 0 && (Cin & 31) == 0 && a.wtaps == 27;
    // 2-D layers (round 5): the reduction of a (tile, cout tile) pair is shared by 4 waves only when the layer has fewer than 512 pairs; from 512
    // pairs on by TWO waves (two tiles per workgroup), from 2 048 on one wave owns a pair (the plain kernel with one cout tile per wave): a wave's
    // fixed cost (index set-up, first operands, the LDS sum and its barrier) is spread over 2-4 x the MFMAs.  tools/ab_mvs_env.sh, two rounds:
    // encoders 1.24 -> 1.19 ms, Equi feature net 1.04 -> 0.96 ms, MVS stage 5.17 -> 5.06 ms.  (Summation order per shape is fixed: results
    // are reproducible, and differ from the 4-way sums in the last bits only.)
    static const int sw2_min = [] { const char* e = getenv("PGRF_CONV_SPLIT_SW2_MIN"); return e ? atoi(e) : 512; }();      // tuning knob
    static const int sw1_min = [] { const char* e = getenv("PGRF_CONV_SPLIT_SW1_MIN"); return e ? atoi(e) : 2048; }();     // tuning knob
    if (a.D == 1 && smt == 1 && !slean && tiles * a.cot_total >= sw1_min) {
      dim3 g1((unsigned)((tiles + 3) / 4), (unsigned)a.cot_total);
      hipLaunchKernelGGL((conv3d_wrap_kernel<1, 1, false, false>), g1, dim3(256), 0, (hipStream_t)stream, a);
      PGRF_CHECK_LAUNCH("conv3d_wrap(one wave per tile)");
      return PGRF_OK;
    }
    if (a.D == 1 && smt == 1 && !slean && tiles * a.cot_total >= sw2_min) {
      dim3 g2((unsigned)((tiles + 1) / 2), (unsigned)a.cot_total);
      hipLaunchKernelGGL((conv3d_wrap_kernel<1, 1, true, false, 2>), g2, dim3(256), 0, (hipStream_t)stream, a);
      PGRF_CHECK_LAUNCH("conv3d_wrap(split 2)");
      return PGRF_OK;
    }
    dim3 sgrid((unsigned)tiles, (unsigned)(a.cot_total / smt));
    if (smt == 2 && slean) hipLaunchKernelGGL((conv3d_wrap_kernel<2, 1, true, true>), sgrid, dim3(256), 0, (hipStream_t)stream, a);
    else if (smt == 2) hipLaunchKernelGGL((conv3d_wrap_kernel<2, 1, true>), sgrid, dim3(256), 0, (hipStream_t)stream, a);
    else if (slean) hipLaunchKernelGGL((conv3d_wrap_kernel<1, 1, true, true>), sgrid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv3d_wrap_kernel<1, 1, true>), sgrid, dim3(256), 0, (hipStream_t)stream, a);
    PGRF_CHECK_LAUNCH("conv3d_wrap(split)");
    return PGRF_OK;
  }
  dim3 grid((unsigned)((tiles + 4 * nt - 1) / (4 * nt)), (unsigned)(a.cot_total / mt));
  bool lean = (Cin & 31) == 0 && a.wtaps == 27;
  if (const char* e = getenv("PGRF_CONV_LEAN")) lean = lean && atoi(e) != 0;                               // tuning knob
#define PGRF_LAUNCH(MT_, NT_)                                                                                               \
  do {                                                                                                                      \
    if (lean) hipLaunchKernelGGL((conv3d_wrap_kernel<MT_, NT_, false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);  \
    else hipLaunchKernelGGL((conv3d_wrap_kernel<MT_, NT_, false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);      \
  } while (0)
  if (mt == 2 && nt == 4) PGRF_LAUNCH(2, 4);
  else if (mt == 2 && nt == 2) PGRF_LAUNCH(2, 2);
  else if (mt == 2) PGRF_LAUNCH(2, 1);
  else if (nt == 4) PGRF_LAUNCH(1, 4);
  else if (nt == 2) PGRF_LAUNCH(1, 2);
  else PGRF_LAUNCH(1, 1);
#undef PGRF_LAUNCH
  PGRF_CHECK_LAUNCH("conv3d_wrap");
  return PGRF_OK;
}

int pgrf_conv3d_wrap(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                     const float* packed, int has_bias, int B, int D, int H, int W, int stride, float lrelu_slope,
                     const float* residual, int res_cs, int res_coff, void* stream) {
  return conv3d_wrap_impl(in, in_cs, in_coff, Cin, out, out_cs, out_coff, Cout, packed, has_bias, B, D, H, W, stride,
                          lrelu_slope, residual, res_cs, res_coff, 1, 0, 0, stream);
}

// ---- 2-D convolutions of the per-frame encoders: the same kernel on a depth-1 volume ------------------------------
// (the depth taps dz = +-1 fall into the zero padding for every lane and are skipped by the wave-level tap mask, so a
// 3x3 conv costs 9 taps; a 1x1 conv is packed as a 3x3 kernel whose only non-zero tap is the centre)
size_t pgrf_conv2d_packed_floats(int Cout, int Cin) { return pgrf_conv3d_packed_floats(Cout, Cin); }

int pgrf_pack_conv2d(const float* w_host, const float* bias_host, int Cout, int Cin, int ksize, float* packed_host) {
  if (!w_host || !packed_host) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv2d: null pointer");
  if (ksize != 1 && ksize != 3) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_pack_conv2d: kernel size %d (1 or 3)", ksize);
  if (Cout < 1 || Cin < 1) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv2d: bad channels");
  float* w3 = (float*)std::calloc((size_t)Cout * Cin * 27, sizeof(float));
  if (!w3) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv2d: out of host memory");
  for (size_t oc = 0; oc < (size_t)Cout * Cin; ++oc) {
    if (ksize == 3) for (int t = 0; t < 9; ++t) w3[oc * 27 + 9 + t] = w_host[oc * 9 + t];
    else w3[oc * 27 + 13] = w_host[oc];
  }
  const int rc = pgrf_pack_conv3d(w3, bias_host, Cout, Cin, packed_host);
  std::free(w3);
  return rc;
}

int pgrf_conv2d(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                const float* packed, int has_bias, int wrap_w, int N, int H, int W, int stride, int act, const float* residual,
                int res_cs, int res_coff, int residual_first, void* stream) {
  if (act < 0 || act > 3) return set_error(PGRF_ERR_ARG, "pgrf_conv2d: act %d (0 none, 1 ReLU, 2 LeakyReLU(0.01), 3 ELU)", act);
  const float slope = act == 0 ? 1.f : (act == 1 ? 0.f : 0.01f);
  return conv3d_wrap_impl(in, in_cs, in_coff, Cin, out, out_cs, out_coff, Cout, packed, has_bias, N, 1, H, W, stride, slope,
                          residual, res_cs, res_coff, wrap_w ? 1 : 0, act == 3 ? 1 : 0, residual_first ? 1 : 0, stream);
}

// ---- narrow layers (conv3d_small_kernel) -------------------------------------------------------------------------
// packed block: [steps][4][64 lanes][4] A operands | [32] bias in accumulator order
static int small_steps(int Cin, int ns) { return ((ns + 2) * 9 * Cin + 31) / 32; }

size_t pgrf_conv3d_small_packed_floats(int Cout, int Cin, int nshift) {
  (void)Cout;
  return (size_t)small_steps(Cin, nshift) * 1024 + 32;
}

int pgrf_pack_conv3d_small(const float* w_host, const float* bias_host, int Cout, int Cin, int nshift, float* packed_host) {
  if (!w_host || !packed_host) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d_small: null pointer");
  if (!(Cin == 8 || Cin == 16 || Cin == 32) || !(nshift == 1 || nshift == 2 || nshift == 4) || Cout < 4 || (Cout & 3) ||
      Cout > 32 / nshift)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_pack_conv3d_small: Cin %d, Cout %d, nshift %d", Cin, Cout, nshift);
  const int steps = small_steps(Cin, nshift), ntap = (nshift + 2) * 9, rps = 32 / nshift;
  std::memset(packed_host, 0, sizeof(float) * pgrf_conv3d_small_packed_floats(Cout, Cin, nshift));
  for (int st = 0; st < steps; ++st)
    for (int s = 0; s < 16; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int row = lane & 31, kk = lane >> 5;
        const int k = st * 32 + kk * 16 + s;             // reduction index = tap * Cin + channel
        const int tap = k / Cin, ci = k % Cin;
        if (tap >= (nshift + 2) * 9) continue;             // padding of the last pair of groups
        const int delta = row / rps, co = row % rps;
        if (tap >= ntap || co >= Cout) continue;
        const int kz = tap / 9 - 1 - delta;               // which of the 3 depth taps of output (delta, co) this plane is
        if (kz < -1 || kz > 1) continue;
        const int t27 = (kz + 1) * 9 + tap % 9;
        packed_host[(((size_t)st * 4 + (s >> 2)) * 64 + lane) * 4 + (s & 3)] = w_host[((size_t)co * Cin + ci) * 27 + t27];
      }
  float* Bv = packed_host + (size_t)steps * 1024;
  for (int h = 0; h < 2; ++h)
    for (int r = 0; r < 16; ++r) {
      const int co = rho_(r, h) % rps;
      Bv[h * 16 + r] = (bias_host && co < Cout) ? bias_host[co] : 0.f;
    }
  return PGRF_OK;
}

int pgrf_conv3d_small(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                      const float* packed, int has_bias, int nshift, int B, int D, int H, int W, int stride, float lrelu_slope,
                      const float* residual, int res_cs, int res_coff, void* stream) {
  if (!in || !out || !packed) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_small: null pointer");
  if (stride != 1 && stride != 2) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small: stride %d", stride);
  if (!(Cin == 8 || Cin == 16 || Cin == 32) || !(nshift == 1 || nshift == 2 || nshift == 4) || Cout < 4 || (Cout & 3) ||
      Cout > 32 / nshift)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small: Cin %d, Cout %d, nshift %d", Cin, Cout, nshift);
  if (B < 1 || D < 1 || H < 1 || W < 2) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_small: bad dims");
  if ((in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3) || (res_cs & 3) || (res_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small: channel strides/offsets must be multiples of 4");
  SmallArgs sa;
  ConvArgs& a = sa.c;
  a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = Cin; a.out = out; a.out_cs = out_cs; a.out_coff = out_coff;
  a.Cout = Cout; a.B = B; a.D = D; a.H = H; a.W = W; a.stride = stride; a.slope = lrelu_slope;
  a.Do = (D - 1) / stride + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
  a.chunks = 1; a.cot_total = 1; a.residual = residual; a.res_cs = res_cs; a.res_coff = res_coff; a.wtaps = 27; a.tapmajor = 0; a.wrap_w = 1; a.elu = 0; a.res_first = 0;
  if (nshift > 1 && (stride != 1 || a.Do % nshift))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small: nshift %d needs stride 1 and D %% nshift == 0", nshift);
  sa.ns = nshift; sa.nsteps = small_steps(Cin, nshift);
  a.wpk = packed;
  a.bias = has_bias ? packed + (size_t)sa.nsteps * 1024 : nullptr;
  const long long nbase = (long long)B * (a.Do / nshift) * a.Ho * a.Wo;
  const long long tiles = (nbase + 31) / 32;
  int sthr = 256;                                   // workgroups (4 waves each) wanted before a wave takes more tiles
  if (const char* e = getenv("PGRF_CONV_STHR")) sthr = atoi(e);                                            // tuning knob
  const int nt = tiles >= 16LL * sthr ? 4 : (tiles >= 8LL * sthr ? 2 : 1);
  dim3 grid((unsigned)((tiles + 4 * nt - 1) / (4 * nt)));
#define PGRF_LS(CIN_, NS_)                                                                                         \
  do {                                                                                                             \
    if (nt == 4) hipLaunchKernelGGL((conv3d_small_kernel<CIN_, NS_, 4>), grid, dim3(256), 0, (hipStream_t)stream, sa);  \
    else if (nt == 2) hipLaunchKernelGGL((conv3d_small_kernel<CIN_, NS_, 2>), grid, dim3(256), 0, (hipStream_t)stream, sa); \
    else hipLaunchKernelGGL((conv3d_small_kernel<CIN_, NS_, 1>), grid, dim3(256), 0, (hipStream_t)stream, sa);     \
  } while (0)
#define PGRF_LS_NS(CIN_)                       \
  do {                                         \
    if (nshift == 4) PGRF_LS(CIN_, 4);         \
    else if (nshift == 2) PGRF_LS(CIN_, 2);    \
    else PGRF_LS(CIN_, 1);                     \
  } while (0)
  if (Cin == 8) PGRF_LS_NS(8);
  else if (Cin == 16) PGRF_LS_NS(16);
  else PGRF_LS_NS(32);
#undef PGRF_LS_NS
#undef PGRF_LS
  PGRF_CHECK_LAUNCH("conv3d_small");
  return PGRF_OK;
}

// ---- narrow layers on the 16-row tile (conv3d_small16_kernel) ------------------------------------------------------
// packed block: [groups][64 lanes][4] A operands | [16] bias by row
static bool small16_ok(int Cout, int Cin, int ns) {
  if (ns == 2) return Cout == 8 && (Cin == 8 || Cin == 16 || Cin == 32);
  if (ns != 1) return false;
  if (Cout == 16 && (Cin == 8 || Cin == 16 || Cin == 32)) return true;
  return Cout >= 16 && Cout % 16 == 0 && Cin >= 16 && Cin % 16 == 0 && Cin <= 1024;       // LDS-brick kernel only (stride 1)
}
static int small16_groups(int Cin, int ns) { return (((ns + 2) * 9 * Cin + 15) / 16 + 1) & ~1; }
static int small16_cgroups(int Cout, int ns) { return ns == 1 ? Cout / 16 : 1; }
#define PGRF_SMALL16_SHAPES "Cout 8 with nshift 2 from Cin 8 | 16 | 32, or Cout % 16 == 0 with nshift 1 from Cin 8 | a multiple of 16"

size_t pgrf_conv3d_small16_packed_floats(int Cout, int Cin, int nshift) {
  if (!small16_ok(Cout, Cin, nshift)) return 0;
  return (size_t)small16_cgroups(Cout, nshift) * small16_groups(Cin, nshift) * 256 + (size_t)(Cout > 16 ? Cout : 16);
}

int pgrf_pack_conv3d_small16(const float* w_host, const float* bias_host, int Cout, int Cin, int nshift, float* packed_host) {
  if (!w_host || !packed_host) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d_small16: null pointer");
  if (!small16_ok(Cout, Cin, nshift))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_pack_conv3d_small16: Cin %d, Cout %d, nshift %d (" PGRF_SMALL16_SHAPES ")", Cin, Cout, nshift);
  const int groups = small16_groups(Cin, nshift), rps = 16 / nshift, ncg = small16_cgroups(Cout, nshift);
  std::memset(packed_host, 0, sizeof(float) * pgrf_conv3d_small16_packed_floats(Cout, Cin, nshift));
  for (int cg = 0; cg < ncg; ++cg)
    for (int G = 0; G < groups; ++G)
      for (int lane = 0; lane < 64; ++lane)
        for (int m = 0; m < 4; ++m) {
          const int row = lane & 15, g = lane >> 4;
          // reduction order: Cin >= 16 -> group G = chunk * ntap + tap carries channels 16 chunk + 4 g + m; Cin = 8 -> k = tap * 8 + c
          const int ntap = (nshift + 2) * 9;
          const int k = G * 16 + 4 * g + m;
          const int tap = Cin >= 16 ? G % ntap : k / Cin;
          const int ci = Cin >= 16 ? 16 * (G / ntap) + 4 * g + m : k % Cin;
          if (Cin >= 16 ? G >= ntap * (Cin / 16) : tap >= ntap) continue;          // padding of the last pair of groups
          const int delta = row / rps, co = cg * 16 + row % rps;
          const int kz = tap / 9 - 1 - delta;                 // which of the 3 depth taps of output (delta, co) this plane is
          if (kz < -1 || kz > 1) continue;
          const int t27 = (kz + 1) * 9 + tap % 9;
          packed_host[(((size_t)cg * groups + G) * 64 + lane) * 4 + m] = w_host[((size_t)co * Cin + ci) * 27 + t27];
        }
  float* bv = packed_host + (size_t)ncg * groups * 256;
  for (int r = 0; r < (Cout > 16 ? Cout : 16); ++r) bv[r] = bias_host ? bias_host[nshift == 2 ? r % rps : r] : 0.f;
  return PGRF_OK;
}

int pgrf_conv3d_small16(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                        const float* packed, int has_bias, int nshift, int B, int D, int H, int W, int stride, float lrelu_slope,
                        const float* residual, int res_cs, int res_coff, void* stream) {
  if (!in || !out || !packed) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_small16: null pointer");
  if (stride != 1 && stride != 2) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: stride %d", stride);
  if (!small16_ok(Cout, Cin, nshift))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: Cin %d, Cout %d, nshift %d (" PGRF_SMALL16_SHAPES ")", Cin, Cout, nshift);
  if (B < 1 || D < 1 || H < 1 || W < 2) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_small16: bad dims");
  if ((in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3) || (res_cs & 3) || (res_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: channel strides/offsets must be multiples of 4");
  SmallArgs sa;
  std::memset(&sa, 0, sizeof(sa));
  ConvArgs& a = sa.c;
  a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = Cin; a.out = out; a.out_cs = out_cs; a.out_coff = out_coff;
  a.Cout = Cout; a.B = B; a.D = D; a.H = H; a.W = W; a.stride = stride; a.slope = lrelu_slope;
  a.Do = (D - 1) / stride + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
  a.residual = residual; a.res_cs = res_cs; a.res_coff = res_coff; a.wrap_w = 1;
  if (nshift > 1 && (stride != 1 || a.Do % nshift))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: nshift %d needs stride 1 and D %% nshift == 0", nshift);
  const int mcs = out_cs > res_cs ? out_cs : res_cs;
  if ((double)B * D * H * W * in_cs >= 2147483648.0 || (double)H * W * in_cs >= 8388608.0 || (double)B * a.Do * a.Ho * a.Wo * (mcs > 1 ? mcs : 1) >= 4294967296.0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: tensor too large (32-bit element offsets, 24-bit plane stride)");
  sa.ns = nshift; sa.nsteps = small16_groups(Cin, nshift);
  const int ncg = small16_cgroups(Cout, nshift);
  sa.grp_floats = sa.nsteps * 256;
  a.wpk = packed;
  a.bias = has_bias ? packed + (size_t)ncg * sa.grp_floats : nullptr;
  const bool no_brick = getenv("PGRF_CONV16_NO_BRICK") != nullptr && ncg == 1 && Cin <= 32;  // A/B switch: B operand from registers
  if (!(Cin >= 16 && stride == 1) && (ncg > 1 || Cin > 32))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: Cout %d / Cin %d need stride 1 (LDS-brick kernel)", Cout, Cin);
  if (Cin >= 16 && stride == 1 && !no_brick) {
    // B operand staged through LDS: bricks of 2 x 4 x 32 output voxels
    sa.in_bytes = (unsigned)(((size_t)B * D * H * W * in_cs - (size_t)in_coff) * sizeof(float));          // < 2^31 elements checked above: fits
    if (sa.in_bytes >= 0x80000000u) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: input beyond 2 GB");
    if ((long long)B * ((D + 1) / 2) > 65535 || (H + 3) / 4 > 65535) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: grid too large");
    sa.nbx = (W + 31) / 32; sa.nch = Cin / 16;
    dim3 grid((unsigned)(sa.nbx * ncg), (unsigned)((H + 3) / 4), (unsigned)(B * ((D + 1) / 2)));
    // small volumes (at most two rounds of 768 resident four-row bricks): two-row bricks, four workgroups per CU, half the chain per wave
    bool yb2 = nshift == 1 && (long long)grid.x * grid.y * grid.z <= 1536;
    if (const char* e = getenv("PGRF_CONV16_YB")) yb2 = nshift == 1 && atoi(e) == 2;          // tuning knob
    if (yb2) {
      if ((H + 1) / 2 > 65535) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_small16: grid too large");
      grid.y = (unsigned)((H + 1) / 2);
      if (Cin == 16) hipLaunchKernelGGL((conv3d_brick16_kernel<16, 1, 2>), grid, dim3(256), 0, (hipStream_t)stream, sa);
      else if (Cin == 32) hipLaunchKernelGGL((conv3d_brick16_kernel<32, 1, 2>), grid, dim3(256), 0, (hipStream_t)stream, sa);
      else hipLaunchKernelGGL((conv3d_brick16_kernel<0, 1, 2>), grid, dim3(256), 0, (hipStream_t)stream, sa);
    } else if (nshift == 2) {
      if (Cin == 16) hipLaunchKernelGGL((conv3d_brick16_kernel<16, 2>), grid, dim3(256), 0, (hipStream_t)stream, sa);
      else hipLaunchKernelGGL((conv3d_brick16_kernel<32, 2>), grid, dim3(256), 0, (hipStream_t)stream, sa);
    } else {
      if (Cin == 16) hipLaunchKernelGGL((conv3d_brick16_kernel<16, 1>), grid, dim3(256), 0, (hipStream_t)stream, sa);
      else if (Cin == 32) hipLaunchKernelGGL((conv3d_brick16_kernel<32, 1>), grid, dim3(256), 0, (hipStream_t)stream, sa);
      else hipLaunchKernelGGL((conv3d_brick16_kernel<0, 1>), grid, dim3(256), 0, (hipStream_t)stream, sa);
    }
    PGRF_CHECK_LAUNCH("conv3d_brick16");
    return PGRF_OK;
  }
  const long long nbase = (long long)B * (a.Do / nshift) * a.Ho * a.Wo;
  const long long tiles = (nbase + 15) / 16;
  int nt = tiles >= 8LL * 4096 ? 8 : (tiles >= 4LL * 4096 ? 4 : 2);
  if (const char* e = getenv("PGRF_CONV16_NT")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8) nt = v; }       // tuning knob
  dim3 grid((unsigned)((tiles + 4 * nt - 1) / (4 * nt)));
#define PGRF_L16(CIN_, NS_)                                                                                                       \
  do {                                                                                                                           \
    if (nt == 8) hipLaunchKernelGGL((conv3d_small16_kernel<CIN_, NS_, 8>), grid, dim3(256), 0, (hipStream_t)stream, sa);          \
    else if (nt == 4) hipLaunchKernelGGL((conv3d_small16_kernel<CIN_, NS_, 4>), grid, dim3(256), 0, (hipStream_t)stream, sa);     \
    else hipLaunchKernelGGL((conv3d_small16_kernel<CIN_, NS_, 2>), grid, dim3(256), 0, (hipStream_t)stream, sa);                  \
  } while (0)
  if (nshift == 2) { if (Cin == 8) PGRF_L16(8, 2); else if (Cin == 16) PGRF_L16(16, 2); else PGRF_L16(32, 2); }
  else { if (Cin == 8) PGRF_L16(8, 1); else if (Cin == 16) PGRF_L16(16, 1); else PGRF_L16(32, 1); }
#undef PGRF_L16
  PGRF_CHECK_LAUNCH("conv3d_small16");
  return PGRF_OK;
}

// ---- up-sampling fused into the following convolution (conv3d_upfused_kernel) ----------------------------------------
// packed block: [27 classes][Cout/4 tiles][steps][4][64 lanes][4] A operands | [Cout] bias
static int upfused_steps(int Cin) { return ((27 * Cin + 31) / 32 + 1) / 2 * 2; }     // even: the kernel walks the steps in pairs
static bool upfused_ok(int Cout, int Cin) { return (Cin == 16 || Cin == 32) && (Cout == 8 || Cout == 16); }

size_t pgrf_conv3d_upfused_packed_floats(int Cout, int Cin) {
  if (!upfused_ok(Cout, Cin)) return 0;
  return (size_t)27 * (Cout / 4) * upfused_steps(Cin) * 1024 + (size_t)Cout;
}

// interpolation-folded 1-D operator of one axis: E[cls][par][nb + 1][tap] = weight of low-res neighbour nb of voxel i (of class
// cls) in up-sampled tap (tap - 1) of output 2i + par.  F.interpolate(scale_factor=2, mode='trilinear', align_corners=False):
// src = max((j + 0.5) / 2 - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, n - 1).  Padding of the UP-SAMPLED axis: zeros or circular.
static void upfused_axis(int n, bool circular, double E[3][2][3][3]) {
  std::memset(E, 0, sizeof(double) * 54);
  const int N = 2 * n;
  for (int cls = 0; cls < 3; ++cls) {
    const int i = cls == 0 ? 0 : (cls == 1 ? 1 : n - 1);
    if (cls == 1 && n < 3) continue;                       // no interior voxel
    for (int par = 0; par < 2; ++par)
      for (int tap = 0; tap < 3; ++tap) {
        int jx = 2 * i + par + tap - 1;
        if (jx < 0 || jx >= N) {
          if (!circular) continue;
          jx = (jx + N) % N;
        }
        double src = (jx + 0.5) / 2.0 - 0.5;
        if (src < 0) src = 0;
        const int i0 = (int)std::floor(src), i1 = i0 + 1 < n ? i0 + 1 : n - 1;
        const double lam = src - i0;
        const int ms[2] = {i0, i1};
        const double ws[2] = {1.0 - lam, lam};
        for (int e = 0; e < 2; ++e) {
          int d = ms[e] - i;
          if (circular) { d = ((d % n) + n) % n; if (d == n - 1 && n > 2) d = -1; }       // n == 2: the other voxel counts as +1
          if (d < -1 || d > 1) continue;                   // cannot happen for the taps above; guards degenerate sizes
          E[cls][par][d + 1][tap] += ws[e];
        }
      }
  }
}

int pgrf_pack_conv3d_upfused(const float* w_host, const float* bias_host, int Cout, int Cin, int D, int H, int W, float* packed_host) {
  if (!w_host || !packed_host) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d_upfused: null pointer");
  if (!upfused_ok(Cout, Cin)) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_pack_conv3d_upfused: Cin %d (16 | 32), Cout %d (8 | 16)", Cin, Cout);
  if (D < 2 || H < 2 || W < 2) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_pack_conv3d_upfused: low-res dims must be >= 2");
  const int steps = upfused_steps(Cin), MT = Cout / 4;
  const size_t combo = (size_t)MT * steps * 1024;
  std::memset(packed_host, 0, sizeof(float) * pgrf_conv3d_upfused_packed_floats(Cout, Cin));
  double Ez[3][2][3][3], Ey[3][2][3][3], Ex[3][2][3][3];
  upfused_axis(D, false, Ez); upfused_axis(H, false, Ey); upfused_axis(W, true, Ex);
  std::vector<double> K((size_t)8 * Cout * 27 * Cin);
  for (int cz = 0; cz < 3; ++cz)
    for (int cy = 0; cy < 3; ++cy)
      for (int cx = 0; cx < 3; ++cx) {
        std::fill(K.begin(), K.end(), 0.0);
        for (int par = 0; par < 8; ++par) {
          const int pz = par >> 2, py = (par >> 1) & 1, px = par & 1;
          for (int nb = 0; nb < 27; ++nb) {
            const int bz = nb / 9, by = (nb / 3) % 3, bx = nb % 3;
            for (int tp = 0; tp < 27; ++tp) {
              const double f = Ez[cz][pz][bz][tp / 9] * Ey[cy][py][by][(tp / 3) % 3] * Ex[cx][px][bx][tp % 3];
              if (f == 0.0) continue;
              for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                  K[(((size_t)par * Cout + co) * 27 + nb) * Cin + ci] += f * (double)w_host[((size_t)co * Cin + ci) * 27 + tp];
            }
          }
        }
        float* blk = packed_host + (size_t)((cz * 3 + cy) * 3 + cx) * combo;
        for (int m = 0; m < MT; ++m)
          for (int st = 0; st < steps; ++st)
            for (int s = 0; s < 16; ++s)
              for (int lane = 0; lane < 64; ++lane) {
                const int row = m * 32 + (lane & 31), kk = lane >> 5;
                const int k = st * 32 + kk * 16 + s;
                const int tap = k / Cin, ci = k % Cin;
                if (tap >= 27) continue;
                blk[((((size_t)m * steps + st) * 4 + (s >> 2)) * 64 + lane) * 4 + (s & 3)] = (float)K[((size_t)row * 27 + tap) * Cin + ci];
              }
      }
  float* bv = packed_host + (size_t)27 * combo;
  for (int co = 0; co < Cout; ++co) bv[co] = bias_host ? bias_host[co] : 0.f;
  return PGRF_OK;
}

int pgrf_conv3d_upfused(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                        const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                        int res_cs, int res_coff, void* stream) {
  if (!in || !out || !packed) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_upfused: null pointer");
  if (!upfused_ok(Cout, Cin)) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_upfused: Cin %d (16 | 32), Cout %d (8 | 16)", Cin, Cout);
  if (B < 1 || D < 2 || H < 2 || W < 2) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_upfused: low-res dims must be >= 2");
  if ((in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3) || (res_cs & 3) || (res_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_upfused: channel strides/offsets must be multiples of 4");
  if ((double)B * D * H * W * in_cs * 4.0 >= 2147483648.0 || (double)B * D * H * W * 8.0 >= 2147483648.0 || (double)H * W * in_cs >= 8388608.0 ||
      (double)B * D * H * W * 8.0 * (out_cs > res_cs ? out_cs : res_cs) >= 4294967296.0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_upfused: tensor too large (32-bit element offsets, 24-bit plane stride)");
  UpArgs ua;
  std::memset(&ua, 0, sizeof(ua));
  ConvArgs& a = ua.c;
  a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = Cin; a.out = out; a.out_cs = out_cs; a.out_coff = out_coff; a.Cout = Cout;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W; a.stride = 1; a.slope = lrelu_slope;
  a.residual = residual; a.res_cs = res_cs; a.res_coff = res_coff; a.wrap_w = 1;
  const int steps = upfused_steps(Cin), MT = Cout / 4;
  ua.nsteps = steps; ua.combo_stride = MT * steps * 1024;
  a.wpk = packed;
  ua.bias4 = has_bias ? packed + (size_t)27 * ua.combo_stride : nullptr;
  int NT = MT == 2 ? 2 : 1;
  if (const char* e = getenv("PGRF_UPF_NT")) { const int v = atoi(e); if (MT == 2 && (v == 1 || v == 2 || v == 4)) NT = v; }      // tuning knob
  const int dims[3] = {D, H, W};
  int total = 0;
  for (int c = 0; c < 27; ++c) {
    const int cl[3] = {c / 9, (c / 3) % 3, c % 3};
    long long vox = B;
    for (int ax = 0; ax < 3; ++ax) {
      const int n = dims[ax];
      ua.lo[c][ax] = cl[ax] == 0 ? 0 : (cl[ax] == 1 ? 1 : n - 1);
      ua.n[c][ax] = cl[ax] == 1 ? (n > 2 ? n - 2 : 0) : 1;
      vox *= ua.n[c][ax];
    }
    ua.item_start[c] = total;
    const long long tiles = (vox + 31) / 32;
    total += (int)((tiles + NT - 1) / NT);
  }
  ua.item_start[27] = total;
  for (int c = 0; c < 27; ++c)       // empty boxes: extent 1 keeps the index arithmetic away from a division by zero
    for (int ax = 0; ax < 3; ++ax) if (ua.n[c][ax] == 0) { ua.n[c][ax] = 1; }
  // (an empty box has zero items, so no wave ever decodes a voxel of it)
  dim3 grid((unsigned)((total + 3) / 4));
  if (Cin == 16 && MT == 2 && NT == 4) hipLaunchKernelGGL((conv3d_upfused_kernel<16, 2, 4>), grid, dim3(256), 0, (hipStream_t)stream, ua);
  else if (Cin == 16 && MT == 2 && NT == 1) hipLaunchKernelGGL((conv3d_upfused_kernel<16, 2, 1>), grid, dim3(256), 0, (hipStream_t)stream, ua);
  else if (Cin == 16 && MT == 2) hipLaunchKernelGGL((conv3d_upfused_kernel<16, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, ua);
  else if (Cin == 32 && MT == 2) hipLaunchKernelGGL((conv3d_upfused_kernel<32, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, ua);
  else if (Cin == 16 && MT == 4) hipLaunchKernelGGL((conv3d_upfused_kernel<16, 4, 1>), grid, dim3(256), 0, (hipStream_t)stream, ua);
  else hipLaunchKernelGGL((conv3d_upfused_kernel<32, 4, 1>), grid, dim3(256), 0, (hipStream_t)stream, ua);
  PGRF_CHECK_LAUNCH("conv3d_upfused");
  return PGRF_OK;
}

// ---- the same step on 16-row tiles with the low-res brick in LDS (conv3d_upbrick_kernel) ----
// packed block: [class 9 = 3 zc + yc][parity set 4 = 2 py + px][tap 27][64 lanes][4] A operands | [16] bias by row
static const size_t kUpBrickA = (size_t)9 * 4 * 27 * 256;
static bool upbrick_ok(int Cout, int Cin) { return Cout == 8 && Cin == 16; }

size_t pgrf_conv3d_upbrick_packed_floats(int Cout, int Cin) {
  if (!upbrick_ok(Cout, Cin)) return 0;
  return kUpBrickA + 16;
}

int pgrf_pack_conv3d_upbrick(const float* w_host, const float* bias_host, int Cout, int Cin, int D, int H, int W, float* packed_host) {
  if (!w_host || !packed_host) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d_upbrick: null pointer");
  if (!upbrick_ok(Cout, Cin)) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_pack_conv3d_upbrick: Cin %d (16), Cout %d (8)", Cin, Cout);
  if (D < 2 || H < 2 || W < 33) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_pack_conv3d_upbrick: low-res D, H >= 2 and W >= 33");
  std::memset(packed_host, 0, sizeof(float) * (kUpBrickA + 16));
  double Ez[3][2][3][3], Ey[3][2][3][3], Ex[3][2][3][3];
  upfused_axis(D, false, Ez); upfused_axis(H, false, Ey); upfused_axis(W, true, Ex);
  for (int cz = 0; cz < 3; ++cz)
    for (int cy = 0; cy < 3; ++cy)
      for (int set = 0; set < 4; ++set)
        for (int nb = 0; nb < 27; ++nb) {
          const int py = set >> 1, px = set & 1, bz = nb / 9, by = (nb / 3) % 3, bx = nb % 3;
          float* blk = packed_host + ((((size_t)(cz * 3 + cy) * 4 + set) * 27 + nb) * 256);
          for (int lane = 0; lane < 64; ++lane)
            for (int m = 0; m < 4; ++m) {
              const int row = lane & 15, pz = row >> 3, co = row & 7, ci = 4 * (lane >> 4) + m;
              double k = 0.0;
              for (int tp = 0; tp < 27; ++tp) {
                const double f = Ez[cz][pz][bz][tp / 9] * Ey[cy][py][by][(tp / 3) % 3] * Ex[1][px][bx][tp % 3];      // x: interior class
                if (f != 0.0) k += f * (double)w_host[((size_t)co * Cin + ci) * 27 + tp];
              }
              blk[lane * 4 + m] = (float)k;
            }
        }
  float* bv = packed_host + kUpBrickA;
  for (int r = 0; r < 16; ++r) bv[r] = bias_host ? bias_host[r & 7] : 0.f;
  return PGRF_OK;
}

int pgrf_conv3d_upbrick(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                        const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                        int res_cs, int res_coff, void* stream) {
  if (!in || !out || !packed) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_upbrick: null pointer");
  if (!upbrick_ok(Cout, Cin)) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_upbrick: Cin %d (16), Cout %d (8)", Cin, Cout);
  // (W >= 33: the first brick, columns W - 1, 0 .. 30, then holds each column once)
  if (B < 1 || D < 2 || H < 2 || W < 33) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_upbrick: low-res D, H >= 2 and W >= 33");
  if ((in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3) || (res_cs & 3) || (res_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_upbrick: channel strides/offsets must be multiples of 4");
  if ((double)B * D * H * W * in_cs * 4.0 >= 2147483648.0 || (double)B * D * H * W * 8.0 * (out_cs > res_cs ? out_cs : res_cs) >= 4294967296.0 ||
      (long long)B * D > 65535 || (H + 3) / 4 > 65535)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_upbrick: tensor too large (32-bit offsets)");
  UpBrickArgs ua;
  std::memset(&ua, 0, sizeof(ua));
  ConvArgs& a = ua.c;
  a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = Cin; a.out = out; a.out_cs = out_cs; a.out_coff = out_coff; a.Cout = Cout;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W; a.stride = 1; a.slope = lrelu_slope;
  a.residual = residual; a.res_cs = res_cs; a.res_coff = res_coff; a.wrap_w = 1;
  a.wpk = packed;
  a.bias = has_bias ? packed + kUpBrickA : nullptr;
  ua.in_bytes = (unsigned)(((size_t)B * D * H * W * in_cs - (size_t)in_coff) * sizeof(float));
  const dim3 grid((unsigned)((W + 31) / 32), (unsigned)((H + 3) / 4), (unsigned)(B * D));
  hipLaunchKernelGGL(conv3d_upbrick_kernel, grid, dim3(256), 0, (hipStream_t)stream, ua);
  PGRF_CHECK_LAUNCH("conv3d_upbrick");
  return PGRF_OK;
}

// Pointwise projection of a 64-channel volume onto the 27 tap vectors, streaming form (round 4): the general kernel in its pointwise mode
// reads 537 MB and writes 226 MB in 244 us at 64x128x256 (3.1 TB/s).  Here the 27 x 64 weights live in registers for the whole launch (A
// operands of `v_mfma_f32_32x32x2_f32`: row = tap), a persistent wave takes 32 consecutive voxels per step -- lane (n, half) loads its voxel's
// channels 8q + 4 half .. with eight `dwordx4`, the next step's loads in flight under this step's 32 MFMAs -- and the accumulator layout
// (lane = voxel, register = tap) makes every plane store 32 consecutive voxels per lane half.  Rows 27..31 of the plane block are not written.
typedef float f32x16p __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256, 3) cout1_project64_kernel(const float* __restrict__ in, int in_cs, int in_coff, const float* __restrict__ wlin,
                                                              float* __restrict__ y, long long nvox) {
  const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * 256) >> 6;
  const long long ntiles = (nvox + 31) / 32;
  float4 a[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    a[q] = n < 27 ? *reinterpret_cast<const float4*>(wlin + n * 64 + 8 * q + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](long long tile, float4 (&x)[8]) {
    long long v = tile * 32 + n;
    v = v < nvox ? v : nvox - 1;
    const float* p = in + v * in_cs + in_coff + 4 * half;
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const float4*>(p + 8 * q);
  };
  float4 xa[8], xb[8];
  auto step = [&](long long tile, float4 (&x)[8]) {
    f32x16p acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, x[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, x[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, x[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, x[q].w, acc, 0, 0, 0);
    }
    const long long v = tile * 32 + n;
    if (v < nvox) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tap = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (tap < 27) y[(long long)tap * nvox + v] = acc[r];
      }
    }
  };
  long long t = wave;
  if (t < ntiles) fetch(t, xa);
  while (t < ntiles) {
    const long long t1 = t + nwaves;
    if (t1 < ntiles) fetch(t1, xb);
    step(t, xa);
    if (t1 >= ntiles) break;
    const long long t2 = t1 + nwaves;
    if (t2 < ntiles) fetch(t2, xa);
    step(t1, xb);
    t = t2;
  }
}

// Cout == 1 convolution in two passes (fast path for wide inputs): y[B,D,H,W,32] = per-voxel projections onto the 27
// tap vectors (MFMA, pointwise), then a 27-tap gather.  y is caller-owned scratch of B*D*H*W*32 floats.
int pgrf_conv3d_cout1_project(const float* in, int in_cs, int in_coff, int Cin, const float* packed, float* y, int B,
                              int D, int H, int W, void* stream) {
  if (!in || !packed || !y) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_cout1_project: null pointer");
  if ((Cin & 3) || (in_cs & 3) || (in_coff & 3)) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_cout1_project: channels/strides must be multiples of 4");
  if ((double)B * D * H * W * in_cs * 4.0 >= 4294967296.0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_cout1_project: input buffer of >= 4 GiB (32-bit byte offsets)");
  ConvArgs a;
  a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = Cin; a.out = y; a.out_cs = 32; a.out_coff = 0; a.Cout = 32;
  a.B = B; a.D = D; a.H = H; a.W = W; a.stride = 1; a.slope = 1.f; a.Do = D; a.Ho = H; a.Wo = W;
  a.chunks = (Cin + 31) / 32; a.cot_total = 1; a.residual = nullptr; a.res_cs = 0; a.res_coff = 0; a.wtaps = 1; a.tapmajor = 1; a.wrap_w = 1; a.elu = 0; a.res_first = 0;
  const size_t a_floats = (size_t)27 * a.chunks * 1024;
  static const bool no_stream = [] { const char* e = getenv("PGRF_COUT1_PROJECT_STREAM"); return e && e[0] == '0'; }();
  if (Cin == 64 && !no_stream) {
    static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const long long nvox = (long long)B * D * H * W, tiles = (nvox + 31) / 32;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 3LL * ncu) blocks = 3LL * ncu;             // persistent: three 4-wave workgroups per CU (~130 registers)
    hipLaunchKernelGGL(cout1_project64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, in_cs, in_coff, packed + a_floats + 32, y, nvox);
    PGRF_CHECK_LAUNCH("conv3d_cout1_project (streaming)");
    return PGRF_OK;
  }
  a.wpk = packed + a_floats + 32 + (((size_t)27 * Cin + 1 + 3) & ~(size_t)3);
  a.bias = nullptr;
  const long long tiles = ((long long)B * D * H * W + 31) / 32;
  const int nt = tiles >= 8192 ? 4 : (tiles >= 4096 ? 2 : 1);
  dim3 grid((unsigned)((tiles + 4 * nt - 1) / (4 * nt)), 1);
  if (nt == 4) hipLaunchKernelGGL((conv3d_wrap_kernel<1, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (nt == 2) hipLaunchKernelGGL((conv3d_wrap_kernel<1, 2, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((conv3d_wrap_kernel<1, 1, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  PGRF_CHECK_LAUNCH("conv3d_cout1_project");
  return PGRF_OK;
}

int pgrf_conv3d_cout1_gather(const float* y, const float* packed, int Cin, int has_bias, float* out, int out_cs, int out_coff,
                             int B, int D, int H, int W, float lrelu_slope, const float* residual, int res_cs, int res_coff,
                             void* stream) {
  if (!y || !packed || !out) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_cout1_gather: null pointer");
  const int chunks = (Cin + 31) / 32;
  const float* bias0 = has_bias ? packed + (size_t)27 * chunks * 1024 + 32 + (size_t)27 * Cin : nullptr;
  const long long nvox = (long long)B * D * H * W;
  hipLaunchKernelGGL(cout1_gather_kernel, dim3(grid_for(nvox)), dim3(256), 0, (hipStream_t)stream, y, bias0, out, out_cs, out_coff, B, D,
                     H, W, lrelu_slope, residual, res_cs, res_coff);
  PGRF_CHECK_LAUNCH("conv3d_cout1_gather");
  return PGRF_OK;
}

int pgrf_conv3d_cout1_gather_up(const float* y_hi, const float* y_lo, const float* packed_hi, int Cin_hi, int has_bias, float* out, int out_cs,
                                int out_coff, int B, int D, int H, int W, float lrelu_slope, const float* residual, int res_cs, int res_coff,
                                void* stream) {
  if (!y_hi || !y_lo || !packed_hi || !out) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_cout1_gather_up: null pointer");
  if ((D & 1) || (H & 1) || (W & 1) || D < 2 || H < 2 || W < 2) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_cout1_gather_up: dims must be even");
  const int chunks = (Cin_hi + 31) / 32;
  const float* bias0 = has_bias ? packed_hi + (size_t)27 * chunks * 1024 + 32 + (size_t)27 * Cin_hi : nullptr;
  const long long nvox = (long long)B * D * H * W;
  // (round 4, measured and removed: the same gather with the 27 low-resolution planes of a 2 x 2 x 64 output block staged in LDS -- 276-319 us
  //  against 181 / 123 us: the kernel is bound by its 243 loads and ~700 VALU instructions per voxel whichever memory they come from)
  hipLaunchKernelGGL(cout1_gather_up_kernel, dim3(grid_for(nvox)), dim3(256), 0, (hipStream_t)stream, y_hi, y_lo, bias0, out, out_cs, out_coff, B, D,
                     H, W, lrelu_slope, residual, res_cs, res_coff);
  PGRF_CHECK_LAUNCH("conv3d_cout1_gather_up");
  return PGRF_OK;
}

int pgrf_avgpool3d_2(const float* in, int in_cs, int in_coff, float* out, int out_cs, int out_coff, int B, int D, int H,
                     int W, int C, void* stream) {
  if (!in || !out) return set_error(PGRF_ERR_ARG, "pgrf_avgpool3d_2: null pointer");
  if ((C & 3) || (in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3) || (D & 1) || (H & 1) || (W & 1))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_avgpool3d_2: C and strides must be multiples of 4, dims even");
  const long long total = (long long)B * (D / 2) * (H / 2) * (W / 2) * (C / 4);
  // (streaming loads for inputs of 64 MB and more: UNet2 at 128x256x64 6.84 -> 6.81 ms)
  const int stream_loads = (double)total * 8.0 * 16.0 >= 64.0 * 1048576.0 ? 1 : 0;
  hipLaunchKernelGGL(avgpool3d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, in_cs, in_coff, out, out_cs,
                     out_coff, B, D, H, W, C, stream_loads);
  PGRF_CHECK_LAUNCH("avgpool3d");
  return PGRF_OK;
}

int pgrf_upsample3d_2(const float* in, int in_cs, int in_coff, float* out, int out_cs, int out_coff, int B, int D, int H,
                      int W, int C, void* stream) {
  if (!in || !out) return set_error(PGRF_ERR_ARG, "pgrf_upsample3d_2: null pointer");
  if ((C & 3) || (in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_upsample3d_2: C and strides must be multiples of 4");
  const long long total = (long long)B * D * H * W * (C / 4);
  // streaming stores for outputs of 64 MB and more (UNet2 at 128x256x64: 6.94 -> 6.85 ms; never slower at 64x128x64)
  const int stream_stores = (double)total * 8.0 * 16.0 >= 64.0 * 1048576.0 ? 1 : 0;
  hipLaunchKernelGGL(upsample3d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, in_cs, in_coff, out, out_cs,
                     out_coff, B, D, H, W, C, stream_stores);
  PGRF_CHECK_LAUNCH("upsample3d");
  return PGRF_OK;
}

}  // extern "C"
