// 3x3x3 wrap-padded convolution (stride 1) by the Winograd minimal-filtering algorithm F(2x2x2, 3x3x3) on
// v_mfma_f32_32x32x2_f32 (the UNet2 regulariser is 1.75 TFLOP per 128x256x64
// volume and the largest FLOP item of a frame).  Per 2x2x2 output tile the 27-tap correlation becomes 64 independent
// element-wise products in the transformed domain, i.e. 64 GEMMs [Cout x Cin] x [Cin x tiles]: 8 MACs per output and
// channel pair instead of 27 (3.375x fewer MFMAs).  Transform matrices (Lavin & Gray, "Fast Algorithms for
// Convolutional Neural Networks", 2016), applied along z, y and x:
//     Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   (input patch 4 -> 4)
//     G  = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]        (filter 3 -> 4; on the host in double, pgrf_pack_conv3d_wino)
//     At = [1 1 1 0; 0 1 -1 -1]                       (products 4 -> outputs 2)
// Only 0 and +-1 on the device side: the input and output transforms are additions.
//
// Workgroup (persistent, one per CU) = 8 MFMA waves + 8 loader waves; a "row" = 32 output tiles (32 x 1 along x, or
// 16 x 2 for the half-width levels) of one (b, z pair) x 64 output channels.  Per 32-channel block of Cin:
//   loader waves  request the 4 x 4 x (2 TX + 2) x 32 input patch of every y tile row (rows outside D/H read the nearest
//            row and are zeroed, circular along W), transform it along z and y in registers while the MFMA waves are still
//            busy with the previous block, and between two barriers store T[fz][fy][tyy][channel quad][x parity][x][4] to
//            LDS (~140 KB; one ds_write_b128 per entry);
//   MFMA wave (fz, cout tile)  for each of its 16 frequencies (fy, fx): the B operand of a k-step is the x transform of two
//            LDS values (one ds_read_b128 per patch column and group of four k-steps, one VALU op per k-step), 16 k-steps
//            accumulate M = U_f . V_f over the block's channels, and M is added with signs into the wave's FOUR partially
//            transformed accumulators R[oy][ox] (the y and x output transforms: 2.25 adds per frequency instead of 3.4 for
//            the full 3-D transform, 64 accumulator registers instead of 128; corner frequencies go straight into R).
// After the last block the z output transform (out[oz=0] = R0 + R1 + R2, out[oz=1] = R1 - R2 - R3 over the four fz
// waves of a cout tile; the bias is the initial value of R1) runs through LDS; LeakyReLU and the stores in the epilogue.
// The patch loads live in loader waves because a wave's vmcnt counter is in order: in the MFMA waves every weight refill
// issued behind them would wait for them.
// Accuracy: fp32 throughout; the transforms add ~1e-6 relative round-off (tests compare with the direct kernel and with
// the oracle at the regulariser tolerances).
#include "pgrf_experiments.h"
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "conv3d_common.h"
#include "conv3d_wino_common.h"

namespace pgrf {

#ifndef PGRF_WINO_TV_PINGPONG
#define PGRF_WINO_TV_PINGPONG 0    // 1: the LDS values of group n+1 are read during group n (8 more registers; the budget is 128)
#endif
#ifdef PGRF_WINO_TIMING   // debug build: per-phase cycle totals of wave 0 (s_memtime), written to the buffer passed as `residual`
#define WT_DECL unsigned long long wt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt_t = __builtin_readcyclecounter();
#define WT_TICK(k) { const unsigned long long wt_n = __builtin_readcyclecounter(); wt_acc[k] += wt_n - wt_t; wt_t = wt_n; }
#else
#define WT_DECL
#define WT_TICK(k)
#endif
// Tile grid of a row: TX x TY = 32 tiles (= the 32 columns of the MFMA tile).  TWO = false: 32 x 1 (64 output voxels along
// x; needs W % 64 == 0); TWO = true: 16 x 2 (32 voxels along x, two y tile rows; W % 32 == 0 and H % 4 == 0 -- the
// half-width levels of a volume).
template <bool TWO> struct WinoGeom {
  static constexpr int TX = TWO ? 16 : 32, TY = TWO ? 2 : 1;
  static constexpr int XW = TX + 1;                    // entries per x-parity array (patch columns p = 0 .. 2 TX + 1)
  static constexpr int kFStride = TY * 8 * 2 * XW * 4; // floats per (fz, fy) slab of T: [tyy][quad 8][parity][XW][4 channels]
  static constexpr int kTFloats = 16 * kFStride;
  static constexpr int kStageFloats = 16 * TY * 16 * 4;   // halo staging: 16 TY items x 16 (dz,dy) float4
  static constexpr int kBiasFloats = 1024;               // the layer's bias in accumulator order (Cout <= 1024), copied once per workgroup
  static constexpr int kLdsBytes = (kTFloats + kStageFloats + kBiasFloats) * 4;
};

template <bool TWO>
__global__ void __launch_bounds__(1024) conv3d_wino_kernel(const ConvArgs a, int nrows) {
  typedef WinoGeom<TWO> G;
  constexpr int TX = G::TX, TY = G::TY, XW = G::XW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* T = lds;
  float* stage = lds + G::kTFloats;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, half = lane >> 5;
  const bool loader = wave >= 8;                         // waves 8..15 feed LDS, waves 0..7 run the MFMAs (two + two per SIMD)
  const int Wg = a.W / (2 * TX), Ht = a.H / (2 * TY), Dt = a.D >> 1, cgroups = a.cot_total >> 1;
  const long long rowstride = (long long)a.W * a.in_cs;              // floats between consecutive y
  const long long planestride = rowstride * a.H;

  // A "row" = 32 consecutive 2x2x2 tiles along x (64 output voxels) of one (b, z pair, y pair) x one group of 64 output
  // channels.  The kernel is persistent: one workgroup per CU (the LDS block allows no more) loops over rows.
  struct Row {               // wave-uniform
    unsigned zyok[TY];       // per y tile row: bit dz*4+dy: that patch row is inside the volume
    int b, z0, y0, x0, cot0;
  };
  auto setup = [&](int rowi) {
    Row r;
    int g = rowi;
    r.cot0 = (g % cgroups) * 2; g /= cgroups;
    const int txg = g % Wg; g /= Wg;
    const int ty = g % Ht; g /= Ht;
    const int tz = g % Dt;
    r.b = g / Dt;
    r.x0 = txg * 2 * TX; r.y0 = ty * 2 * TY; r.z0 = tz * 2;
#pragma unroll
    for (int t = 0; t < TY; ++t) {
      r.zyok[t] = 0;
#pragma unroll
      for (int dz = 0; dz < 4; ++dz)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
          if ((unsigned)(r.z0 - 1 + dz) < (unsigned)a.D && (unsigned)(r.y0 + 2 * t - 1 + dy) < (unsigned)a.H) r.zyok[t] |= 1u << (dz * 4 + dy);
    }
    return r;
  };
  // T index of (fzfy, channel c, patch column p)
  // T index (in floats) of the 4 channels of quad q at (fzfy, y tile row, patch column p): [fzfy][tyy][quad 8][parity][XW][4].
  // Channel quads are innermost so that a loader writes its float4 with ONE ds_write_b128 and an MFMA wave reads the
  // four k-steps of a group (lane half h, group g -> quad 4h + g) with ONE ds_read_b128 per patch column.
  auto tq = [](int fzfy, int tyy, int q, int p) { return ((((fzfy * TY + tyy) * 8 + q) * 2 + (p & 1)) * XW + (p >> 1)) * 4; };
  // z-reduction buffer (reuses T): [m][fz][oy*2+ox][lane][16], a lane's four register quads at slots q ^ ((lane >> 2) & 3): every
  // 16-lane group of a ds_write_b128 / ds_read_b128 then covers 16 different bank quads (round 4; was [..][r][lane] with one
  // 4-byte access per register: 64 writes + 96 reads per lane and row, 12 % of a one-block row)
  float* red = lds;
  const int rsw = (lane >> 2) & 3;
  auto rquad = [&](int fz, int oyox, int m, int q) { return reinterpret_cast<float4*>(red + ((((m * 4 + fz) * 4 + oyox) << 10) + lane * 16 + 4 * (q ^ rsw))); };
  // Epilogue of one row, the 8 MFMA waves: wave w finishes cout tile w >> 2, outputs (oz, oy) = ((w >> 1) & 1, w & 1), both ox:
  // the z output transform over the four fz partials, LeakyReLU, stores.
  auto epilogue = [&](const Row& cur) {
    const int m = wave >> 2, oz = (wave >> 1) & 1, oy = wave & 1;
    const int tyy = TWO ? (j >> 4) : 0, tx = TWO ? (j & 15) : j;            // this lane's tile within the row
    const long long ovrow = (((long long)cur.b * a.D + cur.z0 + oz) * a.H + cur.y0 + 2 * tyy + oy) * a.W + cur.x0 + 2 * tx;
    const int cot = cur.cot0 + m;
#pragma unroll
    for (int ox = 0; ox < 2; ++ox) {
      const long long ov = ovrow + ox;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = cot * 32 + q * 8 + half * 4;
        const float4 p1 = *rquad(1, oy * 2 + ox, m, q), p2 = *rquad(2, oy * 2 + ox, m, q);
        const float4 p03 = *rquad(oz == 0 ? 0 : 3, oy * 2 + ox, m, q);
        float4 v = oz == 0 ? f4add(f4add(p03, p1), p2) : f4sub(f4sub(p1, p2), p03);
        // LeakyReLU only (slope 1 = none): the ELU / residual epilogues of conv3d.hip are not needed by the layers
        // this kernel serves
        v.x = v.x > 0.f ? v.x : v.x * a.slope; v.y = v.y > 0.f ? v.y : v.y * a.slope;
        v.z = v.z > 0.f ? v.z : v.z * a.slope; v.w = v.w > 0.f ? v.w : v.w * a.slope;
        *reinterpret_cast<float4*>(a.out + ov * a.out_cs + a.out_coff + c) = v;          // (non-temporal stores here: UNet2 6.94 -> 7.1-7.2 ms)
      }
    }
  };

  int rowi = blockIdx.x;
  if (rowi >= nrows) return;
  // the bias block in LDS: the first MFMA of a row is a corner frequency that accumulates straight onto it, so a global load
  // at row start is an exposed memory latency per row (6 % of a one-block row)
  float* lbias = lds + G::kTFloats + G::kStageFloats;
  if (a.bias)
    for (int i = tid; i < a.cot_total * 32; i += 1024) lbias[i] = a.bias[i];
  __syncthreads();
  // Barrier protocol per 32-channel block (all 8 waves):  B1 = "T may be overwritten" (the MFMA waves are done with the
  // previous block / everybody is done with the reduction buffer), B2 = "T is complete".  Per row two more around the
  // z reduction: E1 = "T is dead", E2 = "partials are in LDS".
  if (loader) {
    // ------------------------------------------------------------------------------------------------ loader waves
    // They own the global loads of the input patch, so that the in-order vmcnt counter of the MFMA waves only ever
    // sees their short weight refills (one wave doing both had to wait for the whole patch before its first MFMA of
    // every block: 25 % of the kernel).  Thread lt handles the main items (patch column p, channel quad cq) =
    // (lt >> 3, lt & 7) and (32 + (lt >> 3), lt & 7) -- eight consecutive lanes read one 128-byte line -- and one
    // (dz, dy) entry of one halo item (columns 64, 65): item = lt >> 4 (hp = item >> 3, cq = item & 7), (dz, dy) = lt & 15.
    const int lt = tid - 512;                              // 0..511
    const int ityy = TWO ? (lt >> 8) : 0, ip = TWO ? ((lt >> 3) & 31) : (lt >> 3), icq = lt & 7;   // main item: y tile row, patch column 0 .. 2 TX - 1, channel quad
    const bool hthread = lt < 256 * TY;
    const int hitem = lt >> 4, hzy = lt & 15;              // halo item (16 TY of them): y tile row (TWO), column 2 TX + hp, quad
    const int htyy = TWO ? (hitem >> 4) : 0, hp = (hitem >> 3) & 1, hcq = hitem & 7;
    const unsigned ps4 = (unsigned)planestride * 4u, rs4 = (unsigned)rowstride * 4u;       // bytes; one batch element < 4 GiB
    float4 raw[16];
    float4 hraw = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](const Row& r, int blk) {
      const char* eb = reinterpret_cast<const char*>(a.in + (long long)r.b * a.D * planestride + a.in_coff);
      int xm = r.x0 - 1 + ip;
      xm = xm < 0 ? xm + a.W : (xm >= a.W ? xm - a.W : xm);
      const unsigned moff = ((unsigned)xm * (unsigned)a.in_cs + icq * 4 + blk * 32) * 4u;
      // rows outside the volume read the nearest row inside (and are zeroed before the transform): unconditional loads
#pragma unroll
      for (int zy = 0; zy < 16; ++zy) {
        const int zc = min(max(r.z0 - 1 + (zy >> 2), 0), a.D - 1), yc = min(max(r.y0 + 2 * ityy - 1 + (zy & 3), 0), a.H - 1);
        raw[zy] = sload16(eb + ((unsigned)zc * ps4 + (unsigned)yc * rs4), moff);
      }
      if (hthread) {
        int xh = r.x0 + 2 * TX - 1 + hp;
        xh = xh >= a.W ? xh - a.W : xh;
        const int hz = min(max(r.z0 - 1 + (hzy >> 2), 0), a.D - 1), hy = min(max(r.y0 + 2 * htyy - 1 + (hzy & 3), 0), a.H - 1);
        const unsigned hoff = (unsigned)hz * ps4 + (unsigned)hy * rs4 + ((unsigned)xh * (unsigned)a.in_cs + hcq * 4 + blk * 32) * 4u;
        hraw = sload16(eb, hoff);
      }
    };
    // z and y transforms in registers (while the MFMA waves are still busy with the previous block) ...
    auto transform = [&](unsigned zyok) {
      if (zyok != 0xffffu) {                               // volume faces only
#pragma unroll
        for (int zy = 0; zy < 16; ++zy)
          if (!((zyok >> zy) & 1u)) raw[zy] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) bt4(raw[0 * 4 + dy], raw[1 * 4 + dy], raw[2 * 4 + dy], raw[3 * 4 + dy]);   // along z
#pragma unroll
      for (int fz = 0; fz < 4; ++fz) bt4(raw[fz * 4 + 0], raw[fz * 4 + 1], raw[fz * 4 + 2], raw[fz * 4 + 3]);   // along y
      // pin the results here: the scheduler otherwise sinks the whole transform behind the barrier that follows, into
      // the window in which the MFMA waves wait for T
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(raw[q].x), "+v"(raw[q].y), "+v"(raw[q].z), "+v"(raw[q].w));
    };
    // ... and only the LDS stores between the two barriers
    auto store = [&]() {
      // two base addresses + compile-time offsets (a ds_write offset field is 16 bits; with one base the compiler
      // precomputed 16 addresses, spilled half of them and reloaded each from scratch in front of its store)
      float* b0 = T + tq(0, ityy, icq, ip);
      float* b1 = T + tq(8, ityy, icq, ip);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        *reinterpret_cast<float4*>(b0 + f * G::kFStride) = raw[f];
        *reinterpret_cast<float4*>(b1 + f * G::kFStride) = raw[8 + f];
      }
    };
    // halo columns: thread (item, fz, fy) combines the 2 x 2 raw values its transformed entry is made of
    auto halo_store = [&]() {
      if (!hthread) return;
      const int fz = hzy >> 2, fy = hzy & 3;
      // Bt row f = +d[ia] + sb * d[ib]:   0: d0 - d2 | 1: d1 + d2 | 2: d2 - d1 | 3: d1 - d3
      const int za = (fz == 0) ? 0 : (fz == 2 ? 2 : 1), zb = (fz == 2) ? 1 : (fz == 3 ? 3 : 2);
      const int ya = (fy == 0) ? 0 : (fy == 2 ? 2 : 1), yb = (fy == 2) ? 1 : (fy == 3 ? 3 : 2);
      const float szb = (fz == 1) ? 1.f : -1.f, syb = (fy == 1) ? 1.f : -1.f;
      const float* sp = stage + hitem * 64;
      const float4 daa = *reinterpret_cast<const float4*>(sp + (za * 4 + ya) * 4), dab = *reinterpret_cast<const float4*>(sp + (za * 4 + yb) * 4);
      const float4 dba = *reinterpret_cast<const float4*>(sp + (zb * 4 + ya) * 4), dbb = *reinterpret_cast<const float4*>(sp + (zb * 4 + yb) * 4);
      *reinterpret_cast<float4*>(T + tq(hzy, htyy, hcq, 2 * TX + hp)) =
          make_float4((daa.x + syb * dab.x) + szb * (dba.x + syb * dbb.x), (daa.y + syb * dab.y) + szb * (dba.y + syb * dbb.y),
                      (daa.z + syb * dab.z) + szb * (dba.z + syb * dbb.z), (daa.w + syb * dab.w) + szb * (dba.w + syb * dbb.w));
    };
    Row cur = setup(rowi);
#ifndef PGRF_WINO_ABL_NOA
    prefetch(cur, 0);
#endif
    while (true) {
      const int nrow = rowi + gridDim.x;
      const bool has_next = nrow < nrows;
      for (int blk = 0; blk < a.chunks; ++blk) {
        // the halo columns' raw values go to the staging area before B1 (only this wave reads them, after B1)
#ifndef PGRF_WINO_ABL_NOA
        if (hthread) *reinterpret_cast<float4*>(stage + (hitem * 16 + hzy) * 4) = ((cur.zyok[htyy] >> hzy) & 1u) ? hraw : make_float4(0.f, 0.f, 0.f, 0.f);
        transform(cur.zyok[ityy]);
#endif
        lds_barrier();                                     // B1
#ifndef PGRF_WINO_ABL_NOA
        store();
        halo_store();
#endif
        lds_barrier();                                     // B2
#ifndef PGRF_WINO_ABL_NOA
        if (blk + 1 < a.chunks) prefetch(cur, blk + 1);
        else if (has_next) prefetch(setup(nrow), 0);
#endif
      }
      lds_barrier();                                       // E1
      lds_barrier();                                       // E2
      if (!has_next) break;
      rowi = nrow;
      cur = setup(rowi);
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- MFMA waves
  // Wave w owns the 16 frequencies (fz = w & 3, fy, fx) of cout tile w >> 2 of the row's group; two waves per SIMD.
  const int fzw = wave & 3, mw = wave >> 2;
  WT_DECL
  const int tyy = TWO ? (j >> 4) : 0, tx = TWO ? (j & 15) : j;               // this lane's tile (MFMA column j) within the row
  const float* tb = T + tq(fzw * 4, tyy, half * 4, 2 * tx);                  // its T base: fz = fzw, fy = 0, quad = 4*half (+g), even column of tile tx
  const unsigned lane16 = lane * 16;                     // per-lane byte offset into a packed A block; the block offset is scalar
  const char* wb = reinterpret_cast<const char*>(a.wpk);
  auto a_off = [&](int cot0, int blk, int fyfx) {         // byte offset of the packed block (cout tile, frequency, channel block)
    return (((unsigned)(cot0 + mw) * 64u + fzw * 16 + fyfx) * (unsigned)a.chunks + blk) * 4096u;
  };
  auto a_load = [&](unsigned off) { return sload16(wb + off, lane16); };
  float4 A[4];                                           // A operands of the current frequency, refilled in place
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
  Row cur = setup(rowi);
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) A[s4] = a_load(a_off(cur.cot0, 0, 0) + s4 * 1024);
  while (true) {
    const int nrow = rowi + gridDim.x;
    const bool has_next = nrow < nrows;
    const int ncot0 = has_next ? setup(nrow).cot0 : cur.cot0;     // (for the A operands after this row's last frequency)
    // The bias enters through the accumulators of the fz = 1 waves: R1 appears with + in both z outputs
    // (out[oz=0] = R0 + R1 + R2, out[oz=1] = R1 - R2 - R3), so the epilogue has no dependent bias loads.
    f32x16 R[2][2];                                      // [oy][ox]
    {
      f32x16 binit = zero16;
      if (a.bias && fzw == 1) {
        const float4* bp = reinterpret_cast<const float4*>(lbias + (cur.cot0 + mw) * 32 + half * 16);
        const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        binit = (f32x16){b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
      }
#pragma unroll
      for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) R[oy][ox] = binit;
    }
    for (int blk = 0; blk < a.chunks; ++blk) {
      const bool lastblk = blk + 1 == a.chunks;
      WT_TICK(0)                                           // row setup / R init / previous epilogue
      lds_barrier();                                       // B1
      WT_TICK(1)
      lds_barrier();                                       // B2
      WT_TICK(2)
#ifndef PGRF_WINO_ABL_NOB
      // 64 groups of 4 k-steps (frequency-major).  The two T values every k-step needs are read one group ahead
      // (tv ping-pong), so the LDS latency of group n+1 hides under the MFMAs of group n; sched_barriers keep that order.
      float4 tv[PGRF_WINO_TV_PINGPONG ? 2 : 1][2];
      auto tload = [&](auto buf_tag, auto n_tag) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_tag)::value, n = decltype(n_tag)::value;
        constexpr int fyfx = n >> 2, g = n & 3, fy = fyfx >> 2, fx = fyfx & 3;
        const float* tr = tb + fy * G::kFStride + g * (2 * XW * 4);  // even columns of (fy, quad 4*half + g); odd columns at + XW*4
        // Bt along x on patch columns (2j, 2j+1, 2j+2, 2j+3) = (E[j], O[j], E[j+1], O[j+1]); the two operands of the group's
        // difference / sum, four k-steps (channels) each
        constexpr int o0 = (fx == 0) ? 0 : (fx == 2) ? 4 : XW * 4;
        constexpr int o1 = (fx == 0 || fx == 1) ? 4 : (fx == 2) ? XW * 4 : XW * 4 + 4;
        tv[buf][0] = *reinterpret_cast<const float4*>(tr + o0);
        tv[buf][1] = *reinterpret_cast<const float4*>(tr + o1);
      };
      tload(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      f32x16 M;
      // A operands: frequency stride within the block, and where the operands after this block's last frequency are
      // (next block / next row's first block; the very last: this block again)
      const unsigned abase = a_off(cur.cot0, blk, 0);
      const unsigned astride = (unsigned)a.chunks * 4096u;
      const unsigned anext = lastblk ? a_off(ncot0, 0, 0) : abase + 4096u;
      static_for<64>([&](auto n_tag) __attribute__((always_inline)) {
        constexpr int n = decltype(n_tag)::value;
#if PGRF_WINO_TV_PINGPONG
        constexpr int fyfx = n >> 2, g = n & 3, fy = fyfx >> 2, fx = fyfx & 3, buf = n & 1;
        if constexpr (n + 1 < 64) tload(std::integral_constant<int, (buf ^ 1)>{}, std::integral_constant<int, n + 1>{});
#else
        constexpr int fyfx = n >> 2, g = n & 3, fy = fyfx >> 2, fx = fyfx & 3, buf = 0;
#ifdef PGRF_WINO_ABL_NOLDS
        if constexpr (n > 0) { tv[0][0].x += 1.f; tv[0][1].y += 1.f; }
#else
        if constexpr (n > 0) tload(std::integral_constant<int, 0>{}, n_tag);
#endif
#endif
        // The four corner frequencies (fy, fx in {0, 3}) contribute to exactly ONE output (oy, ox) = (fy == 3, fx == 3),
        // with sign (-1)^(fy == 3) (-1)^(fx == 3): their MFMAs accumulate straight into that R, no products to add.
        constexpr bool kCorner = (fy == 0 || fy == 3) && (fx == 0 || fx == 3);
        constexpr int kCy = fy == 3 ? 1 : 0, kCx = fx == 3 ? 1 : 0;
        constexpr bool kCornerNeg = (fy == 3) != (fx == 3);
        const float4 aq = A[g];
        // the x transform of two k-steps per instruction (v_pk_add_f32 through vector types: UNet2 6.80 -> 6.65 ms; the same as inline
        // asm was faster still (6.54) and WRONG: a VALU result needs two wait states before an MFMA reads it as SrcB, which the hazard
        // recogniser does not insert around inline asm; with `s_nop 1` in the asm 6.62, correct on every test -- not worth carrying a
        // hand-maintained hazard for 1 %)
        f32x2_t bpk[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x2_t p0 = h2 == 0 ? (f32x2_t){tv[buf][0].x, tv[buf][0].y} : (f32x2_t){tv[buf][0].z, tv[buf][0].w};
          const f32x2_t p1 = h2 == 0 ? (f32x2_t){tv[buf][1].x, tv[buf][1].y} : (f32x2_t){tv[buf][1].z, tv[buf][1].w};
          if constexpr (fx == 1) bpk[h2] = p0 + p1;       // (vector types: the compiler selects v_pk_add_f32 and knows its hazards)
          else if constexpr (kCorner && kCornerNeg) bpk[h2] = p1 - p0;
          else bpk[h2] = p0 - p1;
        }
#pragma unroll
        for (int kl = 0; kl < 4; ++kl) {
          // (corner frequencies: the sign of their single output-transform term is folded into the operand)
          const float bv = kl == 0 ? bpk[0].x : kl == 1 ? bpk[0].y : kl == 2 ? bpk[1].x : bpk[1].y;
          const float av = kl == 0 ? aq.x : kl == 1 ? aq.y : kl == 2 ? aq.z : aq.w;
#ifdef PGRF_WINO_ABL_NOADD
          M = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, (n == 0 && kl == 0) ? zero16 : M, 0, 0, 0);
#else
          if constexpr (kCorner) R[kCy][kCx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, R[kCy][kCx], 0, 0, 0);
          else M = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, (g == 0 && kl == 0) ? zero16 : M, 0, 0, 0);
#endif
        }
#ifndef PGRF_WINO_ABL_NOALOAD      // timing experiments: stale weights / stale LDS values / one accumulation chain
        A[g] = a_load((fyfx < 15 ? abase + (fyfx + 1) * astride : anext) + g * 1024);
#endif
#ifdef PGRF_WINO_ABL_NOADD
        if constexpr (n == 63) R[0][0] += M;
#else
        if constexpr (g == 3 && !kCorner) {
          // y and x output transforms: At column f contributes to output 0 with + (f = 0, 1, 2) and to output 1 with
          // + (f = 1) or - (f = 2, 3)
#pragma unroll
          for (int oy = 0; oy < 2; ++oy) {
            if ((oy == 0 && fy == 3) || (oy == 1 && fy == 0)) continue;
            const bool ny = (oy == 1 && fy >= 2);
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
              if ((ox == 0 && fx == 3) || (ox == 1 && fx == 0)) continue;
              const bool nx = (ox == 1 && fx >= 2);
              if (ny != nx) sub16(R[oy][ox], M);
              else R[oy][ox] += M;
            }
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
#endif
      WT_TICK(3)                                           // MFMA phase
    }
    lds_barrier();                                         // E1: T is dead
    WT_TICK(4)
#pragma unroll
    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
      for (int ox = 0; ox < 2; ++ox)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *rquad(fzw, oy * 2 + ox, mw, q) = make_float4(R[oy][ox][4 * q], R[oy][ox][4 * q + 1], R[oy][ox][4 * q + 2], R[oy][ox][4 * q + 3]);
    WT_TICK(5)                                             // partials to LDS
    lds_barrier();                                         // E2: partials are in LDS
    WT_TICK(6)
    epilogue(cur);
    WT_TICK(7)
    if (!has_next) break;
    rowi = nrow;
    cur = setup(rowi);
  }
#ifdef PGRF_WINO_TIMING
  if (tid == 0 && a.residual) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.residual));
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&dbg[k], wt_acc[k]);
  }
#endif
}

}  // namespace pgrf

using namespace pgrf;

static constexpr int kMaxDevices = 64;

extern "C" {

size_t pgrf_conv3d_wino_packed_floats(int Cout, int Cin) {
  const size_t cot = (Cout + 31) / 32, chunks = (Cin + 31) / 32;
  return cot * 64 * chunks * 1024 + cot * 32;
}

int pgrf_pack_conv3d_wino(const float* w_host, const float* bias_host, int Cout, int Cin, float* packed_host) {
  if (!w_host || !packed_host) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d_wino: null pointer");
  if (Cout < 1 || Cin < 1) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d_wino: bad channels");
  const int cot = (Cout + 31) / 32, chunks = (Cin + 31) / 32;
  std::memset(packed_host, 0, sizeof(float) * pgrf_conv3d_wino_packed_floats(Cout, Cin));
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  std::vector<float> U((size_t)Cout * Cin * 64);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci) {
      const float* w = w_host + ((size_t)co * Cin + ci) * 27;           // [dz][dy][dx]
      for (int fz = 0; fz < 4; ++fz)
        for (int fy = 0; fy < 4; ++fy)
          for (int fx = 0; fx < 4; ++fx) {
            double s = 0.0;
            for (int dz = 0; dz < 3; ++dz)
              for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) s += G[fz][dz] * G[fy][dy] * G[fx][dx] * (double)w[(dz * 3 + dy) * 3 + dx];
            U[((size_t)co * Cin + ci) * 64 + (fz * 4 + fy) * 4 + fx] = (float)s;
          }
    }
  for (int ct = 0; ct < cot; ++ct)
    for (int f = 0; f < 64; ++f)
      for (int ch = 0; ch < chunks; ++ch)
        for (int s = 0; s < 16; ++s)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = 32 * ct + (lane & 31), ci = 32 * ch + 16 * (lane >> 5) + s;
            if (co < Cout && ci < Cin) packed_host[wpk_index(ct, f, ch, chunks, s, lane, 64)] = U[((size_t)co * Cin + ci) * 64 + f];
          }
  float* B = packed_host + (size_t)cot * 64 * chunks * 1024;
  for (int ct = 0; ct < cot; ++ct)
    for (int h = 0; h < 2; ++h)
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * ct + rho_(r, h);
        B[ct * 32 + h * 16 + r] = (bias_host && co < Cout) ? bias_host[co] : 0.f;
      }
  return PGRF_OK;
}

int pgrf_conv3d_wino_supported(int Cin, int Cout, int D, int H, int W) {
  // rows of 32 x 1 tiles (W % 64 == 0), else of 16 x 2 tiles (W % 32 == 0 and H % 4 == 0)
  const bool geom = (W % 64 == 0 && H % 2 == 0) || (W % 32 == 0 && H % 4 == 0);
  return (Cin % 32 == 0) && (Cout % 64 == 0) && (D % 2 == 0) && geom && D >= 2 && H >= 2 && W >= 32;
}

int pgrf_conv3d_wino_preferred(int Cin, int Cout, int B, int D, int H, int W) {
  if (!pgrf_conv3d_wino_supported(Cin, Cout, D, H, W)) return 0;
  // the kernel is persistent with one workgroup per CU and a row is its unit of work: below ~100 rows most of the chip
  // idles and the split-K direct kernel wins (measured: 512 -> 128 at 16x16x32, 64 rows: 335 vs 277 us)
  const bool two = (W % 64) != 0;
  const long long rows = (long long)B * (D / 2) * (H / (two ? 4 : 2)) * (W / (two ? 32 : 64)) * (Cout / 64);
  return rows >= 96 ? 1 : 0;
}

int pgrf_conv3d_wino(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                     const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                     int res_cs, int res_coff, void* stream) {
  if (!in || !out || !packed) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_wino: null pointer");
  if (!pgrf_conv3d_wino_supported(Cin, Cout, D, H, W))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino: needs Cin a multiple of 32, Cout a multiple of 64, even D, and W % 64 == 0 with even H or W % 32 == 0 with H % 4 == 0 "
                     "(got Cin %d Cout %d D %d H %d W %d); use pgrf_conv3d_wrap", Cin, Cout, D, H, W);
  if (B < 1) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_wino: bad dims");
#ifndef PGRF_WINO_TIMING
  if (residual) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino: no residual epilogue (use pgrf_conv3d_wrap)");
#endif
  if ((double)(Cout / 32) * 64.0 * (Cin / 32) * 4096.0 >= 4294967296.0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino: packed weights of >= 4 GiB");
  if ((double)D * H * W * in_cs * 4.0 >= 4294967296.0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino: one batch element of the input is >= 4 GiB (32-bit byte offsets)");
  if ((in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3) || (res_cs & 3) || (res_coff & 3))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino: channel strides/offsets must be multiples of 4");
  ConvArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = Cin; a.out = out; a.out_cs = out_cs; a.out_coff = out_coff;
  a.Cout = Cout; a.B = B; a.D = D; a.H = H; a.W = W; a.Do = D; a.Ho = H; a.Wo = W; a.stride = 1; a.slope = lrelu_slope;
  a.chunks = Cin / 32; a.cot_total = Cout / 32;
  a.residual = residual; a.res_cs = res_cs; a.res_coff = res_coff; a.wtaps = 64; a.wrap_w = 1;
  a.wpk = packed;
  a.bias = has_bias ? packed + (size_t)a.cot_total * 64 * a.chunks * 1024 : nullptr;
  const bool two = (W % 64) != 0;                        // 16 x 2 tile rows for the half-width levels
  const int tx = two ? 16 : 32, ty = two ? 2 : 1;
  const long long nrows_ll = (long long)B * (D / 2) * (H / (2 * ty)) * (W / (2 * tx)) * (a.cot_total / 2);
  if (nrows_ll >= (1ll << 31)) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino: too many tile rows");
  const int nrows = (int)nrows_ll;
  // per DEVICE (a process may drive several GPUs): the LDS reservation is a per-device function attribute and the
  // persistent grid is sized from that device's CU count
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
    return set_error(PGRF_ERR_LAUNCH, "pgrf_conv3d_wino: cannot query the current device");
  int ncu;
  {
    static std::mutex mu;
    static int ncu_of[kMaxDevices] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (ncu_of[dev] == 0) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WinoGeom<false>::kLdsBytes) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WinoGeom<true>::kLdsBytes) != hipSuccess)
        return set_error(PGRF_ERR_LAUNCH, "pgrf_conv3d_wino: cannot reserve %d bytes of LDS", WinoGeom<true>::kLdsBytes);
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return set_error(PGRF_ERR_LAUNCH, "pgrf_conv3d_wino: cannot query the device");
      ncu_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    ncu = ncu_of[dev];
  }
  // persistent: one workgroup (8 MFMA waves + 8 loader waves) per CU (the ~140 KB LDS block allows no more), each loops over rows
  int wgs = ncu;
  if (const char* e = getenv("PGRF_WINO_WGS")) { const int v = atoi(e); if (v > 0) wgs = v; }          // tuning knob
  dim3 grid((unsigned)(nrows < wgs ? nrows : wgs));
  if (two) hipLaunchKernelGGL(conv3d_wino_kernel<true>, grid, dim3(1024), WinoGeom<true>::kLdsBytes, (hipStream_t)stream, a, nrows);
  else hipLaunchKernelGGL(conv3d_wino_kernel<false>, grid, dim3(1024), WinoGeom<false>::kLdsBytes, (hipStream_t)stream, a, nrows);
  PGRF_CHECK_LAUNCH("conv3d_wino");
  return PGRF_OK;
}

}  // extern "C"
