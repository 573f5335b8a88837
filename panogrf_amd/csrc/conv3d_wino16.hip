// Winograd F(2x2x2, 3x3x3) 3-D convolution, second generation (round 4): TWO INDEPENDENT 8-wave workgroups per CU on 16-tile
// rows and v_mfma_f32_16x16x4_f32, instead of one 16-wave workgroup on 32-tile rows (conv3d_wino.hip).
//
// Why.  The first kernel is one tightly coupled workgroup per CU (its 139 KB LDS block allows no second one): every barrier --
// "T free", "T ready", the z reduction and the epilogue of a row -- stalls BOTH MFMA waves of a SIMD at once, and the phase census
// (tools/wino_timing.py, round 4) has the MFMA phase at 57 % (one-block rows) to 72 % of an MFMA wave's life: 55 % of the
// Winograd-domain MFMA roof overall.  The f32 MFMA shares its issue port with the VALU, so a SIMD's two MFMA waves cannot hide
// each other's transforms either while they run in lock-step.  Here a row is 16 tiles (32 output voxels along x) x 64 output
// channels, its transformed patch T is 70 KB, and a workgroup is 4 MFMA waves + 4 loader waves: two workgroups share a CU (4 waves
// per SIMD, 128 registers each, 2 x 78 KB of LDS), drift against each other, and one's barriers / epilogue / B-operand formation
// run while the other's MFMAs own the pipe.
//
// Per 32-channel block of Cin and row:
//   loader waves (256 threads)  as in conv3d_wino.hip with TX = 16: request the 4 x 4 x 34 x 32 patch (16-byte items, rows outside
//            D / H clamped and zeroed, circular along W), transform along z and y in registers, and between the barriers B1 / B2 store
//            T[fz][fy][channel quad][x parity][17][4] (one ds_write_b128 per entry); the two halo columns through a staging area.
//   MFMA wave fz (4 of them)  owns the 16 frequencies (fz, fy, fx) of ALL FOUR 16-channel output tiles.  Lane (n, kq) = tile column
//            n = lane & 15, k slot kq = lane >> 4.  Per frequency: the B operands of the block's 8 k-steps = the x transform of
//            two ds_read_b128 per group of four k-steps (channel quad 4 g + kq: k-step 4 g + j pairs lane slot kq with channel
//            16 g + 4 kq + j; the packed weights follow the same permutation), formed ONCE and used by the four output tiles
//            (the 32-row kernel formed a B operand per MFMA); per output tile 8 chained MFMAs into a 4-register M, which is added
//            with signs into the partially transformed accumulators R[oy][ox][tile] (y and x output transforms; the four corner
//            frequencies accumulate straight into their single target).  64 + 4 accumulator registers, 32 of A operands (one
//            frequency ahead, refilled in place tile by tile), 8 of B operands.
//   z output transform over the four fz waves through LDS (the buffer reuses T), LeakyReLU, one float4 store per (voxel, tile).
// Applies when Cin % 32 == 0, Cout % 64 == 0, D and H even, W % 32 == 0.
#include "pgrf_experiments.h"
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "conv3d_common.h"
#include "conv3d_wino_common.h"

namespace pgrf {

#ifdef PGRF_WINO_TIMING   // debug build: per-phase cycle totals of wave 0 (s_memtime), written to the buffer passed as `residual`
#define W16_DECL unsigned long long wt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt_t = __builtin_readcyclecounter();
#define W16_TICK(k) { const unsigned long long wt_n = __builtin_readcyclecounter(); wt_acc[k] += wt_n - wt_t; wt_t = wt_n; }
#else
#define W16_DECL
#define W16_TICK(k)
#endif

namespace w16 {
constexpr int TX = 16;                       // tiles per row (= MFMA columns)
constexpr int XW = TX + 1;                   // entries per x-parity array (patch columns 0 .. 2 TX + 1)
// T is a sequence of 17-entry blocks [XW][4 channels] (68 floats = 17 sixteen-byte slots), one per (fz fy, channel quad, x parity):
//   block(fzfy, q, parity) = (fzfy >> 1) * 32 + (q & 1) * 16 + (fzfy & 1) * 8 + (q >> 1) * 2 + parity
// The quads 2 m and 2 m + 1 are SIXTEEN blocks apart = 272 slots = 0 (mod 16): a ds_read_b128 is serviced in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), i.e. eight lanes of k slot kq and eight of kq + 1 with
// complementary tile columns -- with the two quads a multiple of 16 slots apart their 16 slots are all different (the plain
// [fzfy][quad][parity][XW] order put them 34 slots = 2 (mod 16) apart: every B-operand read was a 2-way bank conflict).
constexpr int kBlk = XW * 4;                 // floats per block
constexpr int kTFloats = 8 * 32 * kBlk;      // 17 408 floats = 69.6 KB
__host__ __device__ constexpr int t_block(int fzfy, int q, int parity) { return (fzfy >> 1) * 32 + (q & 1) * 16 + (fzfy & 1) * 8 + (q >> 1) * 2 + parity; }
constexpr int kStageFloats = 16 * 16 * 4;    // halo staging: 16 items x 16 (dz, dy) float4
constexpr int kBiasFloats = 1024;            // the layer's bias (natural order), copied once per workgroup
constexpr int kLdsBytes = (kTFloats + kStageFloats + kBiasFloats) * 4;
constexpr int kThreads = 512;
// packed weights: [cout group of 64][frequency 64][channel block][output tile 4][k group 2][lane 64][4 k-steps]
constexpr int kABlockBytes = 4 * 2 * 64 * 16;   // one (cout group, frequency, block): 8 KB
__host__ __device__ inline size_t wpk16_index(int cg, int f, int blk, int chunks, int t, int g, int lane, int j) {
  return (((((size_t)(cg * 64 + f) * chunks + blk) * 4 + t) * 2 + g) * 64 + lane) * 4 + j;
}
// input channel (within a 32-channel block) that lane slot kq supplies to k-step 4 g + j
__host__ __device__ inline int k_channel(int g, int kq, int j) { return 16 * g + 4 * kq + j; }
}  // namespace w16

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4w __attribute__((ext_vector_type(4)));
// 16-byte load through a buffer descriptor: SGPR descriptor + constant per-lane offset + SCALAR byte offset, so a weight refill
// costs no vector instruction besides the load (a flat `base + offset` made the compiler form a 64-bit per-lane address per load:
// 128 v_lshl_add_u64 per block).  (The clang builtin's b128 form lowers to a one-dword load here: the LLVM intrinsic is bound.)
__device__ f32x4v pgrf_w16_buffer_load_x4(i32x4w rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
// r += m / r -= m on a 4-register accumulator as two packed adds (plain +=, -= on the vector type compile to four scalar adds)
typedef float f32x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void acc_add4(f32x4v& r, const f32x4v& m) {
  f32x2w a = {r[0], r[1]}, b = {r[2], r[3]};
  const f32x2w c = {m[0], m[1]}, d = {m[2], m[3]};
  asm("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
  asm("v_pk_add_f32 %0, %0, %1" : "+v"(b) : "v"(d));
  r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
}
__device__ __forceinline__ void acc_sub4(f32x4v& r, const f32x4v& m) {
  f32x2w a = {r[0], r[1]}, b = {r[2], r[3]};
  const f32x2w c = {m[0], m[1]}, d = {m[2], m[3]};
  asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(c));
  asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(b) : "v"(d));
  r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
}

__global__ void __launch_bounds__(w16::kThreads, 4) conv3d_wino16_kernel(const ConvArgs a, int nrows) {
  using namespace w16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* T = lds;
  float* stage = lds + kTFloats;
  float* lbias = lds + kTFloats + kStageFloats;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;                         // waves 4..7 feed LDS, waves 0..3 run the MFMAs
  const int Wg = (a.W + 2 * TX - 1) / (2 * TX), Ht = a.H >> 1, Dt = a.D >> 1, cgroups = a.cot_total >> 1;
  const long long rowstride = (long long)a.W * a.in_cs;              // floats between consecutive y
  const long long planestride = rowstride * a.H;

  struct Row {               // wave-uniform
    unsigned zyok;           // bit dz*4+dy: that patch row is inside the volume
    int b, z0, y0, x0, cg;
    int c0, c1, split;       // input channel blocks [c0, c1) of this work item (the whole range unless a.ksplit > 1)
  };
  const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
  auto setup = [&](int rowi) {
    Row r;
    int g = rowi;
    r.split = g % ksplit; g /= ksplit;
    r.c0 = (int)((long long)a.chunks * r.split / ksplit); r.c1 = (int)((long long)a.chunks * (r.split + 1) / ksplit);
    r.cg = g % cgroups; g /= cgroups;
    const int txg = g % Wg; g /= Wg;
    const int ty = g % Ht; g /= Ht;
    const int tz = g % Dt;
    r.b = g / Dt;
    r.x0 = txg * 2 * TX; r.y0 = ty * 2; r.z0 = tz * 2;
    r.zyok = 0;
#pragma unroll
    for (int dz = 0; dz < 4; ++dz)
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
        if ((unsigned)(r.z0 - 1 + dz) < (unsigned)a.D && (unsigned)(r.y0 - 1 + dy) < (unsigned)a.H) r.zyok |= 1u << (dz * 4 + dy);
    return r;
  };
  // float index of the 4 channels of quad q at (fzfy, patch column p)
  auto tq = [](int fzfy, int q, int p) { return t_block(fzfy, q, p & 1) * kBlk + (p >> 1) * 4; };

  int rowi = blockIdx.x;
  if (rowi >= nrows) return;
  if (a.bias)
    for (int i = tid; i < a.Cout; i += kThreads) lbias[i] = a.bias[i];
  __syncthreads();

  // Barrier protocol per 32-channel block (all 8 waves):  B1 = "T may be overwritten", B2 = "T is complete"; per row two more
  // around the z reduction: E1 = "T is dead", E2 = "partials are in LDS".
  if (loader) {
    // ------------------------------------------------------------------------------------------------ loader waves
    // thread lt: main item (patch column ip = lt >> 3 in 0..31, channel quad icq = lt & 7) -- eight consecutive lanes read one
    // 128-byte line -- and one (dz, dy) entry of one halo item (columns 32, 33): item = lt >> 4 (hp = item >> 3, cq = item & 7)
    const int lt = tid - 256;                              // 0..255
    const int ip = lt >> 3, icq = lt & 7;
    const int hitem = lt >> 4, hzy = lt & 15, hp = (hitem >> 3) & 1, hcq = hitem & 7;
    const unsigned ps4 = (unsigned)planestride * 4u, rs4 = (unsigned)rowstride * 4u;       // bytes; one batch element < 4 GiB
    float4 raw[16];
    float4 hraw = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](const Row& r, int blk) {
      const char* eb = reinterpret_cast<const char*>(a.in + (long long)r.b * a.D * planestride + a.in_coff);
      int xm = r.x0 - 1 + ip;
      xm = xm < 0 ? xm + a.W : xm;
      while (xm >= a.W) xm -= a.W;                         // (W = 16: a 32-voxel row walks the circular volume twice; its second half is not stored)
      const unsigned moff = ((unsigned)xm * (unsigned)a.in_cs + icq * 4 + blk * 32) * 4u;
#pragma unroll
      for (int zy = 0; zy < 16; ++zy) {
        const int zc = min(max(r.z0 - 1 + (zy >> 2), 0), a.D - 1), yc = min(max(r.y0 - 1 + (zy & 3), 0), a.H - 1);
        raw[zy] = sload16(eb + ((unsigned)zc * ps4 + (unsigned)yc * rs4), moff);
      }
      int xh = r.x0 + 2 * TX - 1 + hp;
      while (xh >= a.W) xh -= a.W;
      const int hz = min(max(r.z0 - 1 + (hzy >> 2), 0), a.D - 1), hy = min(max(r.y0 - 1 + (hzy & 3), 0), a.H - 1);
      const unsigned hoff = (unsigned)hz * ps4 + (unsigned)hy * rs4 + ((unsigned)xh * (unsigned)a.in_cs + hcq * 4 + blk * 32) * 4u;
      hraw = sload16(eb, hoff);
    };
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
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(raw[q].x), "+v"(raw[q].y), "+v"(raw[q].z), "+v"(raw[q].w));
    };
    auto store = [&]() {
      float* b0 = T + tq(0, icq, ip);
      float* b1 = T + tq(8, icq, ip);
#pragma unroll
      for (int f = 0; f < 8; ++f) {                        // (compile-time offsets from two bases: a ds_write offset field is 16 bits)
        *reinterpret_cast<float4*>(b0 + t_block(f, 0, 0) * kBlk) = raw[f];
        *reinterpret_cast<float4*>(b1 + t_block(f, 0, 0) * kBlk) = raw[8 + f];
      }
    };
    // halo columns: thread (item, fz, fy) combines the 2 x 2 raw values its transformed entry is made of.  The values come from
    // the staging area, which only this wave's own lanes write (an item's 16 threads sit in one wave): the arithmetic runs BEFORE
    // barrier B1, only the store into T between the barriers (the section the MFMA waves wait for).
    auto halo_value = [&]() {
      const int fz = hzy >> 2, fy = hzy & 3;
      // Bt row f = +d[ia] + sb * d[ib]:   0: d0 - d2 | 1: d1 + d2 | 2: d2 - d1 | 3: d1 - d3
      const int za = (fz == 0) ? 0 : (fz == 2 ? 2 : 1), zb = (fz == 2) ? 1 : (fz == 3 ? 3 : 2);
      const int ya = (fy == 0) ? 0 : (fy == 2 ? 2 : 1), yb = (fy == 2) ? 1 : (fy == 3 ? 3 : 2);
      const float szb = (fz == 1) ? 1.f : -1.f, syb = (fy == 1) ? 1.f : -1.f;
      const float* sp = stage + hitem * 64;
      const float4 daa = *reinterpret_cast<const float4*>(sp + (za * 4 + ya) * 4), dab = *reinterpret_cast<const float4*>(sp + (za * 4 + yb) * 4);
      const float4 dba = *reinterpret_cast<const float4*>(sp + (zb * 4 + ya) * 4), dbb = *reinterpret_cast<const float4*>(sp + (zb * 4 + yb) * 4);
      return make_float4((daa.x + syb * dab.x) + szb * (dba.x + syb * dbb.x), (daa.y + syb * dab.y) + szb * (dba.y + syb * dbb.y),
                         (daa.z + syb * dab.z) + szb * (dba.z + syb * dbb.z), (daa.w + syb * dab.w) + szb * (dba.w + syb * dbb.w));
    };
    Row cur = setup(rowi);
#ifndef PGRF_WINO_ABL_NOB       // timing experiment (wrong results): the loader waves only keep the barrier protocol
    prefetch(cur, cur.c0);
#endif
    while (true) {
      const int nrow = rowi + gridDim.x;
      const bool has_next = nrow < nrows;
      for (int blk = cur.c0; blk < cur.c1; ++blk) {
#ifndef PGRF_WINO_ABL_NOB
        *reinterpret_cast<float4*>(stage + (hitem * 16 + hzy) * 4) = ((cur.zyok >> hzy) & 1u) ? hraw : make_float4(0.f, 0.f, 0.f, 0.f);
        transform(cur.zyok);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging writes of this wave's lanes (same wave: no barrier needed)
        __builtin_amdgcn_wave_barrier();
        float4 hval = halo_value();
        asm volatile("" : "+v"(hval.x), "+v"(hval.y), "+v"(hval.z), "+v"(hval.w));   // (pinned in front of the barrier)
#endif
        lds_barrier();                                     // B1
#ifndef PGRF_WINO_ABL_NOB
        // (s_setprio 3 around this section -- the one all four MFMA waves wait for -- measured +-0: UNet2 5.67 -> 5.70 ms)
        store();
        *reinterpret_cast<float4*>(T + tq(hzy, hcq, 2 * TX + hp)) = hval;
#endif
        lds_barrier();                                     // B2
#ifndef PGRF_WINO_ABL_NOB
        if (blk + 1 < cur.c1) prefetch(cur, blk + 1);
        else if (has_next) { const Row nr = setup(nrow); prefetch(nr, nr.c0); }
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
  const int fzw = wave;                                    // this wave's fz
  const int n = lane & 15, kq = lane >> 4;
  W16_DECL
  const float* tb = T + tq(fzw * 4, kq, 2 * n);            // fz = fzw, fy = 0, quad kq (+ 4 g), even column of tile n
  const unsigned lane16 = lane * 16;
  const char* wb = reinterpret_cast<const char*>(a.wpk);
  auto a_off = [&](int cg, int blk, int fyfx) {            // byte offset of the packed block (cout group, frequency, channel block)
    return (((unsigned)cg * 64u + fzw * 16 + fyfx) * (unsigned)a.chunks + blk) * (unsigned)kABlockBytes;
  };
  // A operands: a ring of three (output tile, frequency) slots x two k groups, refilled in place two tiles ahead of their use
  // (step s = 4 fyfx + tile uses slot s % 3)
#ifndef PGRF_W16_RING
#define PGRF_W16_RING 3
#endif
  float4 A[PGRF_W16_RING][2];
  const unsigned long long w_bits = (unsigned long long)a.wpk;
  const i32x4w wrsrc = {(int)(unsigned)w_bits, (int)(unsigned)(w_bits >> 32), (int)0x7fffffff, 0x00020000};      // raw buffer, 32-bit data
  auto a_load = [&](unsigned off) {
#ifdef PGRF_WINO_ABL_NOA      // timing experiment (wrong results): every weight refill reads the same 8 KB
    const f32x4v v = pgrf_w16_buffer_load_x4(wrsrc, (int)lane16, (int)(off & 0x1fffu), 0);
#else
    const f32x4v v = pgrf_w16_buffer_load_x4(wrsrc, (int)lane16, (int)off, 0);
#endif
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  // z-reduction buffer (reuses T): [fz][oy*2+ox][tile][lane] float4
  float4* red = reinterpret_cast<float4*>(lds);
  auto ridx = [&](int fz, int oyox, int t) { return ((fz * 4 + oyox) * 4 + t) * 64 + lane; };

  Row cur = setup(rowi);
  // byte offset of step s's operands (s = 4 fyfx + tile, may run past 63: the next block's / row's first steps)
  auto a_step = [&](unsigned abase, unsigned astride, unsigned anext, int st) {
    return (st < 64 ? abase + (unsigned)(st >> 2) * astride : anext) + (unsigned)(st & 3) * 2048u;
  };
#pragma unroll
  for (int st = 0; st < PGRF_W16_RING; ++st)
#pragma unroll
    for (int g = 0; g < 2; ++g) A[st][g] = a_load(a_off(cur.cg, cur.c0, 0) + st * 2048 + g * 1024);
  while (true) {
    const int nrow = rowi + gridDim.x;
    const bool has_next = nrow < nrows;
    const Row nxt = has_next ? setup(nrow) : cur;
    const int ncg = nxt.cg, nc0 = nxt.c0;
    // The bias enters through the accumulators of the fz = 1 wave: R1 appears with + in both z outputs.
    f32x4v R[2][2][4];                                     // [oy][ox][output tile]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4v binit = {0.f, 0.f, 0.f, 0.f};
      if (a.bias && fzw == 1) {
        const float4 b4 = *reinterpret_cast<const float4*>(lbias + cur.cg * 64 + t * 16 + kq * 4);
        binit = (f32x4v){b4.x, b4.y, b4.z, b4.w};
      }
#pragma unroll
      for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) R[oy][ox][t] = binit;
    }
    for (int blk = cur.c0; blk < cur.c1; ++blk) {
      const bool lastblk = blk + 1 == cur.c1;
      W16_TICK(0)
      lds_barrier();                                       // B1
      W16_TICK(1)
      lds_barrier();                                       // B2
      W16_TICK(2)
      const unsigned abase = a_off(cur.cg, blk, 0);
      const unsigned astride = (unsigned)a.chunks * (unsigned)kABlockBytes;
      const unsigned anext = lastblk ? a_off(ncg, nc0, 0) : abase + (unsigned)kABlockBytes;
      // 64 steps (frequency fyfx = s >> 2, output tile t = s & 3), software-pipelined by hand:
      //   * step s issues its 8 chained MFMAs into M[s & 1]; the output-transform adds of step s - 1 (VALU reads of an MFMA result)
      //     come AFTER them, so that no wait states are needed in front of the adds (the first version did the adds right behind
      //     their own chain: `s_nop 8/9` twice per tile and 60 % MFMA duty inside the phase);
      //   * the B operands of frequency f + 1: k group 0 is read from LDS behind tile 1 of frequency f and formed behind tile 2 into
      //     the other half of a two-deep buffer, k group 1 is read behind tile 2 and formed in place behind tile 3's last MFMAs --
      //     every ds_read has a whole tile (8 MFMAs) between its issue and its first use;
      //   * A operands two steps ahead (ring of three slots).
      typedef float f32x2v __attribute__((ext_vector_type(2)));
      float bv0[2][4], bv1[4];                             // k group 0 (two-deep, by frequency parity) and k group 1
      float4 tv[1][2];                                     // one k group's two T operands at a time
      f32x4v M[2];
      auto t_read = [&](auto g_tag, auto f_tag2) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value, fyfx = decltype(f_tag2)::value, fy = fyfx >> 2, fx = fyfx & 3;
        constexpr int o0 = (fx == 0) ? 0 : (fx == 2) ? 4 : XW * 4;
        constexpr int o1 = (fx == 0 || fx == 1) ? 4 : (fx == 2) ? XW * 4 : XW * 4 + 4;
        const float* tr = tb + (t_block(fy, 4 * g, 0)) * kBlk;      // fz = fzw (even slab pair start), quad 4 g + kq: kq's part is in tb
#ifdef PGRF_WINO_ABL_NOLDS    // timing experiment (wrong results): no LDS reads in the MFMA phase
        if constexpr (fyfx == 0 && g == 0) { tv[0][0] = *reinterpret_cast<const float4*>(tr + o0); tv[0][1] = *reinterpret_cast<const float4*>(tr + o1); }
        else { tv[0][0].x += 1.f; tv[0][1].y += 1.f; }
#else
        tv[0][0] = *reinterpret_cast<const float4*>(tr + o0);
        tv[0][1] = *reinterpret_cast<const float4*>(tr + o1);
#endif
      };
      auto b_form = [&](auto g_tag, auto f_tag2) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value, fyfx = decltype(f_tag2)::value, fy = fyfx >> 2, fx = fyfx & 3;
        constexpr bool corner = (fy == 0 || fy == 3) && (fx == 0 || fx == 3), neg = (fy == 3) != (fx == 3);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x2v p0 = h2 == 0 ? (f32x2v){tv[0][0].x, tv[0][0].y} : (f32x2v){tv[0][0].z, tv[0][0].w};
          const f32x2v p1 = h2 == 0 ? (f32x2v){tv[0][1].x, tv[0][1].y} : (f32x2v){tv[0][1].z, tv[0][1].w};
          f32x2v r;
          if constexpr (fx == 1) r = p0 + p1;
          else if constexpr (corner && neg) r = p1 - p0;      // (corner frequencies: the sign of their single output term is folded in)
          else r = p0 - p1;
          if constexpr (g == 0) { bv0[fyfx & 1][2 * h2] = r.x; bv0[fyfx & 1][2 * h2 + 1] = r.y; }
          else { bv1[2 * h2] = r.x; bv1[2 * h2 + 1] = r.y; }
        }
      };
      // output-transform adds of a finished step: At column f contributes to output 0 with + (f = 0, 1, 2) and to output 1 with
      // + (f = 1) or - (f = 2, 3); the four corner frequencies accumulated straight into their single target
      auto finish = [&](auto s_tag) __attribute__((always_inline)) {
        constexpr int st = decltype(s_tag)::value, fyfx = st >> 2, t = st & 3, fy = fyfx >> 2, fx = fyfx & 3;
#ifdef PGRF_WINO_ABL_NOADD    // timing experiment (wrong results): no output-transform adds
        if constexpr (st == 63) R[0][0][0] += M[1];
        return;
#endif
        constexpr bool corner = (fy == 0 || fy == 3) && (fx == 0 || fx == 3);
        if constexpr (corner) {
          R[fy == 3 ? 1 : 0][fx == 3 ? 1 : 0][t] = M[st & 1];
        } else {
#pragma unroll
          for (int oy = 0; oy < 2; ++oy) {
            if ((oy == 0 && fy == 3) || (oy == 1 && fy == 0)) continue;
            const bool ny = (oy == 1 && fy >= 2);
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
              if ((ox == 0 && fx == 3) || (ox == 1 && fx == 0)) continue;
              const bool nx = (ox == 1 && fx >= 2);
              if (ny != nx) acc_sub4(R[oy][ox][t], M[st & 1]); else acc_add4(R[oy][ox][t], M[st & 1]);
            }
          }
        }
      };
      t_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      b_form(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      t_read(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
      b_form(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
      static_for<64>([&](auto s_tag) __attribute__((always_inline)) {
        constexpr int st = decltype(s_tag)::value, fyfx = st >> 2, t = st & 3, fy = fyfx >> 2, fx = fyfx & 3;
        constexpr bool corner = (fy == 0 || fy == 3) && (fx == 0 || fx == 3);
        constexpr int slot = st % PGRF_W16_RING;
        const float4 a0 = A[slot][0], a1 = A[slot][1];
        f32x4v m;
        if constexpr (corner) m = R[fy == 3 ? 1 : 0][fx == 3 ? 1 : 0][t]; else m = (f32x4v){0.f, 0.f, 0.f, 0.f};
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bv0[fyfx & 1][0], m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bv0[fyfx & 1][1], m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bv0[fyfx & 1][2], m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bv0[fyfx & 1][3], m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bv1[0], m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bv1[1], m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bv1[2], m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bv1[3], m, 0, 0, 0);
        M[st & 1] = m;
        if constexpr (fyfx < 15) {
          if constexpr (t == 1) t_read(std::integral_constant<int, 0>{}, std::integral_constant<int, fyfx + 1>{});
          if constexpr (t == 2) {
            b_form(std::integral_constant<int, 0>{}, std::integral_constant<int, fyfx + 1>{});
            t_read(std::integral_constant<int, 1>{}, std::integral_constant<int, fyfx + 1>{});
          }
          if constexpr (t == 3) b_form(std::integral_constant<int, 1>{}, std::integral_constant<int, fyfx + 1>{});
        }
        // this slot's next operands: step st + 3 -- except behind the last three steps: 64 is not a multiple of the ring, so the slots
        // freed by steps 61, 62, 63 (1, 2, 0) take the NEXT block's steps 1, 2, 0, which is where that block looks for them (its
        // step 0 is requested last; the two barriers in front of the next MFMA phase cover its latency)
        {
          // (a ring of four divides 64: plain st + 4)
          constexpr int nst = (PGRF_W16_RING == 4) ? st + 4 : (st <= 60 ? st + 3 : 64 + (st - 60) % 3);
          const unsigned nb = a_step(abase, astride, anext, nst);
          A[slot][0] = a_load(nb);
          A[slot][1] = a_load(nb + 1024);
        }
        // (fence: the adds of the previous step stay BEHIND this step's MFMAs -- left alone the scheduler moves them to the top of
        //  the region, right behind their own chain, and pays the wait states again)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (st > 0) finish(std::integral_constant<int, st - 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      finish(std::integral_constant<int, 63>{});
      W16_TICK(3)                                          // MFMA phase
    }
    lds_barrier();                                         // E1: T is dead
    W16_TICK(4)
#pragma unroll
    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
      for (int ox = 0; ox < 2; ++ox)
#pragma unroll
        for (int t = 0; t < 4; ++t) red[ridx(fzw, oy * 2 + ox, t)] = make_float4(R[oy][ox][t][0], R[oy][ox][t][1], R[oy][ox][t][2], R[oy][ox][t][3]);
    W16_TICK(5)
    lds_barrier();                                         // E2: partials are in LDS
    W16_TICK(6)
    {
      // wave w finishes the 16 output channels t = w of all eight outputs (oz, oy, ox) of its tiles: the z output transform over the
      // four fz partials (out[oz=0] = R0 + R1 + R2, out[oz=1] = R1 - R2 - R3), LeakyReLU, stores.  (Round 4: was "wave = (oz, oy), all
      // four channel tiles" -- the same values, stores and store widths, but this way a lane holds a whole 2 x 2 x 2 output tile of its
      // four channels = one AvgPool3d(2) window: the pooled tensor of the encoder levels is one more store, not one more pass.)
      const int t = wave;
      const long long ov0 = (((long long)cur.b * a.D + cur.z0) * a.H + cur.y0) * a.W + cur.x0 + 2 * n;
      const int coff = a.out_coff + cur.cg * 64 + t * 16 + kq * 4;
      const bool x_in = cur.x0 + 2 * n < a.W;               // (W = 16: tiles 8..15 of the row are the wrapped repeat of tiles 0..7)
      float* const outp = a.out + (long long)cur.split * a.ksplit_stride;       // (split-K: this item's partial-sum buffer)
      float4 pool = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int oyox = 0; oyox < 4; ++oyox) {
        const float4 p0 = red[ridx(0, oyox, t)], p1 = red[ridx(1, oyox, t)], p2 = red[ridx(2, oyox, t)], p3 = red[ridx(3, oyox, t)];
        float4 v0 = f4add(f4add(p0, p1), p2), v1 = f4sub(f4sub(p1, p2), p3);
        v0.x = v0.x > 0.f ? v0.x : v0.x * a.slope; v0.y = v0.y > 0.f ? v0.y : v0.y * a.slope;
        v0.z = v0.z > 0.f ? v0.z : v0.z * a.slope; v0.w = v0.w > 0.f ? v0.w : v0.w * a.slope;
        v1.x = v1.x > 0.f ? v1.x : v1.x * a.slope; v1.y = v1.y > 0.f ? v1.y : v1.y * a.slope;
        v1.z = v1.z > 0.f ? v1.z : v1.z * a.slope; v1.w = v1.w > 0.f ? v1.w : v1.w * a.slope;
        const long long ov = ov0 + (long long)(oyox >> 1) * a.W + (oyox & 1);
        if (x_in) {
          *reinterpret_cast<float4*>(outp + ov * a.out_cs + coff) = v0;
          *reinterpret_cast<float4*>(outp + (ov + (long long)a.H * a.W) * a.out_cs + coff) = v1;
        }
        pool = f4add(pool, f4add(v0, v1));
      }
      if (a.pool && x_in) {
        const long long pv = (((long long)cur.b * (a.D >> 1) + (cur.z0 >> 1)) * (a.H >> 1) + (cur.y0 >> 1)) * (a.W >> 1) + (cur.x0 >> 1) + n;
        *reinterpret_cast<float4*>(a.pool + pv * a.pool_cs + a.pool_coff + cur.cg * 64 + t * 16 + kq * 4) =
            make_float4(pool.x * 0.125f, pool.y * 0.125f, pool.z * 0.125f, pool.w * 0.125f);
      }
    }
    W16_TICK(7)
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

// split-K epilogue: out = LeakyReLU(sum_s partial[s] + bias) over channel-last voxels (four channels per thread)
__global__ void __launch_bounds__(256) wino16_splitk_reduce_kernel(const float* __restrict__ part, long long stride, int S, const float* __restrict__ bias,
                                                                   float* __restrict__ out, int out_cs, int out_coff, long long nvox, int Cout, float slope) {
  const int c4n = Cout >> 2;
  const long long n = nvox * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long v = i / c4n; const int c = 4 * (int)(i % c4n);
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
      const float4 p = *reinterpret_cast<const float4*>(part + s * stride + v * Cout + c);
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    acc.x = acc.x > 0.f ? acc.x : acc.x * slope; acc.y = acc.y > 0.f ? acc.y : acc.y * slope;
    acc.z = acc.z > 0.f ? acc.z : acc.z * slope; acc.w = acc.w > 0.f ? acc.w : acc.w * slope;
    *reinterpret_cast<float4*>(out + v * out_cs + out_coff + c) = acc;
  }
}

static constexpr int kMaxDevices16 = 64;

extern "C" {

size_t pgrf_conv3d_wino16_packed_floats(int Cout, int Cin) {
  const size_t cg = (Cout + 63) / 64, chunks = (Cin + 31) / 32;
  return cg * 64 * chunks * 2048 + cg * 64;
}

int pgrf_pack_conv3d_wino16(const float* w_host, const float* bias_host, int Cout, int Cin, float* packed_host) {
  if (!w_host || !packed_host) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d_wino16: null pointer");
  if (Cout < 1 || Cin < 1) return set_error(PGRF_ERR_ARG, "pgrf_pack_conv3d_wino16: bad channels");
  const int cgs = (Cout + 63) / 64, chunks = (Cin + 31) / 32;
  std::memset(packed_host, 0, sizeof(float) * pgrf_conv3d_wino16_packed_floats(Cout, Cin));
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
  for (int cg = 0; cg < cgs; ++cg)
    for (int f = 0; f < 64; ++f)
      for (int ch = 0; ch < chunks; ++ch)
        for (int t = 0; t < 4; ++t)
          for (int g = 0; g < 2; ++g)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 4; ++j) {
                const int co = 64 * cg + 16 * t + (lane & 15), ci = 32 * ch + w16::k_channel(g, lane >> 4, j);
                if (co < Cout && ci < Cin) packed_host[w16::wpk16_index(cg, f, ch, chunks, t, g, lane, j)] = U[((size_t)co * Cin + ci) * 64 + f];
              }
  float* B = packed_host + (size_t)cgs * 64 * chunks * 2048;
  for (int co = 0; co < cgs * 64; ++co) B[co] = (bias_host && co < Cout) ? bias_host[co] : 0.f;
  return PGRF_OK;
}

int pgrf_conv3d_wino16_supported(int Cin, int Cout, int D, int H, int W) {
  return (Cin % 32 == 0) && (Cout % 64 == 0) && (Cout <= w16::kBiasFloats) && (D % 2 == 0) && (H % 2 == 0) && (W % 32 == 0 || W == 16) && D >= 2 && H >= 2 && W >= 16;
}

int pgrf_conv3d_wino16_preferred(int Cin, int Cout, int B, int D, int H, int W) {
  if (!pgrf_conv3d_wino16_supported(Cin, Cout, D, H, W)) return 0;
  // persistent, two workgroups per CU, a row (16 tiles x 64 output channels) is the unit of work: below ~200 rows most of the
  // chip idles and the split-K direct kernel wins
  const long long rows = (long long)B * (D / 2) * (H / 2) * ((W + 31) / 32) * (Cout / 64);
  return rows >= 192 ? 1 : 0;
}

int pgrf_conv3d_wino16(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                       const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                       int res_cs, int res_coff, void* stream) {
  return pgrf_conv3d_wino16_pool(in, in_cs, in_coff, Cin, out, out_cs, out_coff, Cout, packed, has_bias, B, D, H, W, lrelu_slope, residual, res_cs,
                                 res_coff, nullptr, 0, 0, stream);
}

int pgrf_conv3d_wino16_ksplit(int Cin, int Cout, int B, int D, int H, int W) {
  if (!pgrf_conv3d_wino16_supported(Cin, Cout, D, H, W)) return 0;
  // rows (16 tiles x 64 output channels) are the kernel's unit of work and it wants two workgroups per CU: a layer with fewer than ~1.5 rows
  // per CU gives every row to several work items, each over a share of the 32-channel input blocks
  const long long rows = (long long)B * (D / 2) * (H / 2) * ((W + 31) / 32) * (Cout / 64);
  const int chunks = Cin / 32;
  static const long long max_rows = [] { const char* e = getenv("PGRF_WINO16_KSPLIT_ROWS"); const long long v = e ? atoll(e) : 0; return v > 0 ? v : 384; }();   // tuning knobs (measured: rows < 384 / 512 items 5.42 | 1.62 ms, < 640 / 1024: 5.46 | 1.67, < 1100 / 2048: 5.58 | 1.77 -- UNet2 at 128x256x64 | 64x128x64)
  static const long long items = [] { const char* e = getenv("PGRF_WINO16_KSPLIT_ITEMS"); const long long v = e ? atoll(e) : 0; return v > 0 ? v : 512; }();
  if (rows <= 0 || rows >= max_rows) return 0;
  long long s = (items + rows - 1) / rows;
  if (s > chunks) s = chunks;
  if (s > 8) s = 8;
  return s >= 2 ? (int)s : 0;
}

static int wino16_launch(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                         const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                         int res_cs, int res_coff, float* pool_out, int pool_cs, int pool_coff, int ksplit, float* scratch, void* stream);

int pgrf_conv3d_wino16_pool(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                            const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                            int res_cs, int res_coff, float* pool_out, int pool_cs, int pool_coff, void* stream) {
  return wino16_launch(in, in_cs, in_coff, Cin, out, out_cs, out_coff, Cout, packed, has_bias, B, D, H, W, lrelu_slope, residual, res_cs, res_coff,
                       pool_out, pool_cs, pool_coff, 1, nullptr, stream);
}

int pgrf_conv3d_wino16_splitk(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                              const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, int ksplit,
                              float* scratch, size_t scratch_floats, void* stream) {
  if (ksplit < 2 || ksplit > Cin / 32) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_wino16_splitk: ksplit %d outside [2, Cin / 32 = %d]", ksplit, Cin / 32);
  const size_t need = (size_t)ksplit * B * D * H * W * Cout;
  if (!scratch || scratch_floats < need) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_wino16_splitk: scratch of %zu floats needed", need);
  return wino16_launch(in, in_cs, in_coff, Cin, out, out_cs, out_coff, Cout, packed, has_bias, B, D, H, W, lrelu_slope, nullptr, 0, 0,
                       nullptr, 0, 0, ksplit, scratch, stream);
}

static int wino16_launch(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                         const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                         int res_cs, int res_coff, float* pool_out, int pool_cs, int pool_coff, int ksplit, float* scratch, void* stream) {
  if (!in || !out || !packed) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_wino16: null pointer");
  if (!pgrf_conv3d_wino16_supported(Cin, Cout, D, H, W))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino16: needs Cin a multiple of 32, Cout a multiple of 64 (<= %d), even D and H, W %% 32 == 0 or W == 16 "
                     "(got Cin %d Cout %d D %d H %d W %d); use pgrf_conv3d_wrap", w16::kBiasFloats, Cin, Cout, D, H, W);
  if (B < 1) return set_error(PGRF_ERR_ARG, "pgrf_conv3d_wino16: bad dims");
#ifndef PGRF_WINO_TIMING
  if (residual) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino16: no residual epilogue (use pgrf_conv3d_wrap)");
#endif
  if ((double)(Cout / 64) * 64.0 * (Cin / 32) * 8192.0 >= 2147483647.0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino16: packed weights of >= 2 GiB (buffer descriptor)");
  if ((double)D * H * W * in_cs * 4.0 >= 4294967296.0)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino16: one batch element of the input is >= 4 GiB (32-bit byte offsets)");
  if ((in_cs & 3) || (in_coff & 3) || (out_cs & 3) || (out_coff & 3) || (pool_out && ((pool_cs & 3) || (pool_coff & 3))))
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino16: channel strides/offsets must be multiples of 4");
  ConvArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = in; a.in_cs = in_cs; a.in_coff = in_coff; a.Cin = Cin; a.out = out; a.out_cs = out_cs; a.out_coff = out_coff;
  a.Cout = Cout; a.B = B; a.D = D; a.H = H; a.W = W; a.Do = D; a.Ho = H; a.Wo = W; a.stride = 1; a.slope = lrelu_slope;
  a.chunks = Cin / 32; a.cot_total = Cout / 32;
  a.residual = residual; a.res_cs = res_cs; a.res_coff = res_coff; a.wtaps = 64; a.wrap_w = 1;
  a.wpk = packed;
  a.pool = pool_out; a.pool_cs = pool_cs; a.pool_coff = pool_coff;
  a.bias = has_bias ? packed + (size_t)(Cout / 64) * 64 * a.chunks * 2048 : nullptr;
  const float* bias_dev = a.bias;
  const long long nvox = (long long)B * D * H * W;
  if (ksplit > 1) {          // raw partial sums into the scratch buffers; bias and activation in the reduce kernel behind
    a.out = scratch; a.out_cs = Cout; a.out_coff = 0; a.slope = 1.f; a.bias = nullptr; a.pool = nullptr;
    a.ksplit = ksplit; a.ksplit_stride = nvox * Cout;
  }
  const long long nrows_ll = (long long)B * (D / 2) * (H / 2) * ((W + 31) / 32) * (Cout / 64) * (ksplit > 1 ? ksplit : 1);
  if (nrows_ll >= (1ll << 31)) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_conv3d_wino16: too many tile rows");
  const int nrows = (int)nrows_ll;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices16)
    return set_error(PGRF_ERR_LAUNCH, "pgrf_conv3d_wino16: cannot query the current device");
  int ncu;
  {
    static std::mutex mu;
    static int ncu_of[kMaxDevices16] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (ncu_of[dev] == 0) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wino16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, w16::kLdsBytes) != hipSuccess)
        return set_error(PGRF_ERR_LAUNCH, "pgrf_conv3d_wino16: cannot reserve %d bytes of LDS", w16::kLdsBytes);
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return set_error(PGRF_ERR_LAUNCH, "pgrf_conv3d_wino16: cannot query the device");
      ncu_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    ncu = ncu_of[dev];
  }
  // persistent: two workgroups (4 MFMA waves + 4 loader waves, 78 KB of LDS each) per CU, each loops over rows
  int wgs = 2 * ncu;
  if (const char* e = getenv("PGRF_WINO16_WGS")) { const int v = atoi(e); if (v > 0) wgs = v; }          // tuning knob
  dim3 grid((unsigned)(nrows < wgs ? nrows : wgs));
  hipLaunchKernelGGL(conv3d_wino16_kernel, grid, dim3(w16::kThreads), w16::kLdsBytes, (hipStream_t)stream, a, nrows);
  PGRF_CHECK_LAUNCH("conv3d_wino16");
  if (ksplit > 1) {
    const long long n4 = nvox * (Cout / 4);
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino16_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, a.ksplit_stride, ksplit, bias_dev, out, out_cs,
                       out_coff, nvox, Cout, lrelu_slope);
    PGRF_CHECK_LAUNCH("conv3d_wino16 split-K reduce");
  }
  return PGRF_OK;
}

}  // extern "C"
