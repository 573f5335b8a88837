// What does one more VALU instruction cost a SIMD that is also running f32 MFMAs, with TWO waves per SIMD (the render kernel's
// regime)?  256 workgroups x 512 threads = one workgroup per CU, two waves per SIMD; every wave loops over
//     [ M x v_mfma_f32_32x32x2_f32 on two accumulators ]  [ N x instruction of kind K on 8 independent registers ]
// PHASE = 0: both waves of a SIMD run the same order; PHASE = 1: the younger wave of each SIMD runs [VALU][MFMA] instead, so that one
// wave's VALU block faces the other's MFMA block.  Reported: cycles per iteration and SIMD (2 waves -> wall / (2 x iterations)),
// with the clock calibrated on the MFMA-only loop (64 cycles per MFMA and SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Kind { K_MUL, K_ADD, K_FMA, K_EXP, K_MED3, K_PKFMA, K_SWAP32, K_SWAP16, K_MOV, K_DSREAD, K_RCP, K_MFMA16, K_MFMA4, K_FMAC, K_MAX, K_CNDMASK, K_PKMUL, K_PKADD, K_DSREAD32, K_EXPCLAMP, K_MADAK, K_ADD3OP, K_SUBREV, K_MULDPP, K_LOG, K_COUNT };
static const char* kNames[] = {"v_mul_f32", "v_add_f32", "v_fma_f32", "v_exp_f32", "v_med3_f32", "v_pk_fma_f32", "v_permlane32_swap", "v_permlane16_swap",
                               "v_mov_b32", "ds_read_b128", "v_rcp_f32", "v_mfma_16x16x4_f32", "v_mfma_4x4x1_f32", "v_fmac_f32 (VOP2)", "v_max_f32", "v_cndmask_b32", "v_pk_mul_f32", "v_pk_add_f32", "ds_read_b32", "v_exp_f32 clamp", "v_fmaak_f32", "v_add_f32_e64 (VOP3)", "v_subrev_f32", "v_mul_f32_dpp", "v_log_f32"};

template <int KIND, int N>
__device__ __forceinline__ void valu_block(float (&v)[8], double (&pk)[4], float c, const float4* lds, f32x4 (&d16)[2]) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
    if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j & 7]));
    if (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[j & 3]) : "v"(1.0));
    if (KIND == K_SWAP32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[j & 3]), "+v"(v[4 + (j & 3)]));
    if (KIND == K_SWAP16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v[j & 3]), "+v"(v[4 + (j & 3)]));
    if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j & 7]) : "v"(c));
    if (KIND == K_DSREAD) {
      f32x4 t;
      asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(threadIdx.x * 16)));
      if ((j & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" :: "v"(t));
    }
    if (KIND == K_FMAC) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[j & 3]) : "v"(1.0));
    if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[j & 3]) : "v"(1.0));
    if (KIND == K_EXPCLAMP) asm volatile("v_exp_f32_e64 %0, %0 clamp" : "+v"(v[j & 7]));
    if (KIND == K_MADAK) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800000" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_ADD3OP) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_SUBREV) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_MULDPP) asm volatile("v_mul_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[j & 7]) : "v"(c));
    if (KIND == K_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(v[j & 7]));
    if (KIND == K_DSREAD32) {
      float t;
      asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((unsigned)(threadIdx.x * 4)));
      if ((j & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" :: "v"(t));
    }
    if (KIND == K_MFMA16) d16[j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j & 7], c, d16[j & 1], 0, 0, 0);
    if (KIND == K_MFMA4) d16[j & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[j & 7], c, d16[j & 1], 0, 0, 0);
  }
  if (KIND == K_DSREAD || KIND == K_DSREAD32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int KIND, int M, int N, int PHASE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  __shared__ float4 lds[512];
  lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = threadIdx.x * 0.001f; acc1[r] = threadIdx.x * 0.002f; }
  f32x4 d16[2] = {{0.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}};
  float v[8];
  double pk[4];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  for (int i = 0; i < 4; ++i) pk[i] = 1.0 + i;
  const float a = threadIdx.x * 0.5f, b = 1.0001f, c = 0.999f;
  const bool late = PHASE && (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256);
  if (late) valu_block<KIND, N>(v, pk, c, lds, d16);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < M / 2; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    valu_block<KIND, N>(v, pk, c, lds, d16);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += (float)pk[i];
  for (int i = 0; i < 4; ++i) s += d16[0][i] + d16[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int M, int N, int PHASE>
float run(float* d, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND, M, N, PHASE><<<256, 512>>>(d, 20);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<KIND, M, N, PHASE><<<256, 512>>>(d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

static double g_cyc_per_ms;   // SIMD cycles per millisecond, from the MFMA-only loop

template <int KIND, int M>
void sweep(float* d) {
  const int iters = 3000;
  const double t0 = run<KIND, M, 0, 0>(d, iters);
  const double c0 = t0 * g_cyc_per_ms / (2.0 * iters);
  double c[2][3];
  c[0][0] = run<KIND, M, 16, 0>(d, iters) * g_cyc_per_ms / (2.0 * iters);
  c[0][1] = run<KIND, M, 64, 0>(d, iters) * g_cyc_per_ms / (2.0 * iters);
  c[0][2] = run<KIND, M, 128, 0>(d, iters) * g_cyc_per_ms / (2.0 * iters);
  c[1][0] = run<KIND, M, 16, 1>(d, iters) * g_cyc_per_ms / (2.0 * iters);
  c[1][1] = run<KIND, M, 64, 1>(d, iters) * g_cyc_per_ms / (2.0 * iters);
  c[1][2] = run<KIND, M, 128, 1>(d, iters) * g_cyc_per_ms / (2.0 * iters);
  printf("%-20s M=%2d  base %7.1f | same phase N=16 %7.1f  64 %7.1f  128 %7.1f  -> %5.2f cyc/instr | opposite phase N=16 %7.1f  64 %7.1f  128 %7.1f -> %5.2f cyc/instr\n",
         kNames[KIND], M, c0, c[0][0], c[0][1], c[0][2], (c[0][2] - c[0][0]) / 112.0, c[1][0], c[1][1], c[1][2], (c[1][2] - c[1][0]) / 112.0);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  {
    const int iters = 3000;
    const double t = run<K_MUL, 16, 0, 0>(d, iters);
    g_cyc_per_ms = 2.0 * iters * 16 * 64.0 / t;      // 2 waves x 16 MFMAs x 64 cycles per iteration and SIMD
    printf("calibration: MFMA-only loop %.3f ms -> %.3f GHz if one 32x32x2 f32 MFMA = 64 cycles per SIMD\n", t, g_cyc_per_ms / 1e6);
  }
  sweep<K_MUL, 16>(d); sweep<K_ADD, 16>(d); sweep<K_FMA, 16>(d); sweep<K_EXP, 16>(d); sweep<K_RCP, 16>(d); sweep<K_MED3, 16>(d); sweep<K_PKFMA, 16>(d);
  sweep<K_SWAP32, 16>(d); sweep<K_SWAP16, 16>(d); sweep<K_MOV, 16>(d); sweep<K_DSREAD, 16>(d); sweep<K_MFMA16, 16>(d); sweep<K_MFMA4, 16>(d);
  sweep<K_FMAC, 16>(d); sweep<K_MAX, 16>(d); sweep<K_CNDMASK, 16>(d); sweep<K_PKMUL, 16>(d); sweep<K_PKADD, 16>(d); sweep<K_DSREAD32, 16>(d);
  sweep<K_EXPCLAMP, 16>(d); sweep<K_MADAK, 16>(d); sweep<K_ADD3OP, 16>(d); sweep<K_SUBREV, 16>(d); sweep<K_MULDPP, 16>(d); sweep<K_LOG, 16>(d);
  // no MFMAs at all: the plain VALU rates at two waves per SIMD
  printf("-- M = 0 (no 32x32x2 MFMAs) --\n");
  g_cyc_per_ms = g_cyc_per_ms;
  sweep<K_MUL, 0>(d); sweep<K_FMA, 0>(d); sweep<K_EXP, 0>(d); sweep<K_PKFMA, 0>(d); sweep<K_MFMA16, 0>(d); sweep<K_MFMA4, 0>(d);
  return 0;
}
