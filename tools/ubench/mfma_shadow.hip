// How many single-issue VALU instructions hide in the shadow of one MFMA on gfx950?  Pattern per step:
// [1 MFMA (rotating over NACC accumulators) + NV plain VALU (inline asm, so the compiler cannot SLP-pack them)].
// Reports cycles per step for f32 32x32x2 (64-cycle) and f16 32x32x16 (32-cycle) MFMAs, 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NACC, bool F16, int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = threadIdx.x * 0.001f + a;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  float a32 = threadIdx.x * 0.5f, b32 = 1.0001f;
  f16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(0.5f + i); B[i] = (_Float16)(0.25f * i); }
  const float c = 0.999f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (F16) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[u % NACC], 0, 0, 0);
      else acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a32, b32, acc[u % NACC], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(c));
        if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
        if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&v[(j & 3) * 2]) : "v"(1.0));
      }
    }
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NACC, bool F16, int KIND>
void run(float* d) {
  const int iters = 2000;
  printf("%s NACC=%d kind=%d NV=%2d :", F16 ? "f16 32x32x16" : "f32 32x32x2 ", NACC, KIND, NV);
  for (int bpc = 1; bpc <= 2; ++bpc) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, NACC, F16, KIND><<<256 * bpc, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NV, NACC, F16, KIND><<<256 * bpc, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %dw/SIMD %6.1f cyc/step/wave", bpc, ms * 1e-3 * 2.4e9 / (iters * 16.0 * bpc));
  }
  printf("\n");
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0, 2, false, 0>(d); run<4, 2, false, 0>(d); run<8, 2, false, 0>(d); run<12, 2, false, 0>(d); run<16, 2, false, 0>(d); run<24, 2, false, 0>(d);
  run<0, 1, false, 0>(d); run<8, 1, false, 0>(d); run<16, 1, false, 0>(d);
  run<0, 4, false, 0>(d); run<8, 4, false, 0>(d); run<16, 4, false, 0>(d);
  run<4, 2, false, 1>(d); run<8, 2, false, 1>(d);
  run<4, 2, false, 2>(d); run<8, 2, false, 2>(d);
  run<0, 2, true, 0>(d); run<2, 2, true, 0>(d); run<4, 2, true, 0>(d); run<6, 2, true, 0>(d); run<8, 2, true, 0>(d); run<16, 2, true, 0>(d);
  run<0, 4, true, 0>(d); run<4, 4, true, 0>(d); run<8, 4, true, 0>(d);
  run<4, 2, true, 1>(d);
  return 0;
}
