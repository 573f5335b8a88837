// Do VALU ops overlap with bf16 MFMAs (32x32x16) on gfx950, for dependent accumulator chains? 1 and 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = threadIdx.x * 0.001f + a;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  bf16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (short)(0x3f80 + threadIdx.x + i); B[i] = (short)(0x3f00 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[m % NACC], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int NACC>
void run(int bpc, int iters, float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NM, NV, NACC><<<256 * bpc, 256>>>(d, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NM, NV, NACC><<<256 * bpc, 256>>>(d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("  waves/SIMD=%d NM=%2d NV=%2d NACC=%d : %7.3f ms -> %7.1f cyc per [NM mfma + NV valu] group\n", bpc, NM, NV, NACC, ms,
         ms * 1e-3 * 2.4e9 / (iters * 4.0));
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 2000;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<6, 0, 1>(bpc, iters, d); run<6, 0, 2>(bpc, iters, d); run<0, 32, 1>(bpc, iters, d); run<6, 32, 1>(bpc, iters, d); run<6, 32, 2>(bpc, iters, d);
    run<6, 64, 2>(bpc, iters, d); run<3, 32, 1>(bpc, iters, d); run<12, 64, 2>(bpc, iters, d);
  }
  return 0;
}
