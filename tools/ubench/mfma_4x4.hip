// Rate of the small f32 MFMAs on gfx950: v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4x1, 512 flop) and
// v_mfma_f32_16x16x4_f32 (2048 flop) against v_mfma_f32_32x32x2_f32 (4096 flop); NACC independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x4 a4[NACC]; f32x16 a16[NACC];
  for (int a = 0; a < NACC; ++a) { for (int r = 0; r < 4; ++r) a4[a][r] = threadIdx.x * 0.001f + a; for (int r = 0; r < 16; ++r) a16[a][r] = threadIdx.x * 0.002f + a; }
  float x = threadIdx.x * 0.5f, y = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (KIND == 0) a4[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a4[u % NACC], 0, 0, 0);
      if (KIND == 1) a4[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[u % NACC], 0, 0, 0);
      if (KIND == 2) a16[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a16[u % NACC], 0, 0, 0);
    }
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a) { for (int r = 0; r < 4; ++r) s += a4[a][r]; for (int r = 0; r < 16; ++r) s += a16[a][r]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NACC>
void run(const char* name, float* d, double flop) {
  const int iters = 2000;
  printf("%-28s NACC=%d :", name, NACC);
  for (int bpc = 1; bpc <= 2; ++bpc) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND, NACC><<<256 * bpc, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND, NACC><<<256 * bpc, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 32.0 * bpc);
    printf("  %dw/SIMD %6.2f cyc/MFMA (%5.1f flop/cyc/SIMD)", bpc, cyc, flop / cyc);
  }
  printf("\n");
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0, 1>("v_mfma_f32_4x4x1_16b_f32", d, 512); run<0, 2>("v_mfma_f32_4x4x1_16b_f32", d, 512); run<0, 4>("v_mfma_f32_4x4x1_16b_f32", d, 512); run<0, 8>("v_mfma_f32_4x4x1_16b_f32", d, 512);
  run<1, 1>("v_mfma_f32_16x16x4_f32", d, 2048); run<1, 2>("v_mfma_f32_16x16x4_f32", d, 2048); run<1, 4>("v_mfma_f32_16x16x4_f32", d, 2048);
  run<2, 1>("v_mfma_f32_32x32x2_f32", d, 4096); run<2, 2>("v_mfma_f32_32x32x2_f32", d, 4096);
  return 0;
}
