// Does VALU work issued between dependent f32 MFMAs overlap with them on gfx950?  1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, bool DO_M, bool TRANS>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = threadIdx.x * 0.001f; acc1[r] = threadIdx.x * 0.002f; }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  float a = threadIdx.x * 0.5f, b = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (DO_M) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (TRANS) v[j & 7] = __expf(v[j & 7]) * 0.5f;
        else v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool DO_M, bool TRANS>
float run(int blocks_per_cu, int iters, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, DO_M, TRANS><<<256 * blocks_per_cu, 256>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NV, DO_M, TRANS><<<256 * blocks_per_cu, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 2000;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    printf("waves/SIMD=%d  (per iteration of 8 x [2 MFMA + NV VALU]; cycles @2.4GHz per [2 MFMA + NV valu] group)\n", bpc);
#define R(NV, M, T) { float ms = run<NV, M, T>(bpc, iters, d); printf("  NV=%2d mfma=%d trans=%d : %7.3f ms  -> %6.1f cyc/group/wave-slot\n", NV, M, T, ms, ms * 1e-3 * 2.4e9 / (iters * 8.0) / bpc * bpc); }
    R(0, true, false) R(8, false, false) R(8, true, false) R(16, false, false) R(16, true, false) R(24, true, false) R(32, false, false) R(32, true, false)
    R(8, false, true) R(8, true, true) R(16, true, true)
  }
  return 0;
}
