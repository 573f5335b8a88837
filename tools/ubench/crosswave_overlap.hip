// One 512-thread workgroup per CU = 2 waves per SIMD.  Waves 0-3 run only MFMAs, waves 4-7 only VALU FMAs.
// If the matrix pipe and the VALU are concurrent ACROSS waves, time(both) ~ max(time(mfma), time(valu)).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ void __launch_bounds__(512) k(float* out, int mf_iters, int va_iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float s = 0;
  if (wave < 4) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = threadIdx.x * 0.001f; acc1[r] = threadIdx.x * 0.002f; }
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (short)(0x3f80 + threadIdx.x + i); B[i] = (short)(0x3f00 + i); }
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    for (int it = 0; it < mf_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (BF) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B, A, acc1, 0, 0, 0); }
        else { acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0); }
      }
    }
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    for (int it = 0; it < va_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 64; ++j) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
    }
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool BF>
float run(float* d, int mf, int va) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<BF><<<256, 512>>>(d, mf > 0 ? 10 : 0, va > 0 ? 10 : 0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<BF><<<256, 512>>>(d, mf, va);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  const int mf = 4000;
  for (int bf = 0; bf < 2; ++bf) {
    const float tm = bf ? run<true>(d, mf, 0) : run<false>(d, mf, 0);
    // choose VALU iterations so that VALU-only time ~ MFMA-only time
    const float tv1 = bf ? run<true>(d, 0, 4000) : run<false>(d, 0, 4000);
    const int va = (int)(4000 * tm / tv1);
    const float tv = bf ? run<true>(d, 0, va) : run<false>(d, 0, va);
    const float tb = bf ? run<true>(d, mf, va) : run<false>(d, mf, va);
    printf("%s: mfma-only %.3f ms, valu-only %.3f ms, both (different waves, same SIMDs) %.3f ms -> overlap %.0f %%\n", bf ? "bf16 32x32x16" : "f32 32x32x2 ",
           tm, tv, tb, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
  }
  return 0;
}
