// MFMA/VALU overlap measured in SHADER CYCLES (s_memtime via clock64), immune to DVFS: per-wave cycles for a
// loop of [NM MFMA (f32 32x32x2 or bf16 32x32x16), NV independent v_fma], interleaved (I=1) or blocked (I=0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF, int NM, int NV, bool INTER>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = threadIdx.x * 0.001f; acc1[r] = threadIdx.x * 0.002f; }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  bf16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (short)(0x3f80 + threadIdx.x + i); B[i] = (short)(0x3f00 + i); }
  float a = threadIdx.x * 0.5f, b = 1.0001f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    constexpr int VPM = NM > 0 ? NV / NM : NV;
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
      if (NM > 0) {
        if (BF) { if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0); else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0); }
        else { if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0); else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0); }
      }
      if (INTER) {
#pragma unroll
        for (int j = 0; j < VPM; ++j) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!INTER) {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <bool BF, int NM, int NV, bool INTER>
void run(int bpc, float* d, long long* c) {
  const int iters = 2000;
  k<BF, NM, NV, INTER><<<256 * bpc, 256>>>(d, c, iters);
  (void)hipDeviceSynchronize();
  k<BF, NM, NV, INTER><<<256 * bpc, 256>>>(d, c, iters);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("  w/SIMD=%d %s NM=%2d NV=%2d %s : %8.1f clk64-ticks per iteration\n", bpc, BF ? "bf16" : "f32 ", NM, NV, INTER ? "interleaved" : "blocked    ",
         (double)h / iters);
}

int main() {
  float* d; long long* c; (void)hipMalloc(&d, 256 * 8 * 256 * 4); (void)hipMalloc(&c, 8);
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<false, 4, 0, false>(bpc, d, c); run<false, 0, 32, false>(bpc, d, c); run<false, 4, 32, false>(bpc, d, c); run<false, 4, 32, true>(bpc, d, c);
    run<true, 4, 0, false>(bpc, d, c); run<true, 4, 32, false>(bpc, d, c); run<true, 4, 32, true>(bpc, d, c); run<true, 8, 32, true>(bpc, d, c);
  }
  return 0;
}
