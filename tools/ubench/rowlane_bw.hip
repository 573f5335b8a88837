// Micro-benchmark (round 4): does a "row per lane" access (each lane reads / writes the 16-byte quads of ITS OWN 128-byte row, the pattern of the
// training chain kernels and of lin_mfma_kernel) reach the same HBM rate as a coalesced one (8 lanes per row)?   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>   // 0: row per lane, 1: coalesced
__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, long rows) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  for (long t = wave; t * 64 < rows; t += nw) {         // a tile = 64 rows x 128 B = 8 KB
    const long r0 = t * 64;
    float4 v[8];
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = in[(r0 + lane) * 8 + q];
#pragma unroll
      for (int q = 0; q < 8; ++q) { v[q].x += 1.f; out[(r0 + lane) * 8 + q] = v[q]; }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = in[r0 * 8 + q * 64 + lane];
#pragma unroll
      for (int q = 0; q < 8; ++q) { v[q].x += 1.f; out[r0 * 8 + q * 64 + lane] = v[q]; }
    }
  }
}
int main() {
  const long rows = 1 << 21;                 // 2 M rows x 128 B = 256 MB in, 256 MB out
  float4 *in, *out;
  hipMalloc(&in, rows * 128); hipMalloc(&out, rows * 128);
  hipMemset(in, 0, rows * 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int blocks : {1024, 2048, 4096}) {
      float best = 1e9f;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(copy_kernel<0>, dim3(blocks), dim3(256), 0, 0, in, out, rows);
        else hipLaunchKernelGGL(copy_kernel<1>, dim3(blocks), dim3(256), 0, 0, in, out, rows);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
      printf("%s  blocks %5d: %.1f us = %.2f TB/s (read + write)\n", mode == 0 ? "row per lane" : "coalesced   ", blocks, best * 1e3, 2.0 * rows * 128 / (best * 1e-3) / 1e12);
    }
  return 0;
}
