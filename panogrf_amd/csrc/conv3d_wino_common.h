// Helpers shared by the two Winograd kernels (conv3d_wino.hip: 32-tile rows, one 16-wave workgroup per CU; conv3d_wino16.hip:
// 16-tile rows, two independent 8-wave workgroups per CU).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "conv3d_common.h"

namespace pgrf {

// (the same on packed adds, through vector types: UNet2 6.60 -> 6.68 ms -- the loaders' transform is not on the MFMA waves' path)
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) (the 64-group loop below is too large
// for `#pragma unroll` and must not become a run-time loop: its indices select registers)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// 16-byte global load as  scalar base (SGPR pair) + ONE per-lane 32-bit byte offset  (global_load_dwordx4 v, voff, s[..]).
// The base is made opaque, otherwise the compiler reassociates base + scalar + lane offset into a 64-bit per-lane pointer per
// load, hoists them out of the loops and spills them.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const char __attribute__((address_space(1)))* gptr_t;
__device__ __forceinline__ float4 sload16(const char* sbase, unsigned voff) {
  // (64-bit address arithmetic may have gone through the vector ALU: bring the wave-uniform value into SGPRs explicitly)
  const unsigned long long u = (unsigned long long)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  gptr_t sp = (gptr_t)(((unsigned long long)hi << 32) | lo);
  asm volatile("" : "+s"(sp), "+v"(voff));               // (voff too: its zero-extension must stay next to the load)
  const f32x4_t v = *(const f32x4_t __attribute__((address_space(1)))*)(sp + voff);
  return make_float4(v.x, v.y, v.z, v.w);
}
// r -= m on a 16-register accumulator as 8 packed adds with negated source (plain `-=` compiles to 16 v_sub_f32)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sub16(f32x16& r, const f32x16& m) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x2_t a = {r[2 * i], r[2 * i + 1]};
    const f32x2_t b = {m[2 * i], m[2 * i + 1]};
    asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(b));
    r[2 * i] = a.x; r[2 * i + 1] = a.y;
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL load
// (s_waitcnt vmcnt(0) in front of s_barrier), i.e. for the input patch that was requested a moment earlier precisely so
// that its latency hides under the MFMA phase: measured 25 % of the kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Bt applied to (d0, d1, d2, d3) in place
__device__ __forceinline__ void bt4(float4& d0, float4& d1, float4& d2, float4& d3) {
  const float4 t0 = f4sub(d0, d2), t1 = f4add(d1, d2), t2 = f4sub(d2, d1), t3 = f4sub(d1, d3);
  d0 = t0; d1 = t1; d2 = t2; d3 = t3;
}

}  // namespace pgrf
