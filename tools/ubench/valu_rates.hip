// Issue cost of the VALU ops the render kernel is made of, on gfx950: cycles per wave64 instruction per SIMD, with 1 and
// 2 waves resident per SIMD (independent 8-register chains, no MFMA beside them).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Op { FMA, MUL, PKFMA, PKMUL, EXP, MED3, CVTPK, RCP, ELU4, ELU3, ADD, FMAC, CNDMASK, ELU3PK };

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float v[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.01f + i; p[i] = f32x2{v[i], v[i] + 1.f}; }
  const float c = 0.999f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
        if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(f32x2{c, c}));
        if (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(f32x2{c, c}));
        if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(v[j]) : "v"(c));
        if (OP == CVTPK) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j]));
        if (OP == ELU4) {  // the kernel's ELU: mul, exp, add -1, med3
          float t;
          asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %1\n v_exp_f32 %0, %0\n v_add_f32 %0, -1.0, %0\n v_med3_f32 %1, %1, %0, 0" : "=&v"(t), "+v"(v[j]));
        }
        if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[j]) : "v"(c));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c));
        if (OP == ELU3PK && (j & 1) == 0) {  // two ELUs: 2 exp, 1 pk_fma, 2 med3
          f32x2 t;
          asm volatile("v_exp_f32 %0, %2\n v_exp_f32 %1, %3" : "=&v"(t.x), "=&v"(t.y) : "v"(v[j]), "v"(v[j + 1]));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(f32x2{c, c}), "v"(f32x2{-c, -c}));
          asm volatile("v_med3_f32 %0, %0, %2, 0\n v_med3_f32 %1, %1, %3, 0" : "+v"(v[j]), "+v"(v[j + 1]) : "v"(t.x), "v"(t.y));
        }
        if (OP == ELU3) {  // log2e-scaled carry: exp, fma, med3
          float t;
          asm volatile("v_exp_f32 %0, %1\n v_fma_f32 %0, %0, %2, -%2\n v_med3_f32 %1, %1, %0, 0" : "=&v"(t), "+v"(v[j]) : "v"(c));
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, const char* per, float* d) {
  const int iters = 4000;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<256 * bpc, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<256 * bpc, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: bpc waves each issue iters*128 groups
    printf("%-10s waves/SIMD=%d : %7.3f ms -> %5.2f cyc per %s per wave (SIMD-serial)\n", name, bpc, ms,
           ms * 1e-3 * 2.4e9 / (iters * 128.0 * bpc), per);
  }
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  run<FMA>("v_fma", "inst", d); run<MUL>("v_mul", "inst", d); run<PKFMA>("v_pk_fma", "inst", d); run<PKMUL>("v_pk_mul", "inst", d);
  run<EXP>("v_exp", "inst", d); run<MED3>("v_med3", "inst", d); run<CVTPK>("cvt_pkrtz", "inst", d); run<RCP>("v_rcp", "inst", d);
  run<ELU4>("elu4", "ELU", d); run<ELU3>("elu3", "ELU", d); run<ELU3PK>("elu3pk", "ELU(x0.5: 4 per 8 slots)", d);
  run<ADD>("v_add", "inst", d); run<FMAC>("v_fmac", "inst", d); run<CNDMASK>("v_cndmask", "inst", d);
  return 0;
}
