// Shared by the 3-D convolution kernels (conv3d.hip: direct implicit GEMM; conv3d_wino.hip: Winograd F(2x2x2, 3x3x3)).
#ifndef PGRF_CONV3D_COMMON_H
#define PGRF_CONV3D_COMMON_H
#include <hip/hip_runtime.h>

#include <cstddef>

#include "pgrf_internal.h"

namespace pgrf {

#define PGRF_CHECK_LAUNCH(what)                                                                       \
  do {                                                                                                \
    const hipError_t e__ = hipGetLastError();                                                         \
    if (e__ != hipSuccess) return set_error(PGRF_ERR_LAUNCH, what " launch failed: %s", hipGetErrorString(e__)); \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
// accumulator register r of lane half h of a 32x32 MFMA tile holds output row rho_(r, h)
constexpr int rho_(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct ConvArgs {
  const float* in; int in_cs, in_coff, Cin;
  float* out; int out_cs, out_coff, Cout;
  const float* wpk;       // packed A operands (pgrf_pack_conv3d)
  const float* bias;      // [ceil(Cout/32)*32] in accumulator order (pgrf_pack_conv3d), or null
  int B, D, H, W;         // input dims
  int Do, Ho, Wo;         // output dims
  int stride;
  float slope;            // LeakyReLU negative slope; 1.0 = identity
  int chunks;             // ceil(Cin/32)
  int cot_total;          // ceil(Cout/32)
  const float* residual;  // optional tensor added AFTER the activation (CostRegNet skip adds), same slicing as out
  int res_cs, res_coff;
  int wtaps;              // 27, or 1 = pointwise (1x1x1) product with the centre tap only
  int tapmajor;           // pointwise projection of the Cout == 1 fast path: write out[row][voxel] (row < 27) instead of
                          // channel-last, so that the 27-tap gather that follows reads consecutive voxels of one row
  int wrap_w;             // 1: circular padding along W (ERP wrap); 0: zeros (the 2-D `upconv` blocks, network/ops.py:226-233)
  int elu;                // 1: ELU instead of LeakyReLU(slope) (2-D encoder blocks)
  int res_first;          // 1: the residual is added BEFORE the activation (ResNet BasicBlock), 0: after (CostRegNet skips)
  float* pool;            // conv3d_wino16 only: optional second output = AvgPool3d(2) of `out` (an F(2^3, 3^3) output tile IS a pooling window)
  int pool_cs, pool_coff;
  int ksplit;             // conv3d_wino16 only: > 1 = every row is `ksplit` work items, each over a range of the 32-channel input blocks,
  long long ksplit_stride;//   writing its raw partial sums (no bias, slope 1) to out + item * ksplit_stride floats (pgrf_conv3d_wino16_splitk)
};

__device__ __forceinline__ float conv_act(const ConvArgs& a, float v) {
  return a.elu ? (v > 0.f ? v : expm1f(v)) : (v > 0.f ? v : v * a.slope);
}

// packed weight index: A operand of MFMA k-step s (0..15) of (cout tile, tap, 32-channel chunk) for `lane`
// (row = lane & 31 = output channel within the tile, lane half h = lane >> 5: input channel 32*chunk + 16*h + s)
__host__ __device__ inline size_t wpk_index(int cot, int tap, int chunk, int chunks, int s, int lane, int ntaps = 27) {
  return ((((size_t)(cot * ntaps + tap) * chunks + chunk) * 4 + (s >> 2)) * 64 + lane) * 4 + (s & 3);
}

}  // namespace pgrf
#endif
