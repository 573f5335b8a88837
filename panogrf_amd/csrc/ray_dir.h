// World-space direction of one query ray (not normalised for pinhole queries), shared by the render-pass kernel and by
// the per-ray table kernel of pgrf_render_rays (aux_kernels.hip) so that both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include "render_pass_args.h"

#define PGRF_PI_F 3.14159265358979323846f
#define PGRF_HALF_PI_F 1.57079632679489661923f
#define PGRF_TWO_PI_F 6.28318530717958647692f

namespace pgrf {

// 1-ulp reciprocal (v_rcp_f32) for quantities that are not pixel coordinates: the correctly rounded IEEE
// division costs ~11 VALU instructions, and VALU time adds to f32-MFMA time on this chip.
__device__ __forceinline__ float rcp1(float x) { return __builtin_amdgcn_rcpf(x); }

// GEN = false: m3d ERP convention only; GEN = true: run-time switch over p.dataset / p.perspective
template <bool GEN>
__device__ __forceinline__ void ray_direction(const PassArgs& p, int ray, float& wx, float& wy, float& wz) {
  if (!GEN || !p.perspective) {
    // ERP query: unit direction of the pixel in the dataset's spherical convention (spt_utils.py:45-83, :106-124),
    // normalised (ray_utils.py:13-15), rotated into the world (ray_utils.py:53-71)
    const int ix = (int)p.coords[2 * ray], iy = (int)p.coords[2 * ray + 1];
    const float fx = (float)ix, fy = (float)iy, W1 = (float)(p.W - 1), H1 = (float)(p.H - 1);
    float dx, dy, dz;
    if (!GEN || p.dataset == PGRF_DATASET_M3D) {
      const float theta = fx / W1 * 2.f * PGRF_PI_F - PGRF_HALF_PI_F;
      const float phi = fy / H1 * PGRF_PI_F;
      const float sp = sinf(phi);
      dx = sp * cosf(theta); dy = cosf(phi); dz = sp * sinf(theta);
    } else if (p.dataset == PGRF_DATASET_REPLICA_TEST) {
      const float theta = fx * 2.f * PGRF_PI_F / W1 - PGRF_PI_F;
      const float phi = -fy * PGRF_PI_F / H1 + PGRF_PI_F * 0.5f;
      const float cp = cosf(phi);
      dx = sinf(theta) * cp; dy = -sinf(phi); dz = cosf(theta) * cp;
    } else if (p.dataset == PGRF_DATASET_RESIDENTIAL) {
      const float theta = PGRF_PI_F * (2.f * fx / W1 - 1.5f);
      const float phi = PGRF_PI_F * (0.5f - fy / H1);
      const float cp = cosf(phi);
      dx = cosf(theta) * cp; dz = sinf(theta) * cp; dy = sinf(phi);
    } else {   // CoffeeArea
      // the reference folds these scalars in Python doubles before they meet the float tensor
      const float theta = (float)(-2.0 * 3.14159265358979323846 / (double)(p.W - 1)) * fx + PGRF_TWO_PI_F;
      const float phi = (float)(3.14159265358979323846 / (double)(p.H - 1)) * fy;
      const float sp = sinf(phi);
      dx = sp * cosf(theta); dy = sp * sinf(theta); dz = cosf(phi);
    }
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    { const float rn_ = rcp1(nrm); dx *= rn_; dy *= rn_; dz *= rn_; }
    wx = p.c2w[0] * dx + p.c2w[1] * dy + p.c2w[2] * dz;
    wy = p.c2w[4] * dx + p.c2w[5] * dy + p.c2w[6] * dz;
    wz = p.c2w[8] * dx + p.c2w[9] * dy + p.c2w[10] * dz;
  } else {
    // pinhole query (render_ops.py:35-57): cam = K^-1 [x,y,1]; world = R^T cam + c; direction = world - c, NOT
    // normalised (the sample depth is the camera-space z)
    const float cx_ = p.coords[2 * ray], cy_ = p.coords[2 * ray + 1];
    const float kx = p.kinv[0] * cx_ + p.kinv[1] * cy_ + p.kinv[2];
    const float ky = p.kinv[3] * cx_ + p.kinv[4] * cy_ + p.kinv[5];
    const float kz = p.kinv[6] * cx_ + p.kinv[7] * cy_ + p.kinv[8];
    wx = (p.c2w[0] * kx + p.c2w[1] * ky + p.c2w[2] * kz + p.c2w[3]) - p.c2w[3];
    wy = (p.c2w[4] * kx + p.c2w[5] * ky + p.c2w[6] * kz + p.c2w[7]) - p.c2w[7];
    wz = (p.c2w[8] * kx + p.c2w[9] * ky + p.c2w[10] * kz + p.c2w[11]) - p.c2w[11];
  }
}

}  // namespace pgrf
