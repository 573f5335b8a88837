// Kernel argument block of the fused render pass (by-value kernarg).
#pragma once
#include "pgrf_internal.h"

namespace pgrf {

struct PassArgs {
  // frame
  const float* imgs;        // [V,H,W,4]
  const float* img_feats;   // [V,hf,wf,32]
  const float* ray_feats;   // [V,hr,wr,32]
  const float* prj;         // [V,hr,wr,kPrjCh] per-pixel projection of ray_feats (pgrf_layout.h), this pass's weights
  const float* w2c;         // [V,12]
  const float* ref_dr;      // [V,2]
  float c2w[12];
  float que_near, que_far;
  int dataset;              // PGRF_DATASET_*
  int perspective;          // pinhole query
  float kinv[9];
  int H, W, hf, wf, hr, wr;
  // rays
  const float* coords;      // [rn,2]
  const float* depth;       // [rn,dn]
  const float* raytab;      // [rn,4] world-space ray directions (ray_dir.h) or null = compute per sample
  int rn;
  // processing order: 0 = ray i is processed i-th; else ray ids are visited in tw x th pixel tiles of a row-major grid of width
  // order_w (a bijection on [0, rn): results cannot depend on it, only the gathers' cache locality does)
  int order_w, order_tw, order_th;
  // network
  const float* blob;
  int use_vis;
  // outputs (nullable)
  float* out_pixel;
  float* out_depth;
  float* out_hit;
  float* out_colors;
  float* out_density;
  int* nan_flag;
  PgrfPassDebug dbg;
};

}  // namespace pgrf
