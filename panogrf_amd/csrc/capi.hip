// extern "C" entry points of libpanogrf_hip.so (declared in include/panogrf_hip.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pgrf_internal.h"
#include "pgrf_layout.h"
#include "render_pass_args.h"

namespace pgrf {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int launch_project_ray_feats(const float* rf, const float* blob, int npix, int use_vis, float* out, void* stream);
int launch_pack_map(const float* src, int n, int C, int h, int w, float* dst, void* stream);
int launch_coarse_depth(float near, float far, int use_disp, int rn, int dn, const float* jitter, float* depth, void* stream);
int launch_fine_depth(const float* depth, const float* hit, float qnear, float qfar, int use_disp, int use_all, int rn,
                      int dn, int fdn, const float* urand, float* out, void* stream, int rows_equal);
int launch_merge_c2f(const float* zc, const float* colc, const float* denc, int dnc, const float* zf, const float* colf,
                     const float* denf, int dnf, int rn, const PgrfPassOut& o, void* stream);

int launch_merge_c2f_bwd(const float* zc, const float* colc, const float* denc, int dnc, const float* zf, const float* colf,
                         const float* denf, int dnf, int rn, const PgrfPassGrad& g, float* gcolc, float* gdenc, float* gcolf,
                         float* gdenf, void* stream);
int launch_render_uncert(const float* depth, const float* hit, const float* rdepth, int rn, int dn, float* out, void* stream);

static const char* kRawNames[R_COUNT] = {
    "dist_decoder.mean_decoder.0.weight", "dist_decoder.mean_decoder.0.bias", "dist_decoder.mean_decoder.2.weight",
    "dist_decoder.mean_decoder.2.bias", "dist_decoder.mean_decoder.4.weight", "dist_decoder.mean_decoder.4.bias",
    "dist_decoder.var_decoder.0.weight", "dist_decoder.var_decoder.0.bias", "dist_decoder.var_decoder.2.weight",
    "dist_decoder.var_decoder.2.bias", "dist_decoder.var_decoder.4.weight", "dist_decoder.var_decoder.4.bias",
    "dist_decoder.aw_decoder.0.weight", "dist_decoder.aw_decoder.0.bias", "dist_decoder.aw_decoder.2.weight",
    "dist_decoder.aw_decoder.2.bias", "dist_decoder.aw_decoder.4.weight", "dist_decoder.aw_decoder.4.bias",
    "dist_decoder.vis_decoder.0.weight", "dist_decoder.vis_decoder.0.bias", "dist_decoder.vis_decoder.2.weight",
    "dist_decoder.vis_decoder.2.bias", "dist_decoder.vis_decoder.4.weight", "dist_decoder.vis_decoder.4.bias",
    "agg_net.prob_embed.0.weight", "agg_net.prob_embed.0.bias", "agg_net.prob_embed.2.weight", "agg_net.prob_embed.2.bias",
    "agg_net.agg_impl.ray_dir_fc.0.weight", "agg_net.agg_impl.ray_dir_fc.0.bias", "agg_net.agg_impl.ray_dir_fc.2.weight",
    "agg_net.agg_impl.ray_dir_fc.2.bias",
    "agg_net.agg_impl.base_fc.0.weight", "agg_net.agg_impl.base_fc.0.bias", "agg_net.agg_impl.base_fc.2.weight",
    "agg_net.agg_impl.base_fc.2.bias",
    "agg_net.agg_impl.vis_fc.0.weight", "agg_net.agg_impl.vis_fc.0.bias", "agg_net.agg_impl.vis_fc.2.weight",
    "agg_net.agg_impl.vis_fc.2.bias",
    "agg_net.agg_impl.vis_fc2.0.weight", "agg_net.agg_impl.vis_fc2.0.bias", "agg_net.agg_impl.vis_fc2.2.weight",
    "agg_net.agg_impl.vis_fc2.2.bias",
    "agg_net.agg_impl.geometry_fc.0.weight", "agg_net.agg_impl.geometry_fc.0.bias", "agg_net.agg_impl.geometry_fc.2.weight",
    "agg_net.agg_impl.geometry_fc.2.bias",
    "agg_net.agg_impl.ray_attention.w_qs.weight", "agg_net.agg_impl.ray_attention.w_ks.weight",
    "agg_net.agg_impl.ray_attention.w_vs.weight", "agg_net.agg_impl.ray_attention.fc.weight",
    "agg_net.agg_impl.ray_attention.layer_norm.weight", "agg_net.agg_impl.ray_attention.layer_norm.bias",
    "agg_net.agg_impl.out_geometry_fc.0.weight", "agg_net.agg_impl.out_geometry_fc.0.bias",
    "agg_net.agg_impl.out_geometry_fc.2.weight", "agg_net.agg_impl.out_geometry_fc.2.bias",
    "agg_net.agg_impl.rgb_fc.0.weight", "agg_net.agg_impl.rgb_fc.0.bias", "agg_net.agg_impl.rgb_fc.2.weight",
    "agg_net.agg_impl.rgb_fc.2.bias", "agg_net.agg_impl.rgb_fc.4.weight", "agg_net.agg_impl.rgb_fc.4.bias",
    "agg_net.agg_impl.neuray_fc.0.weight", "agg_net.agg_impl.neuray_fc.0.bias", "agg_net.agg_impl.neuray_fc.2.weight",
    "agg_net.agg_impl.neuray_fc.2.bias",
};

int launch_ray_dirs(const PassArgs& a, float* tab, void* stream);

static int check_frame(const PgrfFrame* f, const char* who) {
  if (!f) return set_error(PGRF_ERR_ARG, "%s: frame is null", who);
  if (f->rfn < 1 || f->rfn > PGRF_MAX_VIEWS) return set_error(PGRF_ERR_ARG, "%s: rfn=%d not in [1,%d]", who, f->rfn, PGRF_MAX_VIEWS);
  if (f->H < 2 || f->W < 2 || f->hf < 1 || f->wf < 1 || f->hr < 1 || f->wr < 1)
    return set_error(PGRF_ERR_ARG, "%s: bad map sizes H=%d W=%d hf=%d wf=%d hr=%d wr=%d", who, f->H, f->W, f->hf, f->wf, f->hr, f->wr);
  if (!f->imgs || !f->img_feats || !f->ray_feats || !f->w2c || !f->ref_depth_range)
    return set_error(PGRF_ERR_ARG, "%s: frame has a null map/pose pointer", who);
  if ((size_t)f->H * f->W >= (1u << 30)) return set_error(PGRF_ERR_ARG, "%s: frame too large", who);
  return PGRF_OK;
}

static size_t pass_ws_bytes(const PgrfFrame* f) { return (size_t)f->rfn * f->hr * f->wr * kPrjCh * sizeof(float); }

// the query-side fields the ray generation reads (ray_dir.h)
static int fill_query(PassArgs& a, const PgrfFrame* frame, const float* coords, int rn, const char* who) {
  if (frame->dataset < PGRF_DATASET_M3D || frame->dataset > PGRF_DATASET_COFFEEAREA)
    return set_error(PGRF_ERR_UNSUPPORTED, "%s: dataset convention %d unknown", who, frame->dataset);
  std::memcpy(a.c2w, frame->que_c2w, sizeof(a.c2w));
  a.dataset = frame->dataset; a.perspective = frame->perspective ? 1 : 0;
  std::memcpy(a.kinv, frame->que_kinv, sizeof(a.kinv));
  a.H = frame->H; a.W = frame->W; a.coords = coords; a.rn = rn;
  return PGRF_OK;
}

static int do_render_pass(const PgrfFrame* frame, const float* blob, int flags, const float* coords, const float* depth,
                          int rn, int dn, const PgrfPassOut* out, const PgrfPassDebug* dbg, void* workspace,
                          size_t ws_bytes, void* stream, const float* raytab = nullptr) {
  const int use_vis = (flags & PGRF_PASS_USE_VIS) ? 1 : 0;
  const int precision = (flags & PGRF_PASS_F16X3) ? 1 : 0;
  int rc = check_frame(frame, "pgrf_render_pass");
  if (rc) return rc;
  if (dbg && precision) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_render_pass: debug dumps are only available at f32 precision");
  if (!blob || !coords || !depth || !out) return set_error(PGRF_ERR_ARG, "pgrf_render_pass: null blob/coords/depth/out");
  if (rn < 0) return set_error(PGRF_ERR_ARG, "pgrf_render_pass: rn=%d < 0", rn);
  if (dn != 32 && dn != 64 && dn != 128)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_render_pass: dn=%d unsupported (32, 64 or 128 samples per ray)", dn);
  if (out->render_uncert && (!out->render_depth || !out->hit_prob))
    return set_error(PGRF_ERR_ARG, "pgrf_render_pass: render_uncert needs the render_depth and hit_prob outputs as well");
  if (rn == 0) return PGRF_OK;
  if (!workspace || ws_bytes < pass_ws_bytes(frame))
    return set_error(PGRF_ERR_ARG, "pgrf_render_pass: workspace too small (%zu < %zu bytes)", ws_bytes, pass_ws_bytes(frame));
  // per-pixel projection of the ray_feats maps with THIS pass's weights (tiny; see pgrf_layout.h kPrjCh)
  rc = launch_project_ray_feats(frame->ray_feats, blob, frame->rfn * frame->hr * frame->wr, use_vis, (float*)workspace, stream);
  if (rc) return rc;
  PassArgs a;
  std::memset(&a, 0, sizeof(a));
  a.imgs = frame->imgs; a.img_feats = frame->img_feats; a.ray_feats = frame->ray_feats;
  a.prj = (const float*)workspace;
  a.w2c = frame->w2c; a.ref_dr = frame->ref_depth_range;
  rc = fill_query(a, frame, coords, rn, "pgrf_render_pass");
  if (rc) return rc;
  a.que_near = frame->que_depth_range[0]; a.que_far = frame->que_depth_range[1];
  a.hf = frame->hf; a.wf = frame->wf; a.hr = frame->hr; a.wr = frame->wr;
  a.depth = depth; a.raytab = raytab; a.blob = blob; a.use_vis = use_vis ? 1 : 0;
  {
    // Rays of a full panorama (or of whole row tiles of one) arrive row-major: visited in 2-D pixel tiles, the workgroups
    // resident on an XCD gather from a compact patch of the source maps instead of a 1-pixel-high strip.  Measured: +1.9 % for
    // 4 views of 1024x2048 (341.4 -> 335.0 ms; one query row's footprint in the maps is ~40 MB there, ten times an XCD's
    // L2), +-0.3 % at 512x1024 x 2 views and at 2 views of 1024x2048, -0.5 % on a fine pass -> on by default from 8 Mpx of source
    // maps, and from 2 Mpx for four views (4 views of 512x1024).  PGRF_RAY_TILE="tw th" forces a tile, "0" turns it off.  Any other ray list keeps its order.
    int tw = 0, th = 0;
    const char* e = getenv("PGRF_RAY_TILE");
    if (e) { if (sscanf(e, "%d %d", &tw, &th) < 2) th = tw; }
    else {
      // gated on the cases that were measured: from 8 Mpx of source maps for any view count (round 4), from 2 Mpx for the four-view
      // kernel (round 5: 4 views of 512x1024 65.07 -> 64.65 ms); 2 views of 1024x2048 (4 Mpx, +-0.3 %) keep the row-major order
      const size_t px = (size_t)frame->rfn * frame->H * frame->W;
      if (px >= ((size_t)8 << 20) || (frame->rfn >= 4 && px >= ((size_t)2 << 20))) tw = th = 16;
    }
    if (tw > 0 && th > 0 && frame->W % tw == 0 && rn % (frame->W * th) == 0 && !frame->perspective) {
      a.order_w = frame->W; a.order_tw = tw; a.order_th = th;
    }
  }
  a.out_pixel = out->pixel_colors; a.out_depth = out->render_depth; a.out_hit = out->hit_prob;
  a.out_colors = out->colors; a.out_density = out->density; a.nan_flag = out->nan_flag;
  if (dbg) a.dbg = *dbg;
  rc = launch_render_pass(a, frame->rfn, dn, dbg != nullptr, precision, stream);
  if (rc || !out->render_uncert) return rc;
  return launch_render_uncert(depth, out->hit_prob, out->render_depth, rn, dn, out->render_uncert, stream);
}

}  // namespace pgrf

using namespace pgrf;

extern "C" {

const char* pgrf_last_error(void) { return g_err; }
// negative when any object of this library was compiled as an experiment (pgrf_experiments.h): never loadable as the product
extern "C" __attribute__((weak)) int pgrf_ablation_marker;
int pgrf_abi_version(void) { return &pgrf_ablation_marker ? -PGRF_ABI_VERSION : PGRF_ABI_VERSION; }

int pgrf_pack_map(const float* src, int n, int C, int h, int w, float* dst, void* stream) {
  if (!src || !dst) return set_error(PGRF_ERR_ARG, "pgrf_pack_map: null pointer");
  if (n < 1 || C < 1 || h < 1 || w < 1) return set_error(PGRF_ERR_ARG, "pgrf_pack_map: bad shape [%d,%d,%d,%d]", n, C, h, w);
  return launch_pack_map(src, n, C, h, w, dst, stream);
}

int pgrf_blob_floats(void) { return kBlobFloats; }
int pgrf_raw_count(void) { return R_COUNT; }
const char* pgrf_raw_name(int i) { return (i >= 0 && i < R_COUNT) ? kRawNames[i] : nullptr; }

int pgrf_pack_weights(const float* const* raw_host, int n_raw, int use_vis, int max_dn, float* blob_host) {
  return pack_weights_host(raw_host, n_raw, use_vis, max_dn, blob_host);
}
int pgrf_pack_map_selftest(int use_vis, int max_dn) { return pack_map_selftest(use_vis, max_dn); }

int pgrf_sample_coarse_depth_train(float min_depth, float max_depth, int use_disp, int rn, int dn, const float* jitter,
                                   float* depth, void* stream) {
  if (!depth) return set_error(PGRF_ERR_ARG, "pgrf_sample_coarse_depth: null output");
  if (dn < 3 || rn < 0) return set_error(PGRF_ERR_ARG, "pgrf_sample_coarse_depth: need dn > 2 (got %d), rn >= 0 (got %d)", dn, rn);
  if (!(min_depth > 0.f) || !(max_depth > min_depth)) return set_error(PGRF_ERR_ARG, "pgrf_sample_coarse_depth: bad depth range");
  if (rn == 0) return PGRF_OK;
  return launch_coarse_depth(min_depth, max_depth, use_disp, rn, dn, jitter, depth, stream);
}

int pgrf_sample_coarse_depth(float min_depth, float max_depth, int use_disp, int rn, int dn, float* depth, void* stream) {
  return pgrf_sample_coarse_depth_train(min_depth, max_depth, use_disp, rn, dn, nullptr, depth, stream);
}

static int sample_fine_checked(const float* depth, const float* hit_prob, float que_near, float que_far, int use_disp,
                               int use_all, int rn, int dn, int fdn, const float* u, float* out, void* stream, int rows_equal) {
  if (!depth || !hit_prob || !out) return set_error(PGRF_ERR_ARG, "pgrf_sample_fine_depth: null pointer");
  if (dn < 2 || dn > PGRF_MAX_DN || fdn < 1 || fdn > PGRF_MAX_DN)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_sample_fine_depth: dn=%d fdn=%d outside [2,%d]/[1,%d]", dn, fdn, PGRF_MAX_DN, PGRF_MAX_DN);
  if (rn < 0) return set_error(PGRF_ERR_ARG, "pgrf_sample_fine_depth: rn < 0");
  if (rn == 0) return PGRF_OK;
  return launch_fine_depth(depth, hit_prob, que_near, que_far, use_disp, use_all, rn, dn, fdn, u, out, stream, rows_equal);
}

int pgrf_sample_fine_depth_train(const float* depth, const float* hit_prob, float que_near, float que_far, int use_disp,
                                 int use_all, int rn, int dn, int fdn, const float* u, float* out, void* stream) {
  return sample_fine_checked(depth, hit_prob, que_near, que_far, use_disp, use_all, rn, dn, fdn, u, out, stream, 0);
}

int pgrf_sample_fine_depth(const float* depth, const float* hit_prob, float que_near, float que_far, int use_disp,
                           int use_all, int rn, int dn, int fdn, float* out, void* stream) {
  return pgrf_sample_fine_depth_train(depth, hit_prob, que_near, que_far, use_disp, use_all, rn, dn, fdn, nullptr, out, stream);
}

size_t pgrf_render_pass_workspace_bytes(const PgrfFrame* frame) {
  if (!frame || frame->rfn < 1 || frame->hr < 1 || frame->wr < 1) return 0;
  return pass_ws_bytes(frame);
}

int pgrf_render_pass(const PgrfFrame* frame, const float* blob, int flags, const float* coords, const float* depth,
                     int rn, int dn, const PgrfPassOut* out, const PgrfPassDebug* dbg, void* workspace, size_t workspace_bytes,
                     void* stream) {
  return do_render_pass(frame, blob, flags, coords, depth, rn, dn, out, dbg, workspace, workspace_bytes, stream);
}

int pgrf_merge_c2f(const float* depth_c, const float* colors_c, const float* density_c, int dn_c, const float* depth_f,
                   const float* colors_f, const float* density_f, int dn_f, int rn, const PgrfPassOut* out, void* stream) {
  if (!depth_c || !colors_c || !density_c || !depth_f || !colors_f || !density_f || !out)
    return set_error(PGRF_ERR_ARG, "pgrf_merge_c2f: null pointer");
  if (dn_c < 1 || dn_f < 1 || dn_c > PGRF_MAX_DN || dn_f > PGRF_MAX_DN)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_merge_c2f: dn_c=%d dn_f=%d outside [1,%d]", dn_c, dn_f, PGRF_MAX_DN);
  if (rn < 0) return set_error(PGRF_ERR_ARG, "pgrf_merge_c2f: rn < 0");
  if (rn == 0) return PGRF_OK;
  return launch_merge_c2f(depth_c, colors_c, density_c, dn_c, depth_f, colors_f, density_f, dn_f, rn, *out, stream);
}

int pgrf_merge_c2f_backward(const float* depth_c, const float* colors_c, const float* density_c, int dn_c, const float* depth_f,
                            const float* colors_f, const float* density_f, int dn_f, int rn, const PgrfPassGrad* grad,
                            float* g_colors_c, float* g_density_c, float* g_colors_f, float* g_density_f, void* stream) {
  if (!depth_c || !colors_c || !density_c || !depth_f || !colors_f || !density_f || !grad || !g_colors_c || !g_density_c || !g_colors_f ||
      !g_density_f)
    return set_error(PGRF_ERR_ARG, "pgrf_merge_c2f_backward: null pointer");
  if (dn_c < 1 || dn_f < 1 || dn_c > PGRF_MAX_DN || dn_f > PGRF_MAX_DN)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_merge_c2f_backward: dn_c=%d dn_f=%d outside [1,%d]", dn_c, dn_f, PGRF_MAX_DN);
  if (rn < 0) return set_error(PGRF_ERR_ARG, "pgrf_merge_c2f_backward: rn < 0");
  if (rn == 0) return PGRF_OK;
  return launch_merge_c2f_bwd(depth_c, colors_c, density_c, dn_c, depth_f, colors_f, density_f, dn_f, rn, *grad, g_colors_c, g_density_c,
                              g_colors_f, g_density_f, stream);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t pgrf_render_workspace_bytes(const PgrfFrame* frame, const PgrfRenderCfg* cfg, int rn) {
  if (!cfg || !frame || rn <= 0) return 0;
  const size_t r = (size_t)rn;
  const int fdn_eff = cfg->fine_use_all ? cfg->dn + cfg->fdn : cfg->fdn;
  size_t b = align256(pass_ws_bytes(frame));     // projection map, reused by the coarse and the fine pass
  b += align256(r * cfg->dn * 4);            // coarse depth
  b += align256(r * cfg->dn * 4);            // coarse hit_prob (when the caller does not ask for it)
  b += align256(r * fdn_eff * 4);            // fine depth
  b += align256(r * 16);                     // per-ray directions (ray_dir.h), shared by both passes
  if (cfg->c2f_all) {
    b += align256(r * cfg->dn * 12) + align256(r * cfg->dn * 4);        // coarse colors, density
    b += align256(r * fdn_eff * 12) + align256(r * fdn_eff * 4);        // fine colors, density
  }
  return b;
}

int pgrf_render_rays_train(const PgrfFrame* frame, const float* blob_coarse, const float* blob_fine, const PgrfRenderCfg* cfg,
                           const float* coords, int rn, const float* coarse_jitter, const float* fine_u,
                           const PgrfPassOut* coarse, const PgrfPassOut* fine, float* coarse_depth_out, float* fine_depth_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (!cfg || !coarse) return set_error(PGRF_ERR_ARG, "pgrf_render_rays: null cfg/coarse");
  if (cfg->hierarchical && (!fine || !blob_fine)) return set_error(PGRF_ERR_ARG, "pgrf_render_rays: hierarchical render needs fine outputs and weights");
  if (rn < 0) return set_error(PGRF_ERR_ARG, "pgrf_render_rays: rn < 0");
  if (rn == 0) return PGRF_OK;
  if (int rc0 = check_frame(frame, "pgrf_render_rays")) return rc0;
  const size_t need = pgrf_render_workspace_bytes(frame, cfg, rn);
  if (!workspace || workspace_bytes < need)
    return set_error(PGRF_ERR_ARG, "pgrf_render_rays: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  const size_t r = (size_t)rn;
  const int fdn_eff = cfg->fine_use_all ? cfg->dn + cfg->fdn : cfg->fdn;
  char* ws = (char*)workspace;
  void* prj_ws = ws; ws += align256(pass_ws_bytes(frame));
  const size_t prj_bytes = pass_ws_bytes(frame);
  float* depth_c = (float*)ws; ws += align256(r * cfg->dn * 4);
  float* hit_c = (float*)ws; ws += align256(r * cfg->dn * 4);
  float* depth_f = (float*)ws; ws += align256(r * fdn_eff * 4);
  float* raytab = (float*)ws; ws += align256(r * 16);
  {
    PassArgs q;
    std::memset(&q, 0, sizeof(q));
    if (int rcq = fill_query(q, frame, coords, rn, "pgrf_render_rays")) return rcq;
    if (int rcq = launch_ray_dirs(q, raytab, stream)) return rcq;
  }
  PgrfPassOut oc = *coarse;
  if (!oc.hit_prob) oc.hit_prob = hit_c;
  PgrfPassOut of;
  std::memset(&of, 0, sizeof(of));
  if (fine) of = *fine;
  float *col_c = nullptr, *den_c = nullptr, *col_f = nullptr, *den_f = nullptr;
  if (cfg->hierarchical && cfg->c2f_all) {
    col_c = (float*)ws; ws += align256(r * cfg->dn * 12);
    den_c = (float*)ws; ws += align256(r * cfg->dn * 4);
    col_f = (float*)ws; ws += align256(r * fdn_eff * 12);
    den_f = (float*)ws; ws += align256(r * fdn_eff * 4);
    if (!oc.colors) oc.colors = col_c; else col_c = oc.colors;
    if (!oc.density) oc.density = den_c; else den_c = oc.density;
  }
  if (coarse_depth_out) depth_c = coarse_depth_out;
  int rc = pgrf_sample_coarse_depth_train(cfg->min_depth, cfg->max_depth, cfg->use_disp, rn, cfg->dn, coarse_jitter, depth_c, stream);
  if (rc) return rc;
  const int pf = cfg->precision == 1 ? PGRF_PASS_F16X3 : 0;
  if (cfg->precision != 0 && cfg->precision != 1) return set_error(PGRF_ERR_ARG, "pgrf_render_rays: precision must be 0 (f32) or 1 (f16x3)");
  rc = do_render_pass(frame, blob_coarse, (cfg->use_vis_coarse ? PGRF_PASS_USE_VIS : 0) | pf, coords, depth_c, rn, cfg->dn, &oc, nullptr, prj_ws, prj_bytes, stream, raytab);
  if (rc || !cfg->hierarchical) return rc;
  float* fd = fine_depth_out ? fine_depth_out : depth_f;
  // the coarse depths generated here without jitter are the same row for every ray, which the lane-per-ray sampler exploits
  rc = sample_fine_checked(depth_c, oc.hit_prob, frame->que_depth_range[0], frame->que_depth_range[1], cfg->use_disp,
                           cfg->fine_use_all, rn, cfg->dn, cfg->fdn, fine_u, fd, stream, coarse_jitter ? 0 : 1);
  if (rc) return rc;
  if (!cfg->c2f_all) return do_render_pass(frame, blob_fine, (cfg->use_vis_fine ? PGRF_PASS_USE_VIS : 0) | pf, coords, fd, rn, fdn_eff, &of, nullptr, prj_ws, prj_bytes, stream, raytab);
  // fine pass into scratch, then one composite over the union (renderer.py:484-518)
  PgrfPassOut tmp;
  std::memset(&tmp, 0, sizeof(tmp));
  tmp.colors = col_f; tmp.density = den_f; tmp.nan_flag = of.nan_flag;
  rc = do_render_pass(frame, blob_fine, (cfg->use_vis_fine ? PGRF_PASS_USE_VIS : 0) | pf, coords, fd, rn, fdn_eff, &tmp, nullptr, prj_ws, prj_bytes, stream, raytab);
  if (rc) return rc;
  return pgrf_merge_c2f(depth_c, col_c, den_c, cfg->dn, fd, col_f, den_f, fdn_eff, rn, &of, stream);
}

int pgrf_render_rays(const PgrfFrame* frame, const float* blob_coarse, const float* blob_fine, const PgrfRenderCfg* cfg,
                     const float* coords, int rn, const PgrfPassOut* coarse, const PgrfPassOut* fine, float* fine_depth_out,
                     void* workspace, size_t workspace_bytes, void* stream) {
  return pgrf_render_rays_train(frame, blob_coarse, blob_fine, cfg, coords, rn, nullptr, nullptr, coarse, fine, nullptr,
                                fine_depth_out, workspace, workspace_bytes, stream);
}

}  // extern "C"
