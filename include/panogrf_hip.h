/* panogrf_hip.h — C ABI of the MI355X (gfx950) PanoGRF inference hot path.
 *
 * Drop-in boundary for the reference's per-ray renderer and spherical cost volume
 * (thucz/PanoGRF).  The reference is pure Python/PyTorch: there is no FFI in it to replace, so
 * each entry point cites the Python function(s) whose work it takes over; INTEGRATION.md shows
 * the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - plain C, no torch types: raw pointers + explicit sizes, float32, contiguous.
 *  - every `*_dev` / unqualified data pointer is DEVICE memory owned by the caller; the library
 *    never allocates, frees or synchronises.  `*_host` pointers are host memory.
 *  - `stream` is a hipStream_t passed as void* (0 = null stream); all work is enqueued
 *    asynchronously on it.
 *  - return value: 0 = PGRF_OK, < 0 = error; pgrf_last_error() returns the message of the last
 *    failing call on the calling thread.  Nothing is printed, nothing blocks.
 */
#ifndef PANOGRF_HIP_H
#define PANOGRF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGRF_OK 0
#define PGRF_ERR_ARG (-1)     /* invalid argument (null pointer, unsupported size, ...) */
#define PGRF_ERR_LAUNCH (-2)  /* HIP reported an error when enqueuing */
#define PGRF_ERR_UNSUPPORTED (-3)

#define PGRF_FEAT_CH 32       /* channels of img_feats / ray_feats (network/ops.py ResUNetLight out_dim) */
#define PGRF_MAX_VIEWS 4
#define PGRF_MAX_DN 128

const char* pgrf_last_error(void);
/* library / kernel ABI version, bumped when struct layouts change (5: PgrfPassOut.render_uncert, pgrf_instnorm scratch).
 * A NEGATIVE value marks a library linked from experiment objects (timing ablations etc., csrc/pgrf_experiments.h):
 * it must never be used as the product; panogrf_amd/_lib.py refuses it unless PGRF_ALLOW_ABLATION_LIB=1. */
#define PGRF_ABI_VERSION 5
int pgrf_abi_version(void);

/* ------------------------------------------------------------------ per-frame source maps -- */
/* Source-view data of one frame, re-laid-out channel-last by pgrf_pack_map so that one bilinear
 * tap of a 32-channel map is one 128-byte line (reference keeps NCHW: utils/imgs_info.py:105-156,
 * network/renderer.py:639-642). */
typedef struct PgrfFrame {
  int rfn;                     /* source views, 1..PGRF_MAX_VIEWS */
  int H, W;                    /* full ERP resolution of the source images (= cfg height,width) */
  int hf, wf;                  /* img_feats resolution  (H/4, W/4 in the shipped config) */
  int hr, wr;                  /* ray_feats resolution  (depth_height/4, depth_width/4) */
  const float* imgs;           /* [rfn,H,W,4]   rgb + zero pad, values in [0,1] */
  const float* img_feats;      /* [rfn,hf,wf,32] */
  const float* ray_feats;      /* [rfn,hr,wr,32] */
  const float* w2c;            /* [rfn,3,4] world->camera of every source view */
  const float* ref_depth_range;/* [rfn,2] near,far of every source view */
  float que_c2w[12];           /* [3,4] camera->world of the query view (host values) */
  float que_depth_range[2];    /* near,far of the query view (host values) */
  int dataset;                 /* PGRF_DATASET_*: spherical convention of network/spt_utils.py (cfg dataset_name) */
  int perspective;             /* 0: the query is an ERP panorama (render_ops.py:76-106 depth2points_spherical);
                                * 1: pinhole query, rays K^-1 [x,y,1] (render_ops.py:35-74 coords2rays /
                                *    depth2points_perspec, is_perspec=True, render_cubes.py) */
  float que_kinv[9];           /* inverse intrinsics of a pinhole query (host values), unused otherwise */
} PgrfFrame;
/* ERP pixel <-> sphere <-> cartesian conventions (network/spt_utils.py:45-83, :106-124, :134-156, :176-192) */
#define PGRF_DATASET_M3D 0
#define PGRF_DATASET_REPLICA_TEST 1
#define PGRF_DATASET_RESIDENTIAL 2
#define PGRF_DATASET_COFFEEAREA 3

/* NCHW [n,C,h,w] -> NHWC [n,h,w,Cpad] (Cpad = 4 for C = 3, else C); replaces nothing in the
 * reference (layout change only). */
int pgrf_pack_map(const float* src_nchw, int n, int C, int h, int w, float* dst_nhwc, void* stream);

/* ------------------------------------------------------------------ network weights -------- */
/* Number of floats of one packed pass blob (coarse or fine). */
int pgrf_blob_floats(void);
/* Number of raw tensors pgrf_pack_weights expects and the state_dict suffix of entry i
 * (e.g. "dist_decoder.mean_decoder.0.weight" with the pass prefix removed: see INTEGRATION.md). */
int pgrf_raw_count(void);
const char* pgrf_raw_name(int i);
/* Pack one pass (dist decoder + aggregation net) from reference-layout HOST tensors
 * (nn.Linear [out,in] + [out]) into the MFMA operand order; raw_host[i] follows pgrf_raw_name(i).
 * Replaces: nothing computed — it is the load_state_dict-time transform
 * (network/dist_decoder.py:53-97, network/aggregate_net.py:16-39, network/ibrnet.py:239-313). */
int pgrf_pack_weights(const float* const* raw_host, int n_raw, int use_vis, int max_dn, float* blob_host);
/* The same blob, bit for bit, built ON THE DEVICE from device-resident tensors: raw_dev is a HOST array of n_raw DEVICE pointers (same
 * order and shapes), blob_dev a device buffer of pgrf_blob_floats() floats; enqueued on `stream`, no host synchronisation (the first
 * call per (device, use_vis, max_dn) builds and uploads a source map).  For training, where the weights change every optimizer step
 * (the reference re-reads its nn.Parameters on every call: network/renderer.py:223-304 runs on the live modules). */
int pgrf_pack_weights_device(const float* const* raw_dev, int n_raw, int use_vis, int max_dn, float* blob_dev, void* stream);
/* Diagnostic (host only, no GPU): evaluates the device packer's source map on the host against pgrf_pack_weights on pseudo-random
 * weights; PGRF_OK when every blob word is reproduced bit for bit. */
int pgrf_pack_map_selftest(int use_vis, int max_dn);

/* ------------------------------------------------------------------ sampling --------------- */
/* Deterministic (eval) stratified depths, depth[rn,dn].  network/render_ops.py:292-339 sample_depth. */
int pgrf_sample_coarse_depth(float min_depth, float max_depth, int use_disp, int rn, int dn, float* depth,
                             void* stream);
/* Inverse-CDF importance sampling + sort.  depth,hit_prob [rn,dn] -> out [rn, fdn] (or
 * [rn, dn+fdn] sorted union when use_all != 0: cfg fine_depth_use_all).
 * network/render_ops.py:413-473 sample_fine_depth + network/renderer.py:469-472. */
int pgrf_sample_fine_depth(const float* depth, const float* hit_prob, float que_near, float que_far, int use_disp,
                           int use_all, int rn, int dn, int fdn, float* out, void* stream);
/* Training-mode forms of the two samplers (is_train=True in network/renderer.py:607,437): the caller supplies the uniform
 * draws the reference takes from torch.rand, so that a seeded run reproduces the reference's sample positions.
 *   jitter [rn, dn-2] in [0,1): interior coarse samples move by (r - 0.5) * 0.999 bins (render_ops.py:310-311,329-330);
 *   u      [rn, fdn]  in [0,1): inverse-CDF positions replacing the bin centres (render_ops.py:443-449).
 * NULL = the deterministic eval behaviour of the functions above. */
int pgrf_sample_coarse_depth_train(float min_depth, float max_depth, int use_disp, int rn, int dn, const float* jitter,
                                   float* depth, void* stream);
int pgrf_sample_fine_depth_train(const float* depth, const float* hit_prob, float que_near, float que_far, int use_disp,
                                 int use_all, int rn, int dn, int fdn, const float* u, float* out, void* stream);

/* ------------------------------------------------------------------ one network pass ------- */
typedef struct PgrfPassOut {   /* any pointer may be NULL = not wanted */
  float* pixel_colors;         /* [rn,3]     'pixel_colors_nr' */
  float* render_depth;         /* [rn]       'render_depth'    */
  float* hit_prob;             /* [rn,dn]    'hit_prob_nr'     */
  float* colors;               /* [rn,dn,3]  'colors_nr'       */
  float* density;              /* [rn,dn]    'density_nr'      */
  int* nan_flag;               /* device int, caller zeroes it: set to 1 when a NaN reaches an output of this pass.  The
                                * reference stops in ipdb there (renderer.py:264-265,459-462; render_ops.py:251-252);
                                * the library never blocks or prints.  NaNs in the source FEATURE maps can be absorbed inside
                                * the MLPs (ELU/ReLU are v_med3/v_max, which return the non-NaN operand) */
  float* render_uncert;        /* [rn]       'render_uncert' = sum_i hit_prob_i (z_i - render_depth)^2 + 1e-5
                                * (cfg render_uncert, network/renderer.py:305-312,507-514); needs render_depth and hit_prob */
} PgrfPassOut;

typedef struct PgrfPassDebug { /* intermediate tensors for stage-level parity tests; NULL = skip */
  float* px;                   /* [rfn,rn,dn,2] projected pixel coords */
  float* depth_v;              /* [rfn,rn,dn]   distance to the source camera */
  float* ddiff;                /* [rfn,rn,dn,4] */
  float* vis;                  /* [rfn,rn,dn] */
  float* hit;                  /* [rfn,rn,dn] */
  float* ray_feats;            /* [rfn,rn,dn,32] */
  float* rgb;                  /* [rfn,rn,dn,3] */
  float* img_feats;            /* [rfn,rn,dn,32] */
  float* neuray;               /* [rn,dn,rfn,32] prob_embed output */
  float* base_x;               /* [rn,dn,rfn,32] base_fc output */
  float* geo;                  /* [rn,dn,16] geometry_fc output */
  float* attn;                 /* [rn,dn,16] ray_attention output */
} PgrfPassDebug;

/* One pass (coarse or fine) of the renderer over rn rays at the given per-ray sample depths:
 * ERP ray generation, projection into the rfn source views, bilinear gathers, mixture-of-logistics
 * visibility, IBRNet aggregation with per-ray attention, alpha compositing.
 * coords [rn,2] (x,y) pixel of each ray in the query ERP (H x W = frame->H, frame->W);
 * depth [rn,dn], dn in {32,64,128}.
 * Replaces NeuralRayBaseRenderer.render_by_depth (network/renderer.py:223-304) and everything it
 * calls: depth2inv_dists, depth2points_spherical, project_points_dict (network/render_ops.py:76-258),
 * predict_proj_ray_prob (renderer.py:120-136), get_img_feats (:180-188), network_rendering (:210-219),
 * DefaultAggregationNet.forward (network/aggregate_net.py:41-89), IBRNetWithNeuRay.forward
 * (network/ibrnet.py:315-373), MixtureLogisticsDistDecoder.forward/compute_prob
 * (network/dist_decoder.py:99-140).
 * `workspace` (>= pgrf_render_pass_workspace_bytes(frame), caller-owned device memory) receives the per-pixel
 * projection of the ray_feats maps through this pass's first linear layers (W*bilinear(f)+b == bilinear(W*f+b)):
 * 160 floats per ray_feats pixel, recomputed by every call. */
size_t pgrf_render_pass_workspace_bytes(const PgrfFrame* frame);
#define PGRF_PASS_USE_VIS 1   /* the dist decoder has the 4th (visibility) MLP: dist_decoder_cfg.use_vis */
#define PGRF_PASS_F16X3 2     /* opt-in: Linear layers as 3 split-f16 MFMAs per product (x = h1 + h2, fp32 accumulate,
                                 ~2^-21 relative per product) instead of exact-fp32 MFMAs; |activations| must stay < 65504 */
int pgrf_render_pass(const PgrfFrame* frame, const float* blob, int flags, const float* coords,
                     const float* depth, int rn, int dn, const PgrfPassOut* out, const PgrfPassDebug* dbg,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ multi-GPU: image all-gather over RCCL --- */
/* For hosts WITHOUT torch.distributed (the Python host uses `panogrf_amd.dist`, i.e. torch.distributed on RCCL): the one
 * collective of the ray-tile split -- every rank renders H/G rows, one ncclAllGather reassembles the image.
 * RCCL is resolved at run time (dlopen; a copy already in the process is reused).  The reference has no counterpart
 * (train/trainer.py:232-234: single GPU).
 *   rank 0: pgrf_comm_unique_id(&id) -> send `id` to the other ranks (any side channel) -> every rank: pgrf_comm_init ->
 *   per frame: pgrf_allgather_image(comm, my_rows, whole_image, floats_per_rank, stream) -> pgrf_comm_destroy */
typedef struct PgrfUniqueId { char internal[128]; } PgrfUniqueId;      /* = ncclUniqueId */
int pgrf_comm_unique_id(PgrfUniqueId* id);
int pgrf_comm_init(const PgrfUniqueId* id, int rank, int nranks, void** comm);
/* shard: `count` floats of this rank; full: nranks * count floats, rank-major (equal shards: H % nranks == 0).  In place when
 * shard == full + rank * count (the rank rendered straight into its rows of the frame buffer: SURVEY.md 8e), as ncclAllGather allows. */
int pgrf_allgather_image(void* comm, const float* shard, float* full, size_t count, void* stream);
/* NCCL_VERSION_CODE of the RCCL library this process resolved (major * 10000 + minor * 100 + patch), for run records */
int pgrf_comm_rccl_version(int* version);
int pgrf_comm_destroy(void* comm);

/* ------------------------------------------------------------------ training: backward of one pass --- */
/* Upstream gradients of one pass's outputs (any pointer may be NULL = zero): what the reference's losses differentiate
 * (network/loss.py: RenderLoss on pixel_colors_nr, DepthLoss on render_depth, the consistency terms on hit_prob_nr). */
typedef struct PgrfPassGrad {
  const float* pixel_colors;   /* [rn,3]    */
  const float* render_depth;   /* [rn]      */
  const float* hit_prob;       /* [rn,dn]   */
  const float* colors;         /* [rn,dn,3] */
  const float* density;        /* [rn,dn]   */
} PgrfPassGrad;

/* Bytes of scratch pgrf_render_pass_backward needs (every activation of a staged forward + its gradient: ~9 KB per
 * (ray, sample, source view)). */
size_t pgrf_render_pass_backward_workspace_bytes(int rfn, int rn, int dn);

/* Backward of pgrf_render_pass (network/renderer.py:223-304 under autograd, as the reference's trainer runs it): the forward is
 * REMATERIALISED stage by stage (the fused kernel keeps no activations), then differentiated in reverse.
 *   raw_dev    pgrf_raw_count() DEVICE pointers to the pass's weights in the reference's state_dict layout ([out,in] row-major
 *              Linear weights, biases; order = pgrf_raw_name(i); vis_decoder entries may be NULL without use_vis)
 *   g_raw_dev  the same table for the gradients, ACCUMULATED (+=) into caller-zeroed buffers; NULL entries are skipped
 *   g_img_feats / g_ray_feats   gradients of the channel-last source maps of `frame` ([rfn,hf,wf,32] / [rfn,hr,wr,32]),
 *              accumulated; NULL = not wanted.  Depths, poses and source colours get no gradient (the reference detaches the
 *              importance sampler's inputs, renderer.py:437).
 *   fwd_out    optional: the staged forward's own outputs (must equal pgrf_render_pass's to round-off; used by the tests) */
int pgrf_render_pass_backward(const PgrfFrame* frame, const float* const* raw_dev, int n_raw, int use_vis, const float* coords,
                              const float* depth, int rn, int dn, const PgrfPassGrad* grad, float* const* g_raw_dev,
                              float* g_img_feats, float* g_ray_feats, const PgrfPassOut* fwd_out, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The same in two calls, for training loops that can hold a pass's activations between forward and backward (the workspace: ~9 KB per
 * (ray, sample, view)): phase 1 = the staged forward alone (fills `workspace`, writes `fwd_out`; grad / g_* may be NULL) -- its outputs ARE the
 * pass's outputs, equal to pgrf_render_pass's to round-off, so the autograd forward needs no fused launch; phase 2 = the reverse sweep over a
 * workspace a phase-1 call with the same frame, weights, coords and depth left behind (the weights must not have changed in between: the
 * reference's autograd has the same rule for its saved tensors).  phase 0 = both = pgrf_render_pass_backward. */
int pgrf_render_pass_backward_phase(const PgrfFrame* frame, const float* const* raw_dev, int n_raw, int use_vis, const float* coords,
                                    const float* depth, int rn, int dn, const PgrfPassGrad* grad, float* const* g_raw_dev,
                                    float* g_img_feats, float* g_ray_feats, const PgrfPassOut* fwd_out, void* workspace,
                                    size_t workspace_bytes, void* stream, int phase);

/* `render_c2f_all` (network/renderer.py:484-518): one composite over the sorted union of the
 * coarse and fine samples.  Inputs [rn,dn_c]/[rn,dn_f]; outputs sized dn_c+dn_f. */
int pgrf_merge_c2f(const float* depth_c, const float* colors_c, const float* density_c, int dn_c,
                   const float* depth_f, const float* colors_f, const float* density_f, int dn_f, int rn,
                   const PgrfPassOut* out, void* stream);

/* Backward of pgrf_merge_c2f (the trainer differentiates network/renderer.py:484-518 like everything else; round 4): the sort is a
 * permutation, so the merged composite's gradient is gathered back to the coarse and the fine pass's per-sample colours and
 * densities (the depths are constants: the importance sampler's inputs are detached, renderer.py:437).  grad: upstream gradients
 * of the MERGED outputs (hit_prob / density [rn,dn_c+dn_f], colors [rn,dn_c+dn_f,3], pixel_colors [rn,3], render_depth [rn]; any
 * entry may be NULL); g_*: written (not accumulated), [rn,dn_c(,3)] / [rn,dn_f(,3)]. */
int pgrf_merge_c2f_backward(const float* depth_c, const float* colors_c, const float* density_c, int dn_c,
                            const float* depth_f, const float* colors_f, const float* density_f, int dn_f, int rn,
                            const PgrfPassGrad* grad, float* g_colors_c, float* g_density_c, float* g_colors_f,
                            float* g_density_f, void* stream);

/* ------------------------------------------------------------------ whole per-batch path --- */
typedef struct PgrfRenderCfg {
  float min_depth, max_depth;  /* cfg min_depth / max_depth */
  int use_disp;                /* cfg use_disp */
  int dn, fdn;                 /* depth_sample_num, fine_depth_sample_num */
  int hierarchical;            /* use_hierarchical_sampling */
  int fine_use_all;            /* fine_depth_use_all */
  int c2f_all;                 /* render_c2f_all */
  int use_vis_coarse, use_vis_fine;
  int precision;               /* 0 = f32 (default), 1 = split-f16 x3 (see PGRF_PASS_F16X3) */
} PgrfRenderCfg;

/* Bytes of scratch pgrf_render_rays needs for rn rays of this frame. */
size_t pgrf_render_workspace_bytes(const PgrfFrame* frame, const PgrfRenderCfg* cfg, int rn);
/* Coarse (+ importance-sampled fine) render of rn rays: NeuralRayBaseRenderer.render_impl
 * (network/renderer.py:567-633, non-DINER branch).  `fine` outputs may be NULL when
 * cfg->hierarchical == 0.  All rays of a frame can be passed at once: the result does not depend
 * on ray batching (the reference's ray_batch_num loop, renderer.py:656-683, exists only to bound
 * memory). */
int pgrf_render_rays(const PgrfFrame* frame, const float* blob_coarse, const float* blob_fine,
                     const PgrfRenderCfg* cfg, const float* coords, int rn, const PgrfPassOut* coarse,
                     const PgrfPassOut* fine, float* fine_depth_out, void* workspace, size_t workspace_bytes,
                     void* stream);
/* The same per-batch path in training mode (is_train=True: render_impl passes random_sample=True to both samplers,
 * network/renderer.py:607,437-438).  coarse_jitter [rn, dn-2] / fine_u [rn, fdn] as in the *_train samplers (NULL = eval
 * behaviour); coarse_depth_out [rn, dn] (optional) returns the jittered coarse depths.  Forward only. */
int pgrf_render_rays_train(const PgrfFrame* frame, const float* blob_coarse, const float* blob_fine,
                           const PgrfRenderCfg* cfg, const float* coords, int rn, const float* coarse_jitter,
                           const float* fine_u, const PgrfPassOut* coarse, const PgrfPassOut* fine,
                           float* coarse_depth_out, float* fine_depth_out, void* workspace, size_t workspace_bytes,
                           void* stream);


/* ------------------------------------------------------------------ spherical cost volume -- */
/* Depth hypotheses of the mono-guided 360-degree MVS: n_guided planes clamp(ref_mu + offs[k]) plus
 * n_uniform shared planes, sorted per pixel.  ref_mu [B,hw] -> depth_volume [B,n_guided+n_uniform,hw].
 * guided_offsets_host[k] = k_list[k]*fixed_sigma (host array, <= 16); uniform_centres is a DEVICE
 * array (ascending, e.g. torch.linspace(min,max,59)).
 * Replaces network/omni_mvsnet/pipeline3_model.py:723-733 + :774-815 (concat + torch.sort). */
int pgrf_depth_hypotheses(const float* ref_mu, int B, int hw, const float* guided_offsets_host, int n_guided,
                          float min_depth, float max_depth, const float* uniform_centres, int n_uniform,
                          float* depth_volume, void* stream);

#define PGRF_COST_ABS_DIFF 0
#define PGRF_COST_DOT 1
#define PGRF_COST_NONE 2
/* Spherical-homography warp of the source-view features onto every depth hypothesis of the
 * reference view + cost.  feats [B,S,h,w,C] channel-last, depth_volume [B,D,h,w], rots [B,S,3,3],
 * trans [B,S,3] (world->camera), out [B,D,h,w,C].  out = sum over the n_src listed source views of
 * cost(warp(view), feats[:,ref_idx]) / src_div.
 *   2-view  (models/spherical_cost_volume.py:231-341 calculate_cost_volume_erp):
 *           S=2, ref_idx=1, src_idx={0}, src_div=1
 *   multi-view (models/spherical_cost_volume_mv.py:219-347 calculate_cost_volume_erp_multiview):
 *           ref_idx=curr_idx, src_idx = {v in 0..S-2, v != curr_idx}, src_div = S-2
 * Both call get_cv_per_depth (:135-230) D times; helpers/my_torch_helpers.py:34-38,62-96.
 * Environment PGRF_CV_STREAM_STORES=1: the volume is written with non-temporal stores (the kernel alone 75 -> 62 us at 128x256x64;
 * a regulariser that reads the volume right behind then finds it in HBM instead of the last-level cache, so it is off by default). */
int pgrf_cost_volume_erp(const float* feats, const float* depth_volume, const float* rots, const float* trans, int B,
                         int S, int h, int w, int C, int D, int ref_idx, const int* src_idx_host, int n_src,
                         float src_div, int cost_type, int dataset /* PGRF_DATASET_* = args["dataset_name"] */,
                         float* out, void* stream);


/* ------------------------------------------------------------------ 3-D cost regularisation -- */
/* Activations are channel-last [B,D,H,W,Cs]; (cs, coff) = channel stride of the buffer and first channel of
 * the slice used, so U-Net concatenations (models/common_blocks.py:232-238) are slices of one buffer. */
/* floats of the packed block of one 3x3x3 conv and the host-side packer (weights [Cout,Cin,3,3,3] as in
 * nn.Conv3d, bias [Cout] or NULL; fold BatchNorm3d running stats into both before packing). */
size_t pgrf_conv3d_packed_floats(int Cout, int Cin);
int pgrf_pack_conv3d(const float* w_host, const float* bias_host, int Cout, int Cin, float* packed_host);
/* 3x3x3 conv, padding 1 with zeros along D,H and CIRCULAR along W, stride 1|2, + bias + LeakyReLU(slope)
 * (slope 1 = none) + optional residual added after the activation.  Replaces WrapPadding3D + nn.Conv3d
 * (+ BatchNorm3d eval + LeakyReLU): models/common_blocks.py:458-503,366-455 (Conv3DBlockv2),
 * models/cost_reg.py:15-57 (ConvBnReLU3DWrap, Conv3DWrap). */
int pgrf_conv3d_wrap(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                     const float* packed, int has_bias, int B, int D, int H, int W, int stride, float lrelu_slope,
                     const float* residual, int res_cs, int res_coff, void* stream);
/* The same convolution (stride 1) by Winograd F(2x2x2, 3x3x3): 8 multiply-adds per output and channel pair instead of
 * 27, fp32 throughout (csrc/conv3d_wino.hip).  For the wide UNet2 layers (models/common_blocks.py:366-455 via
 * models/test_models.py:102-146): needs Cin a multiple of 32, Cout a multiple of 64, even D, and W % 64 == 0 with even H or
 * W % 32 == 0 with H % 4 == 0
 * (pgrf_conv3d_wino_supported).  `packed` from pgrf_pack_conv3d_wino (filters transformed on the host in double).
 * Results equal pgrf_conv3d_wrap to ~1e-6 relative (transform round-off). */
size_t pgrf_conv3d_wino_packed_floats(int Cout, int Cin);
int pgrf_pack_conv3d_wino(const float* w_host, const float* bias_host, int Cout, int Cin, float* packed_host);
int pgrf_conv3d_wino_supported(int Cin, int Cout, int D, int H, int W);
/* 1 when supported AND the layer has enough tile rows to fill the chip (otherwise pgrf_conv3d_wrap is faster) */
int pgrf_conv3d_wino_preferred(int Cin, int Cout, int B, int D, int H, int W);
int pgrf_conv3d_wino(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                     const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                     int res_cs, int res_coff, void* stream);

/* The same algorithm, second kernel generation (round 4; csrc/conv3d_wino16.hip): rows of 16 tiles, v_mfma_f32_16x16x4_f32, TWO
 * independent 8-wave workgroups per CU (78 KB of LDS each) instead of one 16-wave workgroup -- one's barriers, epilogue and operand
 * formation run under the other's MFMAs.  Own packing (k permutation of the 16-row MFMA, bias in natural order).  Needs
 * Cin % 32 == 0, Cout % 64 == 0 (<= 1024), even D and H, W % 32 == 0 or W == 16 (a 32-voxel row then walks the circular volume twice and
 * stores its first half).  Environment switch of the Python host: PANOGRF_WINO=32
 * selects the first generation where both apply. */
size_t pgrf_conv3d_wino16_packed_floats(int Cout, int Cin);
int pgrf_pack_conv3d_wino16(const float* w_host, const float* bias_host, int Cout, int Cin, float* packed_host);
int pgrf_conv3d_wino16_supported(int Cin, int Cout, int D, int H, int W);
int pgrf_conv3d_wino16_preferred(int Cin, int Cout, int B, int D, int H, int W);
int pgrf_conv3d_wino16(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                       const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope,
                       const float* residual, int res_cs, int res_coff, void* stream);
/* The same launch with a second output: pool_out (channel-last, [B, D/2, H/2, W/2, pool_cs] + pool_coff) = nn.AvgPool3d(2) of `out`.
 * An F(2^3, 3^3) output tile is one pooling window, so the encoder levels of UNet2 (models/common_blocks.py:210-218 -> ConvBlock2 :383: two convs, then
 * AvgPool3d(2)) get their pooled tensor from the conv's epilogue instead of one more pass over its output.  pool_out may be NULL. */
/* Layers with few rows (low-resolution levels of UNet2: < ~1.5 rows per CU where the kernel wants two workgroups per CU): every row as
 * `ksplit` work items over shares of the 32-channel input blocks, raw partial sums into `scratch` (ksplit * B*D*H*W*Cout floats), bias +
 * LeakyReLU in a small reduce kernel behind.  pgrf_conv3d_wino16_ksplit() returns the split the host should use (0: none). */
int pgrf_conv3d_wino16_ksplit(int Cin, int Cout, int B, int D, int H, int W);
int pgrf_conv3d_wino16_splitk(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                              const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, int ksplit,
                              float* scratch, size_t scratch_floats, void* stream);
int pgrf_conv3d_wino16_pool(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                            const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope,
                            const float* residual, int res_cs, int res_coff, float* pool_out, int pool_cs, int pool_coff, void* stream);
/* Narrow layers of CostRegNet (models/cost_reg.py:99-175: 8/16/32-channel convs): the same wrap-padded 3x3x3 conv,
 * but the reduction is cut into 32-deep steps over (tap, channel) (two or four taps per step when Cin is 16 or 8) and
 * `nshift` consecutive output depths share one 32-row tile (nshift * Cout <= 32; nshift > 1 needs stride 1 and
 * D % nshift == 0).  Cin in {8,16,32}; Cout a multiple of 4.  `packed` from pgrf_pack_conv3d_small with the same
 * (Cout, Cin, nshift).  Results equal pgrf_conv3d_wrap up to fp32 summation order. */
size_t pgrf_conv3d_small_packed_floats(int Cout, int Cin, int nshift);
int pgrf_pack_conv3d_small(const float* w_host, const float* bias_host, int Cout, int Cin, int nshift, float* packed_host);
int pgrf_conv3d_small(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                      const float* packed, int has_bias, int nshift, int B, int D, int H, int W, int stride, float lrelu_slope,
                      const float* residual, int res_cs, int res_coff, void* stream);

/* pgrf_conv3d_upfused's step for Cin = 16 -> Cout = 8 on 16-row MFMA tiles with the low-resolution brick staged in LDS (z / y
 * position classes select the weight block per brick / wave; the two seam columns of the circular axis get a corrected B operand
 * under the interior x weights).  Same arguments as pgrf_conv3d_upfused; low-res D, H >= 2 and W >= 33. */
size_t pgrf_conv3d_upbrick_packed_floats(int Cout, int Cin);
int pgrf_pack_conv3d_upbrick(const float* w_host, const float* bias_host, int Cout, int Cin, int D, int H, int W, float* packed_host);
int pgrf_conv3d_upbrick(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                        const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                        int res_cs, int res_coff, void* stream);

/* The same narrow layers on the 16-row MFMA tile (v_mfma_f32_16x16x4_f32): Cout = 8 with nshift = 2 (4 input planes per 2
 * output planes: 75 % useful products against 50 % of pgrf_conv3d_small's nshift = 4), or Cout = 16 with nshift = 1 (100 %).
 * Same arguments and epilogue as pgrf_conv3d_small. */
size_t pgrf_conv3d_small16_packed_floats(int Cout, int Cin, int nshift);
int pgrf_pack_conv3d_small16(const float* w_host, const float* bias_host, int Cout, int Cin, int nshift, float* packed_host);
int pgrf_conv3d_small16(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                        const float* packed, int has_bias, int nshift, int B, int D, int H, int W, int stride, float lrelu_slope,
                        const float* residual, int res_cs, int res_coff, void* stream);

/* Trilinear x2 up-sampling (align_corners=False) FUSED into the wrap-padded 3x3x3 convolution behind it -- CostRegNet's decoder
 * layers (models/cost_reg.py:58-94,136-158: conv9 32->16, conv11 16->8): evaluated on the LOW-resolution tensor as an 8-parity
 * convolution (8 * Cout output rows share one operand; the up-sampled volume is never materialised), with per-position-class
 * folded kernels for the border voxels.  in: [B,D,H,W,Cin-slice] low-res; out / residual: [B,2D,2H,2W,.] high-res;
 * epilogue LeakyReLU(slope) then + residual.  Cin in {16, 32}, Cout in {8, 16}, D, H, W >= 2.  The packed block depends on
 * (D, H, W) only through the border classes of axes of length 2. */
size_t pgrf_conv3d_upfused_packed_floats(int Cout, int Cin);
int pgrf_pack_conv3d_upfused(const float* w_host, const float* bias_host, int Cout, int Cin, int D, int H, int W, float* packed_host);
int pgrf_conv3d_upfused(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                        const float* packed, int has_bias, int B, int D, int H, int W, float lrelu_slope, const float* residual,
                        int res_cs, int res_coff, void* stream);
/* Fast path of the same convolution for Cout == 1 and wide inputs (UNet2's last block, 128 -> 1): a pointwise MFMA
 * pass projects every voxel onto the 27 tap vectors (y[B,D,H,W,32], caller-owned scratch), a 27-tap gather finishes
 * (bias, LeakyReLU, residual).  `packed` is the block made by pgrf_pack_conv3d(Cout=1). */
int pgrf_conv3d_cout1_project(const float* in, int in_cs, int in_coff, int Cin, const float* packed, float* y, int B,
                              int D, int H, int W, void* stream);
int pgrf_conv3d_cout1_gather(const float* y, const float* packed, int Cin, int has_bias, float* out, int out_cs, int out_coff,
                             int B, int D, int H, int W, float lrelu_slope, const float* residual, int res_cs, int res_coff,
                             void* stream);
/* The gather for a Cout = 1 convolution over torch.cat((up(x), skip)) (UNet2's last decoder step, models/common_blocks.py:225-241):
 * y_lo = pgrf_conv3d_cout1_project of x at LOW resolution with the weights of the up-sampled channels, y_hi = the projection of the
 * skip half at full resolution; the trilinear x2 interpolation of the 27 low-resolution planes happens inside the gather (the
 * projection commutes with it), so up(x) is never written.  B, D, H, W = the full (even) resolution; packed_hi / Cin_hi / has_bias:
 * the pgrf_pack_conv3d(Cout = 1) block of the skip half (it carries the bias). */
int pgrf_conv3d_cout1_gather_up(const float* y_hi, const float* y_lo, const float* packed_hi, int Cin_hi, int has_bias, float* out, int out_cs,
                                int out_coff, int B, int D, int H, int W, float lrelu_slope, const float* residual, int res_cs, int res_coff,
                                void* stream);
/* nn.AvgPool3d(2) (common_blocks.py:380) and F.interpolate(scale_factor=2, mode="trilinear",
 * align_corners=False) (common_blocks.py:221-236, cost_reg.py:85-88). */
int pgrf_avgpool3d_2(const float* in, int in_cs, int in_coff, float* out, int out_cs, int out_coff, int B, int D, int H,
                     int W, int C, void* stream);
int pgrf_upsample3d_2(const float* in, int in_cs, int in_coff, float* out, int out_cs, int out_coff, int B, int D, int H,
                      int W, int C, void* stream);

/* ------------------------------------------------------------------ per-frame 2-D encoders --- */
/* ResUNetLight image encoder (network/ops.py:235-455) and DefaultVisEncoder
 * (network/vis_encoder.py:6-32) on channel-last maps [N,H,W,Cs] with (cs, coff) channel slices (U-Net concatenations
 * are slices of one buffer).  They run once per frame; `panogrf_amd/encoders_hip.py` is the layer sequence. */
/* 3x3 (pad 1) or 1x1 convolution, stride 1|2, + bias, + optional residual; padding: zeros along H, circular along W
 * when wrap_w != 0 (WrapPadding, models/common_blocks.py:258-292) or zeros (the `upconv` blocks, network/ops.py:226-233).
 * Weights [Cout,Cin,k,k] as in nn.Conv2d, packed once by pgrf_pack_conv2d.  Runs on the implicit-GEMM MFMA kernel of
 * pgrf_conv3d_wrap on a depth-1 volume.  Cin, Cout multiples of 4. */
size_t pgrf_conv2d_packed_floats(int Cout, int Cin);
int pgrf_pack_conv2d(const float* w_host, const float* bias_host, int Cout, int Cin, int ksize, float* packed_host);
int pgrf_conv2d(const float* in, int in_cs, int in_coff, int Cin, float* out, int out_cs, int out_coff, int Cout,
                const float* packed, int has_bias, int wrap_w, int N, int H, int W, int stride,
                int act /* 0 none, 1 ReLU, 2 LeakyReLU(0.01), 3 ELU */, const float* residual, int res_cs, int res_coff,
                int residual_first /* 1: act(conv + residual) as in a ResNet BasicBlock; 0: act(conv) + residual */,
                void* stream);
/* Stem: direct k x k convolution (k odd <= 7, stride 1|2, pad k/2) of an NCHW input with Cin <= 4 into a channel-last
 * map, + optional bias [Cout] and ReLU (a BatchNorm folded into the conv); w_dev [Cout,Cin,k,k] and bias_dev are DEVICE
 * pointers, Cout 16|32|64.  ResUNetLight.conv1 (network/ops.py:262-268), resnet18.conv1 (models/resnet.py:160-164). */
int pgrf_conv2d_direct(const float* in_nchw, int N, int Cin, int H, int W, const float* w_dev, const float* bias_dev, int Cout,
                       int ksize, int stride, int wrap_w, int relu, float* out, int out_cs, int out_coff, void* stream);
/* nn.MaxPool2d(3, 2, 1) with the ERP padding of models/convert_module.py:12-27 (zeros along H, circular along W). */
int pgrf_maxpool3s2_wrap(const float* in, int in_cs, int in_coff, int C, int N, int H, int W, float* out, int out_cs, int out_coff,
                         void* stream);
/* nn.InstanceNorm2d(C, affine=True, track_running_stats=False) + activation (0 none, 1 ReLU, 2 ELU), with an optional
 * residual added BEFORE the activation (BasicBlock: relu(bn2(conv2) + shortcut), network/ops.py:127-196).  gamma/beta
 * may be NULL.  `partial`: caller-owned scratch of N * pgrf_instnorm_chunks(HW) * C * 4 floats (16-byte aligned).  In place is allowed
 * (out == x). */
int pgrf_instnorm_chunks(int HW);
int pgrf_instnorm(const float* x, int cs, int coff, int C, int N, int HW, const float* gamma, const float* beta, float eps,
                  int act, const float* residual, int res_cs, int res_coff, float* out, int out_cs, int out_coff,
                  float* partial, void* stream);
/* F.interpolate(mode="bilinear", align_corners=...) between channel-last maps (network/ops.py:226-233 upconv x2 with
 * align_corners=True; network/vis_encoder.py:27-29 resize to the ray-feature resolution with align_corners=False). */
int pgrf_resize_bilinear(const float* in, int in_cs, int in_coff, int C, int N, int H, int W, float* out, int out_cs, int out_coff,
                         int Ho, int Wo, int align_corners, void* stream);
/* ERP -> cube map [face_w x 6*face_w], faces F R B L U D side by side: replaces `Equirec2Cube.run`
 * (UniFuse-Unidirectional-Fusion/UniFuse/datasets/util.py:72-111), which the reference calls per view and frame on the
 * CPU through scipy.ndimage.map_coordinates (FullPipeline.e2c_process, network/omni_mvsnet/pipeline3_model.py:266-282).
 * erp / out: any strided [N,H,W,C] / [N,face_w,6*face_w,C] views (strides in floats: NCHW and channel-last both work).
 * coor_y / coor_x: the converter's coordinate tables [face_w*6*face_w] (util.py:60-70, built once on the host).
 * order 1 = linear (images), 0 = nearest (depth).  The image is sampled as scipy does: padded below with its last and
 * first rows rolled by W/2, legacy 'wrap' (period len-1), double arithmetic in scipy's tap order -> the same floats.
 * pix_scale (NULL or [face_w*6*face_w]): multiplied in afterwards (the cos map that turns ray distance into z-depth,
 * util.py:22-24,105).  mean / stdv (both NULL or device [C]): (v - mean[c]) / stdv[c] afterwards (normalize_input,
 * pipeline3_model.py:226-244). */
int pgrf_equirec_to_cube(const float* erp, int N, int H, int W, int C, long long in_sn, long long in_sh, long long in_sw, long long in_sc,
                         const float* coor_y, const float* coor_x, int face_w, int order, const float* pix_scale, const float* mean,
                         const float* stdv, float* out, long long out_sn, long long out_sh, long long out_sw, long long out_sc,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PANOGRF_HIP_H */
