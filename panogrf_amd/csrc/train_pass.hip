// Backward pass of ONE network pass of the per-ray renderer: gradients of the reference's
// training losses (network/loss.py: functions of pixel_colors_nr, render_depth, hit_prob_nr; optionally of colors_nr /
// density_nr) with respect to every per-ray network weight (dist decoder + aggregation net, reference state_dict layout) and
// to the two source feature maps (img_feats, ray_feats: from there torch autograd carries on into the 2-D encoders).
//
// The forward the gradients belong to is the fused kernel (render_pass.h), which keeps no activations.  The backward
// REMATERIALISES: one staged forward over the batch's (point, view) rows with every activation kept in a caller-owned
// workspace, then the reverse sweep stage by stage.  Linear layers are plain row-major fp32 GEMMs (train_kernels.h), the
// other stages have hand-written forward / backward kernels.  Sample depths, poses and source colours are constants (the
// reference detaches the importance-sampling inputs, renderer.py:437: hit_prob.detach()), so no gradient flows into geometry.
//
// Rows: P = rn * dn points (ray-major), PV = P * V (point-major, views innermost).  Concatenated layer inputs are real
// buffers whose producers write column slices: pe_in [PV,34] = [ray_feats 32 | hit | vis], xin [PV,207] = [view statistics 140 |
// rgb_feat 35 | neuray 32], xr [PV,37] = [x 32 | vis 1 | ray_diff 4], gin [P,65] = [mean 32 | var 32 | weight mean 1].
#include "pgrf_experiments.h"
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "pgrf_internal.h"
#include "pgrf_layout.h"
#include "render_pass_args.h"
#include "train_kernels.h"

namespace pgrf {
namespace train {

struct GradPtrs { float* p[R_COUNT]; int n[R_COUNT]; };

#define PGRF_TCHECK(what)                                                                              \
  do {                                                                                                \
    const hipError_t e__ = hipGetLastError();                                                         \
    if (e__ != hipSuccess) return set_error(PGRF_ERR_LAUNCH, what " launch failed: %s", hipGetErrorString(e__)); \
  } while (0)

// ---- small glue kernels ----------------------------------------------------------------------------------------------
// dst[p, dc + c] (=|+=) sum_v src[p * V + v, sc + c]     (four columns per thread: cols, strides and offsets multiples of 4)
__global__ void reduce_views_kernel(float* __restrict__ dst, int ldd, int dc, const float* __restrict__ src, int lds, int sc, long P, int V,
                                    int cols, int accumulate) {
  const int c4 = cols >> 2;
  const long n = P * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i; const long p = iu / (unsigned)c4; const int c = 4 * (int)(iu % (unsigned)c4);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int v = 0; v < V; ++v) {
      const float4 t = *reinterpret_cast<const float4*>(src + (p * V + v) * lds + sc + c);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    float4* d = reinterpret_cast<float4*>(dst + p * ldd + dc + c);
    if (accumulate) { const float4 o = *d; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *d = s;
  }
}
// y[r, yc + c] = x[r, xc + c] * (s ? s[r * sstride] : cst)
__global__ void rowscale_kernel(float* __restrict__ y, int ldy, int yc, const float* __restrict__ x, int ldx, int xc, const float* __restrict__ s,
                                int sstride, float cst, long rows, int cols) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i; const long r = iu / (unsigned)cols; const int c = (int)(iu % (unsigned)cols);      // (n < 2^32: host guard; a 64-bit divide is ~100 instructions)
    y[r * ldy + yc + c] = x[r * ldx + xc + c] * (s ? s[r * sstride] : cst);
  }
}
// x2 = x + x_res, v1 = sigmoid(xv[32])  (ibrnet.py:347-350); x2 -> xr[:, 0:32]; xv1 = x2 * v1 (vis_fc2's input) in the same sweep
// (round 5: was this kernel + a row-scale launch).  32 lanes per row.
__global__ void __launch_bounds__(256) visfc_post_kernel(const float* __restrict__ x, const float* __restrict__ xv, long PV, float* __restrict__ xr,
                                                         float* __restrict__ v1, float* __restrict__ xv1) {
  const long r = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int c = threadIdx.x & 31;
  if (r >= PV) return;
  const float s = 1.f / (1.f + expf(-xv[r * 33 + 32]));
  const float t = x[r * 32 + c] + xv[r * 33 + c];
  xr[r * 37 + c] = t;
  xv1[r * 32 + c] = t * s;
  if (c == 0) v1[r] = s;
}
// backward of the two lines above in one sweep (round 5: was a row-scale backward launch + this kernel):
//   t = gxr[:, c] + gxv1[:, c] * v1;  gx = t;  gxv[:, :32] = t;  gv1 = sum_c gxv1[:, c] * xr[:, c];  gxv[:, 32] = gv1 * v1 (1 - v1)
__global__ void __launch_bounds__(256) visfc_post_bwd_kernel(const float* __restrict__ gxr, const float* __restrict__ gxv1, const float* __restrict__ xr,
                                                             const float* __restrict__ v1, long PV, float* __restrict__ gx, float* __restrict__ gxv) {
  const long r = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int c = threadIdx.x & 31;
  if (r >= PV) return;
  const float v = v1[r], g = gxv1[r * 32 + c];
  const float t = gxr[r * 37 + c] + g * v;
  gx[r * 32 + c] = t;
  gxv[r * 33 + c] = t;
  float acc = g * xr[r * 37 + c];
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 32);
  if (c == 0) gxv[r * 33 + 32] = acc * v * (1.f - v);
}
// wgt2 = v2 / (sum_v v2 + 1e-8);  gin[:, 64] = mean_v wgt2   (ibrnet.py:352-355); v2 lives in xr[:, 32]
__global__ void wgt2_kernel(const float* __restrict__ xr, long P, int V, float* __restrict__ wgt2, float* __restrict__ gin) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float S = 0.f;
  for (int v = 0; v < V; ++v) S += xr[(p * V + v) * 37 + 32];
  float m = 0.f;
  for (int v = 0; v < V; ++v) { const float w = xr[(p * V + v) * 37 + 32] / (S + 1e-8f); wgt2[p * V + v] = w; m += w; }
  gin[p * 65 + 64] = m / (float)V;
}
// gv2_v += gw_v / (S + e) - sum_u gw_u v2_u / (S + e)^2, with gw_u = gwgt2_u + ggin[:, 64] / V
__global__ void wgt2_bwd_kernel(const float* __restrict__ xr, const float* __restrict__ gwgt2, const float* __restrict__ ggin, long P, int V,
                                float* __restrict__ gxr) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float S = 0.f;
  for (int v = 0; v < V; ++v) S += xr[(p * V + v) * 37 + 32];
  const float den = S + 1e-8f, gm = ggin[p * 65 + 64] / (float)V;
  float dot = 0.f;
  for (int v = 0; v < V; ++v) dot += (gwgt2[p * V + v] + gm) * xr[(p * V + v) * 37 + 32];
  for (int v = 0; v < V; ++v) gxr[(p * V + v) * 37 + 32] += (gwgt2[p * V + v] + gm) / den - dot / (den * den);
}
// gp = g2 + positional encoding of the sample index (ibrnet.py:356-358); pos [dn,16]
__global__ void posenc_kernel(const float* __restrict__ g2, const float* __restrict__ pos, long P, int dn, float* __restrict__ gp) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P * 16; i += (long)gridDim.x * blockDim.x) {
    const long p = i / 16; const int c = (int)(i % 16);
    gp[i] = g2[i] + pos[(p % dn) * 16 + c];
  }
}
__global__ void fill_kernel(float* __restrict__ x, float v, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void posenc_table_kernel(float* __restrict__ pos, int dn) {     // ibrnet.py:305-313, evaluated in double like numpy
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dn * 16) return;
  const int s = i / 16, j = i % 16;
  const double ang = (double)s / pow(10000.0, 2.0 * (double)(j / 2) / 16.0);
  pos[i] = (float)((j & 1) ? cos(ang) : sin(ang));
}

static inline unsigned nblk(long n, int per = 256) { const long b = (n + per - 1) / per; return (unsigned)(b < 1 ? 1 : (b > 65535L * 16 ? 65535L * 16 : b)); }

// ---- workspace ------------------------------------------------------------------------------------------------------
// Weight-gradient replicas: the dW kernel's workgroups add their partial tiles with float atomics, and with ~768 workgroups per launch
// every one of a layer's <= 13 248 addresses takes ~768 same-address atomics, which the L2 serialises (the launch's tail: 47 us for a 32 x 32
// layer whose row walk is ~15 us).  Workgroup z adds into replica z % kGradReplicas (replica 0 = the caller's buffer, the others in the
// workspace, one kGradSlot-float slot per raw tensor); reduce_replicas_kernel folds them at the end of the pass.
constexpr int kGradReplicas = 8;
constexpr int kGradSlot = 13312;          // >= 64 x 207 (base_fc.0.weight, the largest tensor), multiple of 256
static_assert((size_t)R_COUNT * kGradSlot == kRepStride && kGradReplicas == 8 && kGradSlot >= 64 * 207, "replica layout shared with dw_mfma_kernel");
__global__ void reduce_replicas_kernel(const float* __restrict__ rep, GradPtrs gp) {
  const int t = blockIdx.y;
  float* dst = gp.p[t];
  const int n = gp.n[t];
  if (!dst) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < kGradReplicas - 1; ++r) s += rep[((size_t)r * R_COUNT + t) * kGradSlot + i];
    dst[i] += s;
  }
}

struct Ws {
  // constants
  int *tap_rf_o, *tap_if_o;
  float *tap_rf_w, *tap_if_w, *rgb, *dd, *depth_v, *hw, *w1, *pos, *grep;
  // forward activations
  float *pe_in;
  float *dh1[4], *dh2[4], *dout[4];
  float *vis, *hit, *pe_h, *d1, *dfeat, *n1, *nfs, *w0, *glob, *hb, *xin, *b1, *x, *v1h, *xv, *v1, *xr, *xv1, *f1, *wgt2;
  float *gin, *g1, *g2, *gp, *q, *k, *v, *o, *fo, *ln, *og1, *sigma, *r1, *r2, *logit, *blend, *colors;
  // gradients
  float *gsigma, *gcol, *glogit, *gr2, *gr1, *gxr, *gog1, *gln, *gfo, *go, *gq, *gk, *gv, *ggp, *gg1, *ggin, *gwgt2, *gf1, *gxv1,
      *gx, *gxv, *gv1h, *gb1, *ghb, *gxin, *gglob, *gw0, *gn1, *gdfeat, *gd1, *gpe_h, *gpe_in, *gdout[4], *gdh2[4], *gdh1[4];
};

static size_t plan(Ws& w, long P, int V, int dn, char* base) {
  size_t off = 0;
  auto take = [&](long floats) -> float* {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((size_t)floats * 4 + 255) & ~(size_t)255;
    return p;
  };
  const long PV = P * V;
  w.tap_rf_o = (int*)take(PV * 4); w.tap_if_o = (int*)take(PV * 4);
  w.tap_rf_w = take(PV * 4); w.tap_if_w = take(PV * 4);
  w.rgb = take(PV * 3); w.dd = take(PV * 4); w.depth_v = take(PV); w.hw = take(P * 2); w.w1 = take(PV); w.pos = take((long)dn * 16);
  w.grep = take((long)(kGradReplicas - 1) * kGradSlot * R_COUNT);
  w.pe_in = take(PV * 34);
  for (int m = 0; m < 4; ++m) { w.dh1[m] = take(PV * 32); w.dh2[m] = take(PV * 32); w.dout[m] = take(PV * 2); }
  w.vis = take(PV); w.hit = take(PV); w.pe_h = take(PV * 32); w.d1 = take(PV * 16); w.dfeat = take(PV * 35); w.n1 = take(PV * 8);
  w.nfs = take(PV); w.w0 = take(PV); w.glob = take(P * 140); w.hb = take(P * 64); w.xin = take(PV * 207); w.b1 = take(PV * 64); w.x = take(PV * 32);
  w.v1h = take(PV * 32); w.xv = take(PV * 33); w.v1 = take(PV); w.xr = take(PV * 37); w.xv1 = take(PV * 32);
  w.f1 = take(PV * 32); w.wgt2 = take(PV);
  w.gin = take(P * 65); w.g1 = take(P * 64); w.g2 = take(P * 16); w.gp = take(P * 16); w.q = take(P * 16); w.k = take(P * 16);
  w.v = take(P * 16); w.o = take(P * 16); w.fo = take(P * 16); w.ln = take(P * 16); w.og1 = take(P * 16); w.sigma = take(P);
  w.r1 = take(PV * 16); w.r2 = take(PV * 8); w.logit = take(PV); w.blend = take(PV); w.colors = take(P * 3);
  w.gsigma = take(P); w.gcol = take(P * 3); w.glogit = take(PV); w.gr2 = take(PV * 8); w.gr1 = take(PV * 16); w.gxr = take(PV * 37);
  w.gog1 = take(P * 16); w.gln = take(P * 16); w.gfo = take(P * 16); w.go = take(P * 16); w.gq = take(P * 16); w.gk = take(P * 16);
  w.gv = take(P * 16); w.ggp = take(P * 16); w.gg1 = take(P * 64); w.ggin = take(P * 65); w.gwgt2 = take(PV); w.gf1 = take(PV * 32);
  w.gxv1 = take(PV * 32); w.gx = take(PV * 32); w.gxv = take(PV * 33); w.gv1h = take(PV * 32);
  w.gb1 = take(PV * 64); w.ghb = take(P * 64); w.gxin = take(PV * 207); w.gglob = take(P * 140); w.gw0 = take(PV); w.gn1 = take(PV * 8);
  w.gdfeat = take(PV * 35); w.gd1 = take(PV * 16); w.gpe_h = take(PV * 32); w.gpe_in = take(PV * 34);
  for (int m = 0; m < 4; ++m) { w.gdout[m] = take(PV * 2); w.gdh2[m] = take(PV * 32); w.gdh1[m] = take(PV * 32); }
  (void)take(64);      // the chain kernels' dwordx4 loads may touch 12 bytes past the last row of an activation buffer
  return off;
}

constexpr int kAux = 2;
struct Ctx {
  hipStream_t st;
  const float* const* raw;     // device pointers, reference layouts
  float* const* graw;          // device pointers (accumulated), entries may be null = not wanted
  float* grep;                 // kGradReplicas - 1 further replicas of every weight gradient (workspace, zeroed per pass)
  int* gsize;                  // floats per raw tensor, recorded by lin_bwd for the final fold (host array)
  // The weight-gradient launches on a SIDE STREAM: they only consume (delta, X) pairs and nothing of the pass reads their output before
  // the final fold, and neither they nor the chain kernels they run beside fill the memory pipes alone (2-3 TB/s each).  fork(): the side
  // stream waits for what the main stream has enqueued so far (delta of the layers about to be contracted); join(): the main stream waits
  // for the side stream.  Null = everything on the caller's stream.
  hipStream_t* aux;            // kAux side streams (null = none), used round robin: one per fork()
  hipEvent_t* ev;              // ring of events (re-recording an event does not disturb waits already enqueued on it)
  int* ev_next;
  int ev_count;
  std::mutex* ev_mu;           // the event ring is shared by every pass on the device: slot choice + record + wait are one critical section
                               // (two host threads -- an autograd backward thread and the main thread -- may run passes on one device)
  mutable int cur = 0;         // side stream of the current fork: per-call state (this Ctx), not per-device
  hipStream_t dw_stream() const { return aux ? aux[cur] : st; }
  void fork() const {
    if (!aux) return;
    cur = (cur + 1) % kAux;
    std::lock_guard<std::mutex> lock(*ev_mu);
    hipEvent_t e = ev[(*ev_next)++ % ev_count];
    (void)hipEventRecord(e, st);
    (void)hipStreamWaitEvent(aux[cur], e, 0);
  }
  void join() const {
    if (!aux) return;
    std::lock_guard<std::mutex> lock(*ev_mu);
    for (int i = 0; i < kAux; ++i) {
      hipEvent_t e = ev[(*ev_next)++ % ev_count];
      (void)hipEventRecord(e, aux[i]);
      (void)hipStreamWaitEvent(st, e, 0);
    }
  }
};
// Debug mode PGRF_TRAIN_DETERMINISTIC=1 (tests/test_grad_fixtures.py): every weight-gradient element is produced by ONE workgroup that
// walks all rows in index order and adds once into a zeroed buffer -- no replicas, no side streams, one block for the bias column sums
// and LayerNorm's parameter sums: the result no longer depends on the order in which float atomics happen to land.  Several times
// slower (one 32 x 32 tile = one wave group over all rows); never the default.
static bool train_deterministic() {
  static const bool on = [] { const char* e = getenv("PGRF_TRAIN_DETERMINISTIC"); return e && e[0] == '1'; }();
  return on;
}
// one side stream and event ring per device, created on first use and kept (PGRF_TRAIN_DW_STREAM=0: none)
struct SideStream { hipStream_t s[kAux]; hipEvent_t ev[32]; int next; bool ok; std::mutex mu; };
static SideStream* side_stream() {
  static const bool off = [] { const char* e = getenv("PGRF_TRAIN_DW_STREAM"); return e && e[0] == '0'; }();
  if (off || train_deterministic()) return nullptr;
  static std::mutex mu;
  static SideStream* per_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!per_dev[dev]) {
    SideStream* ss = new SideStream();
    ss->next = 0;
    ss->ok = true;
    for (int i = 0; ss->ok && i < kAux; ++i) ss->ok = hipStreamCreateWithFlags(&ss->s[i], hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ss->ok && i < 32; ++i) ss->ok = hipEventCreateWithFlags(&ss->ev[i], hipEventDisableTiming) == hipSuccess;
    per_dev[dev] = ss;
  }
  return per_dev[dev]->ok ? per_dev[dev] : nullptr;
}

// map gradient += transpose of the bilinear gather: the per-ray merging kernel (PGRF_TRAIN_SCATTER_RAY=0: one atomic per sample and tap)
static void scatter32(hipStream_t st, float* gmap, long view_stride, const int* to, const float* tw, int rn, int dn, int V, const float* g, int ldg, int gc) {
  static const bool per_ray = [] { const char* e = getenv("PGRF_TRAIN_SCATTER_RAY"); return !(e && e[0] == '0'); }();
  const long PV = (long)rn * dn * V;
  if (per_ray) hipLaunchKernelGGL(gather32_bwd_ray_kernel, dim3(nblk((long)rn * V * kGW, 4)), dim3(256), 0, st, gmap, view_stride, to, tw, rn, dn, V, g, ldg, gc);
  else hipLaunchKernelGGL(gather32_bwd_kernel, dim3(nblk(PV * 32)), dim3(256), 0, st, gmap, view_stride, to, tw, PV, V, g, ldg, gc);
}
static int lin_fwd(const Ctx& c, const float* X, int ldx, int rw, int rb, float* Y, int ldy, long rows, int IN, int OUT, int act) {
  // Y = act(X W^T + b) in one launch (bias and activation in the epilogue): 128 rows per workgroup, all OUT <= 64 columns per wave
  const dim3 grid((unsigned)((rows + 127) / 128), 1, 1);
  const float* bias = rb >= 0 ? c.raw[rb] : nullptr;
  if (OUT <= 32) hipLaunchKernelGGL((lin_mfma_kernel<false, 1>), grid, dim3(256), 0, c.st, X, ldx, c.raw[rw], IN, bias, Y, ldy, rows, OUT, IN, act, 0);
  else hipLaunchKernelGGL((lin_mfma_kernel<false, 2>), grid, dim3(256), 0, c.st, X, ldx, c.raw[rw], IN, bias, Y, ldy, rows, OUT, IN, act, 0);
  PGRF_TCHECK("train linear forward");
  return PGRF_OK;
}
// gW += delta^T X and gb += colsum(delta) (one launch): the reduction runs over the ROWS (sample points x views), split over gridDim.z
// so that the launch has ~1 536 workgroups; partial tiles are summed inside a workgroup, then one atomic per element and workgroup
static void L_dw(const Ctx& c, const float* X, int ldx, const float* delta, int ldd, int rw, int rb, long rows, int IN, int OUT, int wld = 0, int wcol = 0, float xscale = 1.f) {
  float* gb = (rb >= 0) ? c.graw[rb] : nullptr;
#ifdef PGRF_TRAIN_ABL_NO_DW   // timing experiment: no weight-gradient products at all (the bound of any fusion of them into the chains)
  return;
#endif
  if (!c.graw[rw]) {
    if (gb) {          // bias gradient alone (the weight's is not wanted): column sums by the elementwise kernel
      int cp = 1;
      while (cp < OUT) cp <<= 1;
      const long nb = (rows * cp + 256 * 8 - 1) / (256 * 8);
      hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)(train_deterministic() ? 1 : (nb < 1 ? 1 : (nb > 1024 ? 1024 : nb)))), dim3(256), 0, c.dw_stream(), const_cast<float*>(delta), ldd, delta, ldd, rows, OUT, cp, (int)ACT_NONE, gb);
    }
    return;
  }
  // n tiles per wave: ONE, whatever the width (measured, 2 048-ray step: base_fc.0's 207 columns as 4 / 2 / 1 tiles per wave: 8.87 / 8.82 / 8.75 ms --
  // more and lighter waves hide the load latency better than sharing delta's rows between a wave's tiles saves traffic)
  static const int wide_nt = [] { const char* e = getenv("PGRF_TRAIN_DW_NT"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 1; }();
  static const int mid_nt = [] { const char* e = getenv("PGRF_TRAIN_DW_NT2"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2) ? v : 1; }();
  const int nt = IN > 64 ? wide_nt : (IN > 32 ? mid_nt : 1);
  const int gx = (IN + 32 * nt - 1) / (32 * nt), gy = (OUT + 31) / 32, groups = gx * gy;
  const int gpw = groups >= 4 ? 4 : (groups == 2 ? 2 : 1);                  // groups per workgroup
  const int nwg = (groups + gpw - 1) / gpw;
  static const int target_wgs = [] { const char* e = getenv("PGRF_TRAIN_DW_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 768; }();   // tuning knob (LDS-staged kernel: 768: 6.74, 1024: 6.82, 1536: 6.85, 2048: 6.98 ms per step; the dword-load kernel it replaced: (with the gradient replicas: 192: 18.6, 384: 15.9, 768: 14.5, 1536: 14.1, 3072: 14.5, 6144: 14.1 ms per 2 048-ray step)
  long chunk = (rows * nwg / target_wgs + 63) / 64 * 64;                    // ~target_wgs workgroups per launch
  if (chunk < 128) chunk = 128;
  if (train_deterministic()) chunk = (rows + 63) / 64 * 64;                 // one workgroup per output tile, rows in index order
  const dim3 grid(nwg, 1, (unsigned)((rows + chunk - 1) / chunk));
  float* rep = c.grep + (size_t)rw * kGradSlot;          // this tensor's slot of replica 1 (replica r at + (r - 1) * R_COUNT * kGradSlot)
  float* brep = gb ? c.grep + (size_t)rb * kGradSlot : nullptr;
  const int ldg = wld ? wld : IN;
  c.gsize[rw] = OUT * ldg;
  if (gb) c.gsize[rb] = OUT;
  static const bool use_lds = [] { const char* e = getenv("PGRF_TRAIN_DW_LDS"); return !(e && e[0] == '0'); }();
  if (nt == 1 && use_lds) {
    hipLaunchKernelGGL(dw_lds_kernel, grid, dim3(256), 0, c.dw_stream(), delta, ldd, X, ldx, c.graw[rw], rep, OUT, IN, rows, (int)chunk, gx, groups, gb, brep, ldg, wcol, xscale);
    return;
  }
  if (nt == 4) hipLaunchKernelGGL((dw_mfma_kernel<4>), grid, dim3(256), 0, c.dw_stream(), delta, ldd, X, ldx, c.graw[rw], rep, OUT, IN, rows, (int)chunk, gx, groups, gb, brep, ldg, wcol, xscale);
  else if (nt == 2) hipLaunchKernelGGL((dw_mfma_kernel<2>), grid, dim3(256), 0, c.dw_stream(), delta, ldd, X, ldx, c.graw[rw], rep, OUT, IN, rows, (int)chunk, gx, groups, gb, brep, ldg, wcol, xscale);
  else hipLaunchKernelGGL((dw_mfma_kernel<1>), grid, dim3(256), 0, c.dw_stream(), delta, ldd, X, ldx, c.graw[rw], rep, OUT, IN, rows, (int)chunk, gx, groups, gb, brep, ldg, wcol, xscale);
}
// dY (a [rows, OUT] slice, ld lddy) is overwritten by delta = dY * act'(Y); then gW += delta^T X, gb += colsum(delta),
// dX (=|+=) delta W
static int lin_bwd(const Ctx& c, const float* X, int ldx, const float* Y, int ldy, float* dY, int lddy, int rw, int rb, float* dX, int lddx,
                   int acc_dx, long rows, int IN, int OUT, int act) {
  if (act != ACT_NONE)
  {
    int cp = 1;
    while (cp < OUT) cp <<= 1;
    if (cp > 64) return set_error(PGRF_ERR_UNSUPPORTED, "train linear backward: %d outputs (at most 64)", OUT);
    const long nb = (rows * cp + 256 * 8 - 1) / (256 * 8);            // ~8 rows per thread
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb)))   /* (<= 1 024 same-address atomics per bias element) */, dim3(256), 0, c.st, dY, lddy, Y, ldy, rows, OUT, cp,
                       act, (float*)nullptr);
  }
  c.fork();
  L_dw(c, X, ldx, dY, lddy, rw, rb, rows, IN, OUT);
  if (dX) {
    // dX (=|+=) delta W: the same kernel with the weight read transposed (W is [OUT, IN]: "M" = IN, "K" = OUT)
    const int mt = IN > 32 ? 2 : 1;
    const dim3 grid((unsigned)((rows + 127) / 128), (unsigned)((IN + 32 * mt - 1) / (32 * mt)), 1);
    if (mt == 2) hipLaunchKernelGGL((lin_mfma_kernel<true, 2>), grid, dim3(256), 0, c.st, dY, lddy, c.raw[rw], IN, (const float*)nullptr, dX, lddx, rows, IN, OUT, (int)ACT_NONE, acc_dx);
    else hipLaunchKernelGGL((lin_mfma_kernel<true, 1>), grid, dim3(256), 0, c.st, dY, lddy, c.raw[rw], IN, (const float*)nullptr, dX, lddx, rows, IN, OUT, (int)ACT_NONE, acc_dx);
  }
  PGRF_TCHECK("train linear backward");
  return PGRF_OK;
}


// A chain of <= 3 Linear layers in one launch (train_kernels.h chain_fwd_kernel / chain_bwd_kernel).  Layer k + 1 reads layer k's
// output; widths after the first layer's input <= 64.
struct Lin { const float* X; int ldx; float* Y; int ldy; float* dY; int lddy; int rw, rb, IN, OUT, act; int wcol = 0, wld = 0; float xscale = 1.f; };     // wcol / wld: columns [wcol, wcol + IN) of a weight with wld columns
static int chain_check(const Lin* l, int n) {
  if (n < 1 || n > 3) return set_error(PGRF_ERR_ARG, "train chain: %d layers", n);
  for (int k = 0; k < n; ++k) {
    if ((l[k].act == ACT_SOFTPLUS || l[k].act == ACT_SOFTPLUS_B) && l[k].OUT > 4)
      return set_error(PGRF_ERR_UNSUPPORTED, "train chain: softplus on %d outputs (at most 4)", l[k].OUT);
    if (l[k].OUT > 64 || (k > 0 && (l[k].IN > 64 || l[k].X != l[k - 1].Y || l[k].ldx != l[k - 1].ldy || l[k].IN != l[k - 1].OUT)))
      return set_error(PGRF_ERR_UNSUPPORTED, "train chain: layer %d does not chain (IN %d, OUT %d)", k, l[k].IN, l[k].OUT);
  }
  return PGRF_OK;
}
// LDS copy of the chain's weights: per layer chain_rows(OUT) rows of stride S, then 64 floats of bias; 64 floats of slack at the end
static size_t chain_layout(ChainArgs& a) {
  int off = 0;
  for (int k = 0; k < a.n; ++k) {
    a.l[k].S = chain_stride(a.l[k].IN);
    a.l[k].off = off;
    off += chain_rows(a.l[k].OUT) * a.l[k].S + 64;
  }
  a.lds_floats = off + 64;
  return (size_t)a.lds_floats * 4;
}
// persistent workgroups: as many as fit the chip at once (LDS- or register-bound), each walking row tiles
// every width a wave keeps in registers <= 32: the QD = 4 kernels
static bool chain_narrow(const Lin* l, int n) {
  for (int k = 0; k < n; ++k)
    if (l[k].OUT > 32 || (k > 0 && l[k].IN > 32)) return false;
  return true;
}
static int chain_wpb() { static const int v = [] { const char* e = getenv("PGRF_TRAIN_CHAIN_WPB"); return (e && atoi(e) == 4) ? 4 : 8; }(); return v; }   // waves per chain workgroup (8: half the workgroups = half the weight stagings of a launch: 5.45 -> 5.39 ms per step; 4 = round 4)
// Grid of a persistent chain launch = resident workgroups of THIS instance (a workgroup stages the chain's weights once and then walks
// its tiles): the runtime's occupancy answer for the instance's registers and LDS, not a constant -- the __launch_bounds__ second
// argument is waves per SIMD, so the eight-wave instances do not keep the four-wave ones' workgroups per CU (fwd<8,8> 166 VGPRs: one
// workgroup per CU, bwd<4,8> 112 VGPRs: two).  PGRF_TRAIN_CHAIN_WGS=n forces n per CU; PGRF_TRAIN_CHAIN_GRID=legacy = the round-5
// constants (max_per_cu counts four-wave workgroups).
static dim3 chain_grid(const void* kernel, int threads, long rows, size_t lds, int max_per_cu) {
  static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  static const int per_cu_env = [] { const char* e = getenv("PGRF_TRAIN_CHAIN_WGS"); return e ? atoi(e) : 0; }();
  static const bool legacy = [] { const char* e = getenv("PGRF_TRAIN_CHAIN_GRID"); return e && !strcmp(e, "legacy"); }();
  const int wpb = threads / 64;
  int per_cu;
  if (legacy) {
    if (wpb == 8) max_per_cu = (max_per_cu + 1) / 2;
    per_cu = (int)(160 * 1024 / (lds + 1024));
    if (per_cu > max_per_cu) per_cu = max_per_cu;
  } else {
    // cached per (instance, LDS bytes rounded up to 4 KB): a handful of distinct values per process
    struct Key { const void* k; size_t l; int v; };
    static std::mutex mu;
    static std::vector<Key> cache;
    const size_t lkey = (lds + 4095) / 4096;
    per_cu = -1;
    {
      std::lock_guard<std::mutex> g(mu);
      for (const Key& e : cache) if (e.k == kernel && e.l == lkey) { per_cu = e.v; break; }
    }
    if (per_cu < 0) {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lkey * 4096) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
      per_cu = n;
      std::lock_guard<std::mutex> g(mu);
      cache.push_back(Key{kernel, lkey, n});
    }
  }
  if (per_cu < 1) per_cu = 1;
  if (per_cu_env > 0) per_cu = per_cu_env;
  const long tiles = (rows + 32 * wpb - 1) / (32 * wpb), cap = (long)ncu * per_cu;
  return dim3((unsigned)(tiles < cap ? tiles : cap));
}
template <typename K>
static dim3 chain_grid_of(K kernel, int threads, long rows, size_t lds, int max_per_cu) {
  return chain_grid(reinterpret_cast<const void*>(kernel), threads, rows, lds, max_per_cu);
}
static int chain_fwd(const Ctx& c, const Lin* l, int n, long rows, const float* add = nullptr, int ldadd = 0, int addV = 1) {
  if (int rc = chain_check(l, n)) return rc;
  ChainArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n = n; a.X = l[0].X; a.ldx = l[0].ldx; a.rows = rows; a.xscale = l[0].xscale;
  for (int k = 0; k < n; ++k)
    a.l[k] = ChainLayer{c.raw[l[k].rw] + l[k].wcol, l[k].rb >= 0 ? c.raw[l[k].rb] : nullptr, l[k].Y, nullptr, l[k].ldy, 0, l[k].IN, l[k].OUT, l[k].act, 0, 0, 0,
                        l[k].wld ? l[k].wld : l[k].IN};
  a.add = add; a.ldadd = ldadd; a.addV = addV > 0 ? addV : 1;
  const size_t lds = chain_layout(a);
  if (lds > 65536) return set_error(PGRF_ERR_UNSUPPORTED, "train chain: %zu bytes of weights do not fit the LDS copy", lds);
  if (chain_wpb() == 8) {
    if (chain_narrow(l, n)) hipLaunchKernelGGL((chain_fwd_kernel<4, 8>), chain_grid_of(chain_fwd_kernel<4, 8>, 512, rows, lds, 6), dim3(512), lds, c.st, a);
    else hipLaunchKernelGGL((chain_fwd_kernel<8, 8>), chain_grid_of(chain_fwd_kernel<8, 8>, 512, rows, lds, 3), dim3(512), lds, c.st, a);
  } else if (chain_narrow(l, n)) hipLaunchKernelGGL(chain_fwd_kernel<4>, chain_grid_of(chain_fwd_kernel<4>, 256, rows, lds, 6), dim3(256), lds, c.st, a);
  else hipLaunchKernelGGL(chain_fwd_kernel<8>, chain_grid_of(chain_fwd_kernel<8>, 256, rows, lds, 3), dim3(256), lds, c.st, a);
  PGRF_TCHECK("train chain forward");
  return PGRF_OK;
}
// every layer's dY is overwritten by its delta; weight and bias gradients accumulated; dX (=|+=) the bottom layer's delta W (null: none)
static int chain_bwd(const Ctx& c, const Lin* l, int n, long rows, float* dX, int lddx, int acc_dx) {
  if (int rc = chain_check(l, n)) return rc;
  ChainArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n = n; a.rows = rows; a.dX = dX; a.lddx = lddx; a.accumulate = acc_dx; a.xscale = l[0].xscale;
  for (int k = 0; k < n; ++k) {          // kernel order: top layer first
    const Lin& s = l[n - 1 - k];
    a.l[k] = ChainLayer{c.raw[s.rw] + s.wcol, nullptr, s.Y, s.dY, s.ldy, s.lddy, s.IN, s.OUT, s.act, (k + 1 < n || dX) ? 1 : 0, 0, 0, s.wld ? s.wld : s.IN};
  }
  const size_t lds = chain_layout(a);
  if (lds > 65536) return set_error(PGRF_ERR_UNSUPPORTED, "train chain: %zu bytes of weights do not fit the LDS copy", lds);
  if (chain_wpb() == 8) {
    if (chain_narrow(l, n)) hipLaunchKernelGGL((chain_bwd_kernel<4, 8>), chain_grid_of(chain_bwd_kernel<4, 8>, 512, rows, lds, 5), dim3(512), lds, c.st, a);
    else hipLaunchKernelGGL((chain_bwd_kernel<8, 8>), chain_grid_of(chain_bwd_kernel<8, 8>, 512, rows, lds, 2), dim3(512), lds, c.st, a);
  } else if (chain_narrow(l, n)) hipLaunchKernelGGL(chain_bwd_kernel<4>, chain_grid_of(chain_bwd_kernel<4>, 256, rows, lds, 5), dim3(256), lds, c.st, a);
  else hipLaunchKernelGGL(chain_bwd_kernel<8>, chain_grid_of(chain_bwd_kernel<8>, 256, rows, lds, 2), dim3(256), lds, c.st, a);
  c.fork();
  for (int k = n - 1; k >= 0; --k) L_dw(c, l[k].X, l[k].ldx, l[k].dY, l[k].lddy, l[k].rw, l[k].rb, rows, l[k].IN, l[k].OUT, l[k].wld, l[k].wcol, l[k].xscale);
  PGRF_TCHECK("train chain backward");
  return PGRF_OK;
}

}  // namespace train
}  // namespace pgrf

using namespace pgrf;
using namespace pgrf::train;

extern "C" {

size_t pgrf_render_pass_backward_workspace_bytes(int rfn, int rn, int dn) {
  if (rfn < 1 || rn < 1 || dn < 1) return 0;
  Ws w;
  return plan(w, (long)rn * dn, rfn, dn, nullptr);
}

int pgrf_render_pass_backward(const PgrfFrame* frame, const float* const* raw_dev, int n_raw, int use_vis, const float* coords,
                              const float* depth, int rn, int dn, const PgrfPassGrad* grad, float* const* g_raw_dev, float* g_img_feats,
                              float* g_ray_feats, const PgrfPassOut* fwd_out, void* workspace, size_t workspace_bytes, void* stream) {
  return pgrf_render_pass_backward_phase(frame, raw_dev, n_raw, use_vis, coords, depth, rn, dn, grad, g_raw_dev, g_img_feats, g_ray_feats, fwd_out, workspace,
                                         workspace_bytes, stream, 0);
}

int pgrf_render_pass_backward_phase(const PgrfFrame* frame, const float* const* raw_dev, int n_raw, int use_vis, const float* coords,
                                    const float* depth, int rn, int dn, const PgrfPassGrad* grad, float* const* g_raw_dev, float* g_img_feats,
                                    float* g_ray_feats, const PgrfPassOut* fwd_out, void* workspace, size_t workspace_bytes, void* stream, int phase) {
  if (phase < 0 || phase > 2) return set_error(PGRF_ERR_ARG, "pgrf_render_pass_backward_phase: phase %d (0 both, 1 staged forward, 2 backward sweep)", phase);
  const bool do_fwd = phase != 2, do_bwd = phase != 1;
  if (rn == 0) return PGRF_OK;             // empty batch: nothing to add to the gradients
  static float* const kNoGrads[R_COUNT] = {};
  if (!do_bwd && !g_raw_dev) g_raw_dev = kNoGrads;       // (the forward phase reads no gradient pointers)
  if (!frame || !raw_dev || !coords || !depth || (do_bwd && !grad) || !g_raw_dev) return set_error(PGRF_ERR_ARG, "pgrf_render_pass_backward: null pointer");
  if (n_raw != R_COUNT) return set_error(PGRF_ERR_ARG, "pgrf_render_pass_backward: expected %d raw tensors, got %d", R_COUNT, n_raw);
  if (frame->rfn < 1 || frame->rfn > PGRF_MAX_VIEWS) return set_error(PGRF_ERR_ARG, "pgrf_render_pass_backward: rfn=%d not in [1,%d]", frame->rfn, PGRF_MAX_VIEWS);
  if (dn < 3 || dn > PGRF_MAX_DN) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_render_pass_backward: dn=%d outside [3,%d]", dn, PGRF_MAX_DN);
  if (rn < 0) return set_error(PGRF_ERR_ARG, "pgrf_render_pass_backward: rn < 0");
  if (frame->dataset < PGRF_DATASET_M3D || frame->dataset > PGRF_DATASET_COFFEEAREA)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_render_pass_backward: dataset convention %d unknown", frame->dataset);
  for (int i = 0; i < R_COUNT; ++i) {
    const bool is_vis = i >= R_DD_VIS_0_W && i <= R_DD_VIS_4_B;
    if (!raw_dev[i] && !(is_vis && !use_vis)) return set_error(PGRF_ERR_ARG, "pgrf_render_pass_backward: raw weight %d is null", i);
  }
  const int V = frame->rfn;
  const long P = (long)rn * dn, PV = P * V;
  // (one-element-per-thread kernels are launched with up to 65535 * 16 workgroups: the bound keeps every such launch complete)
  if (PV > 8000000L || P > 4000000L)
    return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_render_pass_backward: batch too large (rn * dn = %ld > 4 000 000 or rn * dn * rfn = %ld > 8 000 000): split the rays", P, PV);
  Ws w;
  const size_t need = plan(w, P, V, dn, nullptr);
  if (!workspace || workspace_bytes < need)
    return set_error(PGRF_ERR_ARG, "pgrf_render_pass_backward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  plan(w, P, V, dn, (char*)workspace);
  int gsize[R_COUNT] = {};
  SideStream* ss = side_stream();
  Ctx c{(hipStream_t)stream, raw_dev, g_raw_dev, w.grep, gsize, ss ? ss->s : nullptr, ss ? ss->ev : nullptr, ss ? &ss->next : nullptr, 32, ss ? &ss->mu : nullptr};
  struct Joiner { const Ctx& c; ~Joiner() { c.join(); } } joiner{c};      // error returns included: the caller's stream never runs ahead of the side stream
  if (do_bwd) (void)hipMemsetAsync(w.grep, 0, (size_t)(kGradReplicas - 1) * kGradSlot * R_COUNT * 4, c.st);
  hipStream_t st = c.st;
  const int NM = use_vis ? 4 : 3;
  static const int kDD[4][6] = {{R_DD_MEAN_0_W, R_DD_MEAN_0_B, R_DD_MEAN_2_W, R_DD_MEAN_2_B, R_DD_MEAN_4_W, R_DD_MEAN_4_B},
                                {R_DD_VAR_0_W, R_DD_VAR_0_B, R_DD_VAR_2_W, R_DD_VAR_2_B, R_DD_VAR_4_W, R_DD_VAR_4_B},
                                {R_DD_AW_0_W, R_DD_AW_0_B, R_DD_AW_2_W, R_DD_AW_2_B, R_DD_AW_4_W, R_DD_AW_4_B},
                                {R_DD_VIS_0_W, R_DD_VIS_0_B, R_DD_VIS_2_W, R_DD_VIS_2_B, R_DD_VIS_4_W, R_DD_VIS_4_B}};
  static const int kDDout[4] = {2, 2, 1, 1};
  static const int kDDact[4] = {ACT_SOFTPLUS, ACT_SOFTPLUS_B, ACT_SIGMOID, ACT_SIGMOID};
  const float wgt = 1.f / ((float)V + 1e-8f);
  int rc;
#define L(expr) do { rc = (expr); if (rc) return rc; } while (0)

  // ------------------------------------------------------------------ constants: geometry, taps, source colours
  PassArgs a;
  std::memset(&a, 0, sizeof(a));
  a.imgs = frame->imgs; a.img_feats = frame->img_feats; a.ray_feats = frame->ray_feats; a.w2c = frame->w2c; a.ref_dr = frame->ref_depth_range;
  std::memcpy(a.c2w, frame->que_c2w, sizeof(a.c2w));
  a.que_near = frame->que_depth_range[0]; a.que_far = frame->que_depth_range[1];
  a.dataset = frame->dataset; a.perspective = frame->perspective ? 1 : 0;
  std::memcpy(a.kinv, frame->que_kinv, sizeof(a.kinv));
  a.H = frame->H; a.W = frame->W; a.hf = frame->hf; a.wf = frame->wf; a.hr = frame->hr; a.wr = frame->wr;
  a.coords = coords; a.depth = depth; a.rn = rn;
  GeomOut go{w.tap_rf_o, w.tap_rf_w, w.tap_if_o, w.tap_if_w, w.rgb, w.dd, w.depth_v, w.hw};
  if (do_fwd) hipLaunchKernelGGL(geometry_kernel, dim3(nblk(PV)), dim3(256), 0, st, a, dn, V, go);
  if (do_fwd) hipLaunchKernelGGL(fill_kernel, dim3(nblk(PV)), dim3(256), 0, st, w.w1, wgt, PV);
  if (do_fwd) hipLaunchKernelGGL(posenc_table_kernel, dim3(nblk((long)dn * 16)), dim3(256), 0, st, w.pos, dn);
  const long rf_stride = (long)frame->hr * frame->wr * 32, if_stride = (long)frame->hf * frame->wf * 32;
  if (do_fwd) hipLaunchKernelGGL(gather32_kernel, dim3(nblk(PV * 8)), dim3(256), 0, st, frame->ray_feats, rf_stride, w.tap_rf_o, w.tap_rf_w, PV, V, w.pe_in, 34, 0);
  PGRF_TCHECK("train geometry");

  // ------------------------------------------------------------------ forward, every activation kept
  // the MLPs as register-resident chains (chain_fwd_kernel); *_l arrays are reused by the backward
  Lin dd_l[4][3];
  for (int m = 0; m < NM; ++m) {        // dist decoder MLPs on the gathered ray feature (dist_decoder.py:53-107)
    dd_l[m][0] = Lin{w.pe_in, 34, w.dh1[m], 32, w.gdh1[m], 32, kDD[m][0], kDD[m][1], 32, 32, ACT_ELU};
    dd_l[m][1] = Lin{w.dh1[m], 32, w.dh2[m], 32, w.gdh2[m], 32, kDD[m][2], kDD[m][3], 32, 32, ACT_ELU};
    dd_l[m][2] = Lin{w.dh2[m], 32, w.dout[m], 2, w.gdout[m], 2, kDD[m][4], kDD[m][5], 32, kDDout[m], kDDact[m]};
    if (do_fwd) L(chain_fwd(c, dd_l[m], 3, PV));
  }
  ProbIO pio{w.depth_v, w.hw, frame->ref_depth_range, w.dout[0], w.dout[1], w.dout[2], use_vis ? w.dout[3] : nullptr, w.vis, w.hit};
  if (do_fwd) hipLaunchKernelGGL(prob_kernel, dim3(nblk(PV)), dim3(256), 0, st, pio, PV, V, w.pe_in);
  PGRF_TCHECK("train compute_prob");
  const Lin pe_l[2] = {{w.pe_in, 34, w.pe_h, 32, w.gpe_h, 32, R_PE_0_W, R_PE_0_B, 34, 32, ACT_RELU},               // prob_embed (aggregate_net.py:53-61)
                       {w.pe_h, 32, w.xin + 175, 207, w.gxin + 175, 207, R_PE_2_W, R_PE_2_B, 32, 32, ACT_NONE}};   //   -> neuray feature, xin[:, 175:207]
  if (do_fwd) L(chain_fwd(c, pe_l, 2, PV));
  const Lin rd_l[2] = {{w.dd, 4, w.d1, 16, w.gd1, 16, R_RD_0_W, R_RD_0_B, 4, 16, ACT_ELU},                          // ray_dir_fc
                       {w.d1, 16, w.dfeat, 35, w.gdfeat, 35, R_RD_2_W, R_RD_2_B, 16, 35, ACT_ELU}};
  if (do_fwd) L(chain_fwd(c, rd_l, 2, PV));
  if (do_fwd) hipLaunchKernelGGL(gather32_rgbfeat_kernel, dim3(nblk(PV * 8)), dim3(256), 0, st, frame->img_feats, if_stride, w.tap_if_o, w.tap_if_w, PV, V, w.rgb, w.dfeat, w.xin);
  const Lin nf_l[2] = {{w.xin + 175, 207, w.n1, 8, w.gn1, 8, R_NF_0_W, R_NF_0_B, 32, 8, ACT_ELU},                   // neuray_fc -> weight0 (ibrnet.py:337-338)
                       {w.n1, 8, w.nfs, 1, w.gw0, 1, R_NF_2_W, R_NF_2_B, 8, 1, ACT_SIGMOID}};
  if (do_fwd) L(chain_fwd(c, nf_l, 2, PV));
  if (do_fwd) hipLaunchKernelGGL(rowscale_kernel, dim3(nblk(PV)), dim3(256), 0, st, w.w0, 1, 0, w.nfs, 1, 0, (const float*)nullptr, 0, wgt, PV, 1);
  if (do_fwd) hipLaunchKernelGGL(stats_kernel<2>, dim3(nblk(P * 35)), dim3(256), 0, st, w.xin, 207, 140, V, 35, P, w.glob, 140, StatFwd{w.w0, 0, 35}, StatFwd{w.w1, 70, 105});
  PGRF_TCHECK("train view statistics");
  // base_fc.0 in two column blocks, like the fused kernel's L_BG / L_BV: the 140 view statistics are the SAME for the V views of a point,
  // so their product with W[:, 0:140] is formed once per point (hb, P rows) and enters the per-view layer -- W[:, 140:207] on the 67
  // per-view columns -- as an additive term.  The broadcast copy of the statistics into xin[:, 0:140] (147 MB per pass), its columns of the
  // layer's dX and their reduction over the views, and 140 of the 207 columns of the (point x view)-row dW are gone.
  const Lin glob_l[1] = {{w.glob, 140, w.hb, 64, w.ghb, 64, R_BASE_0_W, -1, 140, 64, ACT_NONE, 0, 207}};
  if (do_fwd) L(chain_fwd(c, glob_l, 1, P));
  const Lin base_l[2] = {{w.xin + 140, 207, w.b1, 64, w.gb1, 64, R_BASE_0_W, R_BASE_0_B, 67, 64, ACT_ELU, 140, 207},    // base_fc
                         {w.b1, 64, w.x, 32, w.gx, 32, R_BASE_2_W, R_BASE_2_B, 64, 32, ACT_ELU}};
  if (do_fwd) L(chain_fwd(c, base_l, 2, PV, w.hb, 64, V));
  // vis_fc reads x * weight (ibrnet.py:343, weight = 1 / V here): the scale rides in the chain's first-layer load, in its dX and in its dW
  // (it was a [PV, 32] copy forward and a read-modify-write kernel backward)
  const Lin vf_l[2] = {{w.x, 32, w.v1h, 32, w.gv1h, 32, R_VF_0_W, R_VF_0_B, 32, 32, ACT_ELU, 0, 0, wgt},                      // vis_fc
                       {w.v1h, 32, w.xv, 33, w.gxv, 33, R_VF_2_W, R_VF_2_B, 32, 33, ACT_ELU}};
  if (do_fwd) L(chain_fwd(c, vf_l, 2, PV));
  if (do_fwd) hipLaunchKernelGGL(visfc_post_kernel, dim3(nblk(PV, 8)), dim3(256), 0, st, w.x, w.xv, PV, w.xr, w.v1, w.xv1);
  const Lin v2_l[2] = {{w.xv1, 32, w.f1, 32, w.gf1, 32, R_V2_0_W, R_V2_0_B, 32, 32, ACT_ELU},                       // vis_fc2 -> xr[:, 32]
                       {w.f1, 32, w.xr + 32, 37, w.gxr + 32, 37, R_V2_2_W, R_V2_2_B, 32, 1, ACT_SIGMOID}};
  if (do_fwd) L(chain_fwd(c, v2_l, 2, PV));
  if (do_fwd) hipLaunchKernelGGL(slice_kernel, dim3(nblk(PV * 4)), dim3(256), 0, st, w.xr, 37, 33, w.dd, 4, 0, PV, 4, 0);
  if (do_fwd) hipLaunchKernelGGL(wgt2_kernel, dim3(nblk(P)), dim3(256), 0, st, w.xr, P, V, w.wgt2, w.gin);
  if (do_fwd) hipLaunchKernelGGL(stats_kernel<1>, dim3(nblk(P * 32)), dim3(256), 0, st, w.xr, 37, 0, V, 32, P, w.gin, 65, StatFwd{w.wgt2, 0, 32}, StatFwd{});
  PGRF_TCHECK("train pooled geometry feature");
  const Lin geo_l[2] = {{w.gin, 65, w.g1, 64, w.gg1, 64, R_GEO_0_W, R_GEO_0_B, 65, 64, ACT_ELU},                    // geometry_fc
                        {w.g1, 64, w.g2, 16, w.ggp, 16, R_GEO_2_W, R_GEO_2_B, 64, 16, ACT_ELU}};      // (d(gp)/d(g2) = 1: the gradient arrives in ggp)
  if (do_fwd) L(chain_fwd(c, geo_l, 2, P));
  if (do_fwd) hipLaunchKernelGGL(posenc_kernel, dim3(nblk(P * 16)), dim3(256), 0, st, w.g2, w.pos, P, dn, w.gp);
  if (do_fwd) L(lin_fwd(c, w.gp, 16, R_ATT_Q_W, -1, w.q, 16, P, 16, 16, ACT_NONE));                              // ray attention
  if (do_fwd) L(lin_fwd(c, w.gp, 16, R_ATT_K_W, -1, w.k, 16, P, 16, 16, ACT_NONE));
  if (do_fwd) L(lin_fwd(c, w.gp, 16, R_ATT_V_W, -1, w.v, 16, P, 16, 16, ACT_NONE));
  if (do_fwd) {
    if (dn <= 64 && V > 1) hipLaunchKernelGGL(attn_fwd64_kernel, dim3((unsigned)rn * 4), dim3(64), 0, st, w.q, w.k, w.v, rn, dn, w.o);
    else hipLaunchKernelGGL(attn_fwd_kernel, dim3(nblk((long)rn * 4 * dn)), dim3(256), 0, st, w.q, w.k, w.v, rn, dn, V <= 1 ? 1 : 0, w.o);
  }
  if (do_fwd) L(lin_fwd(c, w.o, 16, R_ATT_FC_W, -1, w.fo, 16, P, 16, 16, ACT_NONE));
  if (do_fwd) hipLaunchKernelGGL(slice_kernel, dim3(nblk(P * 16)), dim3(256), 0, st, w.fo, 16, 0, w.gp, 16, 0, P, 16, 1);           // + residual
  if (do_fwd) hipLaunchKernelGGL(layernorm_kernel, dim3(nblk(P)), dim3(256), 0, st, w.fo, raw_dev[R_ATT_LN_W], raw_dev[R_ATT_LN_B], P, w.ln);
  const Lin og_l[2] = {{w.ln, 16, w.og1, 16, w.gog1, 16, R_OG_0_W, R_OG_0_B, 16, 16, ACT_ELU},                      // out_geometry_fc -> density
                       {w.og1, 16, w.sigma, 1, w.gsigma, 1, R_OG_2_W, R_OG_2_B, 16, 1, ACT_RELU}};
  if (do_fwd) L(chain_fwd(c, og_l, 2, P));
  const Lin rgb_l[3] = {{w.xr, 37, w.r1, 16, w.gr1, 16, R_RGB_0_W, R_RGB_0_B, 37, 16, ACT_ELU},                     // rgb_fc -> blending logits
                        {w.r1, 16, w.r2, 8, w.gr2, 8, R_RGB_2_W, R_RGB_2_B, 16, 8, ACT_ELU},
                        {w.r2, 8, w.logit, 1, w.glogit, 1, R_RGB_4_W, R_RGB_4_B, 8, 1, ACT_NONE}};
  if (do_fwd) L(chain_fwd(c, rgb_l, 3, PV));
  if (do_fwd) hipLaunchKernelGGL(blend_kernel, dim3(nblk(P)), dim3(256), 0, st, w.logit, w.rgb, P, V, w.blend, w.colors);
  PGRF_TCHECK("train forward tail");
  if (fwd_out && do_fwd) {      // the staged forward's own outputs (tests: must equal the fused kernel's to round-off)
    if (do_fwd) hipLaunchKernelGGL(composite_kernel, dim3(nblk(rn, 4)), dim3(256), 0, st, w.sigma, w.colors, depth, rn, dn, fwd_out->hit_prob,
                       fwd_out->pixel_colors, fwd_out->render_depth);
    if (fwd_out->density) (void)hipMemcpyAsync(fwd_out->density, w.sigma, P * 4, hipMemcpyDeviceToDevice, st);
    if (fwd_out->colors) (void)hipMemcpyAsync(fwd_out->colors, w.colors, P * 12, hipMemcpyDeviceToDevice, st);
  }

  // ------------------------------------------------------------------ backward
  if (do_bwd) {
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(nblk(rn, 4)), dim3(256), 0, st, w.sigma, w.colors, depth, rn, dn, grad->pixel_colors,
                     grad->render_depth, grad->hit_prob, grad->colors, grad->density, w.gsigma, w.gcol);
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(nblk(P)), dim3(256), 0, st, w.blend, w.rgb, w.gcol, P, V, w.glogit);
  PGRF_TCHECK("train composite backward");
  L(chain_bwd(c, rgb_l, 3, PV, w.gxr, 37, 0));                                                                      // gxr written (=)
  L(chain_bwd(c, og_l, 2, P, w.gln, 16, 0));
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(train_deterministic() ? 1 : (nblk(P, 512) > 1024 ? 1024 : nblk(P, 512))), dim3(256), 0, st, w.fo, raw_dev[R_ATT_LN_W], w.gln, P, w.gfo,
                     g_raw_dev[R_ATT_LN_W] ? g_raw_dev[R_ATT_LN_W] : w.gq /* scratch sink */, g_raw_dev[R_ATT_LN_B] ? g_raw_dev[R_ATT_LN_B] : w.gq);
  (void)hipMemcpyAsync(w.ggp, w.gfo, P * 64, hipMemcpyDeviceToDevice, st);                                            // residual branch
  L(lin_bwd(c, w.o, 16, w.fo, 16, w.gfo, 16, R_ATT_FC_W, -1, w.go, 16, 0, P, 16, 16, ACT_NONE));
  if (dn <= 64 && V > 1)      // probabilities of a (ray, head) kept in LDS: one exp per (query, key) instead of four; gq / gk / gv are WRITTEN
    hipLaunchKernelGGL(attn_bwd64_kernel, dim3((unsigned)rn * 4), dim3(64), 0, st, w.q, w.k, w.v, w.o, w.go, rn, dn, w.gq, w.gk, w.gv);
  else {
    (void)hipMemsetAsync(w.gq, 0, P * 64, st); (void)hipMemsetAsync(w.gk, 0, P * 64, st); (void)hipMemsetAsync(w.gv, 0, P * 64, st);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)rn * 4), dim3(dn <= 64 ? 64 : 128), 0, st, w.q, w.k, w.v, w.o, w.go, rn, dn,
                       V <= 1 ? 1 : 0, w.gq, w.gk, w.gv);
  }
  PGRF_TCHECK("train attention backward");
  L(lin_bwd(c, w.gp, 16, w.q, 16, w.gq, 16, R_ATT_Q_W, -1, w.ggp, 16, 1, P, 16, 16, ACT_NONE));
  L(lin_bwd(c, w.gp, 16, w.k, 16, w.gk, 16, R_ATT_K_W, -1, w.ggp, 16, 1, P, 16, 16, ACT_NONE));
  L(lin_bwd(c, w.gp, 16, w.v, 16, w.gv, 16, R_ATT_V_W, -1, w.ggp, 16, 1, P, 16, 16, ACT_NONE));
  L(chain_bwd(c, geo_l, 2, P, w.ggin, 65, 0));
  static const bool stats4 = [] { const char* e = getenv("PGRF_TRAIN_STATS4"); return !(e && e[0] == '0'); }();      // four features per lane (5.16 -> 5.10 ms per step); 0 = a lane per feature
  if (stats4) hipLaunchKernelGGL((stats_bwd4_kernel<1, true>), dim3(nblk(P, 16)), dim3(256), 0, st, w.xr, 37, 0, V, 32, P, w.gin, 65, w.ggin, 65, w.gxr, 37, 0,
                                 StatSet{w.wgt2, 0, 32, 0, 32, nullptr}, StatSet{}, 32, 64);      // + the gradient of v2 = xr[:, 32] (was wgt2_bwd_kernel)
  else {
  (void)hipMemsetAsync(w.gwgt2, 0, PV * 4, st);
  hipLaunchKernelGGL(stats_bwd_kernel<1>, dim3(nblk(P, 4 * kStatPts)), dim3(256), 0, st, w.xr, 37, 0, V, 32, P, w.gin, 65, w.ggin, 65, w.gxr, 37, 0,
                     StatSet{w.wgt2, 0, 32, 0, 32, w.gwgt2}, StatSet{});
  hipLaunchKernelGGL(wgt2_bwd_kernel, dim3(nblk(P)), dim3(256), 0, st, w.xr, w.gwgt2, w.ggin, P, V, w.gxr);
  }
  PGRF_TCHECK("train pooled geometry backward");
  L(chain_bwd(c, v2_l, 2, PV, w.gxv1, 32, 0));
  hipLaunchKernelGGL(visfc_post_bwd_kernel, dim3(nblk(PV, 8)), dim3(256), 0, st, w.gxr, w.gxv1, w.xr, w.v1, PV, w.gx, w.gxv);
  L(chain_bwd(c, vf_l, 2, PV, w.gx, 32, 1));                                             // gx += wgt * (delta W)
  PGRF_TCHECK("train visibility backward");
  L(chain_bwd(c, base_l, 2, PV, w.gxin + 140, 207, 0));                                  // d / d(per-view columns); delta of base_fc.0 left in gb1
  hipLaunchKernelGGL(reduce_views_kernel, dim3(nblk(P * 16)), dim3(256), 0, st, w.ghb, 64, 0, w.gb1, 64, 0, P, V, 64, 0);      // d hb = sum over the views
  L(chain_bwd(c, glob_l, 1, P, w.gglob, 140, 0));                                         // d glob = d hb W[:, 0:140]  (+ that block's dW, P rows)
  // both weightings of the per-view features (weight0 = sigmoid(neuray_fc) / V with a gradient, weight1 = 1 / V without) in one sweep;
  // weight0 = sigmoid(neuray_fc) * wgt: the gradient enters the sigmoid output scaled by wgt
  if (stats4) hipLaunchKernelGGL(stats_bwd4_kernel<2>, dim3(nblk(P, 16)), dim3(256), 0, st, w.xin, 207, 140, V, 35, P, w.glob, 140, w.gglob, 140, w.gxin, 207, 140,
                                 StatSet{w.w0, 0, 35, 0, 35, w.gw0, wgt}, StatSet{w.w1, 70, 105, 70, 105, nullptr});      // gw0 written, scaled
  else {
  (void)hipMemsetAsync(w.gw0, 0, PV * 4, st);
  hipLaunchKernelGGL(stats_bwd_kernel<2>, dim3(nblk(P, 4 * kStatPts)), dim3(256), 0, st, w.xin, 207, 140, V, 35, P, w.glob, 140, w.gglob, 140, w.gxin, 207, 140,
                     StatSet{w.w0, 0, 35, 0, 35, w.gw0}, StatSet{w.w1, 70, 105, 70, 105, nullptr});
  hipLaunchKernelGGL(rowscale_kernel, dim3(nblk(PV)), dim3(256), 0, st, w.gw0, 1, 0, w.gw0, 1, 0, (const float*)nullptr, 0, wgt, PV, 1);
  }
  PGRF_TCHECK("train view statistics backward");
  L(chain_bwd(c, nf_l, 2, PV, w.gxin + 175, 207, 1));                                                               // += onto the neuray columns
  hipLaunchKernelGGL(slice_kernel, dim3(nblk(PV * 35)), dim3(256), 0, st, w.gdfeat, 35, 0, w.gxin, 207, 140, PV, 35, 0);
  if (g_img_feats) {      // (nothing of the pass reads the map gradient: beside the remaining chains on a side stream)
    c.fork();
    scatter32(c.dw_stream(), g_img_feats, if_stride, w.tap_if_o, w.tap_if_w, rn, dn, V, w.gxin, 207, 143);
  }
  L(chain_bwd(c, rd_l, 2, PV, (float*)nullptr, 0, 0));
  L(chain_bwd(c, pe_l, 2, PV, w.gpe_in, 34, 0));
  // compute_prob: gradients of (hit, vis) -> mixture parameters
  ProbGrad pg{w.gpe_in, w.gdout[0], w.gdout[1], w.gdout[2], use_vis ? w.gdout[3] : nullptr};
  hipLaunchKernelGGL(prob_bwd_kernel, dim3(nblk(PV)), dim3(256), 0, st, pio, pg, PV, V);
  PGRF_TCHECK("train compute_prob backward");
  for (int m = 0; m < NM; ++m) {
    L(chain_bwd(c, dd_l[m], 3, PV, w.gpe_in, 34, 1));                                                               // += onto the ray feature
  }
  if (g_ray_feats)
    scatter32(st, g_ray_feats, rf_stride, w.tap_rf_o, w.tap_rf_w, rn, dn, V, w.gpe_in, 34, 0);
  PGRF_TCHECK("train gather backward");
  c.join();
  {   // fold the weight-gradient replicas into the caller's buffers
    GradPtrs gp;
    for (int i = 0; i < R_COUNT; ++i) { gp.p[i] = gsize[i] > 0 ? g_raw_dev[i] : nullptr; gp.n[i] = gsize[i]; }
    hipLaunchKernelGGL(reduce_replicas_kernel, dim3(13, R_COUNT), dim3(256), 0, st, w.grep, gp);
    PGRF_TCHECK("train replica fold");
  }
  }   // do_bwd
#undef L
  return PGRF_OK;
}

}  // extern "C"
