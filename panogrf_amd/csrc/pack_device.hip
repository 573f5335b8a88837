// Device-side weight packer: the MFMA-ordered blob of pgrf_pack_weights built on the GPU from DEVICE-resident raw tensors, stream
// ordered, no host round trip.  Training re-packs both passes' weights after every optimizer step; through the host packer that was
// a device->host copy (a stream sync), ~1-3 ms of CPU work and an upload per pass with the GPU idle meanwhile (round 4: ~2.3 ms of
// an 11.9 ms step).
//
// The blob layout is NOT restated here: pack_weights.cpp's build_pack_maps() probes the host packer once per (device, use_vis,
// max_dn) and returns, per 32-bit blob word, where it comes from -- a f32 copy of element e of tensor t, two split-f16 halves, or a
// constant.  The kernel below evaluates that map; the four folded tensors (prob_embed.2 into base_fc.0 / neuray_fc.0) are computed
// on the fly in the host's order of operations (double accumulation, index order: bit-identical, tests/test_pack_device.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "pgrf_experiments.h"
#include "pgrf_internal.h"
#include "pgrf_layout.h"

namespace pgrf {
namespace {

struct RawTable { const float* p[R_COUNT]; };

__device__ __forceinline__ float folded(const RawTable& r, int t, int e) {
  const float* Wp = r.p[R_PE_2_W];
  const float* bp = r.p[R_PE_2_B];
  if (t == X_BASE_FUSED) {
    const int o = e / 207, c = e % 207;
    if (c < 175) return r.p[R_BASE_0_W][e];
    const float* Wc = r.p[R_BASE_0_W] + o * 207 + 175;
    double acc = 0.0;
    for (int k = 0; k < 32; ++k) acc = acc + (double)Wc[k] * (double)Wp[k * 32 + (c - 175)];     // (products of floats: exact in double)
    return (float)acc;
  }
  if (t == X_NF_FUSED) {
    const int o = e / 32, j = e % 32;
    const float* Wc = r.p[R_NF_0_W] + o * 32;
    double acc = 0.0;
    for (int k = 0; k < 32; ++k) acc = acc + (double)Wc[k] * (double)Wp[k * 32 + j];
    return (float)acc;
  }
  const bool base = t == X_BASE_BIAS;
  const float* Wc = base ? r.p[R_BASE_0_W] + e * 207 + 175 : r.p[R_NF_0_W] + e * 32;
  double acc = base ? r.p[R_BASE_0_B][e] : r.p[R_NF_0_B][e];
  for (int k = 0; k < 32; ++k) acc = acc + (double)Wc[k] * (double)bp[k];
  return (float)acc;
}

__device__ __forceinline__ float source(const RawTable& r, const float* const* tab, uint32_t d) {
  const int t = (int)(d >> 17), e = (int)((d >> 3) & 16383u);
  if (t >= R_COUNT) return folded(r, t, e);
  const float* p = tab[t];
  return p ? p[e] : 0.f;                       // (vis_decoder tensors without use_vis: never referenced by the map, guarded anyway)
}
__device__ __forceinline__ uint32_t split_half(const RawTable& r, const float* const* tab, uint32_t d) {
  if (!(d & 1u)) return 0u;
  const float x = source(r, tab, d);
  const _Float16 h1 = (_Float16)x;
  const _Float16 h = (d & 2u) ? (_Float16)(x - (float)h1) : h1;
  return (uint32_t)__builtin_bit_cast(unsigned short, h);
}

__global__ void __launch_bounds__(256) pack_gather_kernel(const RawTable r, const uint32_t* __restrict__ d0, const uint32_t* __restrict__ d1,
                                                          const float* __restrict__ c0, float* __restrict__ blob, int n) {
  __shared__ const float* tab[R_COUNT];
  if (threadIdx.x < R_COUNT) tab[threadIdx.x] = r.p[threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = d0[i];
  if (!(a & 4u)) {
    blob[i] = (a & 1u) ? source(r, tab, a) : c0[i];
    return;
  }
  const uint32_t w = split_half(r, tab, a) | (split_half(r, tab, d1[i]) << 16);
  blob[i] = __uint_as_float(w);
}

struct Maps { uint32_t *d0, *d1; float* c0; };
std::mutex g_mu;
std::map<std::tuple<int, int, int>, Maps> g_maps;

int get_maps(int use_vis, int max_dn, Maps* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return set_error(PGRF_ERR_LAUNCH, "pgrf_pack_weights_device: no current device");
  std::lock_guard<std::mutex> lock(g_mu);
  const auto key = std::make_tuple(dev, use_vis ? 1 : 0, max_dn);
  auto it = g_maps.find(key);
  if (it == g_maps.end()) {
    std::vector<uint32_t> d0, d1;
    std::vector<float> c0;
    if (int rc = build_pack_maps(use_vis, max_dn, d0, d1, c0)) return rc;
    Maps m{nullptr, nullptr, nullptr};
    const size_t nb = (size_t)kBlobFloats * 4;
    if (hipMalloc((void**)&m.d0, nb) != hipSuccess || hipMalloc((void**)&m.d1, nb) != hipSuccess || hipMalloc((void**)&m.c0, nb) != hipSuccess)
      return set_error(PGRF_ERR_LAUNCH, "pgrf_pack_weights_device: hipMalloc of the source map failed");
    // (synchronous copies, once per (device, use_vis, max_dn): the host vectors die with this scope)
    if (hipMemcpy(m.d0, d0.data(), nb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(m.d1, d1.data(), nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m.c0, c0.data(), nb, hipMemcpyHostToDevice) != hipSuccess)
      return set_error(PGRF_ERR_LAUNCH, "pgrf_pack_weights_device: upload of the source map failed");
    it = g_maps.emplace(key, m).first;
  }
  *out = it->second;
  return PGRF_OK;
}

}  // namespace
}  // namespace pgrf

using namespace pgrf;

extern "C" int pgrf_pack_weights_device(const float* const* raw_dev, int n_raw, int use_vis, int max_dn, float* blob_dev, void* stream) {
  if (!raw_dev || !blob_dev) return set_error(PGRF_ERR_ARG, "pgrf_pack_weights_device: null argument");
  if (n_raw != R_COUNT) return set_error(PGRF_ERR_ARG, "pgrf_pack_weights_device: expected %d raw tensors, got %d", R_COUNT, n_raw);
  RawTable r;
  for (int i = 0; i < R_COUNT; ++i) {
    const bool is_vis = (i >= R_DD_VIS_0_W && i <= R_DD_VIS_4_B);
    if (!raw_dev[i] && !(is_vis && !use_vis)) return set_error(PGRF_ERR_ARG, "pgrf_pack_weights_device: raw tensor %d is null", i);
    r.p[i] = raw_dev[i];
  }
  Maps m;
  if (int rc = get_maps(use_vis, max_dn, &m)) return rc;
  hipLaunchKernelGGL(pack_gather_kernel, dim3((kBlobFloats + 255) / 256), dim3(256), 0, (hipStream_t)stream, r, m.d0, m.d1, m.c0, blob_dev, kBlobFloats);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(PGRF_ERR_LAUNCH, "pgrf_pack_weights_device: %s", hipGetErrorString(e));
  return PGRF_OK;
}
