// One instantiation of the fused render-pass kernel per translation unit (compiled in parallel):
//   -DPGRF_KT=<dn/32> -DPGRF_V=<source views> -DPGRF_DBG=<0|1> -DPGRF_PREC=<0 f32 | 1 split-f16>
#include "render_pass.h"

#define PGRF_CAT2(a, b, c, d, e) a##_##b##_##c##_##d##_##e
#define PGRF_CAT(a, b, c, d, e) PGRF_CAT2(a, b, c, d, e)

namespace pgrf {
int PGRF_CAT(launch_rp, PGRF_KT, PGRF_V, PGRF_DBG, PGRF_PREC)(const PassArgs& a, void* stream) {
  return PGRF_PREC_NS::launch_instance<PGRF_KT, PGRF_V, (PGRF_DBG != 0)>(a, stream);
}
}  // namespace pgrf

#ifdef PGRF_STAGE_TIMING
// experiment builds only: read (and clear) the per-stage cycle table
extern "C" int pgrf_exp_stage_cycles(unsigned long long* out32) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(pgrf::PGRF_PREC_NS::g_stage_cycles), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  unsigned long long z[32] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(pgrf::PGRF_PREC_NS::g_stage_cycles), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
