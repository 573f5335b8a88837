// One instantiation of the fused render-pass kernel per translation unit (compiled in parallel):
//   -DPGRF_KT=<dn/32> -DPGRF_V=<source views> -DPGRF_DBG=<0|1>
#include "render_pass.h"

#define PGRF_CAT2(a, b, c, d) a##_##b##_##c##_##d
#define PGRF_CAT(a, b, c, d) PGRF_CAT2(a, b, c, d)

namespace pgrf {
int PGRF_CAT(launch_rp, PGRF_KT, PGRF_V, PGRF_DBG)(const PassArgs& a, void* stream) {
  return launch_instance<PGRF_KT, PGRF_V, (PGRF_DBG != 0)>(a, stream);
}
}  // namespace pgrf
