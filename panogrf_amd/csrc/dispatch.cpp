// Selects the render-pass instantiation for (samples per ray, source views, debug, precision).
#include "pgrf_internal.h"
#include "render_pass_args.h"

namespace pgrf {

#define PGRF_DECL(KT, V, D, P) int launch_rp_##KT##_##V##_##D##_##P(const PassArgs&, void*);
#define PGRF_FOR_ALL(X) \
  X(1, 1, 0, 0) X(2, 1, 0, 0) X(4, 1, 0, 0) X(1, 1, 1, 0) X(1, 1, 0, 1) X(2, 1, 0, 1) X(4, 1, 0, 1)                             \
  X(1, 2, 0, 0) X(2, 2, 0, 0) X(4, 2, 0, 0) X(1, 3, 0, 0) X(2, 3, 0, 0) X(4, 3, 0, 0) X(1, 4, 0, 0) X(2, 4, 0, 0) X(4, 4, 0, 0) \
  X(1, 2, 1, 0) X(2, 2, 1, 0) X(1, 3, 1, 0) X(1, 4, 1, 0)                                                                   \
  X(1, 2, 0, 1) X(2, 2, 0, 1) X(4, 2, 0, 1) X(1, 3, 0, 1) X(2, 3, 0, 1) X(4, 3, 0, 1) X(1, 4, 0, 1) X(2, 4, 0, 1) X(4, 4, 0, 1)
PGRF_FOR_ALL(PGRF_DECL)

int launch_render_pass(const PassArgs& a, int rfn, int dn, bool debug, int precision, void* stream) {
  const int kt = dn / 32;
#define PGRF_CASE(KT, V, D, P) \
  if (kt == KT && rfn == V && (int)debug == D && precision == P) return launch_rp_##KT##_##V##_##D##_##P(a, stream);
  PGRF_FOR_ALL(PGRF_CASE)
  return set_error(PGRF_ERR_UNSUPPORTED,
                   "render pass: no kernel for rfn=%d dn=%d debug=%d precision=%d (rfn in 1..4; dn in {32,64,128}; debug "
                   "dumps only at f32 precision, dn<=64 at rfn=2, dn=32 otherwise)",
                   rfn, dn, (int)debug, precision);
}

}  // namespace pgrf
