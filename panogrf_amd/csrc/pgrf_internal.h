// Internal declarations shared by the translation units of libpanogrf_hip.so.
#pragma once
#include <cstdarg>
#include <cstddef>

#include "../../include/panogrf_hip.h"

namespace pgrf {

// records the message for pgrf_last_error() (thread local) and returns `code`
int set_error(int code, const char* fmt, ...);

// host packer (pack_weights.cpp)
int pack_weights_host(const float* const* raw, int n_raw, int use_vis, int max_dn, float* blob);

// kernel launchers (each in its own .hip translation unit)
struct PassArgs;
int launch_render_pass(const PassArgs& a, int rfn, int dn, bool debug, int precision, void* stream);

}  // namespace pgrf
