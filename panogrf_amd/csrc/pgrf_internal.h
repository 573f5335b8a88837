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
}  // namespace pgrf
#include <cstdint>
#include <vector>
namespace pgrf {
// per blob word: the source of a f32 copy or of two split-f16 halves (d0 / d1: bit 0 valid, bit 1 second piece, bit 2 f16 word,
// bits 3.. = ext id << 14 | element), and the constant words (c0) -- what pack_device.hip's kernel evaluates
int build_pack_maps(int use_vis, int max_dn, std::vector<uint32_t>& d0, std::vector<uint32_t>& d1, std::vector<float>& c0);
int pack_map_selftest(int use_vis, int max_dn);

// kernel launchers (each in its own .hip translation unit)
struct PassArgs;
int launch_render_pass(const PassArgs& a, int rfn, int dn, bool debug, int precision, void* stream);

}  // namespace pgrf
