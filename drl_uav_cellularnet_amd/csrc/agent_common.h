// Shared by the two translation units of libuavagent.so (not part of the C ABI: hidden visibility).
#pragma once
#include <string>

namespace uavagent_internal {
// Records the thread-local message uavagent_last_error() returns and hands `code` back.
__attribute__((visibility("hidden"))) int fail(int code, const std::string &msg);
}  // namespace uavagent_internal

// Softmax numerator exp(x - max) of the ACTION DRAW (uavagent_sample_actions and the actor head's fused draw: the two must stay bit-identical),
// x - max <= 0: the hardware's exp2 (v_exp_f32, ~1 ulp; the scaled argument adds <= 2e-6 relative at x - max = -35) instead of the
// library's expf.  s_memtime stamps of the head (tools/head_stamps.py, profiles/r04hs_*) put 15-17 % of a workgroup's life in its draw, and
// a row's ~800 instructions were half expf (10 per lane, ~40 instructions each).  Underflow to 0 is the exact limit.
#ifdef __HIPCC__
__device__ __forceinline__ float draw_exp(float x_minus_max) { return __builtin_amdgcn_exp2f(x_minus_max * 1.4426950408889634f); }
#endif
