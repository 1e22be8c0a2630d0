// Exact unsigned 32-bit division by a small run-time constant without a divide: q = a / d for every a < 2^32.
// Branch-free scheme of libdivide (Kerr, "labor of division"): q0 = mulhi(a, magic); t = ((a - q0) >> 1) + q0;
// q = t >> shift.  Powers of two use magic = 0, shift = log2(d) - 1.  The generator runs once on the host
// (uavenv_create); the kernels use it for the base-n_act digits of the joint action (Decimal_to_Base_N,
// ue_mobility.py:310-336) instead of four emulated v_rcp-based divisions.  Exhaustively checked against `/` for
// d = 2..9 in tests/native/lean_math_check.cpp.
#pragma once
#include <stdint.h>

#include "philox.h"  // UAVENV_HD

namespace uavk {

UAVENV_HD void u32div_gen(uint32_t d, uint32_t *magic, uint32_t *shift) {
    uint32_t L = 0;
    while ((2u << L) <= d) ++L;                       // floor(log2 d), d >= 2
    if ((d & (d - 1u)) == 0u) { *magic = 0u; *shift = L - 1u; return; }
    const uint64_t two = (uint64_t)1 << (32 + L);
    uint64_t m = two / d;
    const uint64_t rem = two - m * d;
    m += m;
    if (rem + rem >= d) m += 1;
    *magic = (uint32_t)(m + 1);
    *shift = L;
}

// floor(lane / U) for lane in [0, 63] and U in [1, 64] is (lane * M) >> 16 with M = floor(65535 / U) + 1: M = 65536/U + delta,
// 0 < delta <= 1, so the product overshoots lane/U by less than 64/65536 < 1/64 <= 1/U, the smallest gap to the next
// integer.  lane * M < 2^24.  All 64 x 64 cases are checked in tests/native/lean_math_check.cpp.
UAVENV_HD uint32_t lane_div_magic(uint32_t U) { return 65535u / U + 1u; }
UAVENV_HD uint32_t lane_div(uint32_t lane, uint32_t magic) { return (lane * magic) >> 16; }

UAVENV_HD uint32_t u32div(uint32_t a, uint32_t magic, uint32_t shift) {
    const uint32_t q = (uint32_t)(((uint64_t)a * magic) >> 32);
    const uint32_t t = ((a - q) >> 1) + q;
    return t >> shift;
}

}  // namespace uavk
