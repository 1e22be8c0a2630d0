// Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw, SC'11), host + gfx950 device.
// Replaces the reference's process-global MT19937 draws (ue_mobility.py:6,408,508; channel.py:240):
// one independent stream per (env, tick, index, draw site), so results do not depend on scheduling.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define UAVENV_HD __host__ __device__ __forceinline__
#else
#define UAVENV_HD inline
#endif

namespace uavk {

enum DrawSite : uint32_t {  // Philox counter word 3
    DOM_FADING = 1, DOM_HEADING = 2 /* quad mode (B > 8) only; otherwise headings come from the spare words of DOM_FADING calls */, DOM_GROUP_A = 3, DOM_GROUP_B = 4,
    DOM_INIT_UE_A = 5, DOM_INIT_UE_B = 6, DOM_INIT_G_A = 7, DOM_INIT_G_B = 8, DOM_INIT_G_C = 9, DOM_AREA = 10
};

struct U4 { uint32_t x, y, z, w; };

UAVENV_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// 53-bit uniform in [0,1) from two words: ((hi >> 5) * 2^26 + (lo >> 6)) * 2^-53, the construction NumPy's random_sample uses.
// Evaluated as a * 2^-27 + b * 2^-53 with a = hi >> 5 < 2^27 and b = lo >> 6 < 2^26: both u32 -> f64 converts are exact, both
// scalings are by powers of two, and the sum is a 53-bit integer times 2^-53, hence representable -- the fma returns exactly the
// value of the integer formulation (u53_int below; gfx950 has no u64 -> f64 convert, that form costs ~10 instructions, this one 6).
// Equality is checked for edge and random words in tests/native/lean_math_check.cpp; the oracle keeps the integer form.
UAVENV_HD double u53_int(uint32_t hi, uint32_t lo) {
    return (double)(((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6)) * (1.0 / 9007199254740992.0);
}
UAVENV_HD double u53(uint32_t hi, uint32_t lo) {
    return fma((double)(hi >> 5), 1.0 / 134217728.0, (double)(lo >> 6) * (1.0 / 9007199254740992.0));
}

}  // namespace uavk
