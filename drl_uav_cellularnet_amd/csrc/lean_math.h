// Lean float64 primitives for the env kernels (host + gfx950 device).
//
// Why: profiles/r01_v5 shows the step kernel VALU-issue bound; per UE lane it spent 370 of ~2300 VALU
// instructions in four ocml logarithms (97 each: double-double evaluation) and ~220 in sqrt+divide pairs.
// These replacements keep float64 accuracy (measured on the host against long double in
// tests/test_lean_math.py) but drop the special-case handling the env never needs:
// arguments are finite, positive and normal by construction (see each function).
#pragma once
#include <math.h>

#include "philox.h"  // UAVENV_HD

namespace uavk {

// Natural logarithm, x finite, positive, normal.  Algorithm of fdlibm e_log.c (Sun Microsystems, 1993):
// x = 2^k * (1+f), sqrt(1/2) <= 1+f < sqrt(2);  s = f/(2+f);  log(1+f) = f - hfsq + s*(hfsq + R(s^2)),
// R a degree-14 minimax polynomial in s; error < 1 ulp.  Users: Box-Muller radius (argument 1-u in [2^-53, 1])
// and 10*log10(S/(N+I)) (argument in ~[1e-20, 1e13]).
UAVENV_HD double lm_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int k;
    double m = frexp(x, &k);                         // m in [0.5, 1)
    const bool low = m < 0.70710678118654752440;     // bring m into [sqrt(1/2), sqrt(2))
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// 1/sqrt(x), x finite, positive, normal.  Device: ocml rsqrt (v_rsq_f64 + one refinement, 10 VALU against 21+11 for
// sqrt followed by a divide).  Host: the plain expression (used only by the accuracy test).
UAVENV_HD double lm_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrt(x);
#else
    return 1.0 / sqrt(x);
#endif
}

}  // namespace uavk
