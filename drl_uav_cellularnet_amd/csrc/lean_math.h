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

// ---- coefficient block ------------------------------------------------------------------------------------
// Non-trivial float64 constants cannot be encoded in a VALU instruction, and hipcc re-materialises each one with
// two v_mov_b32 at every inlined use (profiles: 271 v_mov_b32 in a 1977-instruction kernel).  The kernels build
// this block ONCE, pin it in VGPRs (lm_pin: an empty asm the compiler cannot see through, hence cannot
// rematerialise) and pass it to lm_exp2 / lm_sincospi / lm_logc.  Values: tools/gen_lean_coeffs.py (60-digit
// decimal arithmetic, correctly rounded, printed as exact hex floats).
struct LeanCoef {
    double e2[13];   // 2^r      = 1 + sum_{k=1..13} e2[k-1] r^k,           |r| <= 1/2     e2[k-1] = ln2^k / k!
    double sp[8];    // sin(pi r) = r * sum_{k=0..7} sp[k] r^(2k),           |r| <= 1/4     (-1)^k pi^(2k+1)/(2k+1)!
    double cp[8];    // cos(pi r) = 1 + sum_{k=1..8} cp[k-1] r^(2k),         |r| <= 1/4     (-1)^k pi^(2k)/(2k)!
    double lg[7], ln2_hi, ln2_lo;   // fdlibm e_log.c
};

// PIN is a launch-time policy, not a constant of the build: pinning costs ~100 VGPRs (occupancy 2 instead of 4).  It wins
// when a launch needs at most two wavefronts per SIMD (4096 envs x 20 UEs: 9.35 vs 9.81 us) and loses when more must be
// resident (65536 envs: 72.1 vs 65.5 us; 4 x 40: 16.0 vs 15.1 us) -- gpurun_out/ab_pin.log, DESIGN.md section 4.
template <bool PIN>
UAVENV_HD void lm_pin(double &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (PIN) asm volatile("" : "+v"(v));
#else
    (void)v;
#endif
}

// The scalar-register counterpart: the value stays in an SGPR pair (a VOP3 float64 FMA may read ONE scalar operand, which is all a
// Horner step fma(p, r, c_k) needs), so a coefficient costs neither a VGPR pair nor the two v_mov per use that an unpinned literal does.
UAVENV_HD void lm_pin_sgpr(double &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(v));
#else
    (void)v;
#endif
}

template <bool PIN>
UAVENV_HD LeanCoef lm_make_coef() {
    LeanCoef c = {
        {0x1.62e42fefa39efp-1, 0x1.ebfbdff82c58fp-3, 0x1.c6b08d704a0c0p-5, 0x1.3b2ab6fba4e77p-7, 0x1.5d87fe78a6731p-10,
         0x1.430912f86c787p-13, 0x1.ffcbfc588b0c7p-17, 0x1.62c0223a5c824p-20, 0x1.b5253d395e7c4p-24,
         0x1.e4cf5158b8ecap-28, 0x1.e8cac7351bb25p-32, 0x1.c3bd650fc2986p-36, 0x1.816193166d0f9p-40},
        {0x1.921fb54442d18p+1, -0x1.4abbce625be53p+2, 0x1.466bc6775aae2p+1, -0x1.32d2cce62bd86p-1,
         0x1.50783487ee782p-4, -0x1.e3074fde8871fp-8, 0x1.e8f434d018d63p-12, -0x1.6fadb9f155744p-16},
        {-0x1.3bd3cc9be45dep+2, 0x1.03c1f081b5ac4p+2, -0x1.55d3c7e3cbffap+0, 0x1.e1f506891babbp-3,
         -0x1.a6d1f2a204a8cp-6, 0x1.f9d38a3763cc3p-10, -0x1.b6e24f44b128fp-14, 0x1.20c62c2f2d7f5p-18},
        {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
         1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01},
        6.93147180369123816490e-01, 1.90821492927058770002e-10};
    for (int k = 0; k < 13; ++k) lm_pin<PIN>(c.e2[k]);
    for (int k = 0; k < 8; ++k) { lm_pin<PIN>(c.sp[k]); lm_pin<PIN>(c.cp[k]); }
    for (int k = 0; k < 7; ++k) lm_pin<PIN>(c.lg[k]);
    lm_pin<PIN>(c.ln2_hi); lm_pin<PIN>(c.ln2_lo);
    return c;
}

// a / b for finite a (either sign, or 0) and positive NORMAL b, quotient normal or 0 (no scaling, no fix-up: the IEEE sequence
// hipcc emits for `/` spends 11 VALU instructions on ranges the env never produces).  r = rcp(b) refined by one Newton
// step, q = a*r corrected by one residual step: 6 instructions, <= 1 ulp (measured in tests/test_lean_math.py).
UAVENV_HD double lm_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(b);
#else
    double r = (double)(float)(1.0 / b);          // host stand-in with a deliberately poor (24-bit) seed
#endif
    r = fma(fma(-b, r, 1.0), r, r);               // Newton: error squared
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);              // residual correction
}

UAVENV_HD double lm_xor_sign(double v, unsigned long long sign_bit) {
    unsigned long long b;
    __builtin_memcpy(&b, &v, 8);
    b ^= sign_bit;
    __builtin_memcpy(&v, &b, 8);
    return v;
}

// 2^x, |x| < 1000.  n = rint(x), r = x - n exact, Taylor of 2^r to degree 13 (truncation 4e-18), scale by 2^n.
UAVENV_HD double lm_exp2(double x, const LeanCoef &c) {
    const double n = rint(x);
    const double r = x - n;
    double p = c.e2[12];
#pragma unroll
    for (int k = 11; k >= 0; --k) p = fma(p, r, c.e2[k]);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// sin(pi x), cos(pi x), |x| < 2^30.  q = rint(2x), r = x - q/2 exact in [-1/4, 1/4]; Taylor in r; quadrant fix-up.
UAVENV_HD void lm_sincospi(double x, const LeanCoef &c, double *s_out, double *c_out) {
    const double q = rint(x + x);
    const double r = fma(q, -0.5, x);
    const double r2 = r * r;
    double ps = c.sp[7], pc = c.cp[7];
#pragma unroll
    for (int k = 6; k >= 0; --k) { ps = fma(ps, r2, c.sp[k]); pc = fma(pc, r2, c.cp[k]); }
    const double sr = ps * r;                 // sin(pi r)
    const double cr = fma(pc, r2, 1.0);       // cos(pi r)
    const int iq = (int)q;
    const bool odd = (iq & 1) != 0;
    const double so = odd ? cr : sr, co = odd ? sr : cr;   // q mod 4: 0 (s,c)  1 (c,-s)  2 (-s,-c)  3 (-c,s)
    // sign flips as integer XORs on the sign bit: per-lane booleans live in SGPR pairs on gfx9 and the kernels are
    // SGPR-bound (spills show up as v_readlane/v_writelane), so only `odd` is a boolean here.
    const unsigned long long ss = (unsigned long long)(unsigned)(iq & 2) << 62;         // bit 1 of q     -> bit 63
    const unsigned long long cs = (unsigned long long)(unsigned)((iq + 1) & 2) << 62;   // bit 1 of (q+1) -> bit 63
    *s_out = lm_xor_sign(so, ss);
    *c_out = lm_xor_sign(co, cs);
}

// lm_log with the fdlibm coefficients taken from the pinned block.
UAVENV_HD double lm_logc(double x, const LeanCoef &c) {
    int k;
    double m = frexp(x, &k);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = lm_div(f, 2.0 + f);        // 2+f in [1.70, 2.42)
    const double z = s * s, w = z * z;
    const double t1 = w * (c.lg[1] + w * (c.lg[3] + w * c.lg[5]));
    const double t2 = z * (c.lg[0] + w * (c.lg[2] + w * (c.lg[4] + w * c.lg[6])));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * c.ln2_hi - ((hfsq - (s * (hfsq + R) + dk * c.ln2_lo)) - f);
}

// 1/sqrt(x), x finite, positive, normal.  Device: v_rsq_f64 + the refinement ocml's rsqrt applies (same operations in the
// same order, so the same bits), without ocml's v_cmp_class guard that keeps a non-finite seed: for x = 0 (walker in the
// UAV's own cell) this returns NaN where ocml returns inf, and rx_power() discards either through its d^2 > pl_dis^2 select.
// Host: the plain expression (used only by the accuracy test).
UAVENV_HD double lm_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(y0 * -x, y0, 1.0);            // 1 - x*y0^2
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);        // y0 * (1 + e/2 + 3e^2/8)
#else
    return 1.0 / sqrt(x);
#endif
}

}  // namespace uavk
