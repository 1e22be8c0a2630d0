// Host accuracy check of csrc/lean_math.h against a long double reference (TEST INFRASTRUCTURE).
// Prints: name samples max_ulp mean_ulp   -- consumed by tests/test_lean_math.py
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../drl_uav_cellularnet_amd/csrc/lean_math.h"
#include "../../drl_uav_cellularnet_amd/csrc/intdiv.h"

static double ulp_of(double ref) {
    int e;
    std::frexp(ref, &e);
    return std::ldexp(1.0, e - 53);
}

template <class F, class R, class G>
static void run(const char *name, F f, R ref, G gen, int n) {
    std::mt19937_64 rng(12345);
    double worst = 0, sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = gen(rng);
        const long double r = ref((long double)x);
        const double got = f(x);
        const double err = std::fabs((double)((long double)got - r)) / ulp_of((double)r);
        if (err > worst) worst = err;
        sum += err;
    }
    std::printf("%s %d %.4f %.4f\n", name, n, worst, sum / n);
}

int main() {
    const int n = 2000000;
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    // Box-Muller radius: log(1-u), u = 53-bit uniform in [0,1)
    run("log_one_minus_u", [](double x) { return uavk::lm_log(x); }, [](long double x) { return std::log(x); },
        [&](std::mt19937_64 &g) { double u = (double)(g() >> 11) * (1.0 / 9007199254740992.0); return 1.0 - u; }, n);
    // SINR ratio: 10^[-20, 13]
    run("log_sinr_ratio", [](double x) { return uavk::lm_log(x); }, [](long double x) { return std::log(x); },
        [&](std::mt19937_64 &g) { return std::pow(10.0, -20.0 + 33.0 * u01(g)); }, n);
    // near 1 (SINR near 0 dB: the region the 1e-5 relative bound of the float32 output is sensitive to)
    run("log_near_one", [](double x) { return uavk::lm_log(x); }, [](long double x) { return std::log(x); },
        [&](std::mt19937_64 &g) { return 1.0 + (u01(g) - 0.5) * 1e-3; }, n);
    const uavk::LeanCoef C = uavk::lm_make_coef<false>();
    run("logc_sinr_ratio", [&](double x) { return uavk::lm_logc(x, C); }, [](long double x) { return std::log(x); },
        [&](std::mt19937_64 &g) { return std::pow(10.0, -20.0 + 33.0 * u01(g)); }, n);
    // 10^(-f/10) = 2^(c*f), f ~ N(0,2) -> |x| < ~4; generic path-loss exponent adds down to ~ -40
    run("exp2_fading", [&](double x) { return uavk::lm_exp2(x, C); }, [](long double x) { return std::exp2(x); },
        [&](std::mt19937_64 &g) { return -4.0 + 8.0 * u01(g); }, n);
    run("exp2_wide", [&](double x) { return uavk::lm_exp2(x, C); }, [](long double x) { return std::exp2(x); },
        [&](std::mt19937_64 &g) { return -60.0 + 70.0 * u01(g); }, n);
    // sincospi on [0,2): reference with the SAME exact reduction (pi*x in long double would lose the zeros of sin)
    auto red = [](long double x, int &iq) { long double q = std::rint(2.0L * x); iq = (int)q; return x - 0.5L * q; };
    run("sinpi_0_2", [&](double x) { double s, c; uavk::lm_sincospi(x, C, &s, &c); return s; },
        [&](long double x) { int iq; long double r = red(x, iq); long double sr = std::sin(3.14159265358979323846264338327950288L * r),
                             cr = std::cos(3.14159265358979323846264338327950288L * r);
                             switch (iq & 3) { case 0: return sr; case 1: return cr; case 2: return -sr; default: return -cr; } },
        [&](std::mt19937_64 &g) { return 2.0 * u01(g); }, n);
    run("cospi_0_2", [&](double x) { double s, c; uavk::lm_sincospi(x, C, &s, &c); return c; },
        [&](long double x) { int iq; long double r = red(x, iq); long double sr = std::sin(3.14159265358979323846264338327950288L * r),
                             cr = std::cos(3.14159265358979323846264338327950288L * r);
                             switch (iq & 3) { case 0: return cr; case 1: return -sr; case 2: return -cr; default: return sr; } },
        [&](std::mt19937_64 &g) { return 2.0 * u01(g); }, n);
    // lm_div on the operand ranges the kernels feed it: f/(2+f) inside the log, S/(N+I) for the SINR
    {
        std::mt19937_64 rng(99);
        double worst = 0, sum = 0;
        const int m = 2000000;
        for (int i = 0; i < m; ++i) {
            const bool logcase = (i & 1) != 0;
            const double f = logcase ? (-0.2928932188134524 + 0.7071067811865476 * u01(rng)) : 0.0;   // m-1, m in [sqrt(1/2), sqrt 2)
            const double a = logcase ? std::fabs(f) + 1e-300 : std::pow(10.0, -18.0 + 18.0 * u01(rng));
            const double b = logcase ? 2.0 + f : std::pow(10.0, -16.0 + 14.0 * u01(rng));
            const long double ref = (long double)a / (long double)b;
            const double err = std::fabs((double)((long double)uavk::lm_div(a, b) - ref)) / ulp_of((double)ref);
            if (err > worst) worst = err;
            sum += err;
        }
        std::printf("lm_div %d %.4f %.4f\n", m, worst, sum / m);
    }
    {   // csrc/intdiv.h: lane / U for every lane of a wavefront and every slot width
        int bad = 0;
        for (uint32_t U = 1; U <= 64; ++U)
            for (uint32_t lane = 0; lane < 64; ++lane)
                if (uavk::lane_div(lane, uavk::lane_div_magic(U)) != lane / U) ++bad;
        std::printf("lane_div_mismatches 4096 %d 0\n", bad);
    }
    {   // csrc/philox.h: the fma form of u53 against the integer form, bit for bit
        std::mt19937_64 rng(2026);
        const uint32_t edge[] = {0u, 1u, 31u, 32u, 63u, 64u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFDFu, 0xFFFFFFE0u, 0xFFFFFFFFu};
        long long bad = 0, total = 0;
        for (uint32_t h : edge) for (uint32_t l : edge) { ++total; if (uavk::u53(h, l) != uavk::u53_int(h, l)) ++bad; }
        for (int i = 0; i < 8000000; ++i) {
            const uint64_t w = rng();
            ++total;
            if (uavk::u53((uint32_t)(w >> 32), (uint32_t)w) != uavk::u53_int((uint32_t)(w >> 32), (uint32_t)w)) ++bad;
        }
        std::printf("u53_mismatches %lld %lld 0\n", total, bad);
    }
    run("rsqrt_dist2", [](double x) { return uavk::lm_rsqrt(x); }, [](long double x) { return 1.0L / std::sqrt(x); },
        [&](std::mt19937_64 &g) { return 25.0 * (double)(1 + g() % 2000000); }, n);
    // exact multiply-shift division used for the action digits: every divisor 2..9, dense + random + edge dividends
    {
        std::mt19937_64 rng(7);
        long long bad = 0, total = 0;
        for (uint32_t d = 2; d <= 9; ++d) {
            uint32_t magic, shift;
            uavk::u32div_gen(d, &magic, &shift);
            auto chk = [&](uint32_t a) { ++total; if (uavk::u32div(a, magic, shift) != a / d) ++bad; };
            for (uint32_t a = 0; a < 3000000u; ++a) chk(a);
            for (uint32_t a = 0xFFFFFFFFu; a > 0xFFFFFFFFu - 3000000u; --a) chk(a);
            for (int i = 0; i < 3000000; ++i) chk((uint32_t)rng());
        }
        std::printf("u32div_mismatches %lld %lld 0\n", total, bad);
    }
    return 0;
}
