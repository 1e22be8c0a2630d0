// Host accuracy check of csrc/lean_math.h against a long double reference (TEST INFRASTRUCTURE).
// Prints: name samples max_ulp mean_ulp   -- consumed by tests/test_lean_math.py
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../drl_uav_cellularnet_amd/csrc/lean_math.h"

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
    run("rsqrt_dist2", [](double x) { return uavk::lm_rsqrt(x); }, [](long double x) { return 1.0L / std::sqrt(x); },
        [&](std::mt19937_64 &g) { return 25.0 * (double)(1 + g() % 2000000); }, n);
    return 0;
}
