// Internal to libuavenv (not installed): the handle and the helpers shared by its translation units (uavenv_capi.hip: everything but
// the gated rollout; uavenv_gated.hip: uavenv_rollout_gated and its kernel instantiations, a file of its own so that neither rebuilds the other).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/uavenv.h"
#include "uavenv_kernels.h"

struct uavenv {
    UavEnvConfig cfg;
    long long N;
    int device;
    uint64_t seed;
    uint32_t env_id_base;
    int bt;  // template bound on B
    bool plc;  // pl_b == 30: cube path-loss kernel variant
    bool packed;  // U <= 64 and U >= max(B, Gr): env_kernel_packed with kp.epw envs per wavefront
    long long n_simd;  // 4 x compute units of the device: wavefront demand per SIMD decides the PIN variant
    char *blob;
    int32_t *bs_init_dev;
    long long *act_pow_dev;
    uint4 *act_dec_dev;  // [B] split decode of the joint action (KParams::act_dec)
    int8_t *gid_dev;  // [max(U,64)] RPGM group of walker u
    int32_t *obs_prev_dev;  // [N, U+B] cells written by the last obs_dense(_update) call; allocated on first use
    const float *obs_last_dev;  // the buffer that call wrote: obs_dense_update refuses any other
    struct RotPlan { int n_steps; int n_launches; long long slots; int4 *dev; };   // rotation schedules built so far (one per n_steps;
                                                                                  // n_launches 0 = none applies: plain launch)
    std::vector<RotPlan> *rot_plans;
    int rotate;         // UAVENV_ROTATE read once at create: -1 unset (automatic), 0 never, 1 whenever a schedule exists (tests)
    long long rot_slots;  // UAVENV_ROTATE_SLOTS (tests: pretend the device has this many SIMDs), else n_simd
    uint32_t *sched_flag_dev;   // [env-wavefronts] hand-off words of the one-launch schedule (zero between calls)
    uint32_t *err_host, *err_dev;   // sticky device-side error word: host-mapped memory, so that every entry point can test it without a HIP call
    uint32_t spin_us;   // hand-off spin budget (UAVENV_HANDOFF_SPIN_US, default 2 s)
    int drop_publish;   // UAVENV_DEBUG_DROP_PUBLISH=1 (test hook): schedules are built WITHOUT their publish bits, so every hand-off times out
    // uavenv_launch_timing: start / stop events attached to the multi-step dispatches themselves (hipExtLaunchKernelGGL: the timestamps of
    // the dispatch packet, no marker packets around it), a ring of kTimedLaunches pairs
    std::vector<hipEvent_t> *tev;
    int timing, n_timed;
    int force_pin;  // UAVENV_FORCE_PIN read ONCE at create (experiments: tools/pin_sweep.sh): -1 unset, 0 / 1 forced
    UavEnvStateLayout lay;
    uavk::KParams kp;  // constants + state pointers, per-call fields patched at launch
};

constexpr int kTimedLaunches = 256;
namespace uavenv_internal {
int fail(int code, const std::string &msg);            // sets uavenv_last_error()'s thread-local message, returns code
int poisoned(const uavenv *h, const char *what);
void fill_call(uavk::KParams &p, const UavEnvInject *inj, const UavEnvOut *out);
bool call_is_fast(const uavk::KParams &p);
}  // namespace uavenv_internal

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(UAVENV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

// Launches go to the handle's device whatever the caller's current device is; restored on return.
struct DeviceGuard {
    int prev = -1, want;
    explicit DeviceGuard(int dev) : want(dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
};

