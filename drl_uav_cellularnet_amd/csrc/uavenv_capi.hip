// C ABI of libuavenv (include/uavenv.h): handle management, state blob, kernel dispatch.
// gfx950 only; built by drl_uav_cellularnet_amd/build.py with hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "uavenv_handle.h"

using namespace uavk;
using uavenv_internal::fail;
using uavenv_internal::poisoned;
using uavenv_internal::fill_call;
using uavenv_internal::call_is_fast;

static_assert(UAVENV_MAX_GROUPS == kMaxGroups && UAVENV_MAX_BS == kMaxBs, "header / kernel bounds differ");

static thread_local std::string g_err;
int uavenv_internal::fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

// A kernel that gave up on a hand-off (uavenv_kernels.h: sched_hand_off_wait) leaves a word in host-mapped memory.  The handle's state
// is then incomplete: every later call on it fails until uavenv_set_state() installs a whole state again.
int uavenv_internal::poisoned(const uavenv *h, const char *what) {
    if (h->err_host && *(volatile uint32_t *)h->err_host != 0u) {
        char buf[200];
        std::snprintf(buf, sizeof buf, "%s: an earlier multi-step launch on this handle failed on the device (code 0x%08x: a hand-off between two "
                      "wavefronts timed out); its state is incomplete -- uavenv_set_state() or a new handle", what, *(volatile uint32_t *)h->err_host);
        return fail(UAVENV_E_DEVICE, buf);
    }
    return UAVENV_OK;
}

extern "C" int uavenv_launch_timing(uavenv_t *h, int enable) {
    if (!h) return fail(UAVENV_E_INVALID, "launch_timing: null handle");
    DeviceGuard guard(h->device);
    if (enable && !h->tev) {
        h->tev = new (std::nothrow) std::vector<hipEvent_t>();
        if (!h->tev) return fail(UAVENV_E_NOMEM, "launch_timing: host allocation failed");
        for (int i = 0; i < 2 * kTimedLaunches; ++i) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return fail(UAVENV_E_HIP, "launch_timing: hipEventCreate failed");
            h->tev->push_back(e);
        }
    }
    h->timing = enable ? 1 : 0;
    h->n_timed = 0;
    return UAVENV_OK;
}

extern "C" int uavenv_launch_times_us(uavenv_t *h, double *us_out, int max_out, int *n_out) {
    if (!h || !n_out || (max_out > 0 && !us_out)) return fail(UAVENV_E_INVALID, "launch_times_us: null argument");
    DeviceGuard guard(h->device);
    const int n = h->n_timed < max_out ? h->n_timed : max_out;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize((*h->tev)[(size_t)i * 2 + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, (*h->tev)[(size_t)i * 2], (*h->tev)[(size_t)i * 2 + 1]));
        us_out[i] = (double)ms * 1e3;
    }
    *n_out = h->n_timed;
    return UAVENV_OK;
}

extern "C" int uavenv_device_error(uavenv_t *h, uint32_t *code) {
    if (!h || !code) return fail(UAVENV_E_INVALID, "device_error: null argument");
    *code = h->err_host ? *(volatile uint32_t *)h->err_host : 0u;
    return UAVENV_OK;
}

extern "C" int uavenv_abi_version(void) { return UAVENV_ABI_VERSION; }
extern "C" const char *uavenv_last_error(void) { return g_err.c_str(); }

extern "C" void uavenv_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const U4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

extern "C" int uavenv_default_config(UavEnvConfig *c, int n_bs, int n_ue, int grid) {
    if (!c || n_bs < 1 || n_bs > UAVENV_MAX_BS || n_ue < 1 || grid < 8)
        return fail(UAVENV_E_INVALID, "default_config: bad n_bs/n_ue/grid");
    std::memset(c, 0, sizeof(*c));
    c->n_bs = n_bs; c->n_ue = n_ue; c->grid = grid;
    c->n_groups = 4;  // mobile_env.py:76: four groups
    int left = n_ue;
    for (int g = 0; g < 4; ++g) { c->group_size[g] = (g < 3) ? n_ue / 4 : left; left -= c->group_size[g]; }
    if (n_bs == 4) {  // mobile_env.py:49-50
        const int q = grid / 4, t = grid * 3 / 4;
        const int xs[4] = {q, q, t, t}, ys[4] = {q, t, q, t};
        for (int b = 0; b < 4; ++b) { c->bs_init_xy[b][0] = xs[b]; c->bs_init_xy[b][1] = ys[b]; }
    } else {  // the reference ctor cannot build n_bs != 4: square lattice, caller may override
        int side = 1;
        while (side * side < n_bs) ++side;
        for (int b = 0; b < n_bs; ++b) {
            c->bs_init_xy[b][0] = grid / (2 * side) + (b / side) * (grid / side);
            c->bs_init_xy[b][1] = grid / (2 * side) + (b % side) * (grid / side);
        }
    }
    c->max_step = 2000; c->bs_step = 2; c->min_bs_dist = 4; c->n_act = 5;
    c->agg_init = 200; c->deagg_len = 100; c->agg_len = 10;
    c->grid_width = 5.0; c->p_bs_dbm = 20.0; c->noise_dbm = -121.0;
    c->pl_a = 38.0; c->pl_b = 30.0; c->pl_dis = 0.0; c->antenna_gain = 2.0; c->eq_loss = 0.0;
    c->shadow_mean = 0.0; c->shadow_sd = 2.0; c->ho_thresh_db = 1.0; c->out_thresh = 0.0;
    c->ue_velocity = 1.0; c->grp_v_min = 0.0; c->grp_v_max = 1.0; c->aggregation = 0.8;
    return UAVENV_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int check_config(const UavEnvConfig &c) {
    if (c.n_bs < 1 || c.n_bs > UAVENV_MAX_BS) return fail(UAVENV_E_INVALID, "config: n_bs out of range [1,32]");
    if (c.n_ue < 1 || c.n_ue > 4096) return fail(UAVENV_E_INVALID, "config: n_ue out of range [1,4096]");
    if (c.n_groups < 1 || c.n_groups > UAVENV_MAX_GROUPS) return fail(UAVENV_E_INVALID, "config: n_groups out of range");
    if (c.grid < 8 || c.grid > 32767) return fail(UAVENV_E_INVALID, "config: grid out of range [8,32767]");
    int s = 0;
    for (int g = 0; g < c.n_groups; ++g) {
        if (c.group_size[g] < 0) return fail(UAVENV_E_INVALID, "config: negative group size");
        s += c.group_size[g];
    }
    if (s != c.n_ue) return fail(UAVENV_E_INVALID, "config: group sizes do not sum to n_ue");
    if (c.n_act < 2 || c.n_act > 9) return fail(UAVENV_E_INVALID, "config: n_act out of range [2,9]");
    {   // the joint action is one int64 (the reference uses unbounded Python ints): n_act^n_bs must fit
        long long pw = 1;
        for (int b = 0; b < c.n_bs; ++b) {
            if (pw > 0x7FFFFFFFFFFFFFFFll / c.n_act)
                return fail(UAVENV_E_INVALID, "config: n_act^n_bs does not fit in int64 (joint action encoding)");
            pw *= c.n_act;
        }
    }
    if (c.max_step < 1 || c.bs_step < 0 || c.min_bs_dist < 0) return fail(UAVENV_E_INVALID, "config: bad step constants");
    // Start cells must lie in [1, G-1]: the reference's boundaries are [1, G] (mobile_env.py:44-45) and cell G has no row in the
    // G x G observation.  bs_move_serial() relies on this (its two-sided range test equals the reference's one-sided ones).
    for (int b = 0; b < c.n_bs; ++b)
        if (c.bs_init_xy[b][0] < 1 || c.bs_init_xy[b][0] >= c.grid || c.bs_init_xy[b][1] < 1 || c.bs_init_xy[b][1] >= c.grid)
            return fail(UAVENV_E_INVALID, "config: UAV start cell outside [1, grid-1]");
    return UAVENV_OK;
}

extern "C" int uavenv_create(const UavEnvConfig *cfg, int64_t n_envs, int device, uint64_t seed, uint32_t env_id_base,
                             uavenv_t **out) {
    if (!cfg || !out || n_envs < 1) return fail(UAVENV_E_INVALID, "create: null argument or n_envs < 1");
    if (int rc = check_config(*cfg)) return rc;
    {   // The kernels address state, outputs and actions as base + 32-bit byte offset (ldx/stx, uavenv_kernels.h): every array
        // indexed that way must stay below 4 GiB.  Packed path (U <= 64, U >= B, U >= Gr): walker records (16 B), group records
        // (48 B), UAV cells (8 B); multi-pass path: only the per-env record (32 B) and per-env outputs go through ldx/stx.  288 GB of HBM hold far larger batches: shard them
        // over several handles (env_id_base keeps the Philox streams those of one big batch).
        const bool packed = (cfg->n_ue <= 64) && (cfg->n_ue >= cfg->n_bs) && (cfg->n_ue >= cfg->n_groups);
        unsigned long long row = sizeof(EnvRec);                                    // bytes of the widest indexed array per env
        if (packed) {
            const unsigned long long cand[] = {(unsigned long long)cfg->n_ue * sizeof(UePos), (unsigned long long)cfg->n_groups * sizeof(GrpRec),
                                               (unsigned long long)cfg->n_bs * 8ull};
            for (unsigned long long c : cand) if (c > row) row = c;
        }
        if ((unsigned long long)n_envs * row > 0xFFFFFFFFull)
            return fail(UAVENV_E_INVALID, "create: n_envs too large for one handle (a state array would exceed 4 GiB); "
                                          "shard the batch over several handles with env_id_base");
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1 || device < 0 || device >= n_dev)
        return fail(UAVENV_E_NODEVICE, "create: no HIP device " + std::to_string(device));
    DeviceGuard guard(device);   // the caller's current device is restored on every return path
    uavenv *h = new (std::nothrow) uavenv();
    if (!h) return fail(UAVENV_E_NOMEM, "create: host allocation failed");
    h->cfg = *cfg; h->N = n_envs; h->device = device; h->seed = seed; h->env_id_base = env_id_base;
    h->force_pin = -1;
    if (const char *f = std::getenv("UAVENV_FORCE_PIN")) h->force_pin = (f[0] == '1') ? 1 : 0;
    h->rotate = -1;
    if (const char *f = std::getenv("UAVENV_ROTATE")) h->rotate = (f[0] == '1') ? 1 : 0;
    h->rot_plans = new (std::nothrow) std::vector<uavenv::RotPlan>();
    h->spin_us = 2000000u;
    if (const char *f = std::getenv("UAVENV_HANDOFF_SPIN_US")) { const long long v = std::atoll(f); if (v > 0 && v < 60000000ll) h->spin_us = (uint32_t)v; }
    if (const char *f = std::getenv("UAVENV_DEBUG_DROP_PUBLISH")) h->drop_publish = (f[0] == '1') ? 1 : 0;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
        h->n_simd = 4ll * cus;
        h->rot_slots = h->n_simd;
        if (const char *f = std::getenv("UAVENV_ROTATE_SLOTS")) { const long long v = std::atoll(f); if (v > 0) h->rot_slots = v; }
    }
    const size_t U = (size_t)cfg->n_ue, B = (size_t)cfg->n_bs;
    const size_t W64 = (U + 63) / 64;
    h->bt = B <= 4 ? 4 : B <= 8 ? 8 : B <= 16 ? 16 : 32;

    // one layout definition shared with the device code (state_layout.h)
    const StateOffsets SO = compute_layout(n_envs, cfg->n_ue, cfg->n_bs, cfg->n_groups);
    UavEnvStateLayout &L = h->lay;
    L.total_bytes = SO.total;
    L.ue_pos = SO.ue_pos; L.ue_aux = SO.ue_aux; L.grp = SO.grp; L.env = SO.env; L.bs_xy = SO.bs_xy; L.out_bits = SO.out_bits;

    // act_pow[b] = n_act^(B-1-b): digit of UAV b in the joint action, most significant first
    // (Decimal_to_Base_N, ue_mobility.py:310-336).  check_config() has verified n_act^B fits in int64.
    long long act_pow[UAVENV_MAX_BS];
    std::memset(act_pow, 0, sizeof(act_pow));
    long long pw = 1;
    for (int b = cfg->n_bs - 1; b >= 0; --b) { act_pow[b] = pw; pw *= cfg->n_act; }
    const long long n_joint = pw;  // n_act^B = action_space_dim (mobile_env.py:104)
    // Split decode (KParams::act_dec): the k_lo least significant digits form `lo`, the others `hi`; usable when the joint action is
    // exact in a double (n_act^B < 2^52) and both halves fit 32 bits.
    uint4 act_dec[UAVENV_MAX_BS];
    std::memset(act_dec, 0, sizeof(act_dec));
    int act_split = 0;
    unsigned long long act_P = 1;
    {
        const int k_lo = cfg->n_bs / 2, k_hi = cfg->n_bs - k_lo;
        unsigned long long p_lo = 1, p_hi = 1;
        for (int i = 0; i < k_lo; ++i) p_lo *= (unsigned long long)cfg->n_act;
        for (int i = 0; i < k_hi; ++i) p_hi *= (unsigned long long)cfg->n_act;
        if (n_joint > 0xFFFFFFFFll && n_joint < (1ll << 52) && p_lo <= 0xFFFFFFFFull && p_hi <= 0xFFFFFFFFull && k_lo >= 1) {
            act_split = 1; act_P = p_lo;
            for (int b = 0; b < cfg->n_bs; ++b) {
                const int h = cfg->n_bs - 1 - b;                       // power index of UAV b's digit (most significant first)
                const int in_hi = h >= k_lo ? 1 : 0;
                unsigned long long pw32 = 1;
                for (int i = 0; i < (in_hi ? h - k_lo : h); ++i) pw32 *= (unsigned long long)cfg->n_act;
                uint32_t magic = 0, shift = 0;
                if (pw32 >= 2) u32div_gen((uint32_t)pw32, &magic, &shift);
                act_dec[b] = uint4{(uint32_t)in_hi, magic, shift, (uint32_t)pw32};
            }
        }
    }
    // group of every walker (ue_mobility.py:417-426 g_ref), padded so that idle lanes read in range
    const size_t n_gid = U < 64 ? 64 : U;
    std::string gid(n_gid, '\0');
    {
        size_t w = 0;
        for (int g = 0; g < cfg->n_groups; ++g)
            for (int i = 0; i < cfg->group_size[g]; ++i) gid[w++] = (char)g;
    }
    // Allocations and uploads: any failure frees what exists and reports; a handle is never returned half-initialised.
    auto bail = [&](int code, const char *what, hipError_t err) {
        const std::string msg = std::string("create: ") + what + ": " + hipGetErrorString(err);
        uavenv_destroy(h);
        return fail(code, msg);
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&h->blob, L.total_bytes)) != hipSuccess) return bail(UAVENV_E_NOMEM, "hipMalloc state", e);
    if ((e = hipMalloc((void **)&h->bs_init_dev, sizeof(int32_t) * 2 * UAVENV_MAX_BS)) != hipSuccess) return bail(UAVENV_E_NOMEM, "hipMalloc bs_init", e);
    if ((e = hipMalloc((void **)&h->act_pow_dev, sizeof(long long) * UAVENV_MAX_BS)) != hipSuccess) return bail(UAVENV_E_NOMEM, "hipMalloc act_pow", e);
    if ((e = hipMalloc((void **)&h->gid_dev, n_gid)) != hipSuccess) return bail(UAVENV_E_NOMEM, "hipMalloc gid table", e);
    if ((e = hipMalloc((void **)&h->act_dec_dev, sizeof(act_dec))) != hipSuccess) return bail(UAVENV_E_NOMEM, "hipMalloc act_dec", e);
    if ((e = hipMemcpy(h->act_dec_dev, act_dec, sizeof(act_dec), hipMemcpyHostToDevice)) != hipSuccess) return bail(UAVENV_E_HIP, "hipMemcpy act_dec", e);
    if ((e = hipMemset(h->blob, 0, L.total_bytes)) != hipSuccess) return bail(UAVENV_E_HIP, "hipMemset state", e);
    if ((e = hipMemcpy(h->bs_init_dev, cfg->bs_init_xy, sizeof(int32_t) * 2 * UAVENV_MAX_BS, hipMemcpyHostToDevice)) != hipSuccess)
        return bail(UAVENV_E_HIP, "hipMemcpy bs_init", e);
    if ((e = hipMemcpy(h->act_pow_dev, act_pow, sizeof(act_pow), hipMemcpyHostToDevice)) != hipSuccess) return bail(UAVENV_E_HIP, "hipMemcpy act_pow", e);
    if ((e = hipMemcpy(h->gid_dev, gid.data(), n_gid, hipMemcpyHostToDevice)) != hipSuccess) return bail(UAVENV_E_HIP, "hipMemcpy gid table", e);
    {   // one-launch schedules (rotation_plan): one hand-off word per env-wavefront, and the sticky error word in host-mapped memory
        const size_t n_flag = (size_t)n_envs + 64;           // (>= env-wavefronts for any envs-per-wavefront)
        if ((e = hipMalloc((void **)&h->sched_flag_dev, n_flag * sizeof(uint32_t))) != hipSuccess) return bail(UAVENV_E_NOMEM, "hipMalloc hand-off words", e);
        if ((e = hipMemset(h->sched_flag_dev, 0, n_flag * sizeof(uint32_t))) != hipSuccess) return bail(UAVENV_E_HIP, "hipMemset hand-off words", e);
        if ((e = hipHostMalloc((void **)&h->err_host, 64, hipHostMallocMapped)) != hipSuccess) return bail(UAVENV_E_NOMEM, "hipHostMalloc error word", e);
        std::memset(h->err_host, 0, 64);
        if ((e = hipHostGetDevicePointer((void **)&h->err_dev, h->err_host, 0)) != hipSuccess) return bail(UAVENV_E_HIP, "hipHostGetDevicePointer error word", e);
    }

    KParams &k = h->kp;
    std::memset(&k, 0, sizeof(k));
    k.U = cfg->n_ue; k.B = cfg->n_bs; k.Gr = cfg->n_groups; k.G = cfg->grid; k.W64 = (int)W64;
    int acc = 0;
    for (int g = 0; g <= kMaxGroups; ++g) {
        k.group_start[g] = acc;
        if (g < cfg->n_groups) acc += cfg->group_size[g];
    }
    k.max_step = cfg->max_step; k.bs_step = cfg->bs_step; k.min_bs_dist2 = cfg->min_bs_dist * cfg->min_bs_dist;
    k.n_act = cfg->n_act; k.agg_init = cfg->agg_init; k.deagg_len = cfg->deagg_len; k.agg_len = cfg->agg_len;
    k.grid_width = cfg->grid_width;
    k.p_bs_watt = std::pow(10.0, cfg->p_bs_dbm / 10.0) * 1e-3;    // channel.py:58
    k.noise_watt = std::pow(10.0, cfg->noise_dbm / 10.0) * 1e-3;  // channel.py:59
    // folded constants of the linear-domain gain (see env_kernel): float64 pow on the host, once
    k.k_pl = k.p_bs_watt * std::pow(10.0, (cfg->antenna_gain - cfg->pl_a - cfg->eq_loss) / 10.0);
    k.k_0 = k.p_bs_watt * std::pow(10.0, (cfg->antenna_gain - cfg->eq_loss) / 10.0);
    k.c_exp = -std::log2(10.0) / 10.0;      // 10^(-f/10) = 2^(c_exp*f)
    k.pl_exp_ln = (cfg->pl_b / 10.0) * 0.5 / std::log(2.0);  // d^(-pl_b/10) = 2^(-pl_exp_ln * ln(d^2))
    k.pl_dis2 = cfg->pl_dis < 0.0 ? -1.0 : cfg->pl_dis * cfg->pl_dis;  // d > pl_dis  <=>  d^2 > pl_dis2 (d >= 0)
    k.db_per_ln = 10.0 / std::log(10.0);     // 10*log10(x) = db_per_ln * ln(x)
    k.inv_U = 1.0 / (double)cfg->n_ue;
    k.inv_U20 = 1.0 / (20.0 * (double)cfg->n_ue);
    h->plc = (cfg->pl_b == 30.0);
    k.pl_a = cfg->pl_a; k.pl_b = cfg->pl_b; k.pl_dis = cfg->pl_dis; k.antenna_gain = cfg->antenna_gain;
    k.eq_loss = cfg->eq_loss; k.shadow_mean = cfg->shadow_mean; k.shadow_sd = cfg->shadow_sd;
    k.ho_thresh_db = cfg->ho_thresh_db; k.out_thresh = cfg->out_thresh; k.ue_velocity = cfg->ue_velocity;
    k.grp_v_min = cfg->grp_v_min; k.grp_v_max = cfg->grp_v_max; k.aggregation = cfg->aggregation;
    k.N = n_envs; k.key0 = (uint32_t)seed; k.key1 = (uint32_t)(seed >> 32); k.env_id_base = env_id_base;
    char *b = h->blob;
    k.ue_pos = (UePos *)(b + L.ue_pos); k.ue_aux = (UeAux *)(b + L.ue_aux); k.grp = (GrpRec *)(b + L.grp);
    k.env = (EnvRec *)(b + L.env); k.bs_xy = (int32_t *)(b + L.bs_xy); k.out_bits = (unsigned long long *)(b + L.out_bits);
    k.bs_init = h->bs_init_dev;
    k.act_pow = h->act_pow_dev;
    k.act_dec = h->act_dec_dev; k.act_split = act_split; k.act_P = (uint32_t)act_P; k.act_inv_P = 1.0 / (double)act_P;
    k.gid_of_u = h->gid_dev;
    k.sched_flag = h->sched_flag_dev; k.sched_err = h->err_dev; k.sched_spin_us = h->spin_us;
    u32div_gen((uint32_t)cfg->n_act, &k.div_magic, &k.div_shift);   // exact digit extraction (intdiv.h)
    k.act32 = (n_joint <= 0xFFFFFFFFll) ? 1 : 0;  // 32-bit digit extraction when every joint action fits
    // Packed kernel: floor(64/U) env instances per wavefront; needs the group / UAV owner lanes inside a slot.
    h->packed = (cfg->n_ue <= 64) && (cfg->n_ue >= cfg->n_bs) && (cfg->n_ue >= cfg->n_groups);
    k.epw = 1;
    if (h->packed) {
        k.epw = 64 / cfg->n_ue;
        if (k.epw > kMaxEpw) k.epw = kMaxEpw;
    }
    *out = h;
    return UAVENV_OK;
}

extern "C" void uavenv_destroy(uavenv_t *h) {
    if (!h) return;
    DeviceGuard guard(h->device);   // runs from __del__ at GC time: must not move the caller's current device
    (void)hipFree(h->blob);
    (void)hipFree(h->bs_init_dev);
    (void)hipFree(h->act_pow_dev);
    if (h->act_dec_dev) (void)hipFree(h->act_dec_dev);
    (void)hipFree(h->gid_dev);
    if (h->obs_prev_dev) (void)hipFree(h->obs_prev_dev);
    if (h->tev) { for (hipEvent_t e : *h->tev) (void)hipEventDestroy(e); delete h->tev; }
    if (h->sched_flag_dev) (void)hipFree(h->sched_flag_dev);
    if (h->err_host) (void)hipHostFree(h->err_host);
    if (h->rot_plans) {
        for (auto &pl : *h->rot_plans) (void)hipFree(pl.dev);
        delete h->rot_plans;
    }
    delete h;
}

extern "C" int uavenv_init(uavenv_t *h, const UavEnvInitInject *inj, void *stream) {
    if (!h) return fail(UAVENV_E_INVALID, "init: null handle");
    if (inj && (!inj->u_x_dev || !inj->u_y_dev || !inj->u_th_dev || !inj->u_g_dev))
        return fail(UAVENV_E_INVALID, "init: injection needs all four arrays");
    DeviceGuard guard(h->device);
    const KParams &k = h->kp;
    InitParams p;
    std::memset(&p, 0, sizeof(p));
    p.U = k.U; p.Gr = k.Gr; p.B = k.B; p.W64 = k.W64; p.G = k.G; p.agg_init = k.agg_init; p.deagg_len = k.deagg_len;
    p.grp_v_min = k.grp_v_min; p.grp_v_max = k.grp_v_max; p.N = k.N; p.key0 = k.key0; p.key1 = k.key1;
    p.env_id_base = k.env_id_base;
    p.ue_pos = k.ue_pos; p.ue_aux = k.ue_aux; p.grp = k.grp; p.env = k.env; p.bs_xy = k.bs_xy; p.out_bits = k.out_bits;
    p.bs_init = k.bs_init;
    if (inj) { p.u_x = inj->u_x_dev; p.u_y = inj->u_y_dev; p.u_th = inj->u_th_dev; p.u_g = inj->u_g_dev; }
    p.per = k.U;
    if (k.Gr > p.per) p.per = k.Gr;
    if (k.B > p.per) p.per = k.B;
    if (k.W64 > p.per) p.per = k.W64;
    const long long total = k.N * p.per;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(init_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return UAVENV_OK;
}

// FAST kernels: no injected draws, all nine standard outputs present, no float64 copies (see UAV_OUT in
// uavenv_kernels.h).  Anything else runs the checked variant of the same kernel.
bool uavenv_internal::call_is_fast(const KParams &p) {
    const OutPtrs &o = p.out;
    return !p.inj_theta && !p.inj_group && !p.inj_fading && o.reward && o.done && o.mean_sinr && o.n_out && o.ue_xy &&
           o.bs_xy && o.serving && o.cur_sinr && o.step_n && !o.cur_sinr_f64 && !o.mean_sinr_f64 && !o.reward_f64;
}

// ---- launch census (test hook, uavenv_debug_variant_*) --------------------------------------------------------------------------
// Every launch of an env kernel is counted under the template instantiation that ran, and variant_selectable() states which
// instantiations launch_env can choose at all.  tests/test_launch_variants_gpu.py drives every one of them against the oracle
// and then asserts that none was left out: a kernel instantiation that ships has a parity test (the unpinned
// multi-step kernel had quoted numbers and no test).  Key = (family, BT, MODE, PLC, variant, MANY).
enum { FAM_PACKED = 0, FAM_MULTIPASS = 1 };
enum { VAR_CHECKED = 0, VAR_FAST = 1, VAR_PIN = 2 };
constexpr int kCensusSlots = 2 * 4 * 5 * 2 * 3 * 3;
static std::atomic<long long> g_census[kCensusSlots];
static constexpr int bt_index(int bt) { return bt == 4 ? 0 : bt == 8 ? 1 : bt == 16 ? 2 : 3; }
static constexpr int census_index(int fam, int bt, int mode, bool plc, int var, int many) {   // many: 0 single step, 1 multi-step, 2 multi-step under a rotation
    return ((((fam * 4 + bt_index(bt)) * 5 + mode) * 2 + (plc ? 1 : 0)) * 3 + var) * 3 + many;  // schedule (SCHED kernels)
}
// The image of launch_env's selection logic (keep the two in step: census_count() refuses a key this predicate rejects).
static bool variant_selectable(int fam, int bt, int mode, bool plc, int var, int many) {
    if (mode == MODE_WARMUP)                               // mobility only: one BT = 4, PLC instantiation per family, FAST or checked
        return bt == 4 && plc && many == 0 && (var == VAR_CHECKED || var == VAR_FAST);
    if (fam == FAM_PACKED) return many == 0 || mode == MODE_STEP;      // every (BT, PLC, variant); MANY / SCHED exist for MODE_STEP only
    if (many != 0 || var == VAR_PIN) return false;              // multi-pass: no PIN variant, uavenv_step_many loops over single steps
    return var == VAR_FAST || bt == 4;                     // the checked multi-pass kernel reads B at run time: BT = 4 serves all
}
static bool census_count(int fam, int bt, int mode, bool plc, int var, int many) {
    if (!variant_selectable(fam, bt, mode, plc, var, many)) return false;
    g_census[census_index(fam, bt, mode, plc, var, many)].fetch_add(1, std::memory_order_relaxed);
    return true;
}
static void census_decode(int i, int &fam, int &bt, int &mode, bool &plc, int &var, int &many) {
    many = i % 3; i /= 3;
    var = i % 3; i /= 3;
    plc = i & 1; i >>= 1;
    mode = i % 5; i /= 5;
    static const int bts[4] = {4, 8, 16, 32};
    bt = bts[i & 3]; fam = i >> 2;
}
extern "C" int uavenv_debug_variant_count(void) { return kCensusSlots; }
extern "C" int uavenv_debug_variant_info(int i, char *name, size_t name_len, int *selectable, long long *launches) {
    if (i < 0 || i >= kCensusSlots) return fail(UAVENV_E_INVALID, "debug_variant_info: index out of range");
    int fam, bt, mode, var, many; bool plc;
    census_decode(i, fam, bt, mode, plc, var, many);
    static const char *modes[5] = {"WARMUP", "RESET", "STEP", "TRACE", "RESET_TRACE"};
    if (name && name_len) {
        if (fam == FAM_PACKED)
            std::snprintf(name, name_len, "env_kernel_packed<BT=%d, %s, PLC=%d, FAST=%d, PIN=%d, MANY=%d%s>", bt, modes[mode], (int)plc,
                          (int)(var != VAR_CHECKED), (int)(var == VAR_PIN), (int)(many != 0), many == 2 ? ", SCHED=1" : "");
        else
            std::snprintf(name, name_len, "env_kernel_multipass<BT=%d, %s, PLC=%d, FAST=%d>%s", bt, modes[mode], (int)plc,
                          (int)(var != VAR_CHECKED), (var == VAR_PIN || many != 0) ? " (no such kernel)" : "");
    }
    if (selectable) *selectable = variant_selectable(fam, bt, mode, plc, var, many) ? 1 : 0;
    if (launches) *launches = g_census[i].load(std::memory_order_relaxed);
    return UAVENV_OK;
}
extern "C" void uavenv_debug_variant_reset(void) {
    for (auto &c : g_census) c.store(0, std::memory_order_relaxed);
}


// MANY_: 0 = one step / reset / tick batch per launch, 1 = uavenv_step_many
template <int MODE, int MANY_ = 0>
static int launch_env(uavenv_t *h, const KParams &p_in, hipStream_t s, long long launch_waves = 0, long long first_env = 0, long long n_range = 0) {
    constexpr bool MANY = MANY_ != 0;
    // one wavefront hosts p.epw env instances (packed) or exactly one (multi-pass); 4 wavefronts per workgroup
    // (n_range > 0: envs [first_env, first_env + n_range) only -- uavenv_step_range has checked that the range starts on a wavefront
    //  boundary and ends on one or at N)
    KParams p = p_in;
    const long long e_lo = n_range > 0 ? first_env : 0, e_hi = n_range > 0 ? first_env + n_range : p.N;
    const int wave0 = (int)(e_lo / p.epw);                                   // first env-wavefront with an env of the range
    const long long waves = (e_hi + p.epw - 1) / p.epw - wave0;              // ... through the last one (either may straddle the range's border)
    p.wave0 = wave0; p.e_end = e_hi;
    // (a launch of a rotation schedule has `launch_waves` slots instead of one wavefront per env-wavefront; p.sched says who does what)
    const unsigned grid = (unsigned)(((launch_waves > 0 ? launch_waves : waves) + kWavesPerBlock - 1) / kWavesPerBlock);
    const dim3 blk(64 * kWavesPerBlock);
    // leading scalar arguments of the packed kernels: delivered in SGPRs at wave launch (kernarg preload), see
    // env_kernel_packed.  The slab base replaces the 19 per-field pointers (state_layout.h).
#define PK_ARGS h->blob, p.actions, p.gid_of_u, p.N, p.U, p.epw, p.Gr, p.B, (int)uavk::lane_div_magic((uint32_t)p.U), wave0, (int)e_lo, (int)e_hi, p
    if (MODE == MODE_WARMUP) {
        // mobility only: independent of B / path loss, so one instantiation per kernel family
        const bool fast = !p.inj_theta && !p.inj_group && (p.B == 4);   // the warm-up instantiation has BT = 4
        if (h->packed) {
            if (fast) hipLaunchKernelGGL((env_kernel_packed<4, MODE_WARMUP, true, true, false>), dim3(grid), blk, 0, s, PK_ARGS);
            else hipLaunchKernelGGL((env_kernel_packed<4, MODE_WARMUP, true, false, false>), dim3(grid), blk, 0, s, PK_ARGS);
        } else if (fast) {
            hipLaunchKernelGGL((env_kernel_multipass<4, MODE_WARMUP, true, true>), dim3(grid), blk, 0, s, p);
        } else {
            hipLaunchKernelGGL((env_kernel_multipass<4, MODE_WARMUP, true, false>), dim3(grid), blk, 0, s, p);
        }
        HIP_TRY(hipGetLastError());
        if (!census_count(h->packed ? FAM_PACKED : FAM_MULTIPASS, 4, MODE_WARMUP, true, fast ? VAR_FAST : VAR_CHECKED, 0))
            return fail(UAVENV_E_INVALID, "launch census: warm-up instantiation outside variant_selectable()");
        return UAVENV_OK;
    }
    constexpr int M = (MODE == MODE_WARMUP) ? MODE_STEP : MODE;  // (never instantiates the channel modes for WARMUP)
    // FAST kernels are compiled for B == BT exactly
    const bool fast = call_is_fast(p) && (p.B == h->bt);
    // PIN variant (constants pinned in VGPRs, occupancy 2) only when the launch puts between one and two wavefronts on a SIMD.
    // Sweep on one box, pinned vs unpinned (profiles/r01_v19_pin_sweep.txt): 0.67 waves/SIMD 7.50 vs 7.23 us, 1.0 tie, 1.33
    // 8.67 vs 9.20, 2.0 9.29 vs 9.77, 2.67 12.87 vs 12.08, 4.0 15.87 vs 14.98: two co-resident waves profit from constants that
    // are not re-read through the scalar path; a lone wave only pays for materialising them; beyond two, occupancy wins.
    bool pin = fast && (waves >= h->n_simd) && (waves <= 2 * h->n_simd);
    // Multi-step launches: the pinned loop wins below one wavefront per SIMD too (1536 envs: 3.47 us per step unpinned; the lone-wave
    // argument above is about materialising constants once per LAUNCH, which a 100-step launch amortises).
    // (a rotation schedule launches S = k x SIMDs slots for its W > S env-wavefronts: the wavefronts that are resident count)
    if (MANY) pin = fast && ((launch_waves > 0 ? launch_waves : waves) <= 2 * h->n_simd);
    // A RANGE launch exists to run beside other kernels of the caller (the A2C rollout's other half: an MFMA workgroup of 8 wavefronts x 128
    // VGPRs per CU).  The pinned variant's 251 VGPRs per wavefront leave no SIMD with two of them room for that workgroup, which then starts
    // only when the env launch drains (rocprofv3 timeline, profiles/r04g_*): ranges run unpinned (90 VGPRs).
    if (n_range > 0) pin = false;
    if (h->force_pin >= 0) pin = fast && (h->force_pin == 1);   // experiments only (read once in uavenv_create)
#define UAVENV_LAUNCH_PKS(BT_, PLC_, SCH_)                                                                       \
    do {                                                                                                         \
        if (MANY && tev0 != nullptr) {   /* (multi-step launches with uavenv_launch_timing on: events on the dispatch itself) */ \
            if (pin) hipExtLaunchKernelGGL((env_kernel_packed<BT_, M, PLC_, true, true, MANY, SCH_>), dim3(grid), blk, 0, s, tev0, tev1, 0, PK_ARGS);     \
            else if (fast) hipExtLaunchKernelGGL((env_kernel_packed<BT_, M, PLC_, true, false, MANY, SCH_>), dim3(grid), blk, 0, s, tev0, tev1, 0, PK_ARGS); \
            else hipExtLaunchKernelGGL((env_kernel_packed<BT_, M, PLC_, false, false, MANY, SCH_>), dim3(grid), blk, 0, s, tev0, tev1, 0, PK_ARGS);       \
        } else                                                                                                   \
        if (pin) hipLaunchKernelGGL((env_kernel_packed<BT_, M, PLC_, true, true, MANY, SCH_>), dim3(grid), blk, 0, s, PK_ARGS);     \
        else if (fast) hipLaunchKernelGGL((env_kernel_packed<BT_, M, PLC_, true, false, MANY, SCH_>), dim3(grid), blk, 0, s, PK_ARGS); \
        else hipLaunchKernelGGL((env_kernel_packed<BT_, M, PLC_, false, false, MANY, SCH_>), dim3(grid), blk, 0, s, PK_ARGS);       \
        counted = census_count(FAM_PACKED, BT_, M, PLC_, pin ? VAR_PIN : (fast ? VAR_FAST : VAR_CHECKED), MANY_ + ((SCH_) ? 1 : 0)); \
    } while (0)
#define UAVENV_LAUNCH_PK(BT_, PLC_)                                                                              \
    do {                                                                                                         \
        if (MANY && p.sched != nullptr) UAVENV_LAUNCH_PKS(BT_, PLC_, MANY); else UAVENV_LAUNCH_PKS(BT_, PLC_, false); \
    } while (0)
#define UAVENV_LAUNCH(BT_)                                                                                       \
    do {                                                                                                         \
        if (h->packed) {                                                                                         \
            if (h->plc) UAVENV_LAUNCH_PK(BT_, true); else UAVENV_LAUNCH_PK(BT_, false);                          \
        } else if (!MANY) {   /* the checked multi-pass variant reads B at run time: one instantiation serves every BT */ \
            if (h->plc) {                                                                                        \
                if (fast) hipLaunchKernelGGL((env_kernel_multipass<BT_, M, true, true>), dim3(grid), blk, 0, s, p);   \
                else hipLaunchKernelGGL((env_kernel_multipass<4, M, true, false>), dim3(grid), blk, 0, s, p);         \
            } else {                                                                                             \
                if (fast) hipLaunchKernelGGL((env_kernel_multipass<BT_, M, false, true>), dim3(grid), blk, 0, s, p);  \
                else hipLaunchKernelGGL((env_kernel_multipass<4, M, false, false>), dim3(grid), blk, 0, s, p);        \
            }                                                                                                    \
            counted = census_count(FAM_MULTIPASS, fast ? BT_ : 4, M, h->plc, fast ? VAR_FAST : VAR_CHECKED, 0); \
        }                                                                                                        \
    } while (0)
    bool counted = false;
    hipEvent_t tev0 = nullptr, tev1 = nullptr;
    if (MANY && h->timing && h->tev && h->n_timed < kTimedLaunches) {
        tev0 = (*h->tev)[(size_t)h->n_timed * 2]; tev1 = (*h->tev)[(size_t)h->n_timed * 2 + 1];
        h->n_timed += 1;
    }
    switch (h->bt) {
        case 4: UAVENV_LAUNCH(4); break;
        case 8: UAVENV_LAUNCH(8); break;
        case 16: UAVENV_LAUNCH(16); break;
        default: UAVENV_LAUNCH(32); break;
    }
#undef UAVENV_LAUNCH
#undef UAVENV_LAUNCH_PK
#undef UAVENV_LAUNCH_PKS
#undef PK_ARGS
    HIP_TRY(hipGetLastError());
    if (!counted) return fail(UAVENV_E_INVALID, "launch census: no kernel launched, or an instantiation outside variant_selectable()");
    return UAVENV_OK;
}

void uavenv_internal::fill_call(KParams &p, const UavEnvInject *inj, const UavEnvOut *out) {
    p.inj_theta = inj ? inj->theta_u_dev : nullptr;
    p.inj_group = inj ? inj->group_u_dev : nullptr;
    p.inj_fading = inj ? inj->fading_dev : nullptr;
    std::memset(&p.out, 0, sizeof(p.out));
    if (out) {
        p.out.reward = out->reward_dev; p.out.done = out->done_dev; p.out.mean_sinr = out->mean_sinr_dev;
        p.out.n_out = out->n_out_dev; p.out.ue_xy = out->ue_xy_dev; p.out.bs_xy = out->bs_xy_dev;
        p.out.serving = out->serving_dev; p.out.cur_sinr = out->cur_sinr_dev; p.out.step_n = out->step_n_dev;
        p.out.cur_sinr_f64 = out->cur_sinr_f64_dev; p.out.mean_sinr_f64 = out->mean_sinr_f64_dev;
        p.out.reward_f64 = out->reward_f64_dev;
    }
}

extern "C" int uavenv_warmup(uavenv_t *h, int n_ticks, const UavEnvInject *inj, void *stream) {
    if (!h || n_ticks < 0) return fail(UAVENV_E_INVALID, "warmup: null handle or negative n_ticks");
    if (n_ticks == 0) return UAVENV_OK;
    if (inj && (inj->theta_u_dev || inj->group_u_dev) && n_ticks != 1)
        return fail(UAVENV_E_INVALID, "warmup: injected draws cover exactly one tick");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "warmup")) return rc_dev;
    KParams p = h->kp;
    fill_call(p, inj, nullptr);
    p.n_ticks = n_ticks;
    return launch_env<MODE_WARMUP>(h, p, (hipStream_t)stream);
}

extern "C" int uavenv_reset(uavenv_t *h, const uint8_t *mask_dev, const UavEnvInject *inj, const UavEnvOut *out,
                            void *stream) {
    if (!h) return fail(UAVENV_E_INVALID, "reset: null handle");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "reset")) return rc_dev;
    KParams p = h->kp;
    fill_call(p, inj, out);
    p.mask = mask_dev; p.n_ticks = 1;
    return launch_env<MODE_RESET>(h, p, (hipStream_t)stream);
}

extern "C" int uavenv_step(uavenv_t *h, const int64_t *actions_dev, const UavEnvInject *inj, const UavEnvOut *out,
                           void *stream) {
    if (!h || !actions_dev) return fail(UAVENV_E_INVALID, "step: null handle or actions");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "step")) return rc_dev;
    KParams p = h->kp;
    fill_call(p, inj, out);
    p.actions = (const long long *)actions_dev; p.n_ticks = 1;
    return launch_env<MODE_STEP>(h, p, (hipStream_t)stream);
}

extern "C" int uavenv_step_range(uavenv_t *h, const int64_t *actions_dev, int64_t first_env, int64_t n_envs, const UavEnvInject *inj,
                                 const UavEnvOut *out, void *stream) {
    if (!h || !actions_dev) return fail(UAVENV_E_INVALID, "step_range: null handle or actions");
    if (first_env < 0 || n_envs < 1 || first_env + n_envs > h->N || h->N > 0x7FFFFFFFll)
        return fail(UAVENV_E_INVALID, "step_range: the range must be non-empty and lie inside the batch");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "step_range")) return rc_dev;
    KParams p = h->kp;
    fill_call(p, inj, out);
    p.actions = (const long long *)actions_dev; p.n_ticks = 1;
    return launch_env<MODE_STEP>(h, p, (hipStream_t)stream, 0, first_env, n_envs);
}

// Output block of step t in the [T][...] arrays of a multi-step call (uavenv_step_many); null members stay null.
static UavEnvOut out_block(const UavEnvOut &o, long long t, long long N, long long U, long long B) {
    UavEnvOut r = o;
    if (r.reward_dev) r.reward_dev += t * N;
    if (r.done_dev) r.done_dev += t * N;
    if (r.mean_sinr_dev) r.mean_sinr_dev += t * N;
    if (r.n_out_dev) r.n_out_dev += t * N;
    if (r.ue_xy_dev) r.ue_xy_dev += t * N * U * 2;
    if (r.bs_xy_dev) r.bs_xy_dev += t * N * B * 2;
    if (r.serving_dev) r.serving_dev += t * N * U;
    if (r.cur_sinr_dev) r.cur_sinr_dev += t * N * U;
    if (r.step_n_dev) r.step_n_dev += t * N;
    if (r.cur_sinr_f64_dev) r.cur_sinr_f64_dev += t * N * U;
    if (r.mean_sinr_f64_dev) r.mean_sinr_f64_dev += t * N;
    if (r.reward_f64_dev) r.reward_f64_dev += t * N;
    return r;
}

// ---- rotation schedule for multi-step launches ----------------------------------------------------------------------------------
// A batch of W env-wavefronts on S slots (S = k x SIMDs, k = floor(W / SIMDs) wavefronts resident per SIMD) with S < W < 2 S leaves
// W - S SIMDs with one wavefront more than the others for the whole launch; they set its time and the others idle part of it
// (BASELINE's 4096 envs x 20 UEs: 1366 wavefronts on 1024 SIMDs).  The W x T wavefront-steps fit S slots in M = ceil(W T / S)
// step-times (McNaughton's wrap-around rule: fill slot after slot; a job that does not fit is split, its LAST steps at the end of this
// slot, its FIRST steps at the start of the next).  M < 2 T, so a slot holds at most three pieces: [first steps of a split job]
// [one whole job] [last steps of another split job].  ONE launch of S persistent wavefronts runs it: the wavefront that ran a job's
// first steps stores the state, releases and sets the job's flag; the wavefront that runs its last steps polls the flag (bounded),
// acquires and loads (uavenv_kernels.h: sched_hand_off_*).  The publishing piece is the FIRST piece of its slot and waits for nothing,
// and the waiting piece starts M - T step-times after the publishing one ended, so in practice nobody waits.
// (Round 3 cut the slots' timelines at D = ceil(M / (M - T)) common boundaries into D launches ordered by the stream; each launch cost
// the ~8 us of load / store / launch phases a launch has, it paid from 48 steps on only and lost to this form at every size measured:
// profiles/r04a_many_ab_three_launch_forms.json.  Removed.)
// Returns the index of the cached / newly built plan in h->rot_plans, or -1 when no schedule applies (then the plain launch runs).
static long long rot_padded_slots(long long S) { return (S + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock; }
// The schedule itself: pure host arithmetic (no device, no handle), so that the CPU test suite can check it for any (W, S, T)
// (uavenv_debug_schedule).  -> table rows [rot_padded_slots(S)][kSchedPieces] of {env-wavefront, first step, steps, SCHED_* bits}; false
// when no valid schedule exists for these numbers.
static bool build_schedule(long long W, long long S, int T, bool drop_publish, std::vector<int4> &table) {
    struct Piece { int ew, t0, nt; long long time; };
    std::vector<std::vector<Piece>> cell((size_t)S);              // [slot] -> pieces
    bool ok = true;
    const long long M = (W * T + S - 1) / S;                       // makespan in step-times
    if (W <= S || T < 2 || M - T < 1 || M >= 2 * (long long)T) return false;
    {   // McNaughton fill.  The FIRST steps of a split job go to the next slot's start.
        long long slot = 0, t = 0;
        for (long long j = 0; j < W && ok; ++j) {
            if (slot >= S) { ok = false; break; }
            if (t + T <= M) {
                cell[(size_t)slot].push_back(Piece{(int)j, 0, T, t});
                t += T;
                if (t == M) { ++slot; t = 0; }
            } else {
                const int a = (int)(M - t);                                         // steps that still fit here: the job's LAST a steps
                if (slot + 1 >= S) { ok = false; break; }
                cell[(size_t)slot + 1].push_back(Piece{(int)j, 0, T - a, 0});
                cell[(size_t)slot].push_back(Piece{(int)j, T - a, a, t});
                ++slot; t = T - a;
            }
        }
    }
    for (auto &c : cell) std::sort(c.begin(), c.end(), [](const Piece &x, const Piece &y) { return x.time < y.time; });
    // Verify what the argument above promises: every job's steps 0..T-1 exactly once, at most kSchedPieces pieces per slot; a job is one
    // whole piece, or two pieces on different slots of which the first one (the one that publishes) LEADS its slot: it can never wait,
    // so every wait ends -- no deadlock whatever the order in which the hardware starts the wavefronts.
    struct Seen { int n, slot0, q0, len0, slot1, t1, len1; };
    std::vector<Seen> seen((size_t)W, Seen{0, -1, -1, 0, -1, 0, 0});
    for (long long sl = 0; sl < S && ok; ++sl) {
        const auto &c = cell[(size_t)sl];
        if (c.size() > (size_t)kSchedPieces) ok = false;
        for (size_t q = 0; q < c.size() && ok; ++q) {
            const Piece &pc = c[q];
            Seen &z = seen[(size_t)pc.ew];
            if (pc.t0 == 0) { z.slot0 = (int)sl; z.q0 = (int)q; z.len0 = pc.nt; }
            else { z.slot1 = (int)sl; z.t1 = pc.t0; z.len1 = pc.nt; }
            z.n += 1;
        }
    }
    for (long long j = 0; j < W && ok; ++j) {
        const Seen &z = seen[(size_t)j];
        if (z.n == 1) ok = z.slot0 >= 0 && z.len0 == T;
        else if (z.n == 2) ok = z.slot0 >= 0 && z.slot1 >= 0 && z.slot0 != z.slot1 && z.q0 == 0 && z.len0 >= 1 && z.t1 == z.len0 && z.len0 + z.len1 == T;
        else ok = false;
    }
    if (!ok) return false;
    // One table row per WAVEFRONT of the launch, not per slot: a launch of S slots has ceil(S / kWavesPerBlock) whole workgroups, and the
    // wavefronts past slot S - 1 of the last one read rows too -- theirs are all-zero (no steps).  (Round 3, first GPU run: with rows
    // per slot those wavefronts read past the allocation: a memory fault at S = 26.)
    const long long Sp = rot_padded_slots(S);
    table.assign((size_t)(Sp * kSchedPieces), int4{0, 0, 0, 0});
    for (long long sl = 0; sl < S; ++sl) {
        const auto &c = cell[(size_t)sl];
        for (size_t q = 0; q < c.size(); ++q) {         // column by kind: 0 publishes (it is the slot's first piece, verified above), 1 whole, 2 waits
            int bits = 0, col = 1;
            if (c[q].t0 > 0) { bits |= SCHED_WAIT; col = 2; }
            if (c[q].t0 + c[q].nt < T) { col = 0; if (!drop_publish) bits |= SCHED_PUBLISH; }
            int4 &cellq = table[(size_t)(sl * kSchedPieces) + (size_t)col];
            if (cellq.z != 0) ok = false;                // (two pieces of one kind in a slot: cannot happen for M < 2 T)
            cellq = int4{c[q].ew, c[q].t0, c[q].nt, bits};
        }
    }
    if (!ok) return false;
    return ok;
}

static int rotation_plan(uavenv_t *h, int T, hipStream_t stream) {
    if (!h->packed || !h->rot_plans || h->rotate == 0 || T < 2) return -1;
    const long long W = (h->N + h->kp.epw - 1) / h->kp.epw;
    const long long k_res = W / h->rot_slots;                      // wavefronts every SIMD hosts for the whole launch
    const long long S = k_res * h->rot_slots;
    if (k_res < 1 || W <= S) return -1;                            // fewer wavefronts than SIMDs, or a balanced batch
    for (size_t i = 0; i < h->rot_plans->size(); ++i)
        if ((*h->rot_plans)[i].n_steps == T) return (*h->rot_plans)[i].dev ? (int)i : -1;
    // A new schedule needs a hipMalloc and a synchronous upload: not inside a stream capture (the plain launch runs instead, and the
    // call is not remembered, so that a later call outside the capture builds it), cf. uavenv_step_many_prepare.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (cap != hipStreamCaptureStatusNone) return -1;
    auto remember = [&](int D, int4 *dev) -> int {
        h->rot_plans->push_back(uavenv::RotPlan{T, D, S, dev});
        return dev ? (int)h->rot_plans->size() - 1 : -1;
    };
    const long long M = (W * T + S - 1) / S;                       // makespan in step-times
    if (M - T < 1 || M >= 2 * (long long)T) return remember(0, nullptr);
    // Automatic use only where it pays (same-box sweeps, us per step plain -> scheduled: profiles/r04a_*, r04k_many_ab_sweep.json).
    // k = 1: one wavefront alone on a SIMD runs a step in ~0.64 of the time two co-resident ones take (2.77 vs 4.31 us), so the schedule
    // wins while W / S <= 1.45 (4096 envs, 1.33: 4.60 -> 3.81; 5400 envs, 1.76: 4.65 -> 4.90).  k = 2: the plain launch of more than two
    // wavefronts per SIMD is the unpinned kernel with a third wavefront queued behind two resident ones; 2 k-resident pinned wavefronts win
    // over the whole range (8192 envs: 7.90 -> 5.92; 9000 envs, 1.46: 7.90 -> 6.47).  k > 2 would need more than two resident wavefronts of
    // the pinned kernel.  Every piece border costs its slot a state store (+ release) and a (poll + acquire +) state load, ~5-8 us, against
    // ~0.7 us gained per step: 16-step calls lose (4.98 -> 5.23), 20-step calls win (4.87 -> 4.52): from 20 steps per call on.
    if (h->rotate == -1 && (k_res > 2 || (k_res == 1 && 100 * W > 145 * S) || T < 20)) return remember(0, nullptr);
    std::vector<int4> table;
    if (!build_schedule(W, S, T, h->drop_publish != 0, table)) return remember(0, nullptr);
    int4 *dev = nullptr;
    if (hipMalloc((void **)&dev, table.size() * sizeof(int4)) != hipSuccess) return remember(0, nullptr);
    if (hipMemcpy(dev, table.data(), table.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dev); return remember(0, nullptr); }
    return remember(1, dev);
}

// Test hook without a device: the schedule of W env-wavefronts on S slots for T steps.  table_out: int32 [rows][3 pieces][4] (may be NULL to
// ask for the size only); *rows_out = rows (slots padded to whole workgroups), *makespan_out = ceil(W T / S).  UAVENV_E_INVALID when no
// schedule exists for these numbers (W <= S, W >= 2 S, T < 2, ...).
extern "C" int uavenv_debug_schedule(int64_t n_wavefronts, int64_t n_slots, int n_steps, int32_t *table_out, int64_t table_capacity_rows,
                                     int64_t *rows_out, int64_t *makespan_out) {
    if (n_wavefronts < 1 || n_slots < 1 || n_steps < 1) return fail(UAVENV_E_INVALID, "debug_schedule: bad argument");
    std::vector<int4> table;
    if (!build_schedule(n_wavefronts, n_slots, n_steps, false, table)) return fail(UAVENV_E_INVALID, "debug_schedule: no schedule for these numbers");
    const long long rows = (long long)table.size() / kSchedPieces;
    if (rows_out) *rows_out = rows;
    if (makespan_out) *makespan_out = (n_wavefronts * n_steps + n_slots - 1) / n_slots;
    if (table_out) {
        if (table_capacity_rows < rows) return fail(UAVENV_E_INVALID, "debug_schedule: table_out too small");
        std::memcpy(table_out, table.data(), table.size() * sizeof(int4));
    }
    return UAVENV_OK;
}

extern "C" int uavenv_step_many_prepare(uavenv_t *h, int n_steps) {
    if (!h || n_steps < 0) return fail(UAVENV_E_INVALID, "step_many_prepare: null handle or negative n_steps");
    DeviceGuard guard(h->device);
    (void)rotation_plan(h, n_steps, nullptr);      // builds + uploads + caches the schedule when one applies
    return UAVENV_OK;
}

// Test hook: the schedule uavenv_step_many would use for n_steps (0 launches = plain launch).
extern "C" int uavenv_debug_rotation_info(uavenv_t *h, int n_steps, int *n_launches, long long *slots) {
    if (!h || n_steps < 0) return fail(UAVENV_E_INVALID, "debug_rotation_info: null handle or negative n_steps");
    DeviceGuard guard(h->device);
    const int i = rotation_plan(h, n_steps, nullptr);
    if (n_launches) *n_launches = i >= 0 ? (*h->rot_plans)[(size_t)i].n_launches : 0;
    if (slots) *slots = i >= 0 ? (*h->rot_plans)[(size_t)i].slots : 0;
    return UAVENV_OK;
}

template <int MANY_>
static int launch_many(uavenv_t *h, KParams &p, int n_steps, hipStream_t s) {
    const int i = rotation_plan(h, n_steps, s);        // (first use of this n_steps outside a capture: builds + uploads the table, synchronously)
    if (i >= 0) {
        const uavenv::RotPlan pl = (*h->rot_plans)[(size_t)i];
        p.sched = pl.dev;
        return launch_env<MODE_STEP, MANY_>(h, p, s, pl.slots);
    }
    p.sched = nullptr;
    return launch_env<MODE_STEP, MANY_>(h, p, s);
}

extern "C" int uavenv_step_many(uavenv_t *h, const int64_t *actions_dev, int n_steps, const UavEnvOut *out, void *stream) {
    if (!h || !actions_dev || n_steps < 0) return fail(UAVENV_E_INVALID, "step_many: null handle / actions or negative n_steps");
    if (n_steps == 0) return UAVENV_OK;
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "step_many")) return rc_dev;
    if (h->packed) {   // one launch: state stays in registers across the steps (env_kernel_packed<..., MANY = true>)
        KParams p = h->kp;
        fill_call(p, nullptr, out);
        p.actions = (const long long *)actions_dev; p.n_ticks = n_steps;
        return launch_many<1>(h, p, n_steps, (hipStream_t)stream);
    }
    // multi-pass handles (n_ue > 64): one single-step launch per step on the same stream, each writing its own output block
    for (int t = 0; t < n_steps; ++t) {
        KParams p = h->kp;
        UavEnvOut blk;
        if (out) blk = out_block(*out, t, h->N, h->cfg.n_ue, h->cfg.n_bs);
        fill_call(p, nullptr, out ? &blk : nullptr);
        p.actions = (const long long *)actions_dev + (long long)t * h->N; p.n_ticks = 1;
        if (int rc = launch_env<MODE_STEP>(h, p, (hipStream_t)stream)) return rc;
    }
    return UAVENV_OK;
}

extern "C" int uavenv_step_seq(uavenv_t *h, const int64_t *actions_dev, int n_steps, const UavEnvOut *out, void *stream) {
    if (!h || !actions_dev || n_steps < 0) return fail(UAVENV_E_INVALID, "step_seq: null handle / actions or negative n_steps");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "step_seq")) return rc_dev;
    KParams p = h->kp;
    fill_call(p, nullptr, out);
    p.n_ticks = 1;
    for (int t = 0; t < n_steps; ++t) {          // n_steps ordinary single-step launches, one host call: ~2 us each instead of the
        p.actions = (const long long *)actions_dev + (long long)t * h->N;   // ~8 us a Python -> ctypes -> launch round trip costs
        if (int rc = launch_env<MODE_STEP>(h, p, (hipStream_t)stream)) return rc;
    }
    return UAVENV_OK;
}

extern "C" int uavenv_step_trace(uavenv_t *h, const int64_t *actions_dev, const int16_t *ue_xy_in_dev,
                                 const UavEnvInject *inj, const UavEnvOut *out, void *stream) {
    if (!h || !actions_dev || !ue_xy_in_dev) return fail(UAVENV_E_INVALID, "step_trace: null handle, actions or trace");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "step_trace")) return rc_dev;
    KParams p = h->kp;
    fill_call(p, inj, out);
    p.actions = (const long long *)actions_dev; p.trace_xy = ue_xy_in_dev; p.n_ticks = 1;
    return launch_env<MODE_TRACE>(h, p, (hipStream_t)stream);
}

extern "C" int uavenv_reset_trace(uavenv_t *h, const uint8_t *mask_dev, const int16_t *ue_xy_in_dev,
                                  const UavEnvInject *inj, const UavEnvOut *out, void *stream) {
    if (!h || !ue_xy_in_dev) return fail(UAVENV_E_INVALID, "reset_trace: null handle or trace");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "reset_trace")) return rc_dev;
    KParams p = h->kp;
    fill_call(p, inj, out);
    p.mask = mask_dev; p.trace_xy = ue_xy_in_dev; p.n_ticks = 1;
    return launch_env<MODE_RESET_TRACE>(h, p, (hipStream_t)stream);
}

static int ensure_obs_prev(uavenv_t *h) {
    if (h->obs_prev_dev) return UAVENV_OK;
    const size_t bytes = (size_t)h->kp.N * (h->kp.U + h->kp.B) * sizeof(int32_t);
    if (hipMalloc((void **)&h->obs_prev_dev, bytes) != hipSuccess) return fail(UAVENV_E_NOMEM, "obs_dense: hipMalloc cell list");
    return UAVENV_OK;
}

extern "C" int uavenv_obs_dense(uavenv_t *h, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(UAVENV_E_INVALID, "obs_dense: null handle or buffer");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "obs_dense")) return rc_dev;
    if (int rc = ensure_obs_prev(h)) return rc;   // (first call only; not inside a captured region)
    const KParams &k = h->kp;
    const size_t bytes = (size_t)k.N * (k.B + 1) * k.G * k.G * sizeof(float);
    HIP_TRY(hipMemsetAsync(obs_dev, 0, bytes, (hipStream_t)stream));
    const long long total = k.N * (k.U + k.B);
    hipLaunchKernelGGL((obs_cells_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       k.N, k.U, k.B, k.G, k.bs_xy, k.ue_aux, h->obs_prev_dev, obs_dev);
    HIP_TRY(hipGetLastError());
    h->obs_last_dev = obs_dev;
    return UAVENV_OK;
}

extern "C" int uavenv_obs_dense_update(uavenv_t *h, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(UAVENV_E_INVALID, "obs_dense_update: null handle or buffer");
    if (!h->obs_prev_dev || h->obs_last_dev != obs_dev)   // deltas against another buffer's cell list would corrupt it silently
        return fail(UAVENV_E_INVALID, "obs_dense_update: obs_dev is not the buffer the last uavenv_obs_dense call of this handle wrote; "
                                      "call uavenv_obs_dense on it first");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "obs_dense_update")) return rc_dev;
    const KParams &k = h->kp;
    const long long total = k.N * (k.U + k.B);
    hipLaunchKernelGGL((obs_cells_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       k.N, k.U, k.B, k.G, k.bs_xy, k.ue_aux, h->obs_prev_dev, obs_dev);
    HIP_TRY(hipGetLastError());
    return UAVENV_OK;
}

extern "C" int uavenv_sinr_area_at(uavenv_t *h, const int32_t *bs_xy_dev, const double *fading_inj_dev, float *out_f32_dev,
                                   double *out_f64_dev, void *stream) {
    if (!h || (!out_f32_dev && !out_f64_dev)) return fail(UAVENV_E_INVALID, "sinr_area: null handle or no output buffer");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "sinr_area_at")) return rc_dev;
    const KParams &k = h->kp;
    const int32_t *cells = bs_xy_dev ? bs_xy_dev : k.bs_xy;   // GetSinrInArea(bsLoc) takes ANY bsLoc (channel.py:411); NULL = the state's
    const size_t n = (size_t)k.N * k.G * k.G;
    if (out_f32_dev) HIP_TRY(hipMemsetAsync(out_f32_dev, 0, n * sizeof(float), (hipStream_t)stream));
    if (out_f64_dev) HIP_TRY(hipMemsetAsync(out_f64_dev, 0, n * sizeof(double), (hipStream_t)stream));
    const long long total = k.N * (long long)(k.G - 1) * (k.G - 1);
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define UAVENV_AREA(BT_)                                                                                              \
    do {                                                                                                              \
        if (h->plc) hipLaunchKernelGGL((sinr_area_kernel<BT_, true>), grid, blk, 0, s, k, cells, fading_inj_dev, out_f32_dev, out_f64_dev);   \
        else hipLaunchKernelGGL((sinr_area_kernel<BT_, false>), grid, blk, 0, s, k, cells, fading_inj_dev, out_f32_dev, out_f64_dev);         \
    } while (0)
    switch (h->bt) {
        case 4: UAVENV_AREA(4); break;
        case 8: UAVENV_AREA(8); break;
        case 16: UAVENV_AREA(16); break;
        default: UAVENV_AREA(32); break;
    }
#undef UAVENV_AREA
    HIP_TRY(hipGetLastError());
    return UAVENV_OK;
}

extern "C" int uavenv_sinr_area(uavenv_t *h, const double *fading_inj_dev, float *out_f32_dev, double *out_f64_dev,
                                void *stream) {
    return uavenv_sinr_area_at(h, nullptr, fading_inj_dev, out_f32_dev, out_f64_dev, stream);
}

// The device code paths of csrc/lean_math.h, callable on arrays: lets tests measure the accuracy of what the kernels execute
// (v_rcp_f64 / v_rsq_f64 seeds, contracted FMAs), not only of the host stand-ins.
__global__ __launch_bounds__(256) void lean_math_kernel(int op, const double *a, const double *b, double *o0, double *o1, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LeanCoef C = lm_make_coef<false>();
    const double x = a[i];
    double r0 = 0.0, r1 = 0.0;
    switch (op) {
        case 0: r0 = lm_div(x, b[i]); break;
        case 1: r0 = lm_rsqrt(x); break;
        case 2: r0 = lm_logc(x, C); break;
        case 3: r0 = lm_exp2(x, C); break;
        case 4: lm_sincospi(x, C, &r0, &r1); break;
        default: break;
    }
    o0[i] = r0;
    if (o1 != nullptr) o1[i] = r1;
}

extern "C" int uavenv_lean_math_eval(int op, const double *a_dev, const double *b_dev, double *out0_dev, double *out1_dev, int64_t n,
                                     void *stream) {
    if (op < 0 || op > 4 || n < 0 || (n > 0 && (!a_dev || !out0_dev)) || (op == 0 && n > 0 && !b_dev) || (op == 4 && n > 0 && !out1_dev))
        return fail(UAVENV_E_INVALID, "lean_math_eval: bad op / null buffer");
    if (n == 0) return UAVENV_OK;
    hipLaunchKernelGGL(lean_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, a_dev, b_dev, out0_dev,
                       out1_dev, (long long)n);
    HIP_TRY(hipGetLastError());
    return UAVENV_OK;
}

#ifdef UAVENV_STAMPS
// Diagnostic builds only (not in include/uavenv.h): where the kernels drop their s_memtime stamps.
extern "C" int uavenv_debug_set_stamp_buffer(uavenv_t *h, void *dev_ptr) {
    if (!h) return fail(UAVENV_E_INVALID, "debug_set_stamp_buffer: null handle");
    h->kp.dbg = (unsigned long long *)dev_ptr;
    return UAVENV_OK;
}
#endif

extern "C" int uavenv_state_layout(const uavenv_t *h, UavEnvStateLayout *layout) {
    if (!h || !layout) return fail(UAVENV_E_INVALID, "state_layout: null argument");
    *layout = h->lay;
    return UAVENV_OK;
}

extern "C" int uavenv_get_state(uavenv_t *h, void *dst, int dst_is_device, void *stream) {
    if (!h || !dst) return fail(UAVENV_E_INVALID, "get_state: null argument");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "get_state")) return rc_dev;
    HIP_TRY(hipMemcpyAsync(dst, h->blob, h->lay.total_bytes, dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           (hipStream_t)stream));
    if (!dst_is_device) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return UAVENV_OK;
}

extern "C" int uavenv_set_state(uavenv_t *h, const void *src, int src_is_device, void *stream) {
    if (!h || !src) return fail(UAVENV_E_INVALID, "set_state: null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(h->blob, src, h->lay.total_bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           (hipStream_t)stream));
    if (!src_is_device) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (h->err_host && *(volatile uint32_t *)h->err_host != 0u) {   // a whole state again: the handle is usable (uavenv_device_error)
        HIP_TRY(hipMemsetAsync(h->sched_flag_dev, 0, ((size_t)h->N + 64) * sizeof(uint32_t), (hipStream_t)stream));
        *(volatile uint32_t *)h->err_host = 0u;
    }
    return UAVENV_OK;
}
