// HIP kernels of the batched UAV-cellular environment for gfx950 (MI355X / CDNA4).
//
// Mapping.  The unit of work is one env instance; its walkers (UEs) are lanes of ONE wavefront, so every
// per-env reduction (mean SINR, outage set/count, bounce flags, UAV collision test) is a wavefront ballot or
// shuffle reduction and never touches LDS atomics or another wave.
//   env_kernel_packed  (U <= 64): a wavefront hosts EPW = floor(64/U) env instances side by side
//                      ("slots": lanes [s*U, (s+1)*U)), e.g. 3 envs at 20 UEs, 1 env at 40 UEs.  Ballots are
//                      masked with the slot's lane mask, sums are segmented shuffle reductions.
//   env_kernel_multipass (U > 64): one env per wavefront, walkers in passes of 64 (200 UEs: 4 passes).
//   In a slot, lanes 0..Gr-1 also own one RPGM group each and lanes 0..B-1 one UAV each (they load / store it).
//   UAV cells and the received powers of the BT UAVs live in registers (BT = template bound on B, every index
//   static).  For B <= 8 every lane replays BS_move for its own env serially (no cross-lane traffic, no LDS); for
//   B > 8 the cooperative one-UAV-per-lane form stages the cells in LDS and copies the row into registers.
//   No MFMA: there is no dense contraction on this path.  Arithmetic is float64 throughout (SURVEY.md H2:
//   float32 breaks the 1e-5 relative bound near 0 dB and flips handover/outage decisions); outputs are
//   rounded to float32 once.  Transcendentals: csrc/lean_math.h (accuracy measured in tests/test_lean_math.py).
//
// Kernel variants (template parameters, chosen per launch in uavenv_capi.hip):
//   BT    bound on B (4/8/16/32)         PLC   path-loss exponent 30 => d^-3 by rsqrt, no log
//   FAST  no injected draws, all nine standard outputs, B == BT: no run-time pointer tests, no `b < B` guards
//   PIN   polynomial coefficients + hot constants pinned in VGPRs (wins iff <= 2 wavefronts per SIMD)
// Structure of the packed kernel (why: DESIGN.md section 4, profiles/r01_v*_phase_stamps.txt): leading scalar
// arguments are kernarg-PRELOADED; every global load is issued from them first (state addresses = slab base +
// csrc/state_layout.h), the parameter struct is fetched while those loads are in flight, all stores form one phase.
//
// What the code follows in the reference (/root/reference):
//   mobility tick      ue_mobility.py:453-523     UAV move   ue_mobility.py:191-271,310-336
//   gains, DL SINR     channel.py:220-269         handover / outage / mean   channel.py:138-216
//   reset              channel.py:113-124, mobile_env.py:115-148      reward  mobile_env.py:163-189
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "intdiv.h"
#include "lean_math.h"
#include "philox.h"
#include "state_layout.h"

namespace uavk {

#ifndef UAVENV_WAVES_PER_BLOCK
#define UAVENV_WAVES_PER_BLOCK 4   // wavefronts per workgroup of the env kernels (build-time knob for A/B runs)
#endif
constexpr int kWavesPerBlock = UAVENV_WAVES_PER_BLOCK;
constexpr int kMaxGroups = 16;
constexpr int kMaxBs = 32;
constexpr int kMaxEpw = 8;  // env instances per wavefront (packed kernel)

enum Mode : int {
    MODE_WARMUP = 0,       // mobility ticks only                                   (mobile_env.py:77-79)
    MODE_RESET = 1,        // UAVs to start cells + one tick + LTEChannel.reset     (mobile_env.py:115-148, "group")
    MODE_STEP = 2,         // tick + BS_move + UpdateDroneNet                       (mobile_env.py:150-194)
    MODE_TRACE = 3,        // step_test, UE cells from a trace, no tick             (mobile_env.py:196-233, "read_trace")
    MODE_RESET_TRACE = 4   // reset, UE cells from a trace, no tick                 (mobile_env.py:128-131)
};
__host__ __device__ constexpr bool is_reset(int m) { return m == MODE_RESET || m == MODE_RESET_TRACE; }
__host__ __device__ constexpr bool is_step(int m) { return m == MODE_STEP || m == MODE_TRACE; }
__host__ __device__ constexpr bool has_mobility(int m) { return m == MODE_WARMUP || m == MODE_RESET || m == MODE_STEP; }

// FAST kernels are launched when the call injects no randomness, passes all nine standard outputs and no
// float64 copies (what BatchedMobiEnv.step does): every optional-pointer test folds away at compile time.
// Why it matters: each runtime test was a scalar kernarg load + s_waitcnt lgkmcnt(0) + branch; the r01_v3
// profile has 59 such drains per wave = 47 % of the wave cycles (profiles/r01_v3_packed_sq_counters.txt).
// FAST additionally means B == BT (host: launch_env), so every `b < B` guard of the unrolled per-UAV loops folds at
// compile time.  Left as run-time tests they were uniform branches, and hipcc parked hoisted kernarg loads in the
// little blocks between them, each with its own s_waitcnt (stamps: ~9 serial scalar round trips in the UAV-move phase).
template <int BT, bool FAST>
__device__ __forceinline__ int uav_count(int runtime_b) { return FAST ? BT : runtime_b; }
#define UAV_INJ(ptr) (!FAST && (ptr) != nullptr)     /* injected draws present            */
#define UAV_OUT(ptr) (FAST || (ptr) != nullptr)      /* standard output requested         */
#define UAV_OUT64(ptr) (!FAST && (ptr) != nullptr)   /* optional float64 copy requested   */

struct OutPtrs {
    float *reward; uint8_t *done; float *mean_sinr; int32_t *n_out;
    int16_t *ue_xy; int32_t *bs_xy; int8_t *serving; float *cur_sinr; int32_t *step_n;
    double *cur_sinr_f64, *mean_sinr_f64, *reward_f64;
};

struct KParams {
    // shape / constants
    int U, B, Gr, G, W64, epw, act32;
    uint32_t div_magic, div_shift;   // exact a / n_act by multiply-shift (intdiv.h)
    int group_start[kMaxGroups + 1];
    int max_step, bs_step, min_bs_dist2, n_act, agg_init, deagg_len, agg_len;
    double grid_width, p_bs_watt, noise_watt, pl_a, pl_b, pl_dis, antenna_gain, eq_loss;
    double k_pl, k_0, c_exp, pl_exp_ln, pl_dis2, db_per_ln;  // folded constants, see rx_power() / sinr_db()
    double inv_U, inv_U20;           // 1/U and 1/(20 U): mean and reward terms by one multiply each (env_finish)
    double shadow_mean, shadow_sd, ho_thresh_db, out_thresh, ue_velocity, grp_v_min, grp_v_max, aggregation;
    long long N;
    uint32_t key0, key1, env_id_base;
    // persistent state: arrays of records, [field][env][...] (state_layout.h)
    UePos *ue_pos; UeAux *ue_aux; GrpRec *grp; EnvRec *env; int32_t *bs_xy; unsigned long long *out_bits;   // state_layout.h
    const int32_t *bs_init;      // [B,2] device copy of the start cells
    const long long *act_pow;    // [B]   n_act^(B-1-b): joint action -> digit of UAV b (most significant first)
    // Split decode of a joint action beyond 32 bits (B > 8; config 5: 5^16): a = hi * act_P + lo with both halves below 2^32, one double
    // multiply + a +-1 correction instead of two emulated 64-bit divisions per UAV lane; then a multiply-shift division per UAV
    // (act_dec[b] = {half: 1 = hi, magic, shift, power}, intdiv.h).  act_split = 0: the plain 64-bit form (n_act^B >= 2^52, or a half >= 2^32).
    const uint4 *act_dec;
    int act_split;
    uint32_t act_P;
    double act_inv_P;
    const int8_t *gid_of_u;      // [U]   RPGM group of walker u (from group_size)
    // per-call inputs
    const double *inj_theta, *inj_group, *inj_fading;
    const long long *actions; const uint8_t *mask; const int16_t *trace_xy; int n_ticks;
    OutPtrs out;
    unsigned long long *dbg;   // diagnostic builds only (UAVENV_STAMPS): [waves][8] s_memtime stamps
    const int4 *sched;         // multi-step launches: work descriptors [launch waves][kSchedPieces] {env-wavefront, first step, steps,
                               // SCHED_* bits} of a rotation schedule (uavenv_capi.hip: rotation_plan), or null = wave w runs
                               // env-wavefront w, all steps
    uint32_t *sched_flag;      // [env-wavefronts] hand-off words of the one-launch schedule: 1 = "the first steps of this env-wavefront
                               // are done and its state is in memory"; set by the producing wavefront, cleared by the consuming one
    uint32_t *sched_err;       // host-mapped sticky error word of the handle: a hand-off that is not signalled within the spin budget
                               // stores UAVENV_DEV_ERR_HANDOFF here and the wavefront exits (the host then fails every later call)
    uint32_t sched_spin_us;    // spin budget of one hand-off wait, in microseconds of s_memrealtime (100 MHz)
    int wave0;                 // multi-pass kernel: first env of the launch (uavenv_step_range; the packed kernel takes it as a scalar argument)
    long long e_end;           // multi-pass kernel: one past the last env of the launch (N, or the end of the range)
};
constexpr int kSchedPieces = 3;          // a slot of a one-launch schedule: [first steps of a split job] [whole job] [last steps of another]
enum : int { SCHED_WAIT = 1,             // piece starts at step t0 > 0: wait for the flag of its env-wavefront, acquire, then load the state
             SCHED_PUBLISH = 2 };        // piece ends before the call's last step: drain the state stores, release, set the flag
constexpr uint32_t kDevErrHandoff = 0x48414e44u;   // "HAND": sched_err value of a hand-off that timed out

struct InitParams {
    int U, Gr, B, W64, G, per; int agg_init, deagg_len;
    double grp_v_min, grp_v_max;
    long long N; uint32_t key0, key1, env_id_base;
    UePos *ue_pos; UeAux *ue_aux; GrpRec *grp; EnvRec *env; int32_t *bs_xy; unsigned long long *out_bits;   // state_layout.h
    const int32_t *bs_init;
    const double *u_x, *u_y, *u_th, *u_g;
};

// Typed pointers to the state fields of one handle.  The packed kernel builds them from the slab base + the shared
// layout function (scalar arithmetic on preloaded arguments); the multi-pass kernel copies them from KParams.
struct StatePtrs {
    UePos *ue_pos; UeAux *ue_aux; GrpRec *grp; EnvRec *env; int32_t *bs_xy; unsigned long long *out_bits;   // state_layout.h
};
// Element access as  uniform base + 32-bit byte offset.  hipcc then emits the SGPR-base form
// (global_load_dwordx2 v[..], v_off, s[base:base+1]) instead of one 64-bit VGPR address per array: arrays indexed alike share
// ONE offset register, the per-array v_lshl_add_u64 / v_mad_u64_u32 address arithmetic disappears and half as many address
// dwords go to the texture addresser per access.  Precondition, enforced by uavenv_create(): every array the packed kernel
// indexes this way is smaller than 4 GiB, so idx * sizeof(T) cannot wrap.
template <class T> __device__ __forceinline__ T ldx(const T *base, uint32_t idx) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + idx * (uint32_t)sizeof(T));
}
template <class T, class V> __device__ __forceinline__ void stx(T *base, uint32_t idx, V v) {
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + idx * (uint32_t)sizeof(T)) = (T)v;
}
// The same accesses made COHERENT at agent scope (hand-off pieces of a one-launch schedule): relaxed agent-scope atomics, i.e.
// global_load / global_store ... sc1 in 8-byte (4-byte) pieces.  Such a load bypasses this CU's L1 and is served by the L2 / the
// coherence point; such a store is written through.  A record moved this way needs no release / acquire FENCE around it: the
// producer waits for its stores (s_waitcnt vmcnt(0)) before it sets the flag, the consumer loads only after its poll has seen the flag
// (MI355X_MICROARCH.md, "Valid forms": sc1 payload -> asm vmcnt(0) -> sc1 flag; every load of the handed-off bytes an sc1 load).
// Measured before building it: the two fences cost a 20-step call 6 us of 91 (profiles/r04l_fence_cost.json).
template <bool COH, class T> __device__ __forceinline__ T ldx_c(const T *base, uint32_t idx) {
    if (!COH) return ldx(base, idx);
    const char *q = reinterpret_cast<const char *>(base) + idx * (uint32_t)sizeof(T);
    union U { T t; unsigned long long w[(sizeof(T) + 7) / 8]; uint32_t d[(sizeof(T) + 3) / 4]; __device__ U() {} } u;
    if (sizeof(T) % 8 == 0) {
#pragma unroll
        for (int i = 0; i < (int)(sizeof(T) / 8); ++i)
            u.w[i] = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        static_assert(sizeof(T) % 8 == 0 || sizeof(T) == 4, "coherent access: 4-byte or 8-byte-multiple records");
        u.d[0] = __hip_atomic_load(reinterpret_cast<const uint32_t *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return u.t;
}
template <bool COH, class T, class V> __device__ __forceinline__ void stx_c(T *base, uint32_t idx, V v) {
    if (!COH) { stx(base, idx, v); return; }
    char *q = reinterpret_cast<char *>(base) + idx * (uint32_t)sizeof(T);
    union U { T t; unsigned long long w[(sizeof(T) + 7) / 8]; uint32_t d[(sizeof(T) + 3) / 4]; __device__ U() {} } u;
    u.t = (T)v;
    if (sizeof(T) % 8 == 0) {
#pragma unroll
        for (int i = 0; i < (int)(sizeof(T) / 8); ++i)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(q) + i, u.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        static_assert(sizeof(T) % 8 == 0 || sizeof(T) == 4, "coherent access: 4-byte or 8-byte-multiple records");
        __hip_atomic_store(reinterpret_cast<uint32_t *>(q), u.d[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// An index for the stores of ONE basic block.  Without it hipcc hoists base + zext(offset) of a store into the kernel prologue
// (v_lshl_add_u64 sgpr_base, vgpr_offset), keeps every such 64-bit address alive in a VGPR pair across the whole compute phase
// and, selecting instructions block by block, can no longer fold it into the SGPR-base form at the store.  The empty asm
// defines the index in the store's own block, so the address is formed there.  No instruction is emitted.
// It is a trade, measured on one box (profiles/r01_v19_ab_addr32_*.txt): the SGPR-base form keeps ~29 base pointers alive in
// SGPR pairs until the store phase.  In the variant that pins its constants in VGPRs (PIN, launches of <= 2 wavefronts per
// SIMD) that costs 32 + 24 SGPR spill moves and is 1 % SLOWER than letting the compiler hoist (8.80 vs 8.71 us at 4096 envs);
// in the unpinned variant it takes 106 -> 90 VGPRs, i.e. 5 instead of 4 wavefronts per SIMD, and is 3 % FASTER (53.7 vs 55.4 us
// at 65536 envs).  Hence LOCAL = !PIN at the call sites.
template <bool LOCAL> __device__ __forceinline__ uint32_t block_local(uint32_t idx) {
    if (LOCAL) asm volatile("" : "+v"(idx));
    return idx;
}

// A (wave-uniform) pointer kept in a VGPR pair: frees two SGPRs where the scalar file is the scarce resource.
template <class T> __device__ __forceinline__ T *vgpr_ptr(T *q) { asm volatile("" : "+v"(q)); return q; }

__device__ __forceinline__ StatePtrs state_from_blob(char *b, long long N, int U, int B, int Gr) {
    const StateOffsets L = compute_layout(N, U, B, Gr);
    StatePtrs s;
    s.ue_pos = (UePos *)(b + L.ue_pos); s.ue_aux = (UeAux *)(b + L.ue_aux); s.grp = (GrpRec *)(b + L.grp);
    s.env = (EnvRec *)(b + L.env); s.bs_xy = (int32_t *)(b + L.bs_xy); s.out_bits = (unsigned long long *)(b + L.out_bits);
    return s;
}

__device__ __forceinline__ StatePtrs state_from_params(const KParams &p) {
    StatePtrs s;
    s.ue_pos = p.ue_pos; s.ue_aux = p.ue_aux; s.grp = p.grp; s.env = p.env; s.bs_xy = p.bs_xy; s.out_bits = p.out_bits;
    return s;
}

// In-kernel phase stamps (diagnostic build -DUAVENV_STAMPS only; MI355X guide, "In-kernel stamps"): one asm
// statement per stamp with its own lgkmcnt(0), scheduling barriers around it.  Values go to p.dbg, never to outputs.
#ifdef UAVENV_STAMPS
#define UAV_STAMP(var)                                                             \
    do {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)
#define UAV_DRAIN_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define UAV_STAMP(var) do { } while (0)
#define UAV_DRAIN_VM() do { } while (0)
#endif

// Kernarg warm-up.  hipcc loads kernarg fields lazily, one scalar-load batch per block that first uses them, and the
// phase stamps (gpurun_out/stamps_*.log) show a cold kernarg fetch costing ~1900 ticks (~1 us):
// the kernarg block is rewritten for every launch, so the first touch of each 64-byte line misses every cache.
// Forcing HOST kernargs (HIP_FORCE_DEV_KERNARG=0) makes the launch 46 % slower, i.e. this latency is first-order.
// Here every line of the block is touched once at kernel entry by independent scalar loads (one round trip for all
// of them); the later lazy loads then hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#ifndef UAVENV_NO_KERNARG_WARM
    typedef const __attribute__((address_space(4))) unsigned int *kptr_t;
    kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    // All loads are consumed by ONE asm statement, so they are issued back to back and waited for once.  (One asm per
    // load made hipcc emit load, s_waitcnt, load, s_waitcnt ...: ten serial round trips instead of one.)
    constexpr int LINES = (BYTES + 63) / 64;
    static_assert(LINES <= 16, "kernarg block larger than the 16 lines this warm-up touches");
    unsigned int v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (i < LINES) ? ka[i * 16] : 0u;
    asm volatile("" ::"s"(v[0]), "s"(v[1]), "s"(v[2]), "s"(v[3]), "s"(v[4]), "s"(v[5]), "s"(v[6]), "s"(v[7]), "s"(v[8]),
                 "s"(v[9]), "s"(v[10]), "s"(v[11]), "s"(v[12]), "s"(v[13]), "s"(v[14]), "s"(v[15]));
#endif
}

// ================================================================================================
// shared device helpers (both env kernels run exactly this arithmetic)
// ================================================================================================

// Hot-path float64 constants, copied once from the kernarg block into PINNED VGPRs.  Left in SGPRs they (15 doubles
// = 30 SGPRs) compete with ~30 pointers and the per-lane booleans for the 102-SGPR file, and the overflow is
// spilled through v_writelane/v_readlane, i.e. paid in VALU issue slots; VGPRs are plentiful here.
struct HotConst {
    double vel, aggr, maxc, sh_mean, sh_sd, c_exp, k_pl, k_0, pl_dis2, pl_exp_ln, noise, db_per_ln, ho_thr, out_thr, gw;
};
template <bool PIN>
__device__ __forceinline__ HotConst make_hot(const KParams &p) {
    HotConst h = {p.ue_velocity, p.aggregation, (double)p.G, p.shadow_mean, p.shadow_sd, p.c_exp, p.k_pl, p.k_0,
                  p.pl_dis2, p.pl_exp_ln, p.noise_watt, p.db_per_ln, p.ho_thresh_db, p.out_thresh, p.grid_width};
    lm_pin<PIN>(h.vel); lm_pin<PIN>(h.aggr); lm_pin<PIN>(h.maxc); lm_pin<PIN>(h.sh_mean); lm_pin<PIN>(h.sh_sd); lm_pin<PIN>(h.c_exp); lm_pin<PIN>(h.k_pl);
    lm_pin<PIN>(h.k_0); lm_pin<PIN>(h.pl_dis2); lm_pin<PIN>(h.pl_exp_ln); lm_pin<PIN>(h.noise); lm_pin<PIN>(h.db_per_ln); lm_pin<PIN>(h.ho_thr);
    lm_pin<PIN>(h.out_thr); lm_pin<PIN>(h.gw);
    return h;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over the U (<= 64) consecutive lanes of a slot; the result is valid in the slot's first lane (ul == 0).
__device__ __forceinline__ double slot_sum(double v, int ul, int U) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_down(v, off, 64);
        if (ul + off < U) v += t;
    }
    return v;
}

__device__ __forceinline__ void philox_u2(const KParams &p, uint32_t env, uint32_t tick, uint32_t idx, uint32_t dom,
                                          double &u0, double &u1) {
    const U4 r = philox4x32_10(p.env_id_base + env, tick, idx, dom, p.key0, p.key1);
    u0 = u53(r.x, r.y);
    u1 = u53(r.z, r.w);
}

// Per-UE draw block of a tick (the oracle defines the same stream, oracle/uavenv_oracle.c):
//   call p of walker u:  q_p = Philox(ctr = (env, tick, u*HB + p, DOM_FADING)),  HB = ceil(B/2)
//   q_p.x, q_p.y -> 53-bit uniform of the Box-Muller radius;  q_p.z * 2^-32 -> angle fraction;  q_p.w -> spare
//   heading uniform drawn this tick = u53(q_0.w, q_1.w)  (HB >= 2)   or   q_0.w * 2^-32  (HB == 1)
// Two Philox calls per UE and tick instead of three: the quarter-rate 32x32->64 multiplies of Philox were ~17 % of
// the step kernel's issue cycles.
// QUAD mode, B > 8 (the 16 / 32-UAV shapes are VALU-issue bound and Philox was 19 % of it): one call serves FOUR UAVs,
//   call c of walker u:  q_c = Philox(ctr = (env, tick, u*QB + c, DOM_FADING)),  QB = ceil(B/4)
//   UAVs 4c, 4c+1: radius uniform q_c.x * 2^-32, angle fraction q_c.y * 2^-32;   UAVs 4c+2, 4c+3: q_c.z and q_c.w
//   heading uniform = u53(h.x, h.y),  h = Philox(ctr = (env, tick, u, DOM_HEADING)).
__host__ __device__ constexpr bool quad_draws(int B) { return B > 8; }
__device__ __forceinline__ U4 philox_raw(const KParams &p, uint32_t env, uint32_t tick, uint32_t idx, uint32_t dom) {
    return philox4x32_10(p.env_id_base + env, tick, idx, dom, p.key0, p.key1);
}
__device__ __forceinline__ double heading_from(const U4 &q0, const U4 &q1, int HB) {
    return (HB >= 2) ? u53(q0.w, q1.w) : (double)q0.w * (1.0 / 4294967296.0);
}

// Digit of UAV b in the joint action: Decimal_to_Base_N (ue_mobility.py:310-336), most significant digit ->
// UAV 0.  One division per UAV lane, all UAVs in parallel (pw = n_act^(B-1-b)).
__device__ __forceinline__ int action_digit(const KParams &p, long long a, long long pw, int b) {
    if (p.act32) return (int)(((uint32_t)a / (uint32_t)pw) % (uint32_t)p.n_act);
    if (p.act_split) {                                   // (0 <= a < n_act^B < 2^52: exact in a double)
        const double ad = fma((double)(uint32_t)((unsigned long long)a >> 32), 4294967296.0, (double)(uint32_t)a);
        uint32_t hi = (uint32_t)(ad * p.act_inv_P);                                   // floor(a / P) or one off
        long long r = a - (long long)((unsigned long long)hi * (unsigned long long)p.act_P);
        if (r < 0) { hi -= 1u; r += (long long)p.act_P; }
        else if (r >= (long long)p.act_P) { hi += 1u; r -= (long long)p.act_P; }
        const uint4 d = p.act_dec[b];
        const uint32_t x = d.x ? hi : (uint32_t)r;
        const uint32_t q = (d.w == 1u) ? x : u32div(x, d.y, d.z);
        return (int)(q - (uint32_t)p.n_act * u32div(q, p.div_magic, p.div_shift));
    }
    return (int)(((unsigned long long)a / (unsigned long long)pw) % (unsigned long long)p.n_act);
}

// Proposed cell of one UAV for digit di (ue_mobility.py:221-253; bounds mobile_env.py:45).
__device__ __forceinline__ void uav_propose(const KParams &p, int xi, int yi, int di, int &nx, int &ny) {
    const int xMin = 1, xMax = p.G, yMin = 1, yMax = p.G;
    const int s = p.bs_step, sl = 2 * p.bs_step;
    nx = xi; ny = yi;
    if (di == 0) { if (xi + s < xMax) nx = xi + s; }
    else if (di == 1) { if (xi - s > xMin) nx = xi - s; }
    else if (di == 2) { if (yi + s < yMax) ny = yi + s; }
    else if (di == 3) { if (yi - s > yMin) ny = yi - s; }
    else if (di == 5) { if (xi + sl < xMax) nx = xi + sl; }
    else if (di == 6) { if (xi - sl > xMin) nx = xi - sl; }
    else if (di == 7) { if (yi + sl < yMax) ny = yi + sl; }
    else if (di == 8) { if (yi - sl > yMin) ny = yi - sl; }
}

// Displacement of digit d in units of BS_STEP, as (k + 2) in bits [3d, 3d+3): digits 0..3 = +x, -x, +y, -y by one step,
// 4 stays, 5..8 the same four directions by two steps (ue_mobility.py:221-253).  check_config() keeps n_act <= 9, so d <= 8.
constexpr uint32_t pack_digit_lut(int k0, int k1, int k2, int k3, int k4, int k5, int k6, int k7, int k8) {
    return (uint32_t)(k0 + 2) | (uint32_t)(k1 + 2) << 3 | (uint32_t)(k2 + 2) << 6 | (uint32_t)(k3 + 2) << 9 | (uint32_t)(k4 + 2) << 12 |
           (uint32_t)(k5 + 2) << 15 | (uint32_t)(k6 + 2) << 18 | (uint32_t)(k7 + 2) << 21 | (uint32_t)(k8 + 2) << 24;
}
constexpr uint32_t kDigitLutX = pack_digit_lut(+1, -1, 0, 0, 0, +2, -2, 0, 0);
constexpr uint32_t kDigitLutY = pack_digit_lut(0, 0, +1, -1, 0, 0, 0, +2, -2);
constexpr int digit_lut_k(uint32_t lut, int d) { return (int)((lut >> (3 * d)) & 7u) - 2; }
// The tables against the compare/select formulation they replace (v1-v16 of this kernel), for every digit the config allows.
constexpr int digit_dir(int d) { return d >= 5 ? d - 5 : d; }
constexpr int digit_len(int d) { return (d == 4 || d > 8) ? 0 : (d >= 5 ? 2 : 1); }
constexpr bool digit_luts_ok() {
    for (int d = 0; d <= 8; ++d) {
        const int kx = digit_dir(d) == 0 ? digit_len(d) : (digit_dir(d) == 1 ? -digit_len(d) : 0);
        const int ky = digit_dir(d) == 2 ? digit_len(d) : (digit_dir(d) == 3 ? -digit_len(d) : 0);
        if (digit_lut_k(kDigitLutX, d) != kx || digit_lut_k(kDigitLutY, d) != ky) return false;
    }
    return true;
}
static_assert(digit_luts_ok(), "digit -> displacement tables disagree with ue_mobility.py:221-253");

// BS_move replayed serially in registers by EVERY lane for its own env (BT <= 8: B*(B-1) integer checks, no
// cross-lane traffic).  profiles/r01_v8: the cooperative version (one UAV per lane, 3 ds_bpermute + 1 ballot per
// sequential round, LDS staging + barrier) took 21 % of a wavefront's lifetime, more than twice the mobility tick.
// Semantics as ue_mobility.py:191-271: UAV i proposes from digit i (most significant digit -> UAV 0, :310-336), the
// collision test uses i's PRE-move cell against the already-updated cells of j < i and the old cells of j > i.
template <int BT, bool FAST>
__device__ __forceinline__ void bs_move_serial(const KParams &p, unsigned a, int (&bsx)[BT], int (&bsy)[BT]) {
    const int B = uav_count<BT, FAST>(p.B);
    const unsigned n = (unsigned)p.n_act;
    const int xMin = 1, xMax = p.G;                            // mobile_env.py:45; the same bounds hold for y
    int dig[BT];
#pragma unroll
    for (int b = BT - 1; b >= 0; --b) {                         // least significant digit -> UAV B-1
        dig[b] = 4;
        if (b < B) {
            const unsigned q = u32div(a, p.div_magic, p.div_shift);
            dig[b] = (int)(a - q * n);
            a = q;
        }
    }
#pragma unroll
    for (int i = 0; i < BT; ++i) {
        if (i < B) {
            // proposal (:221-253), branch-free: one bit-field extract per axis instead of compare/select chains.
            const uint32_t sh = 3u * (uint32_t)dig[i];
            const int kx = (int)((kDigitLutX >> sh) & 7u) - 2, ky = (int)((kDigitLutY >> sh) & 7u) - 2;
            const int ddx = kx * p.bs_step, ddy = ky * p.bs_step;
            const int nx = bsx[i] + ddx, ny = bsy[i] + ddy;
            // Only the MOVED coordinate is range-checked, as in the reference (a UAV on a wall cell may slide along the wall), and
            // at most one of (ddx, ddy) is non-zero; the grid is square (xMin == yMin, xMax == yMax, mobile_env.py:45), so one
            // unsigned test xMin < moved < xMax does it.  The reference tests one side only (x + s < xMax, x - s > xMin); with
            // start cells in [1, G-1] (check_config) and cells changing only through accepted moves, a + move lands on >= 2 and
            // a - move on <= G-2, so the other side always holds and the two-sided test is the same predicate.  Digit 4 /
            // bs_step == 0: nothing moves and the value of `inside` is irrelevant.
            const int moved = (kx != 0) ? nx : ny;
            const int inside = (int)((uint32_t)(moved - (xMin + 1)) < (uint32_t)(xMax - xMin - 1));
            // collision (:256-263): PRE-move cell of i against the current cells of all j != i; integer form of
            // norm <= min_dist, as a running minimum (one comparison, no chain of per-lane booleans)
            int dmin = 0x7FFFFFFF;
#pragma unroll
            for (int j = 0; j < BT; ++j) {
                if (j != i && j < B) {
                    const int dx = bsx[i] - bsx[j], dy = bsy[i] - bsy[j];
                    const int d2 = dx * dx + dy * dy;
                    dmin = d2 < dmin ? d2 : dmin;
                }
            }
            const int go = inside & (int)(dmin > p.min_bs_dist2);                     // :265-266
            bsx[i] += go * ddx;
            bsy[i] += go * ddy;
        }
    }
}

// One next() of reference_point_group for one walker (ue_mobility.py:455-505): own step along the heading
// drawn last tick, group step (+ pull towards the group centre while aggregating), then the four ordered
// bounce tests.  c[k] report which tests fired (they flip the GROUP heading, once per group and test).
__device__ __forceinline__ void walker_move(const HotConst &H, const LeanCoef &C, bool aggregating, double hu, double gx,
                                            double gy, double gv, double gc, double gs, double MAXC, double &x, double &y,
                                            bool c[4]) {
    double sn, cs;
    lm_sincospi(2.0 * hu, C, &sn, &cs);       // theta = 2*pi*u  (:437,508)
    x = x + H.vel * cs;                       // :455
    y = y + H.vel * sn;                       // :456
    // cos/sin of c_theta = arctan2(g_y - y, g_x - x) (:467) are the normalised components of the vector to the
    // group centre; arctan2(0, 0) = 0 gives (1, 0).
    const double dxc = gx - x, dyc = gy - y;
    const double r2 = dxc * dxc + dyc * dyc;
    const double rinv = lm_rsqrt(r2);         // r2 == 0 gives a non-finite value; the selects below discard it
    const double cc = (r2 > 0.0) ? dxc * rinv : 1.0;
    const double sc = (r2 > 0.0) ? dyc * rinv : 0.0;
    x = x + gv * gc;                          // :469 / :483
    y = y + gv * gs;                          // :470 / :484
    if (aggregating) {                        // :461-470 (per-lane select: slots may be in different phases)
        x = x + H.aggr * cc;
        y = y + H.aggr * sc;
    }
    c[0] = x < 0.0;                           // :490-493
    if (c[0]) x = -x;
    c[1] = x > MAXC;                          // :494-497
    if (c[1]) x = 2.0 * MAXC - x;
    c[2] = y < 0.0;                           // :498-501
    if (c[2]) y = -y;
    c[3] = y > MAXC;                          // :502-505
    if (c[3]) y = 2.0 * MAXC - y;
}

// Group owner, end of tick (ue_mobility.py:493-521): bounce flips, remaining flight length, arrival redraw.
template <bool FAST>
__device__ __forceinline__ void group_finish(const KParams &p, const LeanCoef &C, long long e, int g, uint32_t tick, const uint32_t touched[4],
                                             double MAXC, double &ogfl, double &ogv, double &ogc, double &ogs) {
    const uint32_t bit = 1u << g;
    if (touched[0] & bit) ogc = -ogc;
    if (touched[1] & bit) ogc = -ogc;
    if (touched[2] & bit) ogs = -ogs;
    if (touched[3] & bit) ogs = -ogs;
    ogfl = ogfl - ogv;                                    // :513
    if (ogv > 0.0 && ogfl <= 0.0) {                       // :514
        double ut, uf, uv, t1;
        if (UAV_INJ(p.inj_group)) {
            ut = p.inj_group[(e * p.Gr + g) * 3 + 0]; uf = p.inj_group[(e * p.Gr + g) * 3 + 1];
            uv = p.inj_group[(e * p.Gr + g) * 3 + 2];
        } else {
            philox_u2(p, (uint32_t)e, tick, (uint32_t)g, DOM_GROUP_A, ut, uf);
            philox_u2(p, (uint32_t)e, tick, (uint32_t)g, DOM_GROUP_B, uv, t1);
        }
        lm_sincospi(2.0 * ut, C, &ogs, &ogc);             // :517-519
        ogfl = uf * MAXC;                                 // :520 FL_MAX = max(dimensions)
        ogv = uv * (p.grp_v_max - p.grp_v_min) + p.grp_v_min;  // :521
    }
}

// Received power P*gain of every UAV at one walker (channel.py:220-257), linear domain.
// The reference goes through dB and back (loss = a + b*log10(d); gain = 10^((ant-loss-f-eq)/10)).
// Same value with fewer transcendentals:
//     P*gain = k_pl * 10^(-f/10) * d^(-b/10)   for d > pl_dis   (b = 30: d^-3 = rsqrt(d^2)^3, no log: PLC)
//            = k_0  * 10^(-f/10)               otherwise (loss = 0)
// k_pl = P*10^((ant-a-eq)/10), k_0 = P*10^((ant-eq)/10) are folded on the host (float64 pow).
// bsx/bsy: this env's UAV cells, in registers.  f ~ N(mean, sd) per (UE, UAV): injected, or Box-Muller on
// Philox uniforms (one call -> two UAVs), replacing np.random.normal (channel.py:240).
template <int BT, bool PLC, bool FAST, bool PRE>
__device__ __forceinline__ void rx_power(const KParams &p, const HotConst &H, const LeanCoef &C, long long e, uint32_t tick, int u, bool act,
                                         long long iu, int ix, int iy, const int (&bsx)[BT], const int (&bsy)[BT],
                                         const U4 &q0, const U4 &q1, double pg[BT]) {
    // PRE: q0 / q1 are this walker's calls 0 and 1, already made for the heading of the same tick.
    const int B = uav_count<BT, FAST>(p.B);
    U4 qq = {0u, 0u, 0u, 0u};   // quad mode: the current four-UAV call
    (void)qq;
#pragma unroll
    for (int b2 = 0; b2 < BT; b2 += 2) {
        double f0 = 0.0, f1 = 0.0;
        if (b2 < B) {
            if (UAV_INJ(p.inj_fading)) {
                if (act) {
                    f0 = p.inj_fading[iu * B + b2];
                    if (b2 + 1 < B) f1 = p.inj_fading[iu * B + b2 + 1];
                }
            } else if (quad_draws(B)) {
                // quad mode (B > 8; a compile-time fact in the FAST variants, where B == BT): one Philox call per four UAVs,
                // 32-bit radius uniform and angle fraction per pair
                if ((b2 & 3) == 0) qq = philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * ((B + 3) >> 2) + (b2 >> 2)), DOM_FADING);
                const uint32_t wr = (b2 & 3) == 0 ? qq.x : qq.z, wa = (b2 & 3) == 0 ? qq.y : qq.w;
                const double u0 = (double)wr * (1.0 / 4294967296.0);
                const double t = -2.0 * lm_logc(1.0 - u0, C);      // 1-u0 in [2^-32, 1]
                const double r = (t > 0.0) ? t * lm_rsqrt(t) : 0.0;
                double sa, ca;
                lm_sincospi((double)wa * (1.0 / 2147483648.0), C, &sa, &ca);
                f0 = H.sh_mean + H.sh_sd * (r * ca);
                f1 = H.sh_mean + H.sh_sd * (r * sa);
            } else {
                const U4 q = (PRE && b2 == 0) ? q0 : ((PRE && b2 == 2) ? q1 :
                             philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * ((B + 1) >> 1) + (b2 >> 1)), DOM_FADING));
                const double u0 = u53(q.x, q.y);
                const double t = -2.0 * lm_logc(1.0 - u0, C);      // 1-u0 in [2^-53, 1]: positive, normal
                const double r = (t > 0.0) ? t * lm_rsqrt(t) : 0.0;   // sqrt(t); t == 0 only when u0 == 0
                double sa, ca;
                lm_sincospi((double)q.z * (1.0 / 2147483648.0), C, &sa, &ca);   // angle = 2*pi * q.z / 2^32
                f0 = H.sh_mean + H.sh_sd * (r * ca);
                f1 = H.sh_mean + H.sh_sd * (r * sa);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = b2 + k;
            if (b < BT) {
                double g = 0.0;
                if (b < B) {
                    const double f = (k == 0) ? f0 : f1;
                    const double fx = H.gw * (double)(ix - bsx[b]);                   // :221-222
                    const double fy = H.gw * (double)(iy - bsy[b]);
                    const double d2 = fx * fx + fy * fy;                              // d^2, :223 (z ignored); exact
                    if (PLC) {
                        const double rinv = lm_rsqrt(d2);                             // d^-3 = (d2^-1/2)^3
                        g = H.k_pl * lm_exp2(H.c_exp * f, C) * (rinv * rinv * rinv);
                    } else {
                        g = H.k_pl * lm_exp2(H.c_exp * f - H.pl_exp_ln * lm_logc(d2, C), C);  // d^(-b/10) = 2^(-(b/20) log2 d2)
                    }
                    if (!(d2 > H.pl_dis2)) g = H.k_0 * lm_exp2(H.c_exp * f, C);       // d <= pl_dis: loss = 0 (:232-233)
                }
                pg[b] = g;
            }
        }
    }
}

// Best UAV: SINR_b = pg_b / (noise + sum_{j != b} pg_j) is strictly increasing in pg_b (the total is fixed),
// so np.argmax over the dB values (channel.py:141) == first maximum of pg.
// best_pg returns pg[best] (the scan holds it anyway, so the caller need not select it again by index).
template <int BT, bool FAST>
__device__ __forceinline__ int argmax_pg(const KParams &p, const double pg[BT], double &best_pg) {
    const int B = uav_count<BT, FAST>(p.B);
    int best = 0;
    double bp = pg[0];
#pragma unroll
    for (int b = 1; b < BT; ++b)
        if (b < B && pg[b] > bp) { bp = pg[b]; best = b; }
    best_pg = bp;
    return best;
}

// 10*log10(S/(N+I)) for UAV x (channel.py:259-268); interference = the OTHER UAVs summed in index order,
// never total - self (cancellation).  Only two of the B values are ever consumed: best and serving.
// px = pg[x], supplied by the caller.  The masked sum is an fma with m in {0.0, 1.0}: pg*1 + interf rounds exactly like
// interf + pg and pg*0 + interf is interf, so the value is that of `interf += (j != x) ? pg[j] : 0.0`; choosing between 0.0
// and 1.0 takes one v_cndmask (their low words are both zero) where choosing between 0.0 and pg[j] takes two.
template <int BT, bool FAST>
__device__ __forceinline__ double sinr_db_px(const KParams &p, const HotConst &H, const LeanCoef &C, const double pg[BT], int x, double px) {
    const int B = uav_count<BT, FAST>(p.B);
    double interf = 0.0;
#pragma unroll
    for (int j = 0; j < BT; ++j) {
        const double m = (j != x && j < B) ? 1.0 : 0.0;
        interf = fma(pg[j], m, interf);
    }
    return H.db_per_ln * lm_logc(lm_div(px, H.noise + interf), C);   // 10*log10(x) = (10/ln 10) * ln x; both operands normal
}
template <int BT, bool FAST>
__device__ __forceinline__ double sinr_db(const KParams &p, const HotConst &H, const LeanCoef &C, const double pg[BT], int x) {
    double px = 0.0;
#pragma unroll
    for (int j = 0; j < BT; ++j) px = (j == x) ? pg[j] : px;
    return sinr_db_px<BT, FAST>(p, H, C, pg, x, px);
}

// bestBS_buf push + handover decision for one UE (channel.py:148-167).  r0..r2 = FIFO rows, oldest first.
__device__ __forceinline__ void fifo_handover(const HotConst &H, int depth, int best, double bestS, double cur,
                                              int &serving, int &r0, int &r1, int &r2) {
    bool remain;
    if (depth == 1) { r1 = best; remain = (r1 == r0); }                       // append (:148-149)
    else if (depth == 2) { r2 = best; remain = (r1 == r0) && (r2 == r0); }
    else { r0 = r1; r1 = r2; r2 = best; remain = (r1 == r0) && (r2 == r0); }  // FIFO shift (:150-153)
    const bool changed = serving != best;                                     // :156 (newest row == best)
    if (remain && changed && (bestS - cur > H.ho_thr)) serving = best;        // :155-167
}

// Per-env scalars and outputs after a step / reset: reward (mobile_env.py:163-189), done (:186-187).
// `rec` is the env's record as loaded at kernel entry: fields a mode does not own keep their value (the aggregation counters in
// the trace modes, depth and step count during warm-up), and the whole record goes back with one 32-byte store.
// `o`: where this step's outputs go (p.out, or the current step's block of a multi-step launch).  REC = false: outputs only
// (steps 0 .. T-2 of a multi-step launch; the record is stored once, after the last step).
template <int MODE, bool FAST, bool REC = true, bool OUTS = true, bool COH = false>
__device__ __forceinline__ void env_finish(const KParams &p, const OutPtrs &o, const StatePtrs &st, uint32_t e, EnvRec rec, uint32_t tick,
                                           int agg, int deagg, int depth, int step_n, double sum_cur, int n_outage) {
    rec.tick = tick;
    if (has_mobility(MODE)) { rec.agg = agg; rec.deagg = deagg; }
    if (is_reset(MODE)) {
        rec.fifo_depth = 1;                                    // bestBS_buf = [current_BS] (channel.py:115)
        rec.step_n = 0;                                        // mobile_env.py:146
        const double mean = sum_cur * p.inv_U;
        if (OUTS && UAV_OUT(o.step_n)) stx(o.step_n, e, 0);
        if (OUTS && UAV_OUT(o.reward)) stx(o.reward, e, 0.f);
        if (OUTS && UAV_OUT64(o.reward_f64)) stx(o.reward_f64, e, 0.0);
        if (OUTS && UAV_OUT(o.done)) stx(o.done, e, 0);
        if (OUTS && UAV_OUT(o.n_out)) stx(o.n_out, e, 0);
        if (OUTS && UAV_OUT(o.mean_sinr)) stx(o.mean_sinr, e, (float)mean);
        if (OUTS && UAV_OUT64(o.mean_sinr_f64)) stx(o.mean_sinr_f64, e, mean);
    }
    if (is_step(MODE)) {
        rec.fifo_depth = depth < 3 ? depth + 1 : depth;
        const double mean = sum_cur * p.inv_U;                // channel.py:216 (np.mean; <= 1 ulp from sum/U)
        const double r0 = sum_cur * p.inv_U20;                // mobile_env.py:165  mean / 20
        const double r1 = -((double)n_outage * p.inv_U);      // mobile_env.py:167  -1.0 * nOut / nUE
        double reward = (0.0 + r0) + r1;                      // sum(r_dissect)
        if (-1.0 > reward) reward = -1.0;                     // max(.., -1)  mobile_env.py:189
        step_n += 1;                                          // mobile_env.py:181
        rec.step_n = step_n;
        if (OUTS && UAV_OUT(o.step_n)) stx(o.step_n, e, step_n);
        if (OUTS && UAV_OUT(o.done)) stx(o.done, e, (uint8_t)(step_n >= p.max_step));
        if (OUTS && UAV_OUT(o.reward)) stx(o.reward, e, (float)reward);
        if (OUTS && UAV_OUT64(o.reward_f64)) stx(o.reward_f64, e, reward);
        if (OUTS && UAV_OUT(o.mean_sinr)) stx(o.mean_sinr, e, (float)mean);
        if (OUTS && UAV_OUT64(o.mean_sinr_f64)) stx(o.mean_sinr_f64, e, mean);
        if (OUTS && UAV_OUT(o.n_out)) stx(o.n_out, e, n_outage);
    }
    if (REC) stx_c<COH>(st.env, e, rec);
}

// Multi-step launches (uavenv_step_many): every output array holds one block per step, [T][...]; the pointers move on by one
// block after each step (uniform 64-bit adds on the scalar unit).  A null (skipped) output stays null.
template <bool FAST>
__device__ __forceinline__ void out_next_step(OutPtrs &o, long long N, int U, int B) {
    const long long nu = N * U, nb2 = N * B * 2;
    if (UAV_OUT(o.reward)) o.reward += N;
    if (UAV_OUT(o.done)) o.done += N;
    if (UAV_OUT(o.mean_sinr)) o.mean_sinr += N;
    if (UAV_OUT(o.n_out)) o.n_out += N;
    if (UAV_OUT(o.ue_xy)) o.ue_xy += 2 * nu;
    if (UAV_OUT(o.bs_xy)) o.bs_xy += nb2;
    if (UAV_OUT(o.serving)) o.serving += nu;
    if (UAV_OUT(o.cur_sinr)) o.cur_sinr += nu;
    if (UAV_OUT(o.step_n)) o.step_n += N;
    if (UAV_OUT64(o.cur_sinr_f64)) o.cur_sinr_f64 += nu;
    if (UAV_OUT64(o.mean_sinr_f64)) o.mean_sinr_f64 += N;
    if (UAV_OUT64(o.reward_f64)) o.reward_f64 += N;
}

// A segment of a rotation schedule starts at step t0: move every output pointer on by t0 blocks (uniform, once per segment).
template <bool FAST>
__device__ __forceinline__ void out_skip_steps(OutPtrs &o, long long t0, long long N, int U, int B) {
    const long long n = t0 * N, nu = n * U, nb2 = n * B * 2;
    if (UAV_OUT(o.reward)) o.reward += n;
    if (UAV_OUT(o.done)) o.done += n;
    if (UAV_OUT(o.mean_sinr)) o.mean_sinr += n;
    if (UAV_OUT(o.n_out)) o.n_out += n;
    if (UAV_OUT(o.ue_xy)) o.ue_xy += 2 * nu;
    if (UAV_OUT(o.bs_xy)) o.bs_xy += nb2;
    if (UAV_OUT(o.serving)) o.serving += nu;
    if (UAV_OUT(o.cur_sinr)) o.cur_sinr += nu;
    if (UAV_OUT(o.step_n)) o.step_n += n;
    if (UAV_OUT64(o.cur_sinr_f64)) o.cur_sinr_f64 += nu;
    if (UAV_OUT64(o.mean_sinr_f64)) o.mean_sinr_f64 += n;
    if (UAV_OUT64(o.reward_f64)) o.reward_f64 += n;
}

// ================================================================================================
// state construction: ue_mobility.py:433-451 + mobile_env.py:58-59.  `per` threads per env.
// ================================================================================================
static __global__ __launch_bounds__(256) void init_kernel(InitParams p) {   // (static: the header is included by two translation units)
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int U = p.U, Gr = p.Gr, per = p.per;  // per = max(U, Gr, B, W64) threads per env
    if (tid >= p.N * per) return;
    const long long e = tid / per;
    const int u = (int)(tid - e * per);
    const double MAXC = (double)p.G;
    const double two_pi = 2.0 * 3.141592653589793;
    if (u < U) {
        const long long t = e * U + u;
        double ux, uy, ut;
        if (p.u_x) { ux = p.u_x[t]; uy = p.u_y[t]; ut = p.u_th[t]; }
        else {
            U4 r = philox4x32_10(p.env_id_base + (uint32_t)e, 0xFFFFFFFFu, (uint32_t)u, DOM_INIT_UE_A, p.key0, p.key1);
            ux = u53(r.x, r.y); uy = u53(r.z, r.w);
            r = philox4x32_10(p.env_id_base + (uint32_t)e, 0xFFFFFFFFu, (uint32_t)u, DOM_INIT_UE_B, p.key0, p.key1);
            ut = u53(r.x, r.y);
        }
        p.ue_pos[t] = UePos{ux * MAXC, uy * MAXC};         // :434-435
        p.ue_aux[t] = UeAux{ut, 0, 0, 0, 0, 0, 0};         // :437 (cos/sin are taken when the heading is used, :455)
    }
    if (u < Gr) {
        const int g = u;
        double v[5];
        if (p.u_g) { for (int k = 0; k < 5; ++k) v[k] = p.u_g[(e * 5 + k) * Gr + g]; }
        else {
            U4 r = philox4x32_10(p.env_id_base + (uint32_t)e, 0xFFFFFFFFu, (uint32_t)g, DOM_INIT_G_A, p.key0, p.key1);
            v[0] = u53(r.x, r.y); v[1] = u53(r.z, r.w);
            r = philox4x32_10(p.env_id_base + (uint32_t)e, 0xFFFFFFFFu, (uint32_t)g, DOM_INIT_G_B, p.key0, p.key1);
            v[2] = u53(r.x, r.y); v[3] = u53(r.z, r.w);
            r = philox4x32_10(p.env_id_base + (uint32_t)e, 0xFFFFFFFFu, (uint32_t)g, DOM_INIT_G_C, p.key0, p.key1);
            v[4] = u53(r.x, r.y);
        }
        const double th = v[4] * two_pi;   // :446
        // :442 x, :443 y (MAX_X, sic), :444 flight length, :445 speed
        p.grp[e * Gr + g] = GrpRec{v[0] * MAXC, v[1] * MAXC, v[2] * MAXC, v[3] * (p.grp_v_max - p.grp_v_min) + p.grp_v_min, cos(th), sin(th)};
    }
    if (u < p.B) {
        p.bs_xy[(e * p.B + u) * 2] = p.bs_init[2 * u];
        p.bs_xy[(e * p.B + u) * 2 + 1] = p.bs_init[2 * u + 1];
    }
    if (u < p.W64) p.out_bits[e * p.W64 + u] = 0ull;
    if (u == 0) p.env[e] = EnvRec{0u, p.agg_init, p.deagg_len, 0, 0, 0, 0, 0};
}

// ================================================================================================
// Packed env kernel: U <= 64, EPW = p.epw env instances per wavefront (slots of U lanes), one pass.
// Host guarantees U >= max(B, Gr) (owner lanes live inside the slot) and EPW*U <= 64.
// BT: compile-time bound on B.  PLC: pl_b == 30 (channel.py:47) => d^-3 by sqrt.
// ================================================================================================
// MANY (MODE_STEP only): p.n_ticks whole steps in ONE launch -- uavenv_step_many.  State is loaded once, lives in registers
// across the steps and is stored once; step t reads actions[t][N] (the next step's action is prefetched while this step
// computes) and writes block t of every output array.  Exactly the arithmetic of p.n_ticks single-step launches, so results
// are bit-identical (tests/test_step_many_gpu.py); what disappears is the per-step launch, kernarg fetch, state load round
// trip and state store, i.e. the fixed ~5.8 us a 4096-env launch spends outside its arithmetic.
// The kernel proper: `ew` = the env-wavefront this wavefront hosts (envs ew*EPW .. ew*EPW+EPW-1), `t0` / `nt` = first step and
// number of steps (multi-step launches; a plain launch runs env-wavefront = hardware wavefront and all of p.n_ticks).
// HO (pieces of a one-launch schedule): bit 0 = this piece continues a job another wavefront started (state LOADED coherently),
// bit 1 = another wavefront continues this piece's job (state STORED coherently); 0 everywhere else.  Bit 2 (the gated rollout kernel,
// env_kernel_gated): the ACTIONS are loaded coherently too -- another kernel wrote them while this one was running.
// `po` (single-step launches): where the outputs go instead of p.out (the gated rollout kernel moves the reward pointer on per step).
template <int BT, int MODE, bool PLC, bool FAST, bool PIN, bool MANY, int HO = 0>
__device__ __forceinline__ void env_packed_body(char *blob, const long long *actions, const int8_t *gid_of_u, long long N, int U, int EPW,
                                                int Gr, int B_rt, int lane_magic, const KParams &p, int (*s_bs)[kMaxEpw][2 * kMaxBs],
                                                const int wave, const long long ew, const int t0, const int nt, const int e_lo, const int e_hi,
                                                const OutPtrs *po = nullptr) {
    constexpr bool LDC = (HO & 1) != 0, STC = (HO & 2) != 0, ACC = (HO & 4) != 0;
    const OutPtrs &pout = po != nullptr ? *po : p.out;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
    (void)ts0; (void)ts1; (void)ts2; (void)ts3; (void)ts4; (void)ts5; (void)ts6;
    UAV_STAMP(ts0);                                   // wave start
    const int lane = threadIdx.x & 63;
    const int B = uav_count<BT, FAST>(B_rt);
    const StatePtrs st = state_from_blob(blob, N, U, B, Gr);
    const int slot = (int)lane_div((uint32_t)lane, (uint32_t)lane_magic);   // lane / U (intdiv.h); lanes >= EPW*U are not live
    const int base = slot * U;   // first lane of my slot
    const int ul = lane - base;  // walker index inside the env (also: group / UAV index for owner lanes)
    long long e = ew * EPW + slot;
    // envs [e_lo, e_hi) of the batch: the whole batch (0, N), or a range of uavenv_step_range -- which need not start or end on a wavefront
    // boundary: the wavefront that straddles a boundary runs in both launches, each with its own envs live
    bool live = (lane < EPW * U) && (e >= e_lo) && (e < e_hi);
    if (is_reset(MODE)) { if (p.mask != nullptr) live = live && (p.mask[live ? e : 0] != 0); }
    if (__ballot(live) == 0ull) return;
    if (!live) e = 0;            // keep addresses in range; every store below is guarded by `live`
    const unsigned long long slot_mask = ((U >= 64) ? ~0ull : ((1ull << U) - 1ull)) << base;
    const int n_ticks = MANY ? nt : ((MODE == MODE_WARMUP) ? p.n_ticks : 1);   // iterations of the tick / step loop below
    if (MANY) actions += (long long)t0 * N;          // row of this segment's first step
    int *bs_row = s_bs[wave][slot];
    const int u = ul;
    const long long iu = e * U + (live ? u : 0);
    // 32-bit element indices for ldx()/stx(): env, walker [N,U], group [N,Gr], UAV [N,B]
    const uint32_t e32 = (uint32_t)e, iu32 = (uint32_t)iu;
    const uint32_t ig32 = e32 * (uint32_t)Gr + (uint32_t)ul, ib32 = e32 * (uint32_t)B + (uint32_t)ul;
    const bool head = live && (ul == 0);                             // writes the per-env scalars
    const bool bown = (MODE != MODE_WARMUP) && live && (ul < B);     // this lane owns UAV `ul`
    const bool gown = (has_mobility(MODE)) && live && (ul < Gr);     // this lane owns RPGM group `ul`

    UAV_STAMP(ts1);                                   // first kernarg words arrived, lane bookkeeping done
    // ================= load phase: every global read of the launch, issued before any dependent work ======
    constexpr bool REG_MOVE = (BT <= 8);   // serial BS_move in registers; n_act^B <= 9^8 always fits 32 bits here
    int bx = 0, by = 0;                    // the UAV this lane OWNS (store phase)
    int bsx[BT], bsy[BT];                  // all UAV cells of this lane's env (rx_power reads them)
    long long act = 0, apw = 1;
#pragma unroll
    for (int b = 0; b < BT; ++b) { bsx[b] = 0; bsy[b] = 0; }
    if (MODE != MODE_WARMUP) {
        if (REG_MOVE) {
            const int2 *cells = reinterpret_cast<const int2 *>(is_reset(MODE) ? p.bs_init : st.bs_xy);
            const uint32_t c0 = is_reset(MODE) ? 0u : e32 * (uint32_t)B;                 // mobile_env.py:119 on reset
#pragma unroll
            for (int b = 0; b < BT; ++b)
                if (b < B) { const int2 q = is_reset(MODE) ? ldx(cells, c0 + (uint32_t)b) : ldx_c<LDC>(cells, c0 + (uint32_t)b); bsx[b] = q.x; bsy[b] = q.y; }
            if (is_step(MODE)) act = ldx_c<ACC>(actions, e32);
        } else if (bown) {
            if (is_reset(MODE)) { bx = p.bs_init[2 * ul]; by = p.bs_init[2 * ul + 1]; }
            else { const int2 q = ldx_c<LDC>(reinterpret_cast<const int2 *>(st.bs_xy), ib32); bx = q.x; by = q.y; }
            if (is_step(MODE)) { act = ldx_c<ACC>(actions, e32); apw = p.act_pow[ul]; }
        }
    }
    const EnvRec erec = ldx_c<LDC>(st.env, e32);                      // tick, phase counters, FIFO depth, step count: one record
    uint32_t tick = erec.tick;
    int agg = erec.agg, deagg = erec.deagg;
    int depth = erec.fifo_depth;      // (advances only between the steps of a multi-step launch)
    int step_n = erec.step_n;
    double ogx = 0, ogy = 0, ogfl = 0, ogv = 0, ogc = 0, ogs = 0;
    if (gown) {
        const GrpRec g = ldx_c<LDC>(st.grp, ig32);
        ogx = g.x; ogy = g.y; ogfl = g.fl; ogv = g.v; ogc = g.c; ogs = g.s;
    }
    double x = 0, y = 0, hu = 0, hu_inj = 0;
    int ix = 0, iy = 0, gid = 0;
    int serving = 0, r0 = 0, r1 = 0, r2 = 0;
    if (live) {                                                   // heading, integer cell, serving UAV and FIFO rows: one record
        const UeAux a = ldx_c<LDC>(st.ue_aux, iu32);
        hu = a.hu; ix = a.ix; iy = a.iy; serving = a.serving; r0 = a.r0; r1 = a.r1; r2 = a.r2;
    }
    if (has_mobility(MODE)) {
        gid = ldx(gid_of_u, (uint32_t)u);                          // table padded to >= 64 entries: dead lanes have u < 64
        if (live) {
            const UePos q = ldx_c<LDC>(st.ue_pos, iu32);
            x = q.x; y = q.y;
            if (UAV_INJ(p.inj_theta)) hu_inj = p.inj_theta[iu];   // injected draws cover exactly one tick
        }
    } else if (live) {                                            // mobile_env.py:202-203 (read_trace)
        ix = p.trace_xy[2 * iu]; iy = p.trace_xy[2 * iu + 1];
    }
    unsigned long long prev_out = 0ull;
    if (is_step(MODE)) prev_out = ldx_c<LDC>(st.out_bits, e32);           // one 64-bit word per env here (U <= 64)

    // Only now touch the parameter struct: its (cold) kernarg fetch overlaps the global loads issued above.
    __builtin_amdgcn_sched_barrier(0);
    kernarg_warm<(int)sizeof(KParams)>();
    const LeanCoef C = lm_make_coef<PIN>();   // polynomial coefficients, pinned in VGPRs once per kernel (lean_math.h)
    const HotConst H = make_hot<PIN>(p);      // hot kernarg doubles, pinned in VGPRs (frees ~30 SGPRs)
    const double MAXC = H.maxc;
    UAV_DRAIN_VM();
    UAV_STAMP(ts2);                                   // every load of the load phase has returned
    // ================= compute ===============================================================================
#ifdef UAVENV_SKELETON   // diagnostic build (tools/ab_variants.sh): load phase + store phase only, to measure the fixed floor
    double sum_cur = 0.0, cur = x + y + hu + ogx + ogy + ogfl + ogv + ogc + ogs + (double)(bx + by + bsx[0] + bsy[0] + bsx[BT - 1] + bsy[BT - 1] + (int)act + (int)apw + gid);
    int n_outage = (int)(prev_out & 1ull) + serving + r0 + r1 + r2 + depth;
    unsigned long long ob = prev_out;
    tick += 1u;
    OutPtrs om = p.out;
    (void)bs_row; (void)slot_mask; (void)n_ticks; (void)MAXC; (void)hu_inj; (void)C; (void)H; (void)om;
#else
    // One loop body serves all launch kinds: warm-up = n_ticks mobility ticks (no UAV move, no channel update); reset / step =
    // exactly one iteration (compile-time trip count, the loop folds away); multi-step (MANY) = n_ticks whole steps, each
    // storing its outputs, with walker / group / UAV state carried in registers.
    U4 q0 = {0u, 0u, 0u, 0u}, q1 = {0u, 0u, 0u, 0u};   // this walker's Philox calls 0 / 1 of the last tick (heading + fading)
    double sum_cur = 0.0, cur = 0.0;
    int n_outage = 0;
    unsigned long long ob = 0ull;
    OutPtrs om = p.out;                                // MANY: the current step's output blocks (dead code otherwise)
    if (MANY) out_skip_steps<FAST>(om, t0, N, U, B);
    for (int it = 0; it < n_ticks; ++it) {
        long long act_next = 0;
        if (MANY) {                                    // prefetch the next step's action: its round trip hides behind this step
            const long long *nxt = actions + (long long)((it + 1 < n_ticks) ? it + 1 : it) * N;
            if (REG_MOVE || bown) act_next = ldx(nxt, e32);
        }
        // ---- UAV move: Decimal_to_Base_N + BS_move (ue_mobility.py:191-271,310-336) ---------------
        if (MODE != MODE_WARMUP) {
            if (REG_MOVE) {
                if (is_step(MODE)) bs_move_serial<BT, FAST>(p, (unsigned)act, bsx, bsy);
#pragma unroll
                for (int b = 0; b < BT; ++b)
                    if (ul == b) { bx = bsx[b]; by = bsy[b]; }             // the cell this lane writes back
            } else {
                // cooperative form for B > 8: one UAV per lane, sequential rounds, UAV cells staged in LDS
                if (is_step(MODE)) {
                    int dig = 0;
                    if (bown) dig = action_digit(p, act, apw, ul);
                    for (int i = 0; i < B; ++i) {  // sequential: UAV i sees the already-moved UAVs j < i
                        const int xi = __shfl(bx, base + i, 64), yi = __shfl(by, base + i, 64), di = __shfl(dig, base + i, 64);
                        int nx, ny;
                        uav_propose(p, xi, yi, di, nx, ny);
                        // collision on the PRE-move cell of i (:256-263); integer form of norm <= min_dist
                        const int dx = xi - bx, dy = yi - by;
                        const bool near = bown && (ul != i) && (dx * dx + dy * dy <= p.min_bs_dist2);
                        const bool collision = (__ballot(near) & slot_mask) != 0ull;
                        if (!collision && bown && ul == i) { bx = nx; by = ny; }
                    }
                }
                if (MANY) __builtin_amdgcn_wave_barrier();             // the previous step's reads of bs_row are done
                if (bown) { bs_row[2 * ul] = bx; bs_row[2 * ul + 1] = by; }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int b = 0; b < BT; ++b)
                    if (b < B) { bsx[b] = bs_row[2 * b]; bsy[b] = bs_row[2 * b + 1]; }
            }
        }

        if (!MANY) UAV_STAMP(ts3);                        // UAV move done
        // ---- mobility: next(self.mm); walker and group state stay in registers across ticks ----
        if (has_mobility(MODE)) {
            const bool aggregating = agg != 0;
            if (gown) {                                          // ue_mobility.py:458-459
                ogx = ogx + ogv * ogc;
                ogy = ogy + ogv * ogs;
            }
            const int src = base + gid;
            const double gx = __shfl(ogx, src, 64), gy = __shfl(ogy, src, 64);
            const double gv = __shfl(ogv, src, 64), gc = __shfl(ogc, src, 64), gs = __shfl(ogs, src, 64);
            bool c[4];
            walker_move(H, C, aggregating, hu, gx, gy, gv, gc, gs, MAXC, x, y, c);   // ue_mobility.py:455-505
            c[0] = c[0] && live; c[1] = c[1] && live; c[2] = c[2] && live; c[3] = c[3] && live;
            uint32_t touched[4] = {0u, 0u, 0u, 0u};  // per slot: groups bounced at x<0, x>MAX, y<0, y>MAX
            if (__ballot(c[0] || c[1] || c[2] || c[3]) != 0ull) {  // rare, wave-uniform branch
                for (int g = 0; g < Gr; ++g) {
                    const bool mine = gid == g;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((__ballot(mine && c[k]) & slot_mask) != 0ull) touched[k] |= 1u << g;
                }
            }
            if (UAV_INJ(p.inj_theta)) hu = hu_inj;                                       // new heading (:508)
            else {
                if (quad_draws(B)) {                                    // B > 8: the heading has its own call (B is the run-time
                                                                        // count in the checked variants: the warm-up kernel has BT = 4)
                    const U4 hq = philox_raw(p, (uint32_t)e, tick, (uint32_t)u, DOM_HEADING);
                    hu = u53(hq.x, hq.y);
                } else {
                    const int HB = (B + 1) >> 1;
                    q0 = philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * HB), DOM_FADING);
                    q1 = (HB >= 2) ? philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * HB + 1), DOM_FADING) : q0;
                    hu = heading_from(q0, q1, HB);
                }
            }
            if (gown) group_finish<FAST>(p, C, e, ul, tick, touched, MAXC, ogfl, ogv, ogc, ogs);   // :493-521
            if (aggregating) { agg -= 1; if (agg == 0) deagg = p.deagg_len; }            // :472-473
            else { deagg -= 1; if (deagg == 0) agg = p.agg_len; }                        // :486-487
        }
        tick += 1u;
        if (MODE == MODE_WARMUP) continue;                // warm-up: mobility only
        if (has_mobility(MODE)) { ix = (int)x; iy = (int)y; }                            // .astype(int), mobile_env.py:154-155

        if (!MANY) UAV_STAMP(ts4);                        // mobility done
        // ---- channel update (one per reset / step; Philox time = the tick just executed) ------------------
        {
            double pg[BT];
            rx_power<BT, PLC, FAST, FAST && has_mobility(MODE)>(p, H, C, e, tick - 1u, u, live, iu, ix, iy, bsx, bsy, q0, q1, pg);
            double best_pg;
            const int best = argmax_pg<BT, FAST>(p, pg, best_pg);
            const double bestS = sinr_db_px<BT, FAST>(p, H, C, pg, best, best_pg);
            if (is_reset(MODE)) {
                // LTEChannel.reset / GetBestDlBS (channel.py:113-124)
                cur = bestS;
                serving = best;
                r0 = best;                                                                   // bestBS_buf = [current_BS]
            } else {
                // UpdateDroneNet, DL part (channel.py:141-174)
                cur = sinr_db<BT, FAST>(p, H, C, pg, serving);       // serving UAV BEFORE any handover (:145-146)
                fifo_handover(H, depth, best, bestS, cur, serving, r0, r1, r2);
            }
            ob = (__ballot(live && (cur <= H.out_thr)) & slot_mask) >> base;               // :116 / :170
            if (!is_reset(MODE)) n_outage = __popcll(ob & ~prev_out);                      // :171-174 newly outaged
            sum_cur = slot_sum(live ? cur : 0.0, ul, U);
        }
        if (MANY) {
            // ---- this step's outputs (block `it` of every output array), then the hand-over to the next step ------------
            if (live) {
                if (UAV_OUT(om.ue_xy)) { stx(om.ue_xy, 2u * iu32, (int16_t)ix); stx(om.ue_xy, 2u * iu32 + 1u, (int16_t)iy); }
                if (UAV_OUT(om.serving)) stx(om.serving, iu32, (int8_t)serving);
                if (UAV_OUT(om.cur_sinr)) stx(om.cur_sinr, iu32, (float)cur);
                if (UAV_OUT64(om.cur_sinr_f64)) stx(om.cur_sinr_f64, iu32, cur);
            }
            if (bown) { if (UAV_OUT(om.bs_xy)) { stx(om.bs_xy, 2u * ib32, bx); stx(om.bs_xy, 2u * ib32 + 1u, by); } }
            if (it + 1 < n_ticks) {
                if (head) env_finish<MODE, FAST, false>(p, om, st, e32, erec, tick, agg, deagg, depth, step_n, sum_cur, n_outage);
                out_next_step<FAST>(om, N, U, B);
                prev_out = ob;                                    // channel.py:173  self.ue_out = new_out
                depth = depth < 3 ? depth + 1 : depth;            // bestBS_buf grows to hoBufDepth, then shifts (:148-153)
                step_n += 1;                                      // mobile_env.py:181
                act = act_next;
            }
        }
    }
    if (MODE == MODE_WARMUP) { ix = (int)x; iy = (int)y; }                               // cells after the last warm-up tick
#endif  // UAVENV_SKELETON
    UAV_STAMP(ts5);                                   // channel update done

    // ================= store phase: state, then outputs ==========================================================
    if (live) {
        const uint32_t iw = block_local<!PIN>(iu32);
        if (has_mobility(MODE)) stx_c<STC>(st.ue_pos, iw, UePos{x, y});
        // warm-up leaves serving and the FIFO rows as loaded; a reset overwrites serving and row 0 only (depth becomes 1)
        stx_c<STC>(st.ue_aux, iw, UeAux{hu, (int16_t)ix, (int16_t)iy, (int8_t)serving, (int8_t)r0, (int8_t)r1, (int8_t)r2});
        if (MODE != MODE_WARMUP && !MANY) {
            if (UAV_OUT(pout.ue_xy)) { stx(pout.ue_xy, 2u * iw, (int16_t)ix); stx(pout.ue_xy, 2u * iw + 1u, (int16_t)iy); }
            if (UAV_OUT(pout.serving)) stx(pout.serving, iw, (int8_t)serving);
            if (UAV_OUT(pout.cur_sinr)) stx(pout.cur_sinr, iw, (float)cur);
            if (UAV_OUT64(pout.cur_sinr_f64)) stx(pout.cur_sinr_f64, iw, cur);
        }
    }
    if (gown) {
        const uint32_t gw = block_local<!PIN>(ig32);
        stx_c<STC>(st.grp, gw, GrpRec{ogx, ogy, ogfl, ogv, ogc, ogs});
    }
    if (bown) {
        const uint32_t bw = block_local<!PIN>(ib32);
        stx_c<STC>(st.bs_xy, 2u * bw, bx); stx_c<STC>(st.bs_xy, 2u * bw + 1u, by);
        if (!MANY) { if (UAV_OUT(pout.bs_xy)) { stx(pout.bs_xy, 2u * bw, bx); stx(pout.bs_xy, 2u * bw + 1u, by); } }
    }
    if (head) {
        const uint32_t ew = block_local<!PIN>(e32);
        if (MODE != MODE_WARMUP) stx_c<STC>(st.out_bits, ew, ob);                                // :116 / :173
        // (MANY: outputs of the LAST step + the record; depth / step_n are the values that step started from)
        env_finish<MODE, FAST, true, true, STC>(p, MANY ? om : pout, st, ew, erec, tick, agg, deagg, depth, step_n, sum_cur, n_outage);
    }
#ifdef UAVENV_STAMPS
    UAV_STAMP(ts6);                                   // all stores issued (not yet acknowledged)
    UAV_DRAIN_VM();
    unsigned long long ts7 = 0;
    UAV_STAMP(ts7);                                   // all stores acknowledged
    if (p.dbg != nullptr && lane == 0) {
        unsigned long long *d = p.dbg + ((long long)blockIdx.x * kWavesPerBlock + wave) * 8;
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3; d[4] = ts4; d[5] = ts5; d[6] = ts6; d[7] = ts7;
    }
#endif
}

// ---- hand-off between the two wavefronts that share a split job of a one-launch schedule -----------------------------------------
// The state of the job crosses in coherent accesses (ldx_c / stx_c above: the publishing piece stores it write-through, the waiting
// piece loads it past its L1), so the hand-off itself is: producer -- its stores have left (s_waitcnt vmcnt(0)), then the flag;
// consumer -- ONE lane polls the flag with relaxed agent-scope loads, bounded; the state loads are issued after the poll has seen it.
// No agent-scope release / acquire fence (an L2 write-back and an L1 invalidate: 6 us of a 91-us 20-step call,
// profiles/r04l_fence_cost.json).  The flag is cleared by its consumer, so a captured launch replays correctly and no per-call epoch
// is needed.
__device__ __forceinline__ void sched_hand_off_publish(const KParams &p, int ew) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(p.sched_flag + ew, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// false = the flag did not arrive within the budget: the error word is set and the caller must leave the kernel.
__device__ __forceinline__ bool sched_hand_off_wait(const KParams &p, int ew) {
    uint32_t *f = p.sched_flag + ew;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long budget = (unsigned long long)p.sched_spin_us * 100ull;           // s_memrealtime ticks at 100 MHz
    int seen = 0;
    for (;;) {
        uint32_t v = 0u;
        if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = __builtin_amdgcn_readfirstlane((int)v);
        if (seen != 0) break;
        if (__builtin_amdgcn_s_memrealtime() - t_start > budget) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (seen == 0) {
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(p.sched_err, kDevErrHandoff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
    }
    asm volatile("" ::: "memory");                       // (the state loads below stay below the poll)
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    return true;
}

// SCHED (MANY only): the launch runs a rotation schedule (p.sched).  A kernel of its own, not a branch of the plain multi-step kernel: with
// the schedule's three body copies inside it the plain unpinned kernel went from 167 to 169-180 VGPRs, i.e. from three to two wavefronts
// per SIMD, and a 65 536-env call from 46.7 to 52.5 us per step (same-box A/B against the round-3 tree, profiles/r04s_*).
template <int BT, int MODE, bool PLC, bool FAST, bool PIN, bool MANY = false, bool SCHED = false>
__global__ __launch_bounds__(64 * kWavesPerBlock) void env_kernel_packed(char *blob, const long long *actions,
                                                                              const int8_t *gid_of_u, long long N, int U, int EPW,
                                                                              int Gr, int B_rt, int lane_magic, int wave0, int e_lo, int e_hi, const KParams p) {
    static_assert(!MANY || MODE == MODE_STEP, "multi-step launches exist for MobiEnvironment.step only");
    static_assert(!SCHED || MANY, "rotation schedules exist for multi-step launches only");
    // Leading scalars arrive in SGPRs at wave launch (hipcc -mllvm -amdgpu-kernarg-preload-count=16), so the global
    // loads of the load phase are issued from them at once while the parameter struct `p` (constants, output
    // pointers) is still being fetched: the cold kernarg fetch (~2000 ticks) no longer precedes the load round trip.
    __shared__ int s_bs[kWavesPerBlock][kMaxEpw][2 * kMaxBs];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this wavefront's env-wavefront (uniform).  wave0 > 0: a launch over a RANGE of the batch (uavenv_step_range: envs wave0 * EPW ...)
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + wave + wave0;
    if (!MANY) {
        env_packed_body<BT, MODE, PLC, FAST, PIN, MANY>(blob, actions, gid_of_u, N, U, EPW, Gr, B_rt, lane_magic, p, s_bs, wave, gw, 0, 1, e_lo, e_hi);
    } else {
        // Multi-step launch.  Plain: wavefront w hosts env-wavefront w for all p.n_ticks steps.  Rotation schedule (p.sched,
        // uavenv_capi.hip: rotation_plan): this wavefront is a SLOT that works through up to three pieces, each a run of consecutive
        // steps of one env-wavefront: state loaded, nt steps, state stored.
        // (Inlined copies of the body rather than a rolled loop around one: with the rolled loop hipcc allocated 330 VGPRs for the
        // pinned kernel instead of 233 -- one wavefront per SIMD -- and doubled its SGPR spills.)
        if (!SCHED) {
            env_packed_body<BT, MODE, PLC, FAST, PIN, MANY>(blob, actions, gid_of_u, N, U, EPW, Gr, B_rt, lane_magic, p, s_bs, wave, gw, 0, p.n_ticks, e_lo, e_hi);
            return;
        }
        const int4 *sched = p.sched;
        // One-launch schedule: this wavefront is a SLOT with up to kSchedPieces pieces.  A piece that starts inside a job (SCHED_WAIT)
        // waits until the wavefront that ran the job's first steps has published them; a piece that ends inside a job
        // (SCHED_PUBLISH) publishes.  Publishing pieces come FIRST in their slot and wait for nothing, so every wait ends once its
        // producer has been scheduled; the wait is bounded all the same (sched_hand_off_wait): a bug becomes an error code, not a hang.
        // Column q of a slot's table row: 0 = the piece that publishes (first steps of a split job: coherent stores), 1 = a whole job,
        // 2 = the piece that waits (last steps of a split job: coherent loads); nt = 0 = no such piece in this slot.
        auto piece = [&](auto q_c) {
            constexpr int Q = decltype(q_c)::value;
            const int4 d = sched[gw * kSchedPieces + Q];                      // uniform address
            const int ew = __builtin_amdgcn_readfirstlane(d.x), t0 = __builtin_amdgcn_readfirstlane(d.y);
            const int nt = __builtin_amdgcn_readfirstlane(d.z), bits = __builtin_amdgcn_readfirstlane(d.w);
            if (nt <= 0) return true;
            if (Q > 0) __builtin_amdgcn_wave_barrier();                       // (the previous piece's reads of the LDS row are done)
            if (Q == 2) { if (!sched_hand_off_wait(p, ew)) return false; }
            env_packed_body<BT, MODE, PLC, FAST, PIN, MANY, (Q == 0 ? 2 : (Q == 2 ? 1 : 0))>(blob, actions, gid_of_u, N, U, EPW, Gr, B_rt, lane_magic, p, s_bs,
                                                                                                  wave, ew, t0, nt, e_lo, e_hi);
            if (Q == 0 && (bits & SCHED_PUBLISH)) sched_hand_off_publish(p, ew);
            return true;
        };
        if (!piece(std::integral_constant<int, 0>{})) return;
        if (!piece(std::integral_constant<int, 1>{})) return;
        if (!piece(std::integral_constant<int, 2>{})) return;
    }
}

// ================================================================================================
// Multi-pass env kernel: U > 64 (or a slot too narrow for its owner lanes): one env per wavefront, walkers
// in passes of 64 lanes.  Same helpers, wave-uniform env scalars.  BASELINE config 5 (16 UAV x 200 UE) runs here.
//
// What round 2 changed, each from the r02a profile of <16, 2, true> at 8192 envs (214 us, 12 053 VALU instructions per wavefront,
// 204 VGPRs, VALU-issue bound: profiles/r02a_pmc_and_trace_digest.txt):
//   FAST        no injected draws, all nine outputs, B == BT: the per-UAV `b < B` guards and pointer tests fold away;
//   PRE         the two Philox calls that give a walker its new heading ARE its fading calls 0 and 1 (same counters):
//               made once instead of twice (the packed kernel has always done this);
//   UAV cells   read from their LDS row (one broadcast ds_read_b64 per UAV) when BT >= 16 instead of 2 x BT registers;
//   item tail   the last U mod 64 walkers (8 of 200) used to cost a whole pass of HB = B/2 fading iterations at 12 % lane
//               use; they now take ONE iteration with a lane per (walker, UAV pair) and 3-step butterfly reductions for
//               the argmax / interference sums (tail_items(): when HB is a power of two and (U mod 64) * HB <= 64);
//   loads       env record, outage words and group ids are fetched before the first pass, not inside it.
// ================================================================================================
// Two N(mean, sd) shadowing draws from one Philox call (Box-Muller), exactly as rx_power() forms them.
__device__ __forceinline__ void fading_pair(const HotConst &H, const LeanCoef &C, const U4 &q, double &f0, double &f1) {
    const double u0 = u53(q.x, q.y);
    const double t = -2.0 * lm_logc(1.0 - u0, C);
    const double r = (t > 0.0) ? t * lm_rsqrt(t) : 0.0;
    double sa, ca;
    lm_sincospi((double)q.z * (1.0 / 2147483648.0), C, &sa, &ca);
    f0 = H.sh_mean + H.sh_sd * (r * ca);
    f1 = H.sh_mean + H.sh_sd * (r * sa);
}
// Quad mode: the same Box-Muller pair from two 32-bit words (radius uniform, angle fraction).
__device__ __forceinline__ void fading_pair32(const HotConst &H, const LeanCoef &C, uint32_t wr, uint32_t wa, double &f0, double &f1) {
    const double u0 = (double)wr * (1.0 / 4294967296.0);
    const double t = -2.0 * lm_logc(1.0 - u0, C);        // 1 - u0 in [2^-32, 1]
    const double r = (t > 0.0) ? t * lm_rsqrt(t) : 0.0;
    double sa, ca;
    lm_sincospi((double)wa * (1.0 / 2147483648.0), C, &sa, &ca);
    f0 = H.sh_mean + H.sh_sd * (r * ca);
    f1 = H.sh_mean + H.sh_sd * (r * sa);
}
// Received power of one UAV at one walker: the arithmetic of rx_power()'s inner block (channel.py:220-257).
// (xs, ys) = the walker's cell in metres, (bxs, bys) = the UAV's: coordinate * gridWidth first, then the difference, exactly as
// GetDistance does (channel.py:220-226); the UAV's pair comes pre-scaled out of LDS (one cvt + one multiply per UAV and step instead of
// per (walker, UAV) pair: 4 of the ~300 instructions of a fading iteration)
template <bool PLC>
__device__ __forceinline__ double rx_gain(const HotConst &H, const LeanCoef &C, double xs, double ys, double bxs, double bys, double f) {
    const double fx = xs - bxs;
    const double fy = ys - bys;
    const double d2 = fx * fx + fy * fy;
    double g;
    if (PLC) {
        const double rinv = lm_rsqrt(d2);
        g = H.k_pl * lm_exp2(H.c_exp * f, C) * (rinv * rinv * rinv);
    } else {
        g = H.k_pl * lm_exp2(H.c_exp * f - H.pl_exp_ln * lm_logc(d2, C), C);
    }
    if (!(d2 > H.pl_dis2)) g = H.k_0 * lm_exp2(H.c_exp * f, C);
    return g;
}
// Sum / first-maximum over the G consecutive lanes of an aligned group (G a power of two <= 64): xor butterflies, every lane
// of the group ends with the same value.
__device__ __forceinline__ double group_sum(double v, int G) {
    for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ void group_argmax(double &v, int &i, int G) {     // ties: the LOWER index wins (np.argmax)
    for (int off = 1; off < G; off <<= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}
// Item layout for the tail walkers?  -> lanes per walker (HB) or 0.
__device__ __forceinline__ int tail_items(int U, int B) {
    const int R = U & 63, HB = (B + 1) >> 1;
    return (R > 0 && (HB & (HB - 1)) == 0 && HB >= 2 && R * HB <= 64) ? HB : 0;
}

#ifndef UAVENV_MP_VGPR_PTRS
#define UAVENV_MP_VGPR_PTRS 1   // build knob (A/B runs), see env_kernel_multipass
#endif
#ifndef UAVENV_MP_SGPR_COEF
#define UAVENV_MP_SGPR_COEF 2   // build knob (A/B runs): 0 none, 1 = exp2 coefficients pinned in SGPRs, 2 = + sincospi, 3 = + log
#endif
#ifndef UAVENV_MP_WAVES
#define UAVENV_MP_WAVES 3     // occupancy the register allocator must allow (waves per SIMD): 3 -> <= 168 VGPRs, 4 -> <= 128
#endif
template <int BT, int MODE, bool PLC, bool FAST>
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(UAVENV_MP_WAVES)))
void env_kernel_multipass(const KParams p) {
    const StatePtrs st = state_from_params(p);
    constexpr bool PRE = FAST && has_mobility(MODE);
    __shared__ double s_bs[kWavesPerBlock][2 * kMaxBs];          // UAV cells in metres (cell * gridWidth)
    kernarg_warm<(int)sizeof(KParams)>();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long e = (long long)blockIdx.x * kWavesPerBlock + wave + p.wave0;  // wave-uniform (wave0 > 0: uavenv_step_range)
    if (e >= p.e_end) return;
    if (is_reset(MODE)) { if (p.mask != nullptr && p.mask[e] == 0) return; }
    // Constants stay unpinned (SGPRs / literals).  Measured on one box at 8192 envs of 16 x 200 (profiles/r02c_*): pinning them
    // in VGPRs as the packed kernel's PIN variant does: -12 % instructions but 187 VGPRs = 2 wavefronts per SIMD, 151.5 vs
    // 154.5 us; serving the 38 polynomial coefficients of the fading loop from LDS (one broadcast ds_read per use, -11 % VALU in
    // that loop, 167 VGPRs): 154.0 vs 154.1 us -- the LDS round trips it adds to each wave's dependency chain cost what the
    // moves did; forcing 4 wavefronts per SIMD (128 VGPRs, 28 scratch spills): 152.9 us.  None is worth its complexity.
    // Polynomial coefficients in SGPR PAIRS (lm_pin_sgpr): a float64 literal cannot be an operand, so every use of an unpinned
    // coefficient costs a v_mov_b64 (50 of the 300 instructions of a fading iteration); VGPR pinning costs occupancy (above); a Horner
    // step fma(p, r, c_k) may read one scalar operand, so SGPR-resident coefficients cost nothing per use.  The scalar file only has room
    // once the pass loop's seven pointers live in VGPRs (UAVENV_MP_VGPR_PTRS).  One box, 8192 envs of 16 x 200
    // (profiles/r02c_config5_multipass_v2.txt): none 150.5 us; exp2 146.4; exp2 + pointers 145.0; exp2 + sincospi + pointers 142.3
    // (shipped); + log coefficients 146.8 (the scalar file overflows again).
    LeanCoef C = lm_make_coef<false>();
#if UAVENV_MP_SGPR_COEF >= 1
    for (int k = 0; k < 13; ++k) lm_pin_sgpr(C.e2[k]);     // exp2: evaluated twice per fading iteration
#endif
#if UAVENV_MP_SGPR_COEF >= 2
    for (int k = 0; k < 8; ++k) { lm_pin_sgpr(C.sp[k]); lm_pin_sgpr(C.cp[k]); }
#endif
#if UAVENV_MP_SGPR_COEF >= 3
    for (int k = 0; k < 7; ++k) lm_pin_sgpr(C.lg[k]);
    lm_pin_sgpr(C.ln2_hi); lm_pin_sgpr(C.ln2_lo);
#endif
    const HotConst H = make_hot<false>(p);

    const int U = p.U, B = uav_count<BT, FAST>(p.B), Gr = p.Gr;
    const int HB = (B + 1) >> 1;
    const bool quad = quad_draws(B);                   // B > 8: four UAVs per Philox call, heading from its own call
    const int QB = (B + 3) >> 2;
    const double MAXC = H.maxc;
    const int n_full = U >> 6, R = U & 63;
    const int n_pass = n_full + (R ? 1 : 0);
    const int IT = (MODE == MODE_WARMUP) ? 0 : tail_items(U, B);        // lanes per tail walker in the item layout, or 0
    const int n_ticks = (MODE == MODE_WARMUP) ? p.n_ticks : 1;
    double *bs_row = s_bs[wave];

    // ---- loads that do not depend on the pass --------------------------------------------------------------------
    const EnvRec erec = st.env[e];
    unsigned long long prev_w = 0ull;                                     // lane w < W64 holds outage word w of the last update
    if (is_step(MODE) && lane < p.W64) prev_w = st.out_bits[e * p.W64 + lane];
    const bool gown = lane < Gr;
    double ogx = 0, ogy = 0, ogfl = 0, ogv = 0, ogc = 0, ogs = 0;
    if (has_mobility(MODE) && gown) {
        const GrpRec g = st.grp[e * Gr + lane];
        ogx = g.x; ogy = g.y; ogfl = g.fl; ogv = g.v; ogc = g.c; ogs = g.s;
    }

    // ---- UAV move: Decimal_to_Base_N + BS_move, cooperative (one UAV per lane), cells to LDS -------------------------
    if (MODE != MODE_WARMUP) {
        const bool bown = lane < B;
        int bx = 0, by = 0, dig = 0;
        if (bown) {
            if (is_reset(MODE)) { bx = p.bs_init[2 * lane]; by = p.bs_init[2 * lane + 1]; }
            else { const int2 q = reinterpret_cast<const int2 *>(st.bs_xy)[e * B + lane]; bx = q.x; by = q.y; }
        }
        if (is_step(MODE)) {
            if (bown) dig = action_digit(p, p.actions[e], p.act_pow[lane], lane);
            for (int i = 0; i < B; ++i) {
                const int xi = __shfl(bx, i, 64), yi = __shfl(by, i, 64), di = __shfl(dig, i, 64);
                int nx, ny;
                uav_propose(p, xi, yi, di, nx, ny);
                const int dx = xi - bx, dy = yi - by;
                const bool near = bown && (lane != i) && (dx * dx + dy * dy <= p.min_bs_dist2);
                const bool collision = __ballot(near) != 0ull;
                if (!collision && lane == i) { bx = nx; by = ny; }
            }
        }
        if (bown) {
            reinterpret_cast<int2 *>(st.bs_xy)[e * B + lane] = int2{bx, by};
            bs_row[2 * lane] = (double)bx * H.gw; bs_row[2 * lane + 1] = (double)by * H.gw;
            if (UAV_OUT(p.out.bs_xy)) reinterpret_cast<int2 *>(p.out.bs_xy)[e * B + lane] = int2{bx, by};
        }
        __builtin_amdgcn_wave_barrier();
    }
    int agg = erec.agg, deagg = erec.deagg;
    uint32_t tick = erec.tick;
    const int depth = erec.fifo_depth, step_n = erec.step_n;
    double sum_cur = 0.0;
    int n_outage = 0;
#if UAVENV_MP_VGPR_PTRS
    // The pointers the pass loop uses, as (uniform) VGPR pairs: 14 SGPRs fewer live across the fading loop, where the scalar
    // file is wanted for polynomial coefficients (UAVENV_MP_SGPR_COEF); an access then forms its address with one v_lshl_add_u64.
    UeAux *const pv_aux = vgpr_ptr(st.ue_aux);
    UePos *const pv_pos = vgpr_ptr(st.ue_pos);
    unsigned long long *const pv_bits = vgpr_ptr(st.out_bits);
    const int8_t *const pv_gid = vgpr_ptr(p.gid_of_u);
    int16_t *const pv_oxy = vgpr_ptr(p.out.ue_xy);
    int8_t *const pv_osrv = vgpr_ptr(p.out.serving);
    float *const pv_osinr = vgpr_ptr(p.out.cur_sinr);
#else
    UeAux *const pv_aux = st.ue_aux;
    UePos *const pv_pos = st.ue_pos;
    unsigned long long *const pv_bits = st.out_bits;
    const int8_t *const pv_gid = p.gid_of_u;
    int16_t *const pv_oxy = p.out.ue_xy;
    int8_t *const pv_osrv = p.out.serving;
    float *const pv_osinr = p.out.cur_sinr;
#endif

    for (int it = 0; it < n_ticks; ++it) {
        const bool aggregating = agg != 0;
        if (has_mobility(MODE) && gown) {                                 // ue_mobility.py:458-459
            ogx = ogx + ogv * ogc;
            ogy = ogy + ogv * ogs;
        }
        uint32_t touched[4] = {0u, 0u, 0u, 0u};

        for (int pass = 0; pass < n_pass; ++pass) {
            // Lane -> walker.  Full passes and the plain tail: one walker per lane, all HB fading pairs in that lane.
            // Item tail: IT lanes per walker, lane (ul, hb) does fading pair hb of walker 64 * n_full + ul.
            const bool item = (IT != 0) && (pass == n_full);
            const int ul = item ? lane / (IT ? IT : 1) : lane;
            const int hb = item ? lane - ul * IT : 0;
            const int u = pass * 64 + ul;
            const bool act = u < U;
            const bool owner = act && (!item || hb == 0);                  // the lane that stores the walker's results
            const long long iu = e * U + (act ? u : 0);
            int ix = 0, iy = 0;
            UeAux aux = pv_aux[iu];                                    // inactive lanes read walker 0 of the env and store nothing
            U4 h0 = {0u, 0u, 0u, 0u}, h1 = {0u, 0u, 0u, 0u};
            if (has_mobility(MODE)) {
                const int gid = pv_gid[act ? u : 0];                   // table: RPGM group of walker u (ue_mobility.py:417-426)
                const double gx = __shfl(ogx, gid, 64), gy = __shfl(ogy, gid, 64);
                const double gv = __shfl(ogv, gid, 64), gc = __shfl(ogc, gid, 64), gs = __shfl(ogs, gid, 64);
                double x = 0, y = 0, hu = aux.hu;
                if (act) { const UePos q = pv_pos[iu]; x = q.x; y = q.y; }
                bool c[4];
                walker_move(H, C, aggregating, hu, gx, gy, gv, gc, gs, MAXC, x, y, c);
                c[0] = c[0] && act; c[1] = c[1] && act; c[2] = c[2] && act; c[3] = c[3] && act;
                if (__ballot(c[0] || c[1] || c[2] || c[3]) != 0ull) {
                    for (int g = 0; g < Gr; ++g) {
                        const bool mine = gid == g;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (__ballot(mine && c[k]) != 0ull) touched[k] |= 1u << g;
                    }
                }
                if (UAV_INJ(p.inj_theta)) { if (act) hu = p.inj_theta[iu]; }
                else if (quad) {
                    const U4 hq = philox_raw(p, (uint32_t)e, tick, (uint32_t)u, DOM_HEADING);
                    hu = u53(hq.x, hq.y);
                } else {
                    h0 = philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * HB), DOM_FADING);
                    h1 = (HB >= 2) ? philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * HB + 1), DOM_FADING) : h0;
                    hu = heading_from(h0, h1, HB);
                }
                ix = (int)x; iy = (int)y;
                aux.hu = hu;
                if (owner) pv_pos[iu] = UePos{x, y};
            } else if (act) {
                ix = p.trace_xy[2 * iu]; iy = p.trace_xy[2 * iu + 1];
            }
            aux.ix = (int16_t)ix; aux.iy = (int16_t)iy;
            if (MODE == MODE_WARMUP) { if (owner) pv_aux[iu] = aux; continue; }
            if (owner && UAV_OUT(p.out.ue_xy)) reinterpret_cast<int *>(pv_oxy)[iu] = (int)(uint16_t)ix | ((int)(uint16_t)iy << 16);

            // ---- received powers, best UAV, the two SINR values -------------------------------------------------
            const double xs = (double)ix * H.gw, ys = (double)iy * H.gw;       // this walker's cell in metres
            int best, serving = aux.serving;
            double bestS, cur = 0.0;
            if (!item) {
                // A ROLLED loop over the HB fading pairs with running reductions instead of a pg[BT] array: the unrolled form
                // needed 204-206 VGPRs (2 wavefronts per SIMD) and 83-118 SGPR spills at BT = 16.  What SINR needs from the B
                // powers: the first maximum and the sum of the OTHERS in the reference's order -- channel.py:259-268 adds the other UAVs'
                // powers by ascending index -- and for the serving UAV its power and the same sum.  `all` is the plain left-to-right sum
                // of every power so far; while UAV m is the best one, others_b = (sum of the powers before m) + the powers after m, one
                // by one; when a later power displaces m, the sum of everything before it IS `all`.  Nothing is ever subtracted.
                double bp = 0.0, others_b = 0.0, all = 0.0, ps = 0.0, others_s = 0.0;
                best = 0;
                U4 qq = {0u, 0u, 0u, 0u};                                       // quad mode: the current four-UAV call
#pragma unroll 1
                for (int k = 0; k < HB; ++k) {
                    const int b0 = 2 * k, b1 = 2 * k + 1;
                    double f0 = 0.0, f1 = 0.0;
                    if (UAV_INJ(p.inj_fading)) {
                        if (act) { f0 = p.inj_fading[iu * B + b0]; if (b1 < B) f1 = p.inj_fading[iu * B + b1]; }
                    } else if (quad) {
                        if ((k & 1) == 0) qq = philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * QB + (k >> 1)), DOM_FADING);   // (k is wave-uniform)
                        fading_pair32(H, C, (k & 1) ? qq.z : qq.x, (k & 1) ? qq.w : qq.y, f0, f1);
                    } else {
                        U4 q = (k == 0) ? h0 : h1;                              // PRE: this walker's calls 0 / 1 exist already
                        if (!PRE || k >= 2) q = philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * HB + k), DOM_FADING);   // (k is wave-uniform)
                        fading_pair(H, C, q, f0, f1);
                    }
                    const double g0 = rx_gain<PLC>(H, C, xs, ys, bs_row[2 * b0], bs_row[2 * b0 + 1], f0);
                    if (k == 0) { bp = g0; best = 0; all = g0; }
                    else {
                        if (g0 > bp) { others_b = all; bp = g0; best = b0; }
                        else others_b += g0;
                        all += g0;
                    }
                    others_s += (b0 == serving) ? 0.0 : g0;
                    ps = (b0 == serving) ? g0 : ps;
                    if (b1 < B) {
                        const double g1 = rx_gain<PLC>(H, C, xs, ys, bs_row[2 * b1], bs_row[2 * b1 + 1], f1);
                        if (g1 > bp) { others_b = all; bp = g1; best = b1; }
                        else others_b += g1;
                        all += g1;
                        others_s += (b1 == serving) ? 0.0 : g1;
                        ps = (b1 == serving) ? g1 : ps;
                    }
                }
                bestS = H.db_per_ln * lm_logc(lm_div(bp, H.noise + others_b), C);           // channel.py:259-268
                if (!is_reset(MODE)) cur = H.db_per_ln * lm_logc(lm_div(ps, H.noise + others_s), C);
            } else {
                // one fading pair per lane; group = the IT lanes of one walker
                const int b0 = 2 * hb, b1 = 2 * hb + 1;
                double f0 = 0.0, f1 = 0.0;
                if (UAV_INJ(p.inj_fading)) {
                    if (act) { f0 = p.inj_fading[iu * B + b0]; if (b1 < B) f1 = p.inj_fading[iu * B + b1]; }
                } else if (quad) {
                    const U4 q = philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * QB + (hb >> 1)), DOM_FADING);
                    fading_pair32(H, C, (hb & 1) ? q.z : q.x, (hb & 1) ? q.w : q.y, f0, f1);
                } else {
                    const U4 q = (PRE && hb == 0) ? h0 : ((PRE && hb == 1) ? h1 :
                                 philox_raw(p, (uint32_t)e, tick, (uint32_t)(u * HB + hb), DOM_FADING));
                    fading_pair(H, C, q, f0, f1);
                }
                const double g0 = rx_gain<PLC>(H, C, xs, ys, bs_row[2 * b0], bs_row[2 * b0 + 1], f0);
                const double g1 = (b1 < B) ? rx_gain<PLC>(H, C, xs, ys, bs_row[2 * (b1 < B ? b1 : b0)], bs_row[2 * (b1 < B ? b1 : b0) + 1], f1) : 0.0;
                double bp = g0; best = b0;
                if (b1 < B && g1 > g0) { bp = g1; best = b1; }             // first maximum inside the pair, then across the group
                group_argmax(bp, best, IT);
                // interference = the OTHER UAVs, never total - self, added by ascending UAV index like channel.py:259-268: every lane of the
                // group walks the group's IT pairs in order (the sums of the best and of the serving UAV share the shuffles)
                double ib = 0.0, is = 0.0, ps = 0.0;
                const int g_lane0 = lane - hb;
                for (int k = 0; k < IT; ++k) {
                    const double a0 = __shfl(g0, g_lane0 + k, 64), a1 = __shfl(g1, g_lane0 + k, 64);
                    const int j0 = 2 * k, j1 = 2 * k + 1;
                    ib += (j0 == best) ? 0.0 : a0;
                    is += (j0 == serving) ? 0.0 : a0;
                    ps = (j0 == serving) ? a0 : ps;
                    if (j1 < B) {
                        ib += (j1 == best) ? 0.0 : a1;
                        is += (j1 == serving) ? 0.0 : a1;
                        ps = (j1 == serving) ? a1 : ps;
                    }
                }
                bestS = H.db_per_ln * lm_logc(lm_div(bp, H.noise + ib), C);
                if (!is_reset(MODE)) cur = H.db_per_ln * lm_logc(lm_div(ps, H.noise + is), C);
            }

            // ---- handover, outage, stores ------------------------------------------------------------------------
            int r0 = aux.r0, r1 = aux.r1, r2 = aux.r2;
            if (is_reset(MODE)) { cur = bestS; serving = best; r0 = best; }   // LTEChannel.reset / GetBestDlBS (channel.py:113-124)
            else fifo_handover(H, depth, best, bestS, cur, serving, r0, r1, r2);
            unsigned long long ob = __ballot(owner && (cur <= H.out_thr));     // :116 / :170
            if (item) {                                                        // owner lanes sit IT apart: compact to bit = walker
                unsigned long long cb = 0ull;
                for (int k = 0; k < R; ++k) cb |= ((ob >> (k * IT)) & 1ull) << k;
                ob = cb;
            }
            if (is_step(MODE)) {
                const unsigned long long prev = __shfl(prev_w, pass, 64);
                n_outage += __popcll(ob & ~prev);                              // :171-174 newly outaged
            }
            if (lane == 0) pv_bits[e * p.W64 + pass] = ob;
            if (owner) {
                aux.serving = (int8_t)serving; aux.r0 = (int8_t)r0; aux.r1 = (int8_t)r1; aux.r2 = (int8_t)r2;
                pv_aux[iu] = aux;
                if (UAV_OUT(p.out.serving)) pv_osrv[iu] = (int8_t)serving;
                if (UAV_OUT(p.out.cur_sinr)) pv_osinr[iu] = (float)cur;
                if (UAV_OUT64(p.out.cur_sinr_f64)) p.out.cur_sinr_f64[iu] = cur;
            }
            sum_cur += wave_sum(owner ? cur : 0.0);
        }  // passes

        if (has_mobility(MODE)) {
            if (gown) group_finish<FAST>(p, C, e, lane, tick, touched, MAXC, ogfl, ogv, ogc, ogs);
            if (aggregating) { agg -= 1; if (agg == 0) deagg = p.deagg_len; }
            else { deagg -= 1; if (deagg == 0) agg = p.agg_len; }
        }
        tick += 1u;
    }
    if (has_mobility(MODE) && gown) st.grp[e * Gr + lane] = GrpRec{ogx, ogy, ogfl, ogv, ogc, ogs};
    if (lane == 0) env_finish<MODE, FAST>(p, p.out, st, (uint32_t)e, erec, tick, agg, deagg, depth, step_n, sum_cur, n_outage);
}

// ================================================================================================
// env.state planes: GetGridMap (ue_mobility.py:173-188) + GetCurrentAssociationMap (channel.py:387-409).
// ================================================================================================
// Flat index (inside one env's (B+1, G, G) block) of node k: k < B the UAV cells of plane 0, else UE k-B in the plane
// of its serving UAV; -1 when the cell is outside the grid (the reference would raise IndexError).
__device__ __forceinline__ int obs_cell(long long e, int k, int U, int B, int G, const int32_t *bs_xy, const UeAux *ue_aux) {
    int x, y, pl;
    if (k < B) { x = bs_xy[(e * B + k) * 2]; y = bs_xy[(e * B + k) * 2 + 1]; pl = 0; }
    else { const UeAux a = ue_aux[e * U + (k - B)]; x = a.ix; y = a.iy; pl = 1 + a.serving; }
    if (x < 0 || x >= G || y < 0 || y >= G) return -1;
    return (pl * G + x) * G + y;
}

// Full write (after the caller's memset) that also records the written cells, and the in-place update that moves only
// the cells that changed.  Counts are small integers in float32, so +-1.0f is exact and the order of atomics is irrelevant.
template <bool UPDATE>
__global__ __launch_bounds__(256) void obs_cells_kernel(long long N, int U, int B, int G, const int32_t *bs_xy,
                                                        const UeAux *ue_aux, int32_t *prev, float *obs) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = U + B;
    if (t >= N * per) return;
    const long long e = t / per;
    const int k = (int)(t - e * per);
    const long long base = e * (long long)(B + 1) * G * G;
    const int now = obs_cell(e, k, U, B, G, bs_xy, ue_aux);
    if (UPDATE) {
        const int old = prev[t];
        if (old == now) return;
        if (old >= 0) atomicAdd(&obs[base + old], -1.0f);
    }
    if (now >= 0) atomicAdd(&obs[base + now], 1.0f);
    prev[t] = now;
}

// ================================================================================================
// LTEChannel.GetSinrInArea (channel.py:411-433): DL SINR of the NEAREST UAV at every cell (x, y) in [1, G-1]^2 with a
// fresh shadowing draw per (cell, UAV).  One thread per (env, cell); the outputs are zero-filled by the caller, so
// row / column 0 stay 0 like np.zeros((gridX, gridY)).  Same linear-domain arithmetic as rx_power().
// fading_inj: [N, (G-1)^2, B] in the reference's call order per cell (interferers ascending, then the nearest UAV).
// ================================================================================================
template <int BT, bool PLC>
__global__ __launch_bounds__(256) void sinr_area_kernel(const KParams p, const int32_t *bs_xy, const double *fading_inj, float *out32,
                                                        double *out64) {   // bs_xy [N,B,2]: the state's cells or the caller's bsLoc
    const int B = p.B, G = p.G, W = G - 1;
    const long long cells = (long long)W * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.N * cells) return;
    const long long e = t / cells;
    const long long c = t - e * cells;
    const int x = 1 + (int)(c / W), y = 1 + (int)(c % W);            // :416-417 range(xMin, xMax) x range(yMin, yMax)
    const LeanCoef C = lm_make_coef<false>();
    const HotConst H = make_hot<false>(p);
    int d2i[BT];
    int near = 0;                                                   // :418-422 np.argmin(dist): first minimum
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        int v = 0x7FFFFFFF;
        if (b < B) {
            const int dx = x - bs_xy[(e * B + b) * 2], dy = y - bs_xy[(e * B + b) * 2 + 1];
            v = dx * dx + dy * dy;                                   // distance ignores z (GetDistance, :220-226)
        }
        d2i[b] = v;
    }
    int best_d2 = d2i[0];
#pragma unroll
    for (int b = 1; b < BT; ++b)
        if (b < B && d2i[b] < best_d2) { best_d2 = d2i[b]; near = b; }

    double interf = 0.0, own = 0.0;
#pragma unroll
    for (int b2 = 0; b2 < BT; b2 += 2) {
        double f0 = 0.0, f1 = 0.0;
        if (b2 < B) {
            if (fading_inj != nullptr) {
                // position of UAV b in the reference's draw order for this cell: others ascending, `near` last
                const long long base = (e * cells + c) * B;
                const int b0 = b2, b1 = b2 + 1;
                f0 = fading_inj[base + (b0 == near ? B - 1 : (b0 < near ? b0 : b0 - 1))];
                if (b1 < B) f1 = fading_inj[base + (b1 == near ? B - 1 : (b1 < near ? b1 : b1 - 1))];
            } else {
                double u0, u1;
                philox_u2(p, (uint32_t)e, p.env[e].tick, (uint32_t)(c * ((B + 1) >> 1) + (b2 >> 1)), DOM_AREA, u0, u1);
                const double tt = -2.0 * lm_logc(1.0 - u0, C);
                const double r = (tt > 0.0) ? tt * lm_rsqrt(tt) : 0.0;
                double sa, ca;
                lm_sincospi(2.0 * u1, C, &sa, &ca);
                f0 = H.sh_mean + H.sh_sd * (r * ca);
                f1 = H.sh_mean + H.sh_sd * (r * sa);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = b2 + k;
            if (b < BT && b < B) {
                const double f = (k == 0) ? f0 : f1;
                const double d2 = (H.gw * H.gw) * (double)d2i[b];
                double g;
                if (PLC) { const double rinv = lm_rsqrt(d2); g = H.k_pl * lm_exp2(H.c_exp * f, C) * (rinv * rinv * rinv); }
                else g = H.k_pl * lm_exp2(H.c_exp * f - H.pl_exp_ln * lm_logc(d2, C), C);
                if (!(d2 > H.pl_dis2)) g = H.k_0 * lm_exp2(H.c_exp * f, C);
                if (b == near) own = g; else interf += g;            // :423-427 P_interf += P * gain, ascending
            }
        }
    }
    const double s = H.db_per_ln * lm_logc(own / (H.noise + interf), C);   // :429-431
    const long long o = e * (long long)G * G + (long long)x * G + y;
    if (out32) out32[o] = (float)s;
    if (out64) out64[o] = s;
}

}  // namespace uavk
