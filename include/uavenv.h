/*
 * include/uavenv.h -- C ABI of libuavenv: the MI355X (gfx950) batched UAV-cellular environment.
 *
 * The reference (SamKnightGit/DRL_UAV_CellularNet) has no FFI layer: its boundary is the Python
 * class MobiEnvironment (mobile_env.py:35).  This header is what a binding for that class binds
 * (see INTEGRATION.md for the ctypes stub).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *  - plain C, no HIP/torch types: streams travel as void* (a hipStream_t), buffers as raw pointers;
 *  - every *_dev pointer is CALLER-OWNED DEVICE memory on the handle's device; NULL output pointers
 *    are skipped; the library never allocates inside reset/step (graph-capture safe);
 *  - all calls are asynchronous on the given stream and return 0 or a negative UAVENV_E_* code,
 *    never throw; uavenv_last_error() gives a thread-local message;
 *  - one handle is single-threaded; distinct handles are independent;
 *  - N = n_envs, U = n_ue, B = n_bs, Gr = n_groups, G = grid.
 */
#ifndef UAVENV_H
#define UAVENV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVENV_ABI_VERSION 7   /* 2: state blob = arrays of records (UavEnvStateLayout); 3: + uavenv_step_many, uavenv_step_seq;
                                * 4: + uavenv_sinr_area_at, uavenv_step_many_packed / uavenv_unpack_outputs, uavenv_debug_variant_* (launch census), uavenv_debug_rotation_info,
                                *      uavenv_step_many_prepare;
                                * 5: + UAVENV_E_DEVICE, uavenv_device_error (one-launch rotation schedule with bounded hand-offs), uavenv_step_range,
                                *      uavenv_launch_timing / uavenv_launch_times_us;
                                * 6: - uavenv_step_many_packed / uavenv_unpack_outputs (ABI 4's packed output records: no benefit once measured
                                *      under the one-launch schedule, removed with their 24 kernel instantiations per shape);
                                * 7: + uavenv_rollout_gated / UavEnvGatedRollout (a whole rollout as one persistent launch beside a persistent policy kernel) */
#define UAVENV_MAX_GROUPS 16
#define UAVENV_MAX_BS 32

enum {
    UAVENV_OK = 0,
    UAVENV_E_INVALID = -1,   /* bad argument / config            */
    UAVENV_E_HIP = -2,       /* a HIP runtime call failed        */
    UAVENV_E_NODEVICE = -3,  /* no usable gfx950 device          */
    UAVENV_E_NOMEM = -4,
    UAVENV_E_DEVICE = -5     /* an earlier launch on this handle reported a failure from the device (uavenv_device_error) */
};

/* Constants of the reference, as data.  uavenv_default_config() fills the values every reference
 * script runs with; citations: mobile_env.py:18-32,45,49-50,76  channel.py:7,21,36,40,46-55,81-82
 * ue_mobility.py:436,450-451,473,487. */
typedef struct UavEnvConfig {
    int32_t n_bs, n_ue, n_groups, grid;
    int32_t group_size[UAVENV_MAX_GROUPS]; /* walkers per RPGM group, sum == n_ue (mobile_env.py:76) */
    int32_t bs_init_xy[UAVENV_MAX_BS][2];  /* UAV start cells (mobile_env.py:49-50), each in [1, grid-1] */
    int32_t max_step;                      /* MAXSTEP = 2000                                          */
    int32_t bs_step;                       /* BS_STEP = 2                                             */
    int32_t min_bs_dist;                   /* MIN_BS_DIST + BS_STEP = 4 (mobile_env.py:157)           */
    int32_t n_act;                         /* N_ACT = 5                                               */
    int32_t agg_init, deagg_len, agg_len;  /* 200, 100, 10                                            */
    int32_t _pad;
    double grid_width;                     /* metres per cell = 5                                     */
    double p_bs_dbm, noise_dbm;            /* 20, -121                                                */
    double pl_a, pl_b, pl_dis;             /* 38, 30, 0                                               */
    double antenna_gain, eq_loss;          /* 2, 0                                                    */
    double shadow_mean, shadow_sd;         /* 0, 2                                                    */
    double ho_thresh_db, out_thresh;       /* 1, 0                                                    */
    double ue_velocity, grp_v_min, grp_v_max, aggregation; /* 1, 0, 1, 0.8                           */
} UavEnvConfig;

/* Injected randomness (parity mode).  A NULL struct pointer, or NULL members, select the on-device
 * Philox4x32-10 streams (key = seed, counter = (env id, tick, index, draw site)). */
typedef struct UavEnvInitInject {
    const double *u_x_dev, *u_y_dev, *u_th_dev; /* [N,U]    uniforms, ue_mobility.py:434-437        */
    const double *u_g_dev;                      /* [N,5,Gr] g_x,g_y,g_fl,g_v,g_theta  :442-446      */
} UavEnvInitInject;

typedef struct UavEnvInject {
    const double *theta_u_dev; /* [N,U]    heading uniforms drawn this tick      ue_mobility.py:508    */
    const double *group_u_dev; /* [N,Gr,3] (theta,fl,v) uniforms of arriving groups  :517-521          */
    const double *fading_dev;  /* [N,U,B]  N(mean,sd) shadowing draws, UE-major   channel.py:240,254-256 */
} UavEnvInject;

/* Per-step outputs = what MobiEnvironment.step returns / exposes (mobile_env.py:150-194,231-232). */
typedef struct UavEnvOut {
    float *reward_dev;      /* [N]     max(meanSINR/20 - nOut/U, -1)   mobile_env.py:163-189        */
    uint8_t *done_dev;      /* [N]     step_n >= MAXSTEP               mobile_env.py:186-187        */
    float *mean_sinr_dev;   /* [N]     np.mean(current_BS_sinr)        channel.py:216               */
    int32_t *n_out_dev;     /* [N]     newly outaged UEs               channel.py:170-174           */
    int16_t *ue_xy_dev;     /* [N,U,2] env.ueLoc                       mobile_env.py:154-155        */
    int32_t *bs_xy_dev;     /* [N,B,2] env.bsLoc[:, :2]                mobile_env.py:157            */
    int8_t *serving_dev;    /* [N,U]   channel.current_BS (post handover) channel.py:162-167        */
    float *cur_sinr_dev;    /* [N,U]   channel.current_BS_sinr         channel.py:145-146           */
    int32_t *step_n_dev;    /* [N]     env.step_n                                                   */
    double *cur_sinr_f64_dev, *mean_sinr_f64_dev, *reward_f64_dev; /* optional float64 copies       */
} UavEnvOut;

/* Byte offsets of the persistent state arrays inside the state blob (get/set_state).  Arrays of little-endian records (ABI 2;
 * ABI 1 had one array per scalar field -- same bytes, but the step kernel is bound by memory-instruction issue):
 *   ue_pos  [N,U]  16 B  { f64 x, y }                                          walker position in cells
 *   ue_aux  [N,U]  16 B  { f64 heading_uniform; i16 ix, iy; i8 serving, fifo0, fifo1, fifo2 }   fifo0 = oldest bestBS_buf row
 *   grp     [N,Gr] 48 B  { f64 x, y, flight_len, speed, cos, sin }             RPGM group
 *   env     [N]    32 B  { u32 tick; i32 agg, deagg, fifo_depth, step_n; i32 pad[3] }
 *   bs_xy   [N,B,2] i32;   out_bits [N,ceil(U/64)] u64  previous outage set
 * drl_uav_cellularnet_amd.BatchedMobiEnv.state_fields() decodes a blob into named arrays. */
typedef struct UavEnvStateLayout {
    size_t total_bytes;
    size_t ue_pos, ue_aux, grp, env, bs_xy, out_bits;
} UavEnvStateLayout;

typedef struct uavenv uavenv_t;

int uavenv_abi_version(void);
const char *uavenv_last_error(void);

/* Reference constants for (n_bs, n_ue, grid); 4 equal groups; the 4-UAV layout when n_bs == 4. */
int uavenv_default_config(UavEnvConfig *cfg, int n_bs, int n_ue, int grid);

/* Allocate N envs on `device`.  env_id_base offsets the Philox env id (rank * N for sharding).
 * UAVENV_E_INVALID (see uavenv_last_error) for a config check_config rejects, and for a batch too large for ONE handle: the
 * kernels address each state / output array as base + 32-bit byte offset, so n_envs * max(16 n_ue, 48 n_groups) must stay below
 * 4 GiB when n_ue <= 64 (13.4 M envs at n_ue = 20), and n_envs * 32 otherwise.  Larger batches: several handles with consecutive
 * env_id_base ranges, which are bit-identical to one big batch. */
int uavenv_create(const UavEnvConfig *cfg, int64_t n_envs, int device, uint64_t seed, uint32_t env_id_base,
                  uavenv_t **out);
void uavenv_destroy(uavenv_t *h);

/* reference_point_group state construction (ue_mobility.py:433-451) + UAVs to their start cells. */
int uavenv_init(uavenv_t *h, const UavEnvInitInject *inj, void *stream);
/* n_ticks x next(self.mm) without a channel update (mobile_env.py:77-79).  With injection n_ticks must be 1. */
int uavenv_warmup(uavenv_t *h, int n_ticks, const UavEnvInject *inj, void *stream);
/* MobiEnvironment.reset (mobile_env.py:115-148) for envs with mask_dev[e] != 0 (NULL: all).  Also the
 * tail of the constructor: last warm-up tick + LTEChannel.__init__ (mobile_env.py:94-98, channel.py:92-93,110). */
int uavenv_reset(uavenv_t *h, const uint8_t *mask_dev, const UavEnvInject *inj, const UavEnvOut *out, void *stream);
/* MobiEnvironment.step (mobile_env.py:150-194): one fused kernel launch for all N envs. */
int uavenv_step(uavenv_t *h, const int64_t *actions_dev, const UavEnvInject *inj, const UavEnvOut *out,
                void *stream);

/* The same step for the envs [first_env, first_env + n_envs) of the batch only (all pointers are still those of the WHOLE batch: the
 * kernel indexes them by env).  For callers that pipeline parts of a batch on several streams -- the A2C rollout steps one half
 * while the policy network works on the other (a2c_single_thread.py:113-118: the workers are independent of each other).  Any
 * non-empty range inside the batch (a wavefront whose envs straddle a range border runs in both launches, each with its own envs
 * live); ranges that do not overlap may run concurrently on different streams. */
int uavenv_step_range(uavenv_t *h, const int64_t *actions_dev, int64_t first_env, int64_t n_envs, const UavEnvInject *inj,
                      const UavEnvOut *out, void *stream);
/* n_steps consecutive MobiEnvironment.step calls (mobile_env.py:150-194) in ONE launch, for callers whose actions do not depend on
 * the observations in between (a random policy, main.py's warm-up exploration; an action tape): actions_dev is [n_steps, N]
 * (step t uses row t) and every non-NULL member of `out` points to n_steps consecutive blocks of the single-step shape, e.g.
 * reward_dev [n_steps, N], ue_xy_dev [n_steps, N, U, 2]; block t holds what step t returned.  Bit-identical to n_steps calls of
 * uavenv_step with on-device randomness (no draws can be injected); no auto-reset, like the reference: envs that pass MAXSTEP
 * keep stepping with done = 1.  n_ue <= 64: walker / group / UAV state stays in registers across the steps (no per-step launch,
 * state load or state store); n_ue > 64: n_steps single-step launches on `stream`. */
int uavenv_step_many(uavenv_t *h, const int64_t *actions_dev, int n_steps, const UavEnvOut *out, void *stream);
/* The same n_steps steps as n_steps ordinary launches of the single-step kernel issued by ONE host call: step t reads row t of
 * actions_dev [n_steps, N]; `out` is the single-step output set, overwritten by every step (it holds the last step's results
 * afterwards), exactly as n_steps calls of uavenv_step would leave it.  For callers that pay a high price per host call
 * (Python: ~8 us per ctypes round trip, as much as a 4096-env kernel) but want one kernel per step. */
int uavenv_step_seq(uavenv_t *h, const int64_t *actions_dev, int n_steps, const UavEnvOut *out, void *stream);
/* MobiEnvironment.step_test in read_trace mode (mobile_env.py:196-233): UE cells come from ue_xy_in_dev [N,U,2]. */
int uavenv_step_trace(uavenv_t *h, const int64_t *actions_dev, const int16_t *ue_xy_in_dev,
                      const UavEnvInject *inj, const UavEnvOut *out, void *stream);
/* Same tensor, updated IN PLACE: obs_dev must still hold what the previous uavenv_obs_dense / uavenv_obs_dense_update
 * call of this handle wrote into it (the handle remembers those <= U+B cells per env); only cells that changed are
 * touched (-1 at the old cell, +1 at the new one).  First use on a buffer: call uavenv_obs_dense once.  The handle remembers
 * which buffer that was: any other obs_dev is refused with UAVENV_E_INVALID (its deltas would corrupt it silently). */
int uavenv_obs_dense_update(uavenv_t *h, float *obs_dev, void *stream);
/* MobiEnvironment.reset (and the constructor) in read_trace mode (mobile_env.py:85-89,128-131): UE cells come from
 * ue_xy_in_dev [N,U,2] (trace row 0), no mobility tick; UAVs to their start cells, LTEChannel.reset, step_n = 0. */
int uavenv_reset_trace(uavenv_t *h, const uint8_t *mask_dev, const int16_t *ue_xy_in_dev, const UavEnvInject *inj,
                       const UavEnvOut *out, void *stream);
/* env.state: (N, B+1, G, G) float32 count planes (mobile_env.py:139-140,169-170; ue_mobility.py:173-188;
 * channel.py:387-409).  Full rewrite of obs_dev. */
int uavenv_obs_dense(uavenv_t *h, float *obs_dev, void *stream);

/* LTEChannel.GetSinrInArea (channel.py:411-433) for the UAV cells currently in the state: per-cell DL SINR [dB] of the
 * nearest UAV with fresh shadowing.  Outputs [N,G,G] (row/column 0 = 0); at least one of out_f32_dev / out_f64_dev.
 * fading_inj_dev: [N,(G-1)^2,B] draws in the reference's call order per cell (interferers ascending, then the nearest
 * UAV), or NULL for the on-device Philox stream. */
int uavenv_sinr_area(uavenv_t *h, const double *fading_inj_dev, float *out_f32_dev, double *out_f64_dev, void *stream);
/* The same for ANY UAV cells, as the reference's signature GetSinrInArea(bsLoc) allows (channel.py:411): bs_xy_dev int32 [N,B,2]
 * (B = the handle's n_bs; NULL = the cells in the state, i.e. uavenv_sinr_area).  The state is not modified. */
int uavenv_sinr_area_at(uavenv_t *h, const int32_t *bs_xy_dev, const double *fading_inj_dev, float *out_f32_dev,
                        double *out_f64_dev, void *stream);

/* copy.deepcopy(env) (gradient.py:15) / checkpointing: the whole persistent state as one blob. */
int uavenv_state_layout(const uavenv_t *h, UavEnvStateLayout *layout);
int uavenv_get_state(uavenv_t *h, void *dst, int dst_is_device, void *stream);
int uavenv_set_state(uavenv_t *h, const void *src, int src_is_device, void *stream);

/* The float64 primitives the kernels compute with (csrc/lean_math.h), evaluated ON THE DEVICE over arrays of n doubles on the current
 * device: op 0 a/b (lm_div: b positive normal), 1 1/sqrt(a), 2 ln(a), 3 2^a, 4 sin(pi a) -> out0, cos(pi a) -> out1.  Test hook:
 * tests/test_lean_math_gpu.py measures their error against long double. */
int uavenv_lean_math_eval(int op, const double *a_dev, const double *b_dev, double *out0_dev, double *out1_dev, int64_t n, void *stream);

/* Launch census (test hook): the env kernels are templates, and every call picks ONE instantiation from (kernel family, bound on
 * n_bs, mode, path-loss form, checked / fast / pinned variant, multi-step).  Entry i of uavenv_debug_variant_count() entries:
 * its name, whether the dispatch logic can select it at all, and how often this process has launched it since start /
 * uavenv_debug_variant_reset().  tests/test_launch_variants_gpu.py launches every selectable instantiation against the oracle
 * and asserts that none is left at zero. */
int uavenv_debug_variant_count(void);
int uavenv_debug_variant_info(int i, char *name, size_t name_len, int *selectable, long long *launches);
void uavenv_debug_variant_reset(void);

/* Optional: prepare a multi-step call of n_steps ahead of time.  The first uavenv_step_many call with a new n_steps may build and
 * upload a launch schedule (a few hundred microseconds of host time, synchronous); a caller that times the call (bench.py) or must not
 * stall in it builds the schedule here instead.  Idempotent; 0 when there is nothing to prepare for this handle / n_steps. */
int uavenv_step_many_prepare(uavenv_t *h, int n_steps);

/* Test hook: how uavenv_step_many would run n_steps on this handle: *n_launches = 0 for the plain single
 * launch, else the number of launches of the rotation schedule (1 = the one-launch schedule with hand-offs
 * between wavefronts; the several-launch schedule of ABI 4 is gone) and *slots wavefronts per launch.  Environment, read once in
 * uavenv_create: UAVENV_ROTATE=0 never rotate, =1 rotate whenever a valid schedule exists; UAVENV_ROTATE_SLOTS=k plan as if the
 * device had k SIMDs (lets small batches exercise the schedule);
 * UAVENV_HANDOFF_SPIN_US=n spin budget of one hand-off wait (default 2 000 000); UAVENV_DEBUG_DROP_PUBLISH=1 builds schedules
 * whose hand-offs are never signalled (the time-out path's test). */
int uavenv_debug_rotation_info(uavenv_t *h, int n_steps, int *n_launches, long long *slots);

/* Duration of the multi-step launches themselves, for callers that time SHORT calls (bench.py's 20-step region lasts 0.1 ms: a pair of
 * HIP events recorded on the stream around the call are two marker packets that cost it 10 us).  uavenv_launch_timing(h, 1) makes every
 * following uavenv_step_many dispatch carry its own start / stop events (hipExtLaunchKernelGGL: the dispatch packet's
 * timestamps); uavenv_launch_times_us returns the durations of the up-to-256 launches since then in issue order (it waits for them),
 * *n_out = how many were launched.  uavenv_launch_timing(h, 0) switches back to plain launches.  Not capturable. */
int uavenv_launch_timing(uavenv_t *h, int enable);
int uavenv_launch_times_us(uavenv_t *h, double *us_out, int max_out, int *n_out);

/* Test hook that needs NO device: the schedule itself for n_wavefronts env-wavefronts on n_slots slots and n_steps steps (pure host
 * arithmetic; the CPU test suite checks its invariants for many shapes).  table_out: int32 [rows][3][4] = per slot-row its three pieces
 * {env-wavefront, first step, steps, bits (1 = waits for a hand-off, 2 = publishes one)} in the column order publish / whole / wait, or
 * NULL to ask for the size; *rows_out = rows (slots padded to whole workgroups; the padding rows are all-zero), *makespan_out = ceil(W T / S)
 * step-times.  UAVENV_E_INVALID when no schedule exists for these numbers (W <= S, ceil(W T / S) >= 2 T, n_steps < 2, ...). */
int uavenv_debug_schedule(int64_t n_wavefronts, int64_t n_slots, int n_steps, int32_t *table_out, int64_t table_capacity_rows,
                          int64_t *rows_out, int64_t *makespan_out);

/* A whole ROLLOUT -- T x (MobiEnvironment.step, mobile_env.py:150-194, + the observation of mobile_env.py:169-170 encoded for the policy) --
 * as ONE persistent kernel launch that runs BESIDE a persistent policy kernel of the caller (the learner's uavagent_actor_head_gated_f32,
 * include/uavagent.h), for the rollout loop of a2c_single_thread.py:113-133: choose_action -> env.step -> next observation, T times, with no
 * kernel boundary and no host in between.  The two kernels synchronise per BLOCK of UAVENV_GATE_ROWS = 16 consecutive envs through two words
 * in device memory (monotonic step counters; the caller zeroes / presets them before every rollout):
 *     gate_actions_dev[b] >= t + 1 : the actions of step t of block b are in actions_dev[t][16 b .. 16 b + 15]   (policy -> env)
 *     gate_obs_dev[b]     >= t + 1 : the encoded observation BEFORE step t of block b is in enc_out_*[t][16 b ..] (env -> policy; t = 0 is the
 *                                    caller's: it encodes the state the rollout starts from and presets the word to 1)
 * The policy must write actions (and this kernel writes its encodings) with agent-scope coherent stores, wait for them to complete
 * (s_waitcnt vmcnt(0)), then store the counter; readers poll with agent-scope loads.  Every wait is BOUNDED (UAVENV_HANDOFF_SPIN_US, default
 * 2 s): a partner that never arrives -- not launched, not co-resident, crashed -- leaves error word 0x47415445 "GATE", the kernel exits and
 * the handle answers UAVENV_E_DEVICE until uavenv_set_state (uavenv_device_error).  The caller must launch both kernels on DIFFERENT streams
 * (or parallel branches of one graph) so that they can run at the same time; this one occupies min(blocks / 2, CUs) workgroups of 8
 * wavefronts x <= 144 VGPRs, the partner 8 x <= 112: one workgroup of each then fills a CU exactly.  Correctness does not depend on that
 * placement: both kernels CLAIM their pairs of blocks from a counter in arrival order (claim_dev here, the partner's own word there), so the
 * lowest unfinished pairs are always held by resident workgroups on both sides.
 * The encoder is the first dense layer of main.py:147 / :153 applied to the raveled one-hot state of main.py:190 without forming it:
 * per env the B + U observation nodes (UAV k in plane 0 at its cell, UE in plane 1 + serving UAV) select rows (plane * G + x) * G + y of one
 * or two float32 tables [n_rows][hidden] (hidden a multiple of 4, <= 256; rows summed in node order, + bias, optionally relu6; a node off the
 * grid or a row >= n_rows contributes nothing): bit-identical to uavagent_first_layer_from_obs_f32.  enc_out_* are [T][N][hidden]: slot t + 1
 * is written after step t, for t + 1 < T.  idx_out_dev (optional) is [T + 1][N][B + U] int64: slot t + 1 = the row indices after step t.
 * reward_dev (optional) [T][N]: the reward of step t; every other output of `out` is overwritten each step as by uavenv_step, and `out`
 * must hold all nine standard outputs.  State, outputs and rewards are bit-identical to T calls of uavenv_step with the same actions.
 * n_ue <= 64 and n_bs == 4 (the reference's shape), on-device randomness only; UAVENV_E_INVALID otherwise. */
#define UAVENV_GATE_ROWS 16
typedef struct UavEnvGatedRollout {
    int32_t n_steps;
    const int64_t *actions_dev;          /* [T][N], written by the policy kernel while this launch runs */
    uint32_t *gate_actions_dev;          /* [ceil(N / 16)] */
    uint32_t *gate_obs_dev;              /* [ceil(N / 16)] */
    uint32_t *claim_dev;                 /* one word, zero before the launch (pairs of blocks are handed out in arrival order) */
    float *reward_dev;                   /* [T][N] or NULL */
    const float *enc_table_a_dev, *enc_bias_a_dev;   /* [n_rows][hidden], [hidden] (bias may be NULL) */
    float *enc_out_a_dev;                /* [T][N][hidden] */
    const float *enc_table_c_dev, *enc_bias_c_dev;   /* a second table over the same rows, or NULL */
    float *enc_out_c_dev;
    int64_t *idx_out_dev;                /* [T + 1][N][B + U] or NULL */
    int64_t enc_rows;                    /* n_rows of the tables */
    int32_t enc_hidden;                  /* floats per table row */
    int32_t enc_relu6;
} UavEnvGatedRollout;
int uavenv_rollout_gated(uavenv_t *h, const UavEnvGatedRollout *r, const UavEnvOut *out, void *stream);

/* Sticky device-side error of a handle: *code = 0, or the word a kernel left when it gave up (0x48414e44 "HAND": a wavefront of a
 * one-launch schedule waited longer than the spin budget for the wavefront that runs the first steps of the same envs -- never seen
 * outside the test hook, but a bounded wait is what turns a scheduling bug into an error code instead of a hung GPU).  The word lives
 * in host-mapped memory: reading it costs no HIP call, and it is meaningful once the stream of the failing launch has been
 * synchronised.  While it is non-zero every stepping / reset / state-reading entry point of the handle returns UAVENV_E_DEVICE;
 * uavenv_set_state() installs a whole state again and clears it. */
int uavenv_device_error(uavenv_t *h, uint32_t *code);

/* Philox4x32-10 of one counter/key on the HOST (known-answer tests of the generator the kernels use). */
void uavenv_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* UAVENV_H */
