/*
 * oracle/uavenv_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's hot path
 *   MobiEnvironment.__init__/reset/step      /root/reference/mobile_env.py:37-194
 *   LTEChannel DL path                        /root/reference/channel.py:13-269,387-409
 *   reference_point_group / BS_move / ...     /root/reference/ue_mobility.py:173-336,408-523
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (drl_uav_cellularnet_amd/) never includes, links or calls anything here.
 *
 * Parity status: PINNED -- checked bit-for-bit (ints) / <=1e-12 (float64) against
 * golden vectors captured from the real reference (tests/golden/make_golden.py).
 */
#ifndef UAVENV_ORACLE_H
#define UAVENV_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVO_MAX_GROUPS 16
#define UAVO_MAX_BS 32

typedef struct {
    int32_t n_bs, n_ue, n_groups, grid;
    int32_t group_size[UAVO_MAX_GROUPS];
    int32_t bs_init_xy[UAVO_MAX_BS][2];
    int32_t max_step;      /* MAXSTEP      mobile_env.py:18  */
    int32_t bs_step;       /* BS_STEP      mobile_env.py:32  */
    int32_t min_bs_dist;   /* MIN_BS_DIST + BS_STEP, mobile_env.py:157 */
    int32_t n_act;         /* N_ACT        mobile_env.py:21  */
    int32_t agg_init, deagg_len, agg_len; /* 200,100,10  ue_mobility.py:450-451,473,487 */
    int32_t _pad;
    double grid_width;     /* channel.py:21 */
    double p_bs_dbm;       /* channel.py:36 */
    double noise_dbm;      /* channel.py:40 */
    double pl_a, pl_b, pl_dis; /* channel.py:46-48 */
    double antenna_gain, eq_loss; /* channel.py:50,52 */
    double shadow_mean, shadow_sd; /* channel.py:54-55 */
    double ho_thresh_db;   /* channel.py:82 */
    double out_thresh;     /* channel.py:7  */
    double ue_velocity;    /* ue_mobility.py:436 */
    double grp_v_min, grp_v_max; /* mobile_env.py:76 velocity=(0,1) */
    double aggregation;    /* mobile_env.py:76 aggregation=0.8 */
} UavoConfig;

/* Per-env persistent state, struct-of-arrays, caller-owned (numpy) memory. */
typedef struct {
    int64_t n_envs;
    uint64_t seed;
    uint32_t env_id_base;
    uint32_t _pad;
    double *ue_x, *ue_y, *ue_hu;                    /* [N,U]  hu = heading uniform for the NEXT move */
    double *g_x, *g_y, *g_fl, *g_v, *g_cos, *g_sin; /* [N,Gr] */
    int32_t *agg, *deagg;                           /* [N] */
    uint32_t *tick;                                 /* [N] mobility ticks since init (Philox time) */
    int32_t *bs_xy;                                 /* [N,B,2] */
    int8_t *serving;                                /* [N,U] */
    int8_t *fifo;                                   /* [N,3,U] oldest row first */
    int32_t *fifo_depth;                            /* [N] */
    uint64_t *out_bits;                             /* [N,ceil(U/64)] previous-step outage set */
    int32_t *step_n;                                /* [N] */
    int16_t *ue_xy;                                 /* [N,U,2] last integer UE cells (mobile_env.py:154-155) */
} UavoState;

/* Injected randomness (parity mode).  Any pointer NULL => Philox4x32-10. */
typedef struct {
    const double *u_x, *u_y, *u_th; /* [N,U] */
    const double *u_g;              /* [N,5,Gr]  g_x,g_y,g_fl,g_v,g_theta */
} UavoInitInject;

typedef struct {
    const double *theta_u; /* [N,U]     heading uniforms drawn this tick           */
    const double *group_u; /* [N,Gr,3]  (theta,fl,v) uniforms for arriving groups  */
    const double *fading;  /* [N,U,B]   N(mean,sd) shadowing draws, UE-major       */
} UavoInject;

typedef struct {
    float *reward; uint8_t *done; float *mean_sinr; int32_t *n_out;  /* [N] */
    int16_t *ue_xy;    /* [N,U,2] */
    int32_t *bs_xy;    /* [N,B,2] */
    int8_t *serving;   /* [N,U]   */
    float *cur_sinr;   /* [N,U]   */
    int32_t *step_n;   /* [N]     */
    double *cur_sinr_f64; double *mean_sinr_f64; double *reward_f64; /* optional float64 copies */
} UavoOut;

void uavo_default_config(UavoConfig *cfg, int n_bs, int n_ue, int grid);
int uavo_init(const UavoConfig *cfg, UavoState *st, const UavoInitInject *inj);
int uavo_warmup(const UavoConfig *cfg, UavoState *st, const UavoInject *inj);          /* one mobility tick */
int uavo_reset(const UavoConfig *cfg, UavoState *st, const uint8_t *mask, const UavoInject *inj, UavoOut *out);
int uavo_step(const UavoConfig *cfg, UavoState *st, const int64_t *actions, const UavoInject *inj, UavoOut *out);
/* step_test with read_trace: UE ints come from the trace, no mobility tick (mobile_env.py:202-203) */
int uavo_step_trace(const UavoConfig *cfg, UavoState *st, const int64_t *actions, const int16_t *ue_xy_in,
                    const UavoInject *inj, UavoOut *out);
/* reset with read_trace: UE ints from the trace, no mobility tick (mobile_env.py:128-131) */
int uavo_reset_trace(const UavoConfig *cfg, UavoState *st, const uint8_t *mask, const int16_t *ue_xy_in,
                     const UavoInject *inj, UavoOut *out);
int uavo_obs_dense(const UavoConfig *cfg, const UavoState *st, float *obs);
/* LTEChannel.GetSinrInArea (channel.py:411-433): per-cell DL SINR of the NEAREST UAV with fresh fading, for the UAV
 * cells currently in st->bs_xy.  out [N,G,G] float64 (row/column 0 stay 0: the loops start at xMin = yMin = 1).
 * fading_inj [N,(G-1)*(G-1),B] in the reference's call order per cell (interferers ascending, then the serving UAV),
 * or NULL for Philox (draw site DOM_AREA, time = st->tick). */
int uavo_sinr_area(const UavoConfig *cfg, const UavoState *st, const double *fading_inj, double *out);
void uavo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double uavo_np_pairwise_sum(const double *a, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
