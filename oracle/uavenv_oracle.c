/*
 * oracle/uavenv_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Scalar, float64, libm restatement of the reference's env hot path, loop for loop.
 * See uavenv_oracle.h for scope and parity status.  Every function cites the
 * reference lines it follows (paths relative to /root/reference).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no fast-math: the reference is
 * NumPy/CPython float64 without FMA contraction).
 */
#include "uavenv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TWO_PI_NP (2.0 * 3.141592653589793) /* 2*np.pi  ue_mobility.py:437,508,517 */

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123 v1.09 constants) ------------------
 * Not in the reference: replaces the process-global MT19937 (ue_mobility.py:6, channel.py:240)
 * for the production (non-injected) mode.  Known-answer vectors: tests/test_philox.py. */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void uavo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

enum { /* Philox counter word 3: draw site */
    DOM_FADING = 1, DOM_HEADING = 2 /* quad mode (n_bs > 8) only; otherwise headings come from the spare words of DOM_FADING calls */, DOM_GROUP_A = 3, DOM_GROUP_B = 4,
    DOM_INIT_UE_A = 5, DOM_INIT_UE_B = 6, DOM_INIT_G_A = 7, DOM_INIT_G_B = 8, DOM_INIT_G_C = 9, DOM_AREA = 10
};

/* 53-bit uniform in [0,1) from two 32-bit words (same construction as numpy's random_double) */
static inline double u53(uint32_t hi, uint32_t lo) {
    return (double)(((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

static void philox_u2(const UavoState *st, int64_t e, uint32_t tick, uint32_t idx, uint32_t dom, double u[2]) {
    uint32_t ctr[4] = {st->env_id_base + (uint32_t)e, tick, idx, dom};
    uint32_t key[2] = {(uint32_t)st->seed, (uint32_t)(st->seed >> 32)};
    uint32_t o[4];
    uavo_philox4x32_10(ctr, key, o);
    u[0] = u53(o[0], o[1]);
    u[1] = u53(o[2], o[3]);
}

/* Per-UE draw block of a tick (Philox mode; the product's csrc/uavenv_kernels.h defines the same stream):
 *   call p of walker u:  q_p = Philox(ctr = (env, tick, u*HB + p, DOM_FADING)),  HB = ceil(B/2)
 *   q_p.x, q_p.y -> 53-bit uniform of the Box-Muller radius;  q_p.z * 2^-32 -> angle fraction;  q_p.w -> spare
 *   heading uniform drawn this tick = u53(q_0.w, q_1.w)   (HB >= 2)   or   q_0.w * 2^-32   (HB == 1)
 * Two Philox calls per UE and tick instead of three (a separate heading call): the 32x32->64 multiplies of Philox
 * are quarter-rate on CDNA4 and were ~17 % of the step kernel's issue cycles.
 * QUAD mode, n_bs > 8 (round 2; the 16-UAV shapes are bound by VALU issue and Philox was 19 % of it): one call serves FOUR UAVs,
 *   call c of walker u:  q_c = Philox(ctr = (env, tick, u*QB + c, DOM_FADING)),  QB = ceil(B/4)
 *   UAVs 4c, 4c+1: radius uniform q_c.x * 2^-32, angle fraction q_c.y * 2^-32;   UAVs 4c+2, 4c+3: q_c.z and q_c.w
 *   (32-bit radius uniforms: the normal's tail ends at sqrt(64 ln 2) = 6.66 sigma instead of 8.57)
 *   heading uniform = u53(h.x, h.y),  h = Philox(ctr = (env, tick, u, DOM_HEADING)). */
static void philox_raw(const UavoState *st, int64_t e, uint32_t tick, uint32_t idx, uint32_t dom, uint32_t o[4]) {
    uint32_t ctr[4] = {st->env_id_base + (uint32_t)e, tick, idx, dom};
    uint32_t key[2] = {(uint32_t)st->seed, (uint32_t)(st->seed >> 32)};
    uavo_philox4x32_10(ctr, key, o);
}

static int quad_mode(const UavoConfig *cfg) { return cfg->n_bs > 8; }

static double heading_uniform(const UavoConfig *cfg, const UavoState *st, int64_t e, uint32_t tick, int u) {
    const int HB = (cfg->n_bs + 1) / 2;
    uint32_t o0[4], o1[4];
    if (quad_mode(cfg)) {
        philox_raw(st, e, tick, (uint32_t)u, DOM_HEADING, o0);
        return u53(o0[0], o0[1]);
    }
    philox_raw(st, e, tick, (uint32_t)(u * HB + 0), DOM_FADING, o0);
    if (HB >= 2) {
        philox_raw(st, e, tick, (uint32_t)(u * HB + 1), DOM_FADING, o1);
        return u53(o0[3], o1[3]);
    }
    return (double)o0[3] * (1.0 / 4294967296.0);
}

/* ---- numpy add.reduce order (np.sum / np.mean at channel.py:216,265): pairwise_sum ---- */
double uavo_np_pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return uavo_np_pairwise_sum(a, n2) + uavo_np_pairwise_sum(a + n2, n - n2);
    }
}

void uavo_default_config(UavoConfig *c, int n_bs, int n_ue, int grid) {
    memset(c, 0, sizeof(*c));
    c->n_bs = n_bs; c->n_ue = n_ue; c->grid = grid;
    c->n_groups = 4;
    for (int g = 0; g < 4; ++g) c->group_size[g] = n_ue / 4; /* mobile_env.py:76 uses [10,10,10,10] */
    /* mobile_env.py:49-50: x in {G//4,G//4,3G//4,3G//4}, y in {G//4,3G//4,G//4,3G//4} (only nBS==4) */
    if (n_bs == 4) {
        int q = grid / 4, t = grid * 3 / 4;
        int xs[4] = {q, q, t, t}, ys[4] = {q, t, q, t};
        for (int b = 0; b < 4; ++b) { c->bs_init_xy[b][0] = xs[b]; c->bs_init_xy[b][1] = ys[b]; }
    }
    c->max_step = 2000; c->bs_step = 2; c->min_bs_dist = 4; c->n_act = 5;
    c->agg_init = 200; c->deagg_len = 100; c->agg_len = 10;
    c->grid_width = 5.0; c->p_bs_dbm = 20.0; c->noise_dbm = -121.0;
    c->pl_a = 38.0; c->pl_b = 30.0; c->pl_dis = 0.0;
    c->antenna_gain = 2.0; c->eq_loss = 0.0;
    c->shadow_mean = 0.0; c->shadow_sd = 2.0;
    c->ho_thresh_db = 1.0; c->out_thresh = 0.0;
    c->ue_velocity = 1.0; c->grp_v_min = 0.0; c->grp_v_max = 1.0; c->aggregation = 0.8;
}

static int check_cfg(const UavoConfig *c) {
    if (c->n_bs < 1 || c->n_bs > UAVO_MAX_BS || c->n_ue < 1 || c->n_groups < 1 || c->n_groups > UAVO_MAX_GROUPS)
        return -1;
    int s = 0;
    for (int g = 0; g < c->n_groups; ++g) s += c->group_size[g];
    return s == c->n_ue ? 0 : -1;
}

/* U = lambda MIN, MAX, SAMPLES: rand(...) * (MAX - MIN) + MIN      ue_mobility.py:408 */
static inline double U_(double mn, double mx, double r) { return r * (mx - mn) + mn; }

/* ---- reference_point_group: state construction, ue_mobility.py:433-451 ---------------- */
int uavo_init(const UavoConfig *cfg, UavoState *st, const UavoInitInject *inj) {
    if (check_cfg(cfg)) return -1;
    const int U = cfg->n_ue, Gr = cfg->n_groups, B = cfg->n_bs;
    const double MAX_X = cfg->grid, MAX_Y = cfg->grid, FL_MAX = cfg->grid; /* :428,433 dimensions=(G,G) */
    const int W64 = (U + 63) / 64;
    for (int64_t e = 0; e < st->n_envs; ++e) {
        for (int u = 0; u < U; ++u) {
            double ux, uy, ut, t[2];
            if (inj && inj->u_x) { ux = inj->u_x[e * U + u]; uy = inj->u_y[e * U + u]; ut = inj->u_th[e * U + u]; }
            else {
                philox_u2(st, e, 0xFFFFFFFFu, (uint32_t)u, DOM_INIT_UE_A, t); ux = t[0]; uy = t[1];
                philox_u2(st, e, 0xFFFFFFFFu, (uint32_t)u, DOM_INIT_UE_B, t); ut = t[0];
            }
            st->ue_x[e * U + u] = U_(0, MAX_X, ux);  /* :434 */
            st->ue_y[e * U + u] = U_(0, MAX_Y, uy);  /* :435 */
            st->ue_hu[e * U + u] = ut;               /* :437 theta = U(0, 2*pi) (cos/sin taken at use, :455) */
        }
        for (int g = 0; g < Gr; ++g) {
            double v[5], t[2];
            if (inj && inj->u_g) { for (int k = 0; k < 5; ++k) v[k] = inj->u_g[(e * 5 + k) * Gr + g]; }
            else {
                philox_u2(st, e, 0xFFFFFFFFu, (uint32_t)g, DOM_INIT_G_A, t); v[0] = t[0]; v[1] = t[1];
                philox_u2(st, e, 0xFFFFFFFFu, (uint32_t)g, DOM_INIT_G_B, t); v[2] = t[0]; v[3] = t[1];
                philox_u2(st, e, 0xFFFFFFFFu, (uint32_t)g, DOM_INIT_G_C, t); v[4] = t[0];
            }
            st->g_x[e * Gr + g] = U_(0, MAX_X, v[0]);   /* :442 */
            st->g_y[e * Gr + g] = U_(0, MAX_X, v[1]);   /* :443 (MAX_X, sic) */
            st->g_fl[e * Gr + g] = U_(0, FL_MAX, v[2]); /* :444 */
            st->g_v[e * Gr + g] = U_(cfg->grp_v_min, cfg->grp_v_max, v[3]); /* :445 */
            double th = U_(0, TWO_PI_NP, v[4]);         /* :446 */
            st->g_cos[e * Gr + g] = cos(th);            /* :447 */
            st->g_sin[e * Gr + g] = sin(th);            /* :448 */
        }
        st->agg[e] = cfg->agg_init;   /* :450 */
        st->deagg[e] = cfg->deagg_len;/* :451 */
        st->tick[e] = 0;
        for (int b = 0; b < B; ++b) { /* mobile_env.py:58-59 */
            st->bs_xy[(e * B + b) * 2 + 0] = cfg->bs_init_xy[b][0];
            st->bs_xy[(e * B + b) * 2 + 1] = cfg->bs_init_xy[b][1];
        }
        for (int u = 0; u < U; ++u) st->serving[e * U + u] = 0;
        memset(st->fifo + e * 3 * U, 0, (size_t)3 * U);
        st->fifo_depth[e] = 0;
        for (int w = 0; w < W64; ++w) st->out_bits[e * W64 + w] = 0;
        st->step_n[e] = 0;
        for (int u = 0; u < 2 * U; ++u) st->ue_xy[e * 2 * U + u] = 0;
    }
    return 0;
}

/* ---- one next() of reference_point_group, ue_mobility.py:453-523 ---------------------- */
static void mobility_tick(const UavoConfig *cfg, UavoState *st, int64_t e, const UavoInject *inj) {
    const int U = cfg->n_ue, Gr = cfg->n_groups;
    const double MAX_X = cfg->grid, MAX_Y = cfg->grid, FL_MAX = cfg->grid;
    double *x = st->ue_x + e * U, *y = st->ue_y + e * U, *hu = st->ue_hu + e * U;
    double *g_x = st->g_x + e * Gr, *g_y = st->g_y + e * Gr, *g_fl = st->g_fl + e * Gr;
    double *g_v = st->g_v + e * Gr, *g_cos = st->g_cos + e * Gr, *g_sin = st->g_sin + e * Gr;
    const double velocity = cfg->ue_velocity;
    uint32_t tick = st->tick[e];

    for (int u = 0; u < U; ++u) {          /* :455-456, costheta/sintheta from the previous draw */
        double th = U_(0, TWO_PI_NP, hu[u]);
        x[u] = x[u] + velocity * cos(th);
        y[u] = y[u] + velocity * sin(th);
    }
    for (int g = 0; g < Gr; ++g) {         /* :458-459 */
        g_x[g] = g_x[g] + g_v[g] * g_cos[g];
        g_y[g] = g_y[g] + g_v[g] * g_sin[g];
    }
    int u0 = 0;
    if (st->agg[e]) {                       /* :461-473 */
        for (int g = 0; g < Gr; ++g) {
            for (int u = u0; u < u0 + cfg->group_size[g]; ++u) {
                double c = atan2(g_y[g] - y[u], g_x[g] - x[u]);                    /* :467 */
                double nx = x[u] + g_v[g] * g_cos[g] + cfg->aggregation * cos(c); /* :469 */
                double ny = y[u] + g_v[g] * g_sin[g] + cfg->aggregation * sin(c); /* :470 */
                x[u] = nx; y[u] = ny;
            }
            u0 += cfg->group_size[g];
        }
        st->agg[e] -= 1;
        if (st->agg[e] == 0) st->deagg[e] = cfg->deagg_len;
    } else {                                /* :475-487 */
        for (int g = 0; g < Gr; ++g) {
            for (int u = u0; u < u0 + cfg->group_size[g]; ++u) {
                x[u] = x[u] + g_v[g] * g_cos[g];  /* :483 */
                y[u] = y[u] + g_v[g] * g_sin[g];  /* :484 */
            }
            u0 += cfg->group_size[g];
        }
        st->deagg[e] -= 1;
        if (st->deagg[e] == 0) st->agg[e] = cfg->agg_len;
    }
    /* :489-505 bounce; each test flips the heading of every group touched, once (np.unique).
     * The node-heading flips at :492,496,500,504 are dead: theta is redrawn at :508. */
    for (int pass = 0; pass < 4; ++pass) {
        int touched[UAVO_MAX_GROUPS] = {0};
        u0 = 0;
        for (int g = 0; g < Gr; ++g) {
            for (int u = u0; u < u0 + cfg->group_size[g]; ++u) {
                if (pass == 0 && x[u] < 0) { x[u] = -x[u]; touched[g] = 1; }
                else if (pass == 1 && x[u] > MAX_X) { x[u] = 2 * MAX_X - x[u]; touched[g] = 1; }
                else if (pass == 2 && y[u] < 0) { y[u] = -y[u]; touched[g] = 1; }
                else if (pass == 3 && y[u] > MAX_Y) { y[u] = 2 * MAX_Y - y[u]; touched[g] = 1; }
            }
            u0 += cfg->group_size[g];
        }
        for (int g = 0; g < Gr; ++g) if (touched[g]) {
            if (pass < 2) g_cos[g] = -g_cos[g]; else g_sin[g] = -g_sin[g];
        }
    }
    for (int u = 0; u < U; ++u) {          /* :508 theta = U(0, 2*pi, NODES) */
        if (inj && inj->theta_u) hu[u] = inj->theta_u[e * U + u];
        else hu[u] = heading_uniform(cfg, st, e, tick, u);
    }
    for (int g = 0; g < Gr; ++g) {         /* :513-521 */
        g_fl[g] = g_fl[g] - g_v[g];
        if (g_v[g] > 0. && g_fl[g] <= 0.) {
            double ut, uf, uv;
            if (inj && inj->group_u) {
                ut = inj->group_u[(e * Gr + g) * 3 + 0]; uf = inj->group_u[(e * Gr + g) * 3 + 1];
                uv = inj->group_u[(e * Gr + g) * 3 + 2];
            } else {
                double t[2];
                philox_u2(st, e, tick, (uint32_t)g, DOM_GROUP_A, t); ut = t[0]; uf = t[1];
                philox_u2(st, e, tick, (uint32_t)g, DOM_GROUP_B, t); uv = t[0];
            }
            double th = U_(0, TWO_PI_NP, ut);
            g_cos[g] = cos(th);
            g_sin[g] = sin(th);
            g_fl[g] = U_(0, FL_MAX, uf);
            g_v[g] = U_(cfg->grp_v_min, cfg->grp_v_max, uv);
        }
    }
    st->tick[e] = tick + 1;
    /* mobile_env.py:154-155 .astype(int): truncation toward zero */
    for (int u = 0; u < U; ++u) {
        st->ue_xy[(e * U + u) * 2 + 0] = (int16_t)(int64_t)x[u];
        st->ue_xy[(e * U + u) * 2 + 1] = (int16_t)(int64_t)y[u];
    }
}

int uavo_warmup(const UavoConfig *cfg, UavoState *st, const UavoInject *inj) {
    for (int64_t e = 0; e < st->n_envs; ++e) mobility_tick(cfg, st, e, inj);
    return 0;
}

/* ---- Decimal_to_Base_N + BS_move, ue_mobility.py:191-271,310-336 ----------------------- */
static void bs_move(const UavoConfig *cfg, int32_t *loc /*[B,2]*/, int64_t action) {
    const int B = cfg->n_bs;
    int digits[UAVO_MAX_BS];
    int64_t a = action;
    for (int i = B - 1; i >= 0; --i) { digits[i] = (int)(a % cfg->n_act); a /= cfg->n_act; } /* MSD -> UAV 0 */
    const int xMin = 1, xMax = cfg->grid, yMin = 1, yMax = cfg->grid; /* mobile_env.py:45 */
    const int s = cfg->bs_step, sl = 2 * cfg->bs_step;
    for (int i = 0; i < B; ++i) {
        int x = loc[2 * i], y = loc[2 * i + 1];
        switch (digits[i]) {                            /* :221-253 */
            case 0: if (x + s < xMax) x += s; break;
            case 1: if (x - s > xMin) x -= s; break;
            case 2: if (y + s < yMax) y += s; break;
            case 3: if (y - s > yMin) y -= s; break;
            case 5: if (x + sl < xMax) x += sl; break;
            case 6: if (x - sl > xMin) x -= sl; break;
            case 7: if (y + sl < yMax) y += sl; break;
            case 8: if (y - sl > yMin) y -= sl; break;
            default: break;
        }
        int collision = 0;                              /* :256-263: PRE-move loc[i], updated loc[j<i] */
        for (int j = 0; j < B; ++j) if (j != i) {
            int64_t dx = loc[2 * i] - loc[2 * j], dy = loc[2 * i + 1] - loc[2 * j + 1];
            double dist = sqrt((double)(dx * dx + dy * dy)); /* z equal: 3-D norm == 2-D norm */
            if (dist <= (double)cfg->min_bs_dist) collision = 1;
        }
        if (!collision) { loc[2 * i] = x; loc[2 * i + 1] = y; } /* :265-266 */
    }
}

/* ---- channel.py:220-269: gains and DL SINR in dB for one env --------------------------- */
static void dl_sinr_db(const UavoConfig *cfg, const int16_t *ue_xy, const int32_t *bs_xy,
                       const double *fading /*[U,B]*/, double *sinr /*[U,B]*/) {
    const int U = cfg->n_ue, B = cfg->n_bs;
    const double P_bs_watt = pow(10.0, cfg->p_bs_dbm / 10.0) * 1e-3;   /* channel.py:58 */
    const double noise_watt = pow(10.0, cfg->noise_dbm / 10.0) * 1e-3; /* channel.py:59 */
    double gain[UAVO_MAX_BS], tmp[UAVO_MAX_BS];
    for (int u = 0; u < U; ++u) {
        for (int b = 0; b < B; ++b) {
            /* GetDistance :220-226: coord[:2]*gridWidth, z ignored (Q1) */
            double dx = (double)ue_xy[2 * u] * cfg->grid_width - (double)bs_xy[2 * b] * cfg->grid_width;
            double dy = (double)ue_xy[2 * u + 1] * cfg->grid_width - (double)bs_xy[2 * b + 1] * cfg->grid_width;
            double d = sqrt(dx * dx + dy * dy);
            double loss = 0;                               /* GetPassLoss :230-235 (Q2: d=0 -> 0) */
            if (d > cfg->pl_dis) loss = cfg->pl_a + cfg->pl_b * log10(d);
            double f = fading[u * B + b];                  /* :240 */
            double gdb = cfg->antenna_gain - loss - f - cfg->eq_loss; /* :245 */
            gain[b] = pow(10.0, gdb / 10.0);               /* :246 */
        }
        for (int b = 0; b < B; ++b) {                      /* GetDLSinrAllDb :259-269 */
            int n = 0;
            for (int j = 0; j < B; ++j) if (j != b) tmp[n++] = P_bs_watt * gain[j]; /* :265 */
            double P_interf = uavo_np_pairwise_sum(tmp, n);
            double s = P_bs_watt * gain[b] / (noise_watt + P_interf); /* :266 */
            sinr[u * B + b] = 10 * log10(s);               /* :268 */
        }
    }
}

static void draw_fading(const UavoConfig *cfg, const UavoState *st, int64_t e, uint32_t tick,
                        const UavoInject *inj, double *fading) {
    const int U = cfg->n_ue, B = cfg->n_bs, HB = (B + 1) / 2;
    if (inj && inj->fading) { memcpy(fading, inj->fading + e * U * B, sizeof(double) * U * B); return; }
    /* np.random.normal(mean, sd) (channel.py:240) -> Box-Muller on Philox uniforms, two BSs per call (four in quad mode) */
    if (quad_mode(cfg)) {
        const int QB = (B + 3) / 4;
        for (int u = 0; u < U; ++u)
            for (int c = 0; c < QB; ++c) {
                uint32_t o[4];
                philox_raw(st, e, tick, (uint32_t)(u * QB + c), DOM_FADING, o);
                for (int h = 0; h < 2; ++h) {
                    const int b0 = 4 * c + 2 * h;
                    if (b0 >= B) break;
                    double r = sqrt(-2.0 * log(1.0 - (double)o[2 * h] * (1.0 / 4294967296.0)));
                    double a = TWO_PI_NP * ((double)o[2 * h + 1] * (1.0 / 4294967296.0));
                    fading[u * B + b0] = cfg->shadow_mean + cfg->shadow_sd * (r * cos(a));
                    if (b0 + 1 < B) fading[u * B + b0 + 1] = cfg->shadow_mean + cfg->shadow_sd * (r * sin(a));
                }
            }
        return;
    }
    for (int u = 0; u < U; ++u)
        for (int p = 0; p < HB; ++p) {
            uint32_t o[4];
            philox_raw(st, e, tick, (uint32_t)(u * HB + p), DOM_FADING, o);
            double r = sqrt(-2.0 * log(1.0 - u53(o[0], o[1])));
            double a = TWO_PI_NP * ((double)o[2] * (1.0 / 4294967296.0));
            fading[u * B + 2 * p] = cfg->shadow_mean + cfg->shadow_sd * (r * cos(a));
            if (2 * p + 1 < B) fading[u * B + 2 * p + 1] = cfg->shadow_mean + cfg->shadow_sd * (r * sin(a));
        }
}

static void write_common_out(const UavoConfig *cfg, const UavoState *st, int64_t e, UavoOut *out,
                             const double *cur) {
    const int U = cfg->n_ue, B = cfg->n_bs;
    if (!out) return;
    if (out->ue_xy) memcpy(out->ue_xy + e * 2 * U, st->ue_xy + e * 2 * U, sizeof(int16_t) * 2 * U);
    if (out->bs_xy) memcpy(out->bs_xy + e * 2 * B, st->bs_xy + e * 2 * B, sizeof(int32_t) * 2 * B);
    if (out->serving) memcpy(out->serving + e * U, st->serving + e * U, (size_t)U);
    if (out->step_n) out->step_n[e] = st->step_n[e];
    for (int u = 0; u < U; ++u) {
        if (out->cur_sinr) out->cur_sinr[e * U + u] = (float)cur[u];
        if (out->cur_sinr_f64) out->cur_sinr_f64[e * U + u] = cur[u];
    }
}

/* ---- LTEChannel.reset / GetBestDlBS, channel.py:113-124 -------------------------------- */
static void channel_reset(const UavoConfig *cfg, UavoState *st, int64_t e, const UavoInject *inj,
                          UavoOut *out, double *fading, double *sinr, double *cur) {
    const int U = cfg->n_ue, B = cfg->n_bs, W64 = (U + 63) / 64;
    draw_fading(cfg, st, e, st->tick[e] - 1u, inj, fading);
    dl_sinr_db(cfg, st->ue_xy + e * 2 * U, st->bs_xy + e * 2 * B, fading, sinr);
    for (int w = 0; w < W64; ++w) st->out_bits[e * W64 + w] = 0;
    for (int u = 0; u < U; ++u) {
        int best = 0;                                      /* np.argmax: first maximum (Q10) */
        for (int b = 1; b < B; ++b) if (sinr[u * B + b] > sinr[u * B + best]) best = b;
        st->serving[e * U + u] = (int8_t)best;             /* :114,122 */
        cur[u] = sinr[u * B + best];                       /* :123 */
        st->fifo[(e * 3 + 0) * U + u] = (int8_t)best;      /* :115 bestBS_buf = [current_BS] */
        if (cur[u] <= cfg->out_thresh) st->out_bits[e * W64 + u / 64] |= (1ull << (u % 64)); /* :116 */
    }
    st->fifo_depth[e] = 1;
    write_common_out(cfg, st, e, out, cur);
}

int uavo_reset(const UavoConfig *cfg, UavoState *st, const uint8_t *mask, const UavoInject *inj, UavoOut *out) {
    const int U = cfg->n_ue, B = cfg->n_bs;
    double *fading = (double *)malloc(sizeof(double) * U * B * 2 + sizeof(double) * U);
    double *sinr = fading + U * B, *cur = sinr + U * B;
    for (int64_t e = 0; e < st->n_envs; ++e) {
        if (mask && !mask[e]) continue;
        for (int b = 0; b < B; ++b) {                      /* mobile_env.py:119 bsLoc = initBsLoc */
            st->bs_xy[(e * B + b) * 2 + 0] = cfg->bs_init_xy[b][0];
            st->bs_xy[(e * B + b) * 2 + 1] = cfg->bs_init_xy[b][1];
        }
        mobility_tick(cfg, st, e, inj);                    /* mobile_env.py:122-127 (Q8) */
        channel_reset(cfg, st, e, inj, out, fading, sinr, cur); /* mobile_env.py:137 */
        st->step_n[e] = 0;                                 /* mobile_env.py:146 */
        if (out) {
            if (out->step_n) out->step_n[e] = 0;
            if (out->reward) out->reward[e] = 0.f;
            if (out->done) out->done[e] = 0;
            if (out->n_out) out->n_out[e] = 0;
            double m = uavo_np_pairwise_sum(cur, U) / (double)U;
            if (out->mean_sinr) out->mean_sinr[e] = (float)m;
            if (out->mean_sinr_f64) out->mean_sinr_f64[e] = m;
            if (out->reward_f64) out->reward_f64[e] = 0.0;
        }
    }
    free(fading);
    return 0;
}

/* reset() in read_trace mode, mobile_env.py:119,128-131,137,146: ueLoc = ueLoc_trace[0], no next(self.mm) */
int uavo_reset_trace(const UavoConfig *cfg, UavoState *st, const uint8_t *mask, const int16_t *ue_xy_in,
                     const UavoInject *inj, UavoOut *out) {
    const int U = cfg->n_ue, B = cfg->n_bs;
    double *fading = (double *)malloc(sizeof(double) * U * B * 2 + sizeof(double) * U);
    double *sinr = fading + U * B, *cur = sinr + U * B;
    for (int64_t e = 0; e < st->n_envs; ++e) {
        if (mask && !mask[e]) continue;
        for (int b = 0; b < B; ++b) {
            st->bs_xy[(e * B + b) * 2 + 0] = cfg->bs_init_xy[b][0];
            st->bs_xy[(e * B + b) * 2 + 1] = cfg->bs_init_xy[b][1];
        }
        memcpy(st->ue_xy + e * 2 * U, ue_xy_in + e * 2 * U, sizeof(int16_t) * 2 * U);
        st->tick[e] += 1; /* Philox time advances once per channel update */
        channel_reset(cfg, st, e, inj, out, fading, sinr, cur);
        st->step_n[e] = 0;
        if (out) {
            if (out->step_n) out->step_n[e] = 0;
            if (out->reward) out->reward[e] = 0.f;
            if (out->done) out->done[e] = 0;
            if (out->n_out) out->n_out[e] = 0;
            double m = uavo_np_pairwise_sum(cur, U) / (double)U;
            if (out->mean_sinr) out->mean_sinr[e] = (float)m;
            if (out->mean_sinr_f64) out->mean_sinr_f64[e] = m;
            if (out->reward_f64) out->reward_f64[e] = 0.0;
        }
    }
    free(fading);
    return 0;
}

/* ---- UpdateDroneNet (DL part), channel.py:138-216, + reward, mobile_env.py:163-189 ----- */
static void channel_update(const UavoConfig *cfg, UavoState *st, int64_t e, const UavoInject *inj,
                           UavoOut *out, double *fading, double *sinr, double *cur) {
    const int U = cfg->n_ue, B = cfg->n_bs, W64 = (U + 63) / 64;
    draw_fading(cfg, st, e, st->tick[e] - 1u, inj, fading);
    dl_sinr_db(cfg, st->ue_xy + e * 2 * U, st->bs_xy + e * 2 * B, fading, sinr); /* :139-140 */
    int8_t *serving = st->serving + e * U;
    int8_t *f0 = st->fifo + (e * 3 + 0) * U, *f1 = f0 + U, *f2 = f1 + U;
    int depth = st->fifo_depth[e];
    int n_outage = 0;
    uint64_t newbits[8] = {0};
    uint64_t *nb = W64 <= 8 ? newbits : (uint64_t *)calloc((size_t)W64, 8);
    for (int u = 0; u < U; ++u) {
        int best = 0;                                      /* :141-142 */
        for (int b = 1; b < B; ++b) if (sinr[u * B + b] > sinr[u * B + best]) best = b;
        double bestS = sinr[u * B + best];
        cur[u] = sinr[u * B + serving[u]];                 /* :145-146 serving BS BEFORE handover (Q4) */
        int8_t *newest;
        int remain;
        if (depth < 3) {                                   /* :148-149 append */
            int8_t *row = depth == 1 ? f1 : f2;
            row[u] = (int8_t)best;
            newest = row;
            remain = depth == 1 ? (f1[u] == f0[u]) : (f1[u] == f0[u] && f2[u] == f0[u]);
        } else {                                           /* :150-153 FIFO shift */
            f0[u] = f1[u]; f1[u] = f2[u]; f2[u] = (int8_t)best;
            newest = f2;
            remain = (f1[u] == f0[u] && f2[u] == f0[u]);   /* :155 */
        }
        int changed = serving[u] != newest[u];             /* :156 */
        int need = remain && changed && (bestS - cur[u] > cfg->ho_thresh_db); /* :158-159 */
        if (need) serving[u] = newest[u];                  /* :162-167 (cur_sinr not refreshed) */
        if (cur[u] <= cfg->out_thresh) {                   /* :170 */
            nb[u / 64] |= (1ull << (u % 64));
            if (!((st->out_bits[e * W64 + u / 64] >> (u % 64)) & 1ull)) n_outage++; /* :171-174 (Q3) */
        }
    }
    if (depth < 3) st->fifo_depth[e] = depth + 1;
    for (int w = 0; w < W64; ++w) st->out_bits[e * W64 + w] = nb[w]; /* :173 */
    if (nb != newbits) free(nb);

    double mean = uavo_np_pairwise_sum(cur, U) / (double)U; /* :216 np.mean */
    /* mobile_env.py:163-189 */
    double r0 = mean / 20;
    double r1 = -1.0 * n_outage / cfg->n_ue;
    st->step_n[e] += 1;
    int done = st->step_n[e] >= cfg->max_step;
    double reward = (0 + r0) + r1; /* sum([r0, r1]) */
    if (-1 > reward) reward = -1; /* max(sum(r_dissect), -1) */
    write_common_out(cfg, st, e, out, cur);
    if (out) {
        if (out->reward) out->reward[e] = (float)reward;
        if (out->reward_f64) out->reward_f64[e] = reward;
        if (out->done) out->done[e] = (uint8_t)done;
        if (out->mean_sinr) out->mean_sinr[e] = (float)mean;
        if (out->mean_sinr_f64) out->mean_sinr_f64[e] = mean;
        if (out->n_out) out->n_out[e] = n_outage;
    }
}

int uavo_step(const UavoConfig *cfg, UavoState *st, const int64_t *actions, const UavoInject *inj, UavoOut *out) {
    const int U = cfg->n_ue, B = cfg->n_bs;
    double *fading = (double *)malloc(sizeof(double) * U * B * 2 + sizeof(double) * U);
    double *sinr = fading + U * B, *cur = sinr + U * B;
    for (int64_t e = 0; e < st->n_envs; ++e) {
        mobility_tick(cfg, st, e, inj);                       /* mobile_env.py:152-155 */
        bs_move(cfg, st->bs_xy + e * 2 * B, actions[e]);      /* mobile_env.py:157 */
        channel_update(cfg, st, e, inj, out, fading, sinr, cur); /* mobile_env.py:158 */
    }
    free(fading);
    return 0;
}

int uavo_step_trace(const UavoConfig *cfg, UavoState *st, const int64_t *actions, const int16_t *ue_xy_in,
                    const UavoInject *inj, UavoOut *out) {
    const int U = cfg->n_ue, B = cfg->n_bs;
    double *fading = (double *)malloc(sizeof(double) * U * B * 2 + sizeof(double) * U);
    double *sinr = fading + U * B, *cur = sinr + U * B;
    for (int64_t e = 0; e < st->n_envs; ++e) {
        memcpy(st->ue_xy + e * 2 * U, ue_xy_in + e * 2 * U, sizeof(int16_t) * 2 * U); /* mobile_env.py:202-203 */
        st->tick[e] += 1; /* Philox time still advances once per channel update */
        bs_move(cfg, st->bs_xy + e * 2 * B, actions[e]);
        channel_update(cfg, st, e, inj, out, fading, sinr, cur);
    }
    free(fading);
    return 0;
}

/* ---- LTEChannel.GetSinrInArea, channel.py:411-433 -------------------------------------------------- */
int uavo_sinr_area(const UavoConfig *cfg, const UavoState *st, const double *fading_inj, double *out) {
    const int B = cfg->n_bs, G = cfg->grid, HB = (B + 1) / 2, W = G - 1;
    const double P_bs_watt = pow(10.0, cfg->p_bs_dbm / 10.0) * 1e-3;
    const double noise_watt = pow(10.0, cfg->noise_dbm / 10.0) * 1e-3;
    double f[UAVO_MAX_BS];
    for (int64_t e = 0; e < st->n_envs; ++e) {
        const int32_t *bs = st->bs_xy + e * 2 * B;
        double *o = out + e * (int64_t)G * G;
        memset(o, 0, sizeof(double) * (size_t)G * G);                      /* :413 np.zeros((gridX, gridY)) */
        for (int x = 1; x < G; ++x)                                          /* :416 range(xMin, xMax) */
            for (int y = 1; y < G; ++y) {                                    /* :417 */
                const int64_t cell = (int64_t)(x - 1) * W + (y - 1);
                int near = 0;                                                /* :418-422 argmin of GetDistance (first minimum) */
                double dmin = 0;
                for (int b = 0; b < B; ++b) {
                    double dx = (double)x * cfg->grid_width - (double)bs[2 * b] * cfg->grid_width;
                    double dy = (double)y * cfg->grid_width - (double)bs[2 * b + 1] * cfg->grid_width;
                    double d = sqrt(dx * dx + dy * dy);
                    if (b == 0 || d < dmin) { dmin = d; near = b; }
                }
                /* shadowing of UAV b at this cell */
                if (fading_inj) {                                            /* call order :425-429: others ascending, own last */
                    int k = 0;
                    for (int b = 0; b < B; ++b) if (b != near) f[b] = fading_inj[(e * W * W + cell) * B + k++];
                    f[near] = fading_inj[(e * W * W + cell) * B + (B - 1)];
                } else {
                    for (int p = 0; p < HB; ++p) {
                        double t[2];
                        philox_u2(st, e, st->tick[e], (uint32_t)(cell * HB + p), DOM_AREA, t);
                        double r = sqrt(-2.0 * log(1.0 - t[0])), a = TWO_PI_NP * t[1];
                        f[2 * p] = cfg->shadow_mean + cfg->shadow_sd * (r * cos(a));
                        if (2 * p + 1 < B) f[2 * p + 1] = cfg->shadow_mean + cfg->shadow_sd * (r * sin(a));
                    }
                }
                double P_interf = 0;                                         /* :423 */
                double own = 0;
                for (int b = 0; b < B; ++b) {
                    double dx = (double)x * cfg->grid_width - (double)bs[2 * b] * cfg->grid_width;
                    double dy = (double)y * cfg->grid_width - (double)bs[2 * b + 1] * cfg->grid_width;
                    double d = sqrt(dx * dx + dy * dy);
                    double loss = 0;
                    if (d > cfg->pl_dis) loss = cfg->pl_a + cfg->pl_b * log10(d);
                    double gain = pow(10.0, (cfg->antenna_gain - loss - f[b] - cfg->eq_loss) / 10.0);   /* :237-247 */
                    if (b == near) own = gain; else P_interf += P_bs_watt * gain;        /* :425-427 sequential += */
                }
                double s = P_bs_watt * own / (noise_watt + P_interf);        /* :429 */
                o[(int64_t)x * G + y] = 10 * log10(s);                       /* :431 */
            }
    }
    return 0;
}

/* ---- state planes: GetGridMap ue_mobility.py:173-188, GetCurrentAssociationMap
 *      channel.py:387-409, mobile_env.py:139-140,169-170 ---------------------------------- */
int uavo_obs_dense(const UavoConfig *cfg, const UavoState *st, float *obs) {
    const int U = cfg->n_ue, B = cfg->n_bs, G = cfg->grid;
    const int64_t plane = (int64_t)G * G;
    for (int64_t e = 0; e < st->n_envs; ++e) {
        float *o = obs + e * (B + 1) * plane;
        memset(o, 0, sizeof(float) * (size_t)((B + 1) * plane));
        for (int b = 0; b < B; ++b) {
            int x = st->bs_xy[(e * B + b) * 2], y = st->bs_xy[(e * B + b) * 2 + 1];
            if (x >= 0 && x < G && y >= 0 && y < G) o[(int64_t)x * G + y] += 1.f;
        }
        for (int u = 0; u < U; ++u) {
            int x = st->ue_xy[(e * U + u) * 2], y = st->ue_xy[(e * U + u) * 2 + 1];
            int b = st->serving[e * U + u];
            if (x >= 0 && x < G && y >= 0 && y < G) o[(1 + b) * plane + (int64_t)x * G + y] += 1.f;
        }
    }
    return 0;
}
