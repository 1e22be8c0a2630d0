// libuavenv: the gated rollout kernel (uavenv_rollout_gated, include/uavenv.h).  Included by uavenv_gated.hip only; the step body it runs
// is env_packed_body of uavenv_kernels.h.
#pragma once
#include "uavenv_kernels.h"

namespace uavk {

// ================================================================================================
// Gated rollout kernel (uavenv_rollout_gated): T steps of MobiEnvironment.step in ONE persistent launch whose actions are written by
// ANOTHER kernel while this one runs -- a policy (the learner's actor head) that is itself persistent and waits for this kernel's
// observations.  The two exchange nothing through the host or through kernel boundaries: per BLOCK of kGateRows = 16 consecutive envs there
// are two words in device memory,
//     gate_act[b] >= t + 1   "the actions of step t of block b's envs are in actions[t][...]"      (set by the policy, awaited here)
//     gate_obs[b] >= t + 1   "the encoded observation BEFORE step t of block b's envs is in out_a[t] / out_c[t]"  (set here, awaited there;
//                             t = 0 is the caller's: it encodes the observation the rollout starts from and presets the word to 1)
// and the protocol of sched_hand_off_*: payload in coherent stores -> s_waitcnt vmcnt(0) -> the word; the reader polls ONE lane, bounded,
// and loads the payload coherently after it.  A wait that does not end within the budget stores kDevErrGate in the handle's error word
// and the wavefront leaves the kernel; the host then fails every later call (UAVENV_E_DEVICE), it never hangs.
//
// A workgroup of 8 wavefronts owns a PAIR of blocks (32 envs) for the whole rollout and alternates between them: while the policy
// works on one block this workgroup steps and encodes the other.  Per block and step: (1) wait for the actions; (2) the single-step body
// of env_kernel_packed for the env-wavefronts that hold the block's envs (a wavefront that straddles the block border runs for both blocks,
// each time with its own envs live -- uavenv_step_range's rule); (3) the observation ENCODER: per env the B + U nodes of the observation
// (mobile_env.py:169-170: UAV k in plane 0, UE in plane 1 + serving UAV) as rows of one or two float tables, summed in node order, + bias,
// optionally relu6 -- the first dense layer of main.py:147 / :153 applied to the raveled one-hot state without forming it.  One wavefront per
// env, lane = float4 column group, UNR rows (x tables) in flight: the arithmetic and its order are those of the learner's
// sparse_rows_sum_kernel (agent_kernels.hip), results are bit-identical; (4) publish.
// (Tried and not kept, all bit-identical, profiles/r04g_*: four wavefronts stepping while four encode, counters in LDS between them -- a step
// of the pair took 84 us instead of 76: half the wavefronts have half the row pieces in flight; the same with EIGHT encoders + four steppers, 12
// wavefronts at <= 96 VGPRs beside the policy kernel's 2 x 112: the pair 4.43-4.54 against 3.75 ms, the step body spills 68 registers at 96; one wavefront per (env, table) with all 24 pieces
// issued at once: the same time as 16 at once; every second pair starting 8-36 us late so that the CUs' encoder phases do not coincide: 1 %;
// non-temporal table loads: the encoder alone 4.19 instead of 3.13 ms per rollout.)
// Launch: min(pairs, CUs) workgroups of kGateWaves = 8 wavefronts, at most UAVENV_GATE_VGPRS = 144 VGPRs (amdgpu_num_vgpr, which counts register
// PAIRS on gfx90a and later); the partner at most 112: one workgroup of each fills a CU's register file exactly (2 x 144 + 2 x 112 per SIMD lane), and that is how an idle chip is filled -- one pair per
// CU.  Residency: nothing here DEPENDS on that placement.  Pairs of blocks are claimed from a counter in arrival order by both kernels, so
// whichever workgroups are resident hold the lowest unfinished pairs on both sides; if the dispatcher ever puts two of these workgroups on one
// CU (and the partner's therefore on none), the pairs left over simply start when a workgroup has finished its rollout -- slower, never stuck.
// (A first version assigned pairs by workgroup index and kept two of its workgroups off one CU by using 88..96 VGPRs on purpose, 12
// wavefronts; at 96 VGPRs both kernels spilled: this kernel alone 3.12 against 2.47 ms per 8192 x 50 rollout, the partner 2.89 against 1.96,
// profiles/r04g_gated_kernels_alone_vgpr_caps.txt.)
// ================================================================================================
constexpr int kGateRows = 16, kGateWaves = 8;
#ifndef UAVENV_GATE_UNR24
#define UAVENV_GATE_UNR24 8      /* rows (x tables) in flight per wavefront at 24 nodes; 12: A/B build */
#endif
// What the step body loads coherently (env_packed_body's HO bits): the ACTIONS -- another kernel wrote them.  Not the env state and not the
// observation the encoder reads back: only wavefronts of this workgroup touch them during the launch, they share one CU and its L1 (work-group
// scope needs no cache bypass on gfx950), and the launch boundary took care of everything older.  UAVENV_GATE_COHERENT_STATE=1 (build flag,
// A/B): state and observation past the L1 as well.
#ifdef UAVENV_GATE_COHERENT_STATE
constexpr int kGateHO = 5;
#define GATE_OBS_LOAD(ptr) __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
constexpr int kGateHO = 4;
#define GATE_OBS_LOAD(ptr) (*(ptr))
#endif
#ifdef UAVENV_GATE_STAMPS       /* diagnostic build (tools/gated_timeline.py): s_memrealtime (100 MHz, one clock for the whole chip) of pair 0's events */
__device__ unsigned long long *g_gate_dbg;    // [T][2 halves][8]: slot 2 = actions seen, 3 = env step done (outputs in L2), 4 = encoded rows published
#define GATE_STAMP(t, half, k) do { if (g_gate_dbg != nullptr && pair == 0 && threadIdx.x == 0) g_gate_dbg[((t) * 2 + (half)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GATE_STAMP(t, half, k) do { } while (0)
#endif
constexpr uint32_t kDevErrGate = 0x47415445u;      // "GATE": error word of a gate wait that timed out
struct GatedParams {
    int T, n_blocks;
    const long long *actions;      // [T][N]
    uint32_t *gate_act, *gate_obs; // [n_blocks]
    uint32_t *claim;               // one word, zero before the launch: the next pair of blocks nobody has taken yet
    float *reward;                 // [T][N] or null (then p.out.reward is overwritten every step like every other output)
    const float *wa, *ba, *wc, *bc;// tables [n_rows][4 H4] and biases [4 H4] (wc / bc null: one table)
    float *oa, *oc;                // [T][N][4 H4]: slot t + 1 is written after step t (t + 1 < T)
    long long *idx_out;            // [T + 1][N][B + U] or null: slot t + 1 = row indices of the observation after step t (-1: node off the grid)
    long long n_rows;
    int H4, relu6, G;
};

__device__ __forceinline__ bool gate_wait(uint32_t *word, uint32_t need, const KParams &p) {
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long budget = (unsigned long long)p.sched_spin_us * 100ull;           // s_memrealtime ticks at 100 MHz
    bool ok = false;
    for (;;) {
        uint32_t v = 0u;
        if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)v) >= need) { ok = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t_start > budget) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (!ok && (threadIdx.x & 63) == 0) __hip_atomic_store(p.sched_err, kDevErrGate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("" ::: "memory");
    return ok;
}

__device__ __forceinline__ void enc_fma4(float4 &s, const float4 &v, float w) {
    s.x = __builtin_fmaf(v.x, w, s.x); s.y = __builtin_fmaf(v.y, w, s.y); s.z = __builtin_fmaf(v.z, w, s.z); s.w = __builtin_fmaf(v.w, w, s.w);
}
__device__ __forceinline__ float4 enc_relu6(float4 v) {
    v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f); v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
    return v;
}
__device__ __forceinline__ void enc_store4(float *dst, const float4 &v) {      // written through: the reader is another kernel, maybe another XCD
    union { float4 f; unsigned long long w[2]; } u;
    u.f = v;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst), u.w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst) + 1, u.w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A table row piece (an ordinary cached load: as non-temporal loads -- so as not to evict the partner kernel's weights from the L2 -- the
// encoder alone took 4.19 instead of 3.13 ms per rollout, profiles/r04g_gated_kernels_alone_vgpr_caps.txt).
__device__ __forceinline__ float4 enc_load4(const char *q) { return *reinterpret_cast<const float4 *>(q); }

// The encoder, one wavefront for NS envs of the block: m, m + m_stride, ... (n_here <= NS of them exist).  KT > 0: B + U known at compile time (the
// row loop unrolls by UNR around v_readlane).  NS = 2 (2 KT <= 64): the observation of BOTH envs is read and turned into row indices first --
// lanes [0, K) the first env, [K, 2 K) the second -- so that a wavefront pays that round trip once per block and step, not once per env; the rows
// are then gathered env by env.
template <int KT, int UNR, bool TWO, int NS>
__device__ __forceinline__ void encode_env(const GatedParams &g, const OutPtrs &obs, long long m, long long m_stride, int n_here, int U, int B, long long N,
                                           int t_slot, bool gather) {
    static_assert(NS == 1 || (KT > 0 && NS * KT <= 64), "several envs per wavefront: their nodes share its 64 lanes");
    const int lane = threadIdx.x & 63;
    const int K = KT > 0 ? KT : U + B;
    const int sl = NS > 1 ? lane / K : 0, k = NS > 1 ? lane - sl * K : lane;
    long long mine = -1ll;
    if (lane < NS * K && sl < n_here) {
        const long long me = m + sl * m_stride;
        int x, y, pl;
        if (k < B) {     // (written by other wavefronts of this workgroup a moment ago: see kGateHO)
            union { unsigned long long w; int2 c; } q;
            q.w = GATE_OBS_LOAD(reinterpret_cast<const unsigned long long *>(obs.bs_xy) + (me * B + k));
            x = q.c.x; y = q.c.y; pl = 0;
        } else {
            const long long iu = me * U + (k - B);
            union { uint32_t w; short2 c; } q;
            q.w = GATE_OBS_LOAD(reinterpret_cast<const uint32_t *>(obs.ue_xy) + iu);
            x = q.c.x; y = q.c.y;
            pl = 1 + (int)GATE_OBS_LOAD(obs.serving + iu);
        }
        const bool ok = x >= 0 && x < g.G && y >= 0 && y < g.G && pl >= 0 && pl <= B;
        mine = ok ? ((long long)pl * g.G + x) * g.G + y : -1ll;
        if (g.idx_out != nullptr) g.idx_out[((long long)t_slot * N + me) * K + k] = mine;
    }
    if (!gather) return;
    const float wgt = (mine >= 0 && mine < g.n_rows) ? 1.f : 0.f;     // (a row index outside the table means "no row", like an off-grid node)
    mine = (wgt != 0.f) ? mine : 0;
    const int H4 = g.H4;
    const uint32_t row_bytes = (uint32_t)H4 * 16u;
    const uint32_t my_row_off = (uint32_t)mine * row_bytes;          // < 4 GiB: checked by the host entry point
    const bool on = lane < H4;
    const uint32_t lane_off = (uint32_t)(on ? lane : H4 - 1) * 16u;  // every lane loads unconditionally (see sparse_rows_sum_kernel)
    auto one_env = [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        if (S >= n_here) return;
        float4 sa = {0.f, 0.f, 0.f, 0.f}, sc = {0.f, 0.f, 0.f, 0.f};
        if (KT > 0) {
#pragma unroll 1
            for (int k0 = 0; k0 < KT; k0 += UNR) {
                float4 va[UNR], vc[UNR];
#pragma unroll
                for (int j = 0; j < UNR; ++j) {
                    const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_row_off, S * KT + k0 + j) + lane_off;
                    va[j] = enc_load4(reinterpret_cast<const char *>(g.wa) + off);
                    if (TWO) vc[j] = enc_load4(reinterpret_cast<const char *>(g.wc) + off);
                }
#pragma unroll
                for (int j = 0; j < UNR; ++j) {                        // k ascending
                    const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), S * KT + k0 + j));
                    enc_fma4(sa, va[j], w);
                    if (TWO) enc_fma4(sc, vc[j], w);
                }
            }
        } else {
            for (int kk = 0; kk < K; ++kk) {
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_row_off, kk) + lane_off;
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), kk));
                enc_fma4(sa, enc_load4(reinterpret_cast<const char *>(g.wa) + off), w);
                if (TWO) enc_fma4(sc, enc_load4(reinterpret_cast<const char *>(g.wc) + off), w);
            }
        }
        if (!on) return;
        const unsigned long long o = ((unsigned long long)t_slot * (unsigned long long)N + (unsigned long long)(m + S * m_stride)) * row_bytes + lane_off;
        if (g.ba != nullptr) { const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(g.ba) + lane_off); sa.x += b.x; sa.y += b.y; sa.z += b.z; sa.w += b.w; }
        if (g.relu6) sa = enc_relu6(sa);
        enc_store4(reinterpret_cast<float *>(reinterpret_cast<char *>(g.oa) + o), sa);
        if (TWO) {
            if (g.bc != nullptr) { const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(g.bc) + lane_off); sc.x += b.x; sc.y += b.y; sc.z += b.z; sc.w += b.w; }
            if (g.relu6) sc = enc_relu6(sc);
            enc_store4(reinterpret_cast<float *>(reinterpret_cast<char *>(g.oc) + o), sc);
        }
    };
    one_env(std::integral_constant<int, 0>{});
    if (NS > 1) one_env(std::integral_constant<int, (NS > 1 ? 1 : 0)>{});
}

#ifdef UAVENV_GATE_NOCAP        /* timing experiment: the kernel without its register cap (such a build cannot run beside its partner) */
#define UAVENV_GATE_CAP
#else
#ifndef UAVENV_GATE_VGPRS
#define UAVENV_GATE_VGPRS 144       /* this kernel's share of a SIMD lane's 512 registers is 2 x this; the policy kernel has the rest (112) */
#endif
#define UAVENV_GATE_CAP __attribute__((amdgpu_num_vgpr(UAVENV_GATE_VGPRS / 2)))       /* (gfx90a and later: the attribute counts VGPR + AGPR pairs) */
#endif
template <int BT, bool PLC, int KT, bool TWO>
__global__ __launch_bounds__(64 * kGateWaves) UAVENV_GATE_CAP void env_kernel_gated(char *blob, const int8_t *gid_of_u, long long N, int U, int EPW, int Gr, int B_rt,
                                                                                     int lane_magic, const GatedParams g, const KParams p) {
    __shared__ int s_bs[kGateWaves][kMaxEpw][2 * kMaxBs];
    __shared__ int s_pair;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_pairs = (g.n_blocks + 1) >> 1;
    OutPtrs po = p.out;
    for (;;) {
        // pairs are CLAIMED, not assigned by workgroup index: the p-th workgroup of this kernel to arrive works with the p-th of the partner
        // kernel, so the lowest unfinished pairs are always held by resident workgroups on both sides (see "Residency" above)
        __syncthreads();
        if (threadIdx.x == 0) s_pair = (int)atomicAdd(g.claim, 1u);
        __syncthreads();
        const int pair = s_pair;
        if (pair >= n_pairs) break;
        if (g.reward != nullptr) po.reward = g.reward;
        for (int t = 0; t < g.T; ++t) {
            for (int half = 0; half < 2; ++half) {
                const int blk = 2 * pair + half;
                if (blk >= g.n_blocks) continue;
                const int e_lo = blk * kGateRows;
                const int e_hi = (long long)(e_lo + kGateRows) < N ? e_lo + kGateRows : (int)N;
                if (!gate_wait(g.gate_act + blk, (uint32_t)t + 1u, p)) return;
                GATE_STAMP(t, half, 2);
                const int w_lo = e_lo / EPW, n_w = (e_hi - 1) / EPW - w_lo + 1;
                for (int w = wave; w < n_w; w += kGateWaves) {
                    env_packed_body<BT, MODE_STEP, PLC, true, false, false, kGateHO>(blob, g.actions + (long long)t * N, gid_of_u, N, U, EPW, Gr, B_rt, lane_magic, p,
                                                                               s_bs, wave, (long long)(w_lo + w), 0, 1, e_lo, e_hi, &po);
                    __builtin_amdgcn_wave_barrier();
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's output stores are in the L2
                __syncthreads();
                GATE_STAMP(t, half, 3);
                const bool gather = t + 1 < g.T;
                // (UNR as in sparse_rows_sum_kernel: 24 = 3 x 8, 44 = 11 x 4; at 24 nodes a wavefront takes its two envs of the block, m and m + 8, at once)
                constexpr int NS = (KT > 0 && 2 * KT <= 64) ? 2 : 1;
                for (int m = e_lo + wave; m < e_hi; m += NS * kGateWaves)
                    encode_env<KT, (KT == 44 ? 4 : UAVENV_GATE_UNR24), TWO, NS>(g, p.out, m, kGateWaves, (NS > 1 && m + kGateWaves < e_hi) ? 2 : 1, U, BT, N, t + 1, gather);
                if (gather) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the encoded rows have left
                    __syncthreads();
                    if (threadIdx.x == 0) __hip_atomic_store(g.gate_obs + blk, (uint32_t)t + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    GATE_STAMP(t, half, 4);
                }
            }
            if (g.reward != nullptr) po.reward += N;
        }
    }
}

}  // namespace uavk
