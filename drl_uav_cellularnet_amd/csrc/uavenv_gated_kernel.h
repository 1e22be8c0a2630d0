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
// A workgroup of 12 wavefronts owns a PAIR of blocks (32 envs) for the whole rollout and alternates between them: while the policy
// works on one block this workgroup steps and encodes the other.  Per block and step: (1) wait for the actions; (2) the single-step body
// of env_kernel_packed for the env-wavefronts that hold the block's envs (a wavefront that straddles the block border runs for both blocks,
// each time with its own envs live -- uavenv_step_range's rule); (3) the observation ENCODER: per env the B + U nodes of the observation
// (mobile_env.py:169-170: UAV k in plane 0, UE in plane 1 + serving UAV) as rows of one or two float tables, summed in node order, + bias,
// optionally relu6 -- the first dense layer of main.py:147 / :153 applied to the raveled one-hot state without forming it.  One wavefront per
// env, lane = float4 column group, UNR rows (x tables) in flight: the arithmetic and its order are those of the learner's
// sparse_rows_sum_kernel (agent_kernels.hip), results are bit-identical; (4) publish.
// Launch: min(pairs, CUs) workgroups of kGateWaves = 12 wavefronts (three per SIMD).  Residency: the kernel must be co-resident with its
// partner whatever order the dispatcher meets the two kernels in, and the register file settles that: this kernel uses at least
// kGateVgprFloor = 88 VGPRs ON PURPOSE and at most 96 (amdgpu_waves_per_eu(5, 5)), the partner at most 96 with 8 wavefronts per workgroup.
// Two of THESE workgroups never fit one CU (6 x 88 > 512 VGPRs per SIMD lane), two partner workgroups never do either (135 KB of LDS each),
// and one of each always does (3 x 96 + 2 x 96 = 480): with min(pairs, CUs) workgroups of each kind every CU ends up with one pair.
// ================================================================================================
constexpr int kGateRows = 16, kGateWaves = 12;
constexpr uint32_t kDevErrGate = 0x47415445u;      // "GATE": error word of a gate wait that timed out
struct GatedParams {
    int T, n_blocks;
    const long long *actions;      // [T][N]
    uint32_t *gate_act, *gate_obs; // [n_blocks]
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

// The encoder for env m (one wavefront).  KT > 0: B + U known at compile time (the row loop unrolls by UNR around v_readlane).
template <int KT, int UNR, bool TWO>
__device__ __forceinline__ void encode_env(const GatedParams &g, const OutPtrs &obs, long long m, int U, int B, long long N, int t_slot, bool gather) {
    const int lane = threadIdx.x & 63;
    const int K = KT > 0 ? KT : U + B;
    long long mine = 0;
    if (lane < K) {
        int x, y, pl;
        if (lane < B) {     // coherent loads: these words were written by other wavefronts of this workgroup a moment ago (past this CU's L1)
            union { unsigned long long w; int2 c; } q;
            q.w = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(obs.bs_xy) + (m * B + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x = q.c.x; y = q.c.y; pl = 0;
        } else {
            const long long iu = m * U + (lane - B);
            union { uint32_t w; short2 c; } q;
            q.w = __hip_atomic_load(reinterpret_cast<const uint32_t *>(obs.ue_xy) + iu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x = q.c.x; y = q.c.y;
            pl = 1 + (int)__hip_atomic_load(obs.serving + iu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool ok = x >= 0 && x < g.G && y >= 0 && y < g.G && pl >= 0 && pl <= B;
        mine = ok ? ((long long)pl * g.G + x) * g.G + y : -1ll;
        if (g.idx_out != nullptr) g.idx_out[((long long)t_slot * N + m) * K + lane] = mine;
    }
    if (!gather) return;
    const float wgt = (mine >= 0 && mine < g.n_rows) ? 1.f : 0.f;
    mine = (wgt != 0.f) ? mine : 0;
    const int H4 = g.H4;
    const uint32_t row_bytes = (uint32_t)H4 * 16u;
    const uint32_t my_row_off = (uint32_t)mine * row_bytes;          // < 4 GiB: checked by the host entry point
    const bool on = lane < H4;
    const uint32_t lane_off = (uint32_t)(on ? lane : H4 - 1) * 16u;  // every lane loads unconditionally (see sparse_rows_sum_kernel)
    float4 sa = {0.f, 0.f, 0.f, 0.f}, sc = {0.f, 0.f, 0.f, 0.f};
    if (KT > 0) {
#pragma unroll 1
        for (int k0 = 0; k0 < KT; k0 += UNR) {
            float4 va[UNR], vc[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_row_off, k0 + j) + lane_off;
                va[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(g.wa) + off);
                if (TWO) vc[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(g.wc) + off);
            }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {                            // k ascending
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), k0 + j));
                enc_fma4(sa, va[j], w);
                if (TWO) enc_fma4(sc, vc[j], w);
            }
        }
    } else {
        for (int k = 0; k < K; ++k) {
            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_row_off, k) + lane_off;
            const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), k));
            enc_fma4(sa, *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(g.wa) + off), w);
            if (TWO) enc_fma4(sc, *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(g.wc) + off), w);
        }
    }
    if (!on) return;
    const unsigned long long o = ((unsigned long long)t_slot * (unsigned long long)N + (unsigned long long)m) * row_bytes + lane_off;
    if (g.ba != nullptr) { const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(g.ba) + lane_off); sa.x += b.x; sa.y += b.y; sa.z += b.z; sa.w += b.w; }
    if (g.relu6) sa = enc_relu6(sa);
    enc_store4(reinterpret_cast<float *>(reinterpret_cast<char *>(g.oa) + o), sa);
    if (TWO) {
        if (g.bc != nullptr) { const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(g.bc) + lane_off); sc.x += b.x; sc.y += b.y; sc.z += b.z; sc.w += b.w; }
        if (g.relu6) sc = enc_relu6(sc);
        enc_store4(reinterpret_cast<float *>(reinterpret_cast<char *>(g.oc) + o), sc);
    }
}

constexpr int kGateVgprFloor = 88;
template <int BT, bool PLC, int KT, bool TWO>
__global__ __launch_bounds__(64 * kGateWaves) __attribute__((amdgpu_waves_per_eu(5, 5))) void env_kernel_gated(char *blob, const int8_t *gid_of_u, long long N, int U, int EPW, int Gr, int B_rt,
                                                                     int lane_magic, const GatedParams g, const KParams p) {
    __shared__ int s_bs[kGateWaves][kMaxEpw][2 * kMaxBs];
    asm volatile("v_mov_b32 v87, 0" ::: "v87");                      // (residency: see kGateVgprFloor above)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_pairs = (g.n_blocks + 1) >> 1;
    OutPtrs po = p.out;
    for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        if (g.reward != nullptr) po.reward = g.reward;
        for (int t = 0; t < g.T; ++t) {
            for (int half = 0; half < 2; ++half) {
                const int blk = 2 * pair + half;
                if (blk >= g.n_blocks) continue;
                const int e_lo = blk * kGateRows;
                const int e_hi = (long long)(e_lo + kGateRows) < N ? e_lo + kGateRows : (int)N;
                if (!gate_wait(g.gate_act + blk, (uint32_t)t + 1u, p)) return;
                const int w_lo = e_lo / EPW, n_w = (e_hi - 1) / EPW - w_lo + 1;
                for (int w = wave; w < n_w; w += kGateWaves) {
                    env_packed_body<BT, MODE_STEP, PLC, true, false, false, 5>(blob, g.actions + (long long)t * N, gid_of_u, N, U, EPW, Gr, B_rt, lane_magic, p,
                                                                               s_bs, wave, (long long)(w_lo + w), 0, 1, e_lo, e_hi, &po);
                    __builtin_amdgcn_wave_barrier();
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's output stores are in the L2
                __syncthreads();
                const bool gather = t + 1 < g.T;
                for (int m = e_lo + wave; m < e_hi; m += kGateWaves) encode_env<KT, 8, TWO>(g, p.out, m, U, BT, N, t + 1, gather);
                if (gather) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the encoded rows have left
                    __syncthreads();
                    if (threadIdx.x == 0) __hip_atomic_store(g.gate_obs + blk, (uint32_t)t + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (g.reward != nullptr) po.reward += N;
        }
    }
}

}  // namespace uavk
