// libuavagent.so: the sparse first layer of the actor / critic trunks (main.py:143-156 on the raveled state of main.py:190) for gfx950.
// Interface and rationale: include/uavagent.h.  Reference implementation and CPU path: agent.py (F.embedding_bag + bias).
//
// One wavefront per sample.  Lane k < K fetches index k of its sample (one coalesced load); v_readlane turns each index into a
// SCALAR row base, so every row read is  global_load_dwordx4 v, v_lane_offset, s[row]  (SGPR-base form, no per-lane 64-bit
// address arithmetic).  Lanes 0 .. H/4-1 own one float4 column group each: a 200-float row is 50 lanes x 16 B, 7 cache lines.
// The tables (N_S x H x 4 B = 40 MB each at the reference's sizes) stay resident in L2 / Infinity Cache, so the kernel is
// bound by cache bandwidth and load latency, not HBM: rows are read in groups of UNR (x2 tables) so that many reads are in flight.
// Sum order: k ascending in fp32, bias last -- embedding_bag(idx, W, mode="sum") + b.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/uavagent.h"
#include "agent_common.h"

namespace {
thread_local std::string g_err;
}
namespace uavagent_internal {
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}  // namespace uavagent_internal

namespace {
using uavagent_internal::fail;

__device__ __forceinline__ void add4(float4 &s, const float4 &v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
// s += v * w with w in {0.0f, 1.0f} held in an SGPR: one v_fmac per component, like the add it replaces; v * 1 + s rounds exactly
// like s + v, and v * 0 + s is s for the finite table entries (a skipped row still costs its read, of row 0, but adds nothing).
__device__ __forceinline__ void fma4(float4 &s, const float4 &v, float w) {
    s.x = __builtin_fmaf(v.x, w, s.x); s.y = __builtin_fmaf(v.y, w, s.y); s.z = __builtin_fmaf(v.z, w, s.z); s.w = __builtin_fmaf(v.w, w, s.w);
}
__device__ __forceinline__ float lane_weight(float w, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), k)); }

// KT > 0: K known at compile time (24 = 4 UAV + 20 UE, 44 = 4 + 40), so the row loop unrolls by UNR with no remainder --
// v_readlane is a convergent operation and hipcc will not unroll a loop around it when the trip count is a run-time value.
// KT == 0: any K, one row (per table) in flight at a time.
// RELU6: the layer's activation applied to the stored result, h = relu6(sum + b) (tf.nn.relu6, main.py:147-148,153): saves the
// separate elementwise pass over [M, H] on both the acting and the update path.
__device__ __forceinline__ float4 relu6_4(float4 v) {
    v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f); v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
    return v;
}
// OBS: the index list is not read but BUILT, from the env's compact observation of sample (= env) m -- agent.obs_to_indices /
// obs_indices_kernel (agent_learner.hip) folded into the gather: node k < B is UAV k in plane 0, node k >= B is UE k - B in plane
// 1 + its serving UAV (mobile_env.py:169-170), row = (plane * G + x) * G + y, -1 for a node off the grid; the list is also stored
// (idx_out [M, K], the update's sample record) unless idx_out is null.  One launch less per rollout step.
struct ObsSrc {
    const int16_t *ue_xy;       // [M, U, 2]
    const int32_t *bs_xy;       // [M, B, 2]
    const int8_t *serving;      // [M, U]
    long long *idx_out;         // [M, U + B] or null
    int U, B, G;
};
template <bool TWO, int KT, int UNR, bool RELU6, bool OBS>
__global__ __launch_bounds__(256) void sparse_rows_sum_kernel(const float *__restrict__ wa, const float *__restrict__ ba,
                                                              float *__restrict__ oa, const float *__restrict__ wc,
                                                              const float *__restrict__ bc, float *__restrict__ oc,
                                                              const long long *__restrict__ idx, long long M, int K_rt, int H4,
                                                              long long n_rows, const ObsSrc src) {
    static_assert(KT == 0 || KT % UNR == 0, "the row groups must tile K exactly");
    const int K = KT > 0 ? KT : K_rt;
    const int lane = threadIdx.x & 63;
    const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // the same for all lanes of a wavefront
    if (m >= M) return;
    long long mine = 0;
    if (OBS) {
        if (lane < K) {
            int x, y, pl;
            if (lane < src.B) {
                const int2 c = reinterpret_cast<const int2 *>(src.bs_xy)[m * src.B + lane];
                x = c.x; y = c.y; pl = 0;
            } else {
                const long long iu = m * src.U + (lane - src.B);
                const short2 c = reinterpret_cast<const short2 *>(src.ue_xy)[iu];
                x = c.x; y = c.y; pl = 1 + src.serving[iu];
            }
            const bool ok = x >= 0 && x < src.G && y >= 0 && y < src.G && pl >= 0 && pl <= src.B;
            mine = ok ? ((long long)pl * src.G + x) * src.G + y : -1ll;
            if (src.idx_out != nullptr) src.idx_out[m * K + lane] = mine;
        }
    } else if (lane < K) mine = idx[m * K + lane];
    // An index outside [0, n_rows) means "no row" (agent.obs_to_indices writes -1 for a walker off the grid, and an all -1 list is
    // the reference's all-zero first state): it contributes nothing and is never dereferenced (include/uavagent.h).
    const float wgt = (mine >= 0 && mine < n_rows) ? 1.f : 0.f;
    mine = (wgt != 0.f) ? mine : 0;
    const uint32_t row_bytes = (uint32_t)H4 * 16u;
    const uint32_t my_row_off = (uint32_t)mine * row_bytes;          // < 4 GiB: checked by the host entry point
    // Every lane loads UNCONDITIONALLY: a conditional float4 load is split by hipcc into four exec-masked dword loads with a
    // vmcnt(0) wait after each.  Lanes >= H4 re-read the last column group (an in-range address); their sums are never stored.
    const bool on = lane < H4;
    const uint32_t lane_off = (uint32_t)(on ? lane : H4 - 1) * 16u;
    float4 sa = {0.f, 0.f, 0.f, 0.f}, sc = {0.f, 0.f, 0.f, 0.f};
    if (KT > 0) {
        // The outer loop stays a loop: fully unrolled, the scheduler hoists all 2*KT row reads to the top (156 VGPRs at KT = 24,
        // 258 at KT = 44).  UNR reads per table in flight is the intent; KT % UNR == 0, so there is no remainder.
#pragma unroll 1
        for (int k0 = 0; k0 < KT; k0 += UNR) {
            float4 va[UNR], vc[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {                            // UNR (x2 tables) row reads in flight
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_row_off, k0 + j) + lane_off;
                va[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wa) + off);
                if (TWO) vc[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wc) + off);
            }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {                            // k ascending
                const float w = lane_weight(wgt, k0 + j);
                fma4(sa, va[j], w);
                if (TWO) fma4(sc, vc[j], w);
            }
        }
    } else {
        for (int k = 0; k < K; ++k) {
            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_row_off, k) + lane_off;
            const float w = lane_weight(wgt, k);
            fma4(sa, *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wa) + off), w);
            if (TWO) fma4(sc, *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wc) + off), w);
        }
    }
    if (!on) return;
    const unsigned long long o = (unsigned long long)m * row_bytes;
    if (ba != nullptr) add4(sa, *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(ba) + lane_off));
    if (RELU6) sa = relu6_4(sa);
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(oa) + o + lane_off) = sa;
    if (TWO) {
        if (bc != nullptr) add4(sc, *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(bc) + lane_off));
        if (RELU6) sc = relu6_4(sc);
        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(oc) + o + lane_off) = sc;
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int uavagent_abi_version(void) { return 5; }
extern "C" const char *uavagent_last_error(void) { return g_err.c_str(); }

extern "C" int uavagent_first_layer_f32(const float *w_a, const float *bias_a, float *out_a, const float *w_c,
                                        const float *bias_c, float *out_c, const int64_t *idx, int64_t m_rows, int32_t k,
                                        int32_t h, int64_t n_rows, int32_t relu6, void *stream);

extern "C" int uavagent_sparse_rows_sum_f32(const float *w_a, const float *bias_a, float *out_a, const float *w_c,
                                            const float *bias_c, float *out_c, const int64_t *idx, int64_t m_rows, int32_t k,
                                            int32_t h, int64_t n_rows, void *stream) {
    return uavagent_first_layer_f32(w_a, bias_a, out_a, w_c, bias_c, out_c, idx, m_rows, k, h, n_rows, 0, stream);
}

namespace {
int first_layer_launch(const float *w_a, const float *bias_a, float *out_a, const float *w_c, const float *bias_c, float *out_c,
                       const int64_t *idx, const ObsSrc *obs, int64_t m_rows, int32_t k, int32_t h, int64_t n_rows, int32_t relu6,
                       void *stream);
}

extern "C" int uavagent_first_layer_f32(const float *w_a, const float *bias_a, float *out_a, const float *w_c,
                                        const float *bias_c, float *out_c, const int64_t *idx, int64_t m_rows, int32_t k,
                                        int32_t h, int64_t n_rows, int32_t relu6, void *stream) {
    if (m_rows > 0 && !idx) return fail(UAVAGENT_E_INVALID, "sparse_rows_sum: null table, output or index pointer");
    return first_layer_launch(w_a, bias_a, out_a, w_c, bias_c, out_c, idx, nullptr, m_rows, k, h, n_rows, relu6, stream);
}

extern "C" int uavagent_first_layer_from_obs_f32(const float *w_a, const float *bias_a, float *out_a, const float *w_c,
                                                 const float *bias_c, float *out_c, const int16_t *ue_xy, const int32_t *bs_xy,
                                                 const int8_t *serving, int64_t n_envs, int32_t n_ue, int32_t n_bs, int32_t grid,
                                                 int32_t h, int64_t n_rows, int32_t relu6, int64_t *idx_out, void *stream) {
    if (n_ue < 1 || n_bs < 1 || grid < 1 || n_ue + n_bs > 64)
        return fail(UAVAGENT_E_INVALID, "first_layer_from_obs: need n_ue, n_bs, grid >= 1 and n_ue + n_bs <= 64 (one lane per node)");
    if (n_envs > 0 && (!ue_xy || !bs_xy || !serving)) return fail(UAVAGENT_E_INVALID, "first_layer_from_obs: null observation pointer");
    if ((reinterpret_cast<uintptr_t>(ue_xy) & 3u) || (reinterpret_cast<uintptr_t>(bs_xy) & 7u))
        return fail(UAVAGENT_E_INVALID, "first_layer_from_obs: ue_xy must be 4-byte and bs_xy 8-byte aligned (one cell per load)");
    if ((long long)(n_bs + 1) * grid * grid > n_rows) return fail(UAVAGENT_E_INVALID, "first_layer_from_obs: the table has fewer than (n_bs + 1) * grid^2 rows");
    const ObsSrc src = {ue_xy, bs_xy, serving, reinterpret_cast<long long *>(idx_out), n_ue, n_bs, grid};
    return first_layer_launch(w_a, bias_a, out_a, w_c, bias_c, out_c, nullptr, &src, n_envs, n_ue + n_bs, h, n_rows, relu6, stream);
}

namespace {
int first_layer_launch(const float *w_a, const float *bias_a, float *out_a, const float *w_c, const float *bias_c, float *out_c,
                       const int64_t *idx, const ObsSrc *obs, int64_t m_rows, int32_t k, int32_t h, int64_t n_rows, int32_t relu6,
                       void *stream) {
    if (m_rows < 0 || n_rows < 1 || k < 1 || k > 64) return fail(UAVAGENT_E_INVALID, "sparse_rows_sum: need m_rows >= 0, n_rows >= 1, 1 <= k <= 64");
    if (h < 4 || h > 256 || (h & 3)) return fail(UAVAGENT_E_INVALID, "sparse_rows_sum: h must be a multiple of 4 in [4, 256]");
    if ((unsigned long long)n_rows * (unsigned long long)h * 4ull > 0xFFFFFFFFull)
        return fail(UAVAGENT_E_INVALID, "sparse_rows_sum: a table must be smaller than 4 GiB (rows are addressed by 32-bit byte offsets)");
    if (m_rows == 0) return UAVAGENT_OK;      // an empty batch: idx and the outputs may legitimately be null (torch's empty tensors are)
    if ((w_c != nullptr) != (out_c != nullptr)) return fail(UAVAGENT_E_INVALID, "sparse_rows_sum: w_c and out_c go together");
    if (!w_a || !out_a) return fail(UAVAGENT_E_INVALID, "sparse_rows_sum: null table, output or index pointer");
    if (!aligned16(w_a) || !aligned16(out_a) || !aligned16(w_c) || !aligned16(out_c) || !aligned16(bias_a) || !aligned16(bias_c))
        return fail(UAVAGENT_E_INVALID, "sparse_rows_sum: tables, biases and outputs must be 16-byte aligned");
    const long long blocks = (m_rows + 3) / 4;
    if (blocks > 0x7FFFFFFFll) return fail(UAVAGENT_E_INVALID, "sparse_rows_sum: m_rows too large for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long *ix = reinterpret_cast<const long long *>(idx);
    const ObsSrc none = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
#define UAVAGENT_LAUNCH_(TWO_, KT_, UNR_, R6_, OBS_)                                                                       \
    hipLaunchKernelGGL((sparse_rows_sum_kernel<TWO_, KT_, UNR_, R6_, OBS_>), dim3((unsigned)blocks), dim3(256), 0, s, w_a, bias_a, out_a, \
                       w_c, bias_c, out_c, ix, (long long)m_rows, (int)k, (int)(h / 4), (long long)n_rows, OBS_ ? *obs : none)
#define UAVAGENT_LAUNCH(TWO_, KT_, UNR_)                                                                                  \
    do {                                                                                                                  \
        if (obs) {                                                                                                        \
            if (relu6) UAVAGENT_LAUNCH_(TWO_, KT_, UNR_, true, true); else UAVAGENT_LAUNCH_(TWO_, KT_, UNR_, false, true);  \
        } else {                                                                                                          \
            if (relu6) UAVAGENT_LAUNCH_(TWO_, KT_, UNR_, true, false); else UAVAGENT_LAUNCH_(TWO_, KT_, UNR_, false, false); \
        }                                                                                                                 \
    } while (0)
    if (w_c) {
        if (k == 24) UAVAGENT_LAUNCH(true, 24, 8);
        else if (k == 44) UAVAGENT_LAUNCH(true, 44, 4);
        else UAVAGENT_LAUNCH(true, 0, 1);
    } else {
        if (k == 24) UAVAGENT_LAUNCH(false, 24, 8);
        else if (k == 44) UAVAGENT_LAUNCH(false, 44, 4);
        else UAVAGENT_LAUNCH(false, 0, 1);
    }
#undef UAVAGENT_LAUNCH_
#undef UAVAGENT_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(UAVAGENT_E_HIP, std::string("sparse_rows_sum launch: ") + hipGetErrorString(e));
    return UAVAGENT_OK;
}
}  // namespace
