// libuavagent.so, part 2: the learner-side kernels around the PyTorch GEMMs (interface: include/uavagent.h, ABI 2).
//
// The MLP actor-critic stays PyTorch-ROCm (north_star): its dense layers are rocBLAS / hipBLASLt GEMMs called from agent.py.
// What is here is everything ELSE an A2C update and a rollout step spend their time on (profiles/r01_a2c_profile_after_*.txt:
// of 46.8 ms per update, 15 ms ATen's dense embedding backward, 6.8 ms column reductions, ~6 ms elementwise passes of autograd
// over [M, 625] tensors; of 292 us per rollout step, ~40 small launches):
//   obs_indices      compact observation -> row indices of the non-zero state cells (agent.obs_to_indices)      main.py:190,202
//   sample_actions   softmax + inverse-CDF draw, one wavefront per env (np.random.choice)                       main.py:165-169
//   a2c_loss_grad    softmax, loss terms and d(loss)/d(logits), d(loss)/dv in one pass over the logits          main.py:64-74
//   relu6_bwd        dx = dy * (0 < y < 6) with the bias gradient (column sums) in the same pass                main.py:147-148,153
//   critic_head_bwd  the [200 -> 1] value head backwards (outer product + two column sums)                      main.py:155
//   rows_grad        d(loss)/dW1: stable sort of (row, sample) pairs + segmented row sums, deterministic         (x^T g for 0/1 x)
//   rmsprop_tf1      tf.train.RMSPropOptimizer step, fused, on flat buffers                                     main.py:300-301
// All reductions are two-stage with a fixed grid, so every result is bit-reproducible from run to run (no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/uavagent.h"
#include "agent_common.h"

namespace {

int fail2(int code, const std::string &msg) { return uavagent_internal::fail(code, msg); }   // one error string for both parts

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// obs_indices: one thread per (env, node).  Node k < B: UAV k, plane 0; else UE k-B in plane 1 + serving UAV.
// A cell outside [0, G)^2 has no row: -1 (agent.obs_to_indices; the reference raises IndexError there).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void obs_indices_kernel(const int16_t *__restrict__ ue_xy, const int32_t *__restrict__ bs_xy,
                                                          const int8_t *__restrict__ serving, long long N, int U, int B, int G,
                                                          long long *__restrict__ idx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = U + B;
    if (t >= N * per) return;
    const long long n = t / per;
    const int k = (int)(t - n * per);
    int x, y, pl;
    if (k < B) { x = bs_xy[(n * B + k) * 2]; y = bs_xy[(n * B + k) * 2 + 1]; pl = 0; }
    else { const long long iu = n * U + (k - B); x = ue_xy[2 * iu]; y = ue_xy[2 * iu + 1]; pl = 1 + serving[iu]; }
    const bool ok = x >= 0 && x < G && y >= 0 && y < G && pl >= 0 && pl <= B;
    idx[t] = ok ? ((long long)pl * G + x) * G + y : -1ll;
}

// ---------------------------------------------------------------------------------------------------------------------
// sample_actions: one wavefront per row of logits [N, A].  Lane l owns the PER consecutive columns [l*PER, (l+1)*PER), so
// the running sum over columns (the CDF of np.random.choice) is a lane-local prefix plus an exclusive scan over lanes.
// action = first i with cdf[i] > u * cdf[A-1]  (searchsorted(side="right") on the normalised CDF, main.py:167-168).
// (Tried: coalesced row loads staged through LDS, then the same per-lane columns: 10.3 us against 9.0 us at [8192, 625] -- the
// strided direct loads are served by the vector L1 once the first lane's line is in; not kept.)
// ---------------------------------------------------------------------------------------------------------------------
template <int PER>
__global__ __launch_bounds__(256) void sample_actions_kernel(const float *__restrict__ logits, long long ld, const float *__restrict__ uni,
                                                             long long N, int A, long long *__restrict__ action,
                                                             float *__restrict__ prob) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const float *row = logits + r * ld;
    float v[PER];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = lane * PER + k;
        v[k] = (c < A) ? row[c] : -3.0e38f;
        mx = fmaxf(mx, v[k]);
    }
    mx = wave_max_f(mx);
    float loc = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = lane * PER + k;
        v[k] = (c < A) ? draw_exp(v[k] - mx) : 0.f;   // softmax numerator (tf.nn.softmax, main.py:150)
        loc += v[k];
    }
    float incl = loc;                                   // inclusive scan over lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    const float total = __shfl(incl, 63, 64);
    if (prob != nullptr) {
        const float inv = 1.f / total;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane * PER + k;
            if (c < A) prob[r * A + c] = v[k] * inv;
        }
    }
    const float target = uni[r] * total;
    const unsigned long long over = __ballot(incl > target);
    int a = A - 1;                                      // u * total rounding up to total: the last action (the reference clamps too)
    if (over != 0ull) {
        const int first = __ffsll((long long)over) - 1;
        // every lane walks its OWN PER values from its own exclusive prefix; the walk of the lane that holds the crossing is the answer
        // (the same sums in the same order as walking that lane's values by broadcast, without PER dependent cross-lane reads)
        float c = incl - loc;
        int fk = -1;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            c += v[k];
            if (fk < 0 && c > target) fk = k;
        }
        fk = __shfl(fk, first, 64);
        a = fk < 0 ? first * PER + PER - 1 : first * PER + fk;
        if (a > A - 1) a = A - 1;
    }
    if (lane == 0) action[r] = a;
}

// ---------------------------------------------------------------------------------------------------------------------
// a2c_loss_grad (main.py:64-74), rows = samples, grid-stride over rows, one wavefront per row at a time:
//   p = softmax(logits);  td = v_target - v;  c_loss = td^2;  a_loss = -(beta * H + log(p[a] + 1e-5) * stop_gradient(td)),
//   H = -sum p log(p + 1e-5).   d a_loss / d p_j = beta (log(p_j + e) + p_j / (p_j + e)) - [j == a] td / (p_a + e)   =: gp_j
//   d a_loss / d logit_i = p_i (gp_i - sum_j p_j gp_j);   d c_loss / d v = -2 td;   both scaled by 1/M (the means of :66,:74).
// The logits are overwritten with their gradient.  Column sums of that gradient (= the gradient of the output bias) and the
// loss sums are accumulated per wavefront and reduced by colsum_reduce_kernel in a fixed order.  Nothing here needs a running
// sum over columns, so lane l owns columns l, l + 64, ... (the sampling kernel's contiguous-per-lane layout made every load
// instruction touch a 2.5 KB span for 256 useful bytes: 0.99 ms for 2 GB of traffic).
// ---------------------------------------------------------------------------------------------------------------------
template <int PER>
__global__ __launch_bounds__(256) void a2c_loss_grad_kernel(float *__restrict__ logits, const float *__restrict__ v,
                                                            const float *__restrict__ target, const long long *__restrict__ act,
                                                            long long M, int A, long long ld, float beta, float inv_m, float *__restrict__ dv,
                                                            float *__restrict__ col_partial, double *__restrict__ loss_partial) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    float csum[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) csum[k] = 0.f;
    double la = 0.0, lc = 0.0, sdv = 0.0;
    for (long long r = wave; r < M; r += n_waves) {
        float *row = logits + r * ld;
        float p[PER];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = k * 64 + lane;          // coalesced: a load instruction reads 64 consecutive floats
            p[k] = (c < A) ? row[c] : -3.0e38f;
            mx = fmaxf(mx, p[k]);
        }
        mx = wave_max_f(mx);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = k * 64 + lane;          // coalesced: a load instruction reads 64 consecutive floats
            p[k] = (c < A) ? expf(p[k] - mx) : 0.f;
            s += p[k];
        }
        const float inv = 1.f / wave_sum_f(s);
        const float td = target[r] - v[r];
        const int a = (int)act[r];
        float gp[PER];
        float h = 0.f, dot = 0.f, lpa = 0.f, pa = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = k * 64 + lane;          // coalesced: a load instruction reads 64 consecutive floats
            p[k] *= inv;
            const float lp = logf(p[k] + 1e-5f);
            h -= p[k] * lp;                                         // entropy term (:71-72); p == 0 on padding lanes
            gp[k] = beta * (lp + p[k] / (p[k] + 1e-5f));
            if (c == a) { lpa = lp; pa = p[k]; }
        }
        lpa = wave_sum_f(lpa);
        pa = wave_sum_f(pa);
        h = wave_sum_f(h);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = k * 64 + lane;          // coalesced: a load instruction reads 64 consecutive floats
            if (c == a) gp[k] -= td / (pa + 1e-5f);
            dot += p[k] * gp[k];
        }
        dot = wave_sum_f(dot);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = k * 64 + lane;          // coalesced: a load instruction reads 64 consecutive floats
            const float g = p[k] * (gp[k] - dot) * inv_m;
            if (c < A) { row[c] = g; csum[k] += g; }
        }
        if (lane == 0) {
            const float g = -2.f * td * inv_m;
            dv[r] = g;
            sdv += (double)g;
            la += (double)(-(beta * h + lpa * td));                 // :73-74
            lc += (double)(td * td);                                // :66
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = k * 64 + lane;          // coalesced: a load instruction reads 64 consecutive floats
        if (c < A) col_partial[wave * A + c] = csum[k];
    }
    if (lane == 0) { loss_partial[wave * 3] = la; loss_partial[wave * 3 + 1] = lc; loss_partial[wave * 3 + 2] = sdv; }
}

// The same pass for rows that start 16-byte aligned (ld a multiple of 4: the learner's logits live in rows of 640 floats with a zero tail):
// a lane moves whole float4s (float4 index l, l + 64, ...: a wave instruction touches 1 KB of one row) and the per-element
// transcendentals are the hardware's (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp) instead of the library's correctly-handled-everywhere
// expf / logf / IEEE division: at 625 columns the first form spends ~50 VALU instructions per element -- 0.33 ms of pure issue for the
// update's 2.56e8 elements -- and ran at 3.3-3.9 TB/s; this one is bound by its 2 x 1.05 GB of traffic.  Operand ranges make the fast
// forms safe: x - max <= 0 (no overflow; underflow to 0 is the exact limit), p + 1e-5 in [1e-5, 1.00001] (normal, no special cases).
// Padding elements of the last float4 (columns >= A) are read (zero tail) but excluded from every sum and written back as zeros.
template <int PERV>
__global__ __launch_bounds__(256) void a2c_loss_grad_vec_kernel(float *__restrict__ logits, const float *__restrict__ v,
                                                                const float *__restrict__ target, const long long *__restrict__ act,
                                                                long long M, int A, long long ld, float beta, float inv_m, float *__restrict__ dv,
                                                                float *__restrict__ col_partial, double *__restrict__ loss_partial) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    const int nv = (A + 3) >> 2;                         // float4s of a row that hold at least one real column
    float4 csum[PERV];
#pragma unroll
    for (int k = 0; k < PERV; ++k) csum[k] = float4{0.f, 0.f, 0.f, 0.f};
    double la = 0.0, lc = 0.0, sdv = 0.0;
    constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    // the NEXT row of this wavefront is loaded while the current one is computed and stored (two rows in flight per wavefront)
    float4 nxt[PERV];
    auto load_row = [&](long long r) {
        const float4 *row = reinterpret_cast<const float4 *>(logits + r * ld);
#pragma unroll
        for (int k = 0; k < PERV; ++k) {
            const int i4 = k * 64 + lane;
            nxt[k] = float4{0.f, 0.f, 0.f, 0.f};
            if (i4 < nv) nxt[k] = row[i4];
        }
    };
    if (wave < M) load_row(wave);
    for (long long r = wave; r < M; r += n_waves) {
        float4 *row = reinterpret_cast<float4 *>(logits + r * ld);
        float x[PERV][4];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < PERV; ++k) {
            const float4 q = nxt[k];
            x[k][0] = q.x; x[k][1] = q.y; x[k][2] = q.z; x[k][3] = q.w;
        }
        const float td = target[r] - v[r];
        const int a = (int)act[r];
        if (r + n_waves < M) load_row(r + n_waves);
#pragma unroll
        for (int k = 0; k < PERV; ++k) {
            const int i4 = k * 64 + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (i4 * 4 + j >= A) x[k][j] = -3.0e38f;
                mx = fmaxf(mx, x[k][j]);
            }
        }
        mx = wave_max_f(mx);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PERV; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = (k * 64 + lane) * 4 + j < A;
                x[k][j] = ok ? __builtin_amdgcn_exp2f((x[k][j] - mx) * kLog2e) : 0.f;
                s += x[k][j];
            }
        const float inv = __builtin_amdgcn_rcpf(wave_sum_f(s));
        float gp[PERV][4];
        float h = 0.f, lpa = 0.f, pa = 0.f;
#pragma unroll
        for (int k = 0; k < PERV; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = (k * 64 + lane) * 4 + j;
                const float p = x[k][j] * inv;
                x[k][j] = p;
                const float q = p + 1e-5f;
                const float lp = __builtin_amdgcn_logf(q) * kLn2;
                h -= p * lp;                                             // entropy term (:71-72); p == 0 on padding elements
                gp[k][j] = beta * (lp + p * __builtin_amdgcn_rcpf(q));
                if (c == a) { lpa = lp; pa = p; }
            }
        lpa = wave_sum_f(lpa);
        pa = wave_sum_f(pa);
        h = wave_sum_f(h);
        const float tda = td * __builtin_amdgcn_rcpf(pa + 1e-5f);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < PERV; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = (k * 64 + lane) * 4 + j;
                if (c == a) gp[k][j] -= tda;
                dot += x[k][j] * gp[k][j];
            }
        dot = wave_sum_f(dot);
#pragma unroll
        for (int k = 0; k < PERV; ++k) {
            const int i4 = k * 64 + lane;
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = (i4 * 4 + j < A) ? x[k][j] * (gp[k][j] - dot) * inv_m : 0.f;
            if (i4 < nv) {
                row[i4] = float4{g[0], g[1], g[2], g[3]};
                csum[k].x += g[0]; csum[k].y += g[1]; csum[k].z += g[2]; csum[k].w += g[3];
            }
        }
        if (lane == 0) {
            const float g = -2.f * td * inv_m;
            dv[r] = g;
            sdv += (double)g;
            la += (double)(-(beta * h + lpa * td));                 // :73-74
            lc += (double)(td * td);                                // :66
        }
    }
#pragma unroll
    for (int k = 0; k < PERV; ++k) {
        const int c0 = (k * 64 + lane) * 4;
        const float cs[4] = {csum[k].x, csum[k].y, csum[k].z, csum[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < A) col_partial[wave * A + c0 + j] = cs[j];
    }
    if (lane == 0) { loss_partial[wave * 3] = la; loss_partial[wave * 3 + 1] = lc; loss_partial[wave * 3 + 2] = sdv; }
}

// out[c] = sum_w partial[w][c] in a FIXED order (deterministic): a block owns 64 columns; its 16 slab-threads per column each add
// a contiguous slab of the partial rows in ascending order (64 consecutive floats per load instruction: coalesced), then the
// 16 slab sums are added in slab order.  (The first version, one thread per column over all 4096 partial rows, took 0.99 ms per
// call, 5.9 ms per update: profiles/r02b_a2c_profile_fused_first.txt.)
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float *__restrict__ partial, long long n_part, int C,
                                                             float *__restrict__ out) {
    __shared__ float slab_sum[16][64];
    const int cl = threadIdx.x & 63, slab = threadIdx.x >> 6;          // 16 slabs x 64 columns
    const int c = blockIdx.x * 64 + cl;
    const long long per = (n_part + 15) / 16;
    const long long w0 = slab * per, w1 = (w0 + per < n_part) ? w0 + per : n_part;
    float s = 0.f;
    if (c < C) {
        long long w = w0;
        for (; w + 4 <= w1; w += 4) {                                   // four independent loads in flight, added in order
            const float a0 = partial[w * C + c], a1 = partial[(w + 1) * C + c], a2 = partial[(w + 2) * C + c], a3 = partial[(w + 3) * C + c];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; w < w1; ++w) s += partial[w * C + c];
    }
    slab_sum[slab][cl] = s;
    __syncthreads();
    if (slab == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += slab_sum[k][cl];
        out[c] = t;
    }
}
// One wavefront: lane l adds the partials w = l, l + 64, ... in ascending order, then a fixed shuffle tree (deterministic).
// (A single thread over all 4096 partials took 0.53 ms: profiles/r02b_a2c_profile_fused_second.txt.)
__global__ __launch_bounds__(64) void loss_reduce_kernel(const double *__restrict__ partial, long long n_part, double inv_m,
                                                         double *__restrict__ out) {
    const int lane = threadIdx.x;
    double a = 0.0, c = 0.0, d = 0.0;
    for (long long w = lane; w < n_part; w += 64) { a += partial[w * 3]; c += partial[w * 3 + 1]; d += partial[w * 3 + 2]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); c += __shfl_xor(c, off, 64); d += __shfl_xor(d, off, 64); }
    if (lane == 0) { out[0] = a * inv_m; out[1] = c * inv_m; out[2] = d; }   // mean actor loss, mean critic loss, sum of dv (= d loss / d b3c)
}

// ---------------------------------------------------------------------------------------------------------------------
// relu6 backwards with the bias gradient:  dx[m, c] = dy[m, c] * (0 < y[m, c] < 6)   (tf.nn.relu6 gradient),
// col_partial[wave][c] = sum over the wave's rows of dx.  C % 4 == 0, C <= 256: lane q < C/4 owns the float4 column group q.
// OUTER (the critic's value head, main.py:155: v = h2c @ w3 + b3 with w3 [C, 1]):  dy[m, c] = dv[m] * w3[c] is formed on the
// fly and a second column sum, sum_m y[m, c] * dv[m] = d loss / d w3, is produced in the same pass.
// dx is written with row stride ldx (so two results can share one [M, 2C] buffer for rows_grad).
// ---------------------------------------------------------------------------------------------------------------------
template <bool OUTER>
__global__ __launch_bounds__(256) void relu6_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                        const float *__restrict__ dv, const float *__restrict__ w3, long long M,
                                                        int C4, float *__restrict__ dx, long long ldx,
                                                        float *__restrict__ col_partial, float *__restrict__ col_partial_w) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    const bool on = lane < C4;
    const int q = on ? lane : 0;
    float4 cs = {0.f, 0.f, 0.f, 0.f}, cw = {0.f, 0.f, 0.f, 0.f};
    float4 w = {0.f, 0.f, 0.f, 0.f};
    if (OUTER) w = reinterpret_cast<const float4 *>(w3)[q];
    for (long long r = wave; r < M; r += n_waves) {
        const float4 yy = reinterpret_cast<const float4 *>(y + r * (long long)C4 * 4)[q];
        float4 g;
        if (OUTER) {
            const float d = dv[r];
            g = {d * w.x, d * w.y, d * w.z, d * w.w};
            cw.x += yy.x * d; cw.y += yy.y * d; cw.z += yy.z * d; cw.w += yy.w * d;
        } else {
            g = reinterpret_cast<const float4 *>(dy + r * (long long)C4 * 4)[q];
        }
        g.x = (yy.x > 0.f && yy.x < 6.f) ? g.x : 0.f;
        g.y = (yy.y > 0.f && yy.y < 6.f) ? g.y : 0.f;
        g.z = (yy.z > 0.f && yy.z < 6.f) ? g.z : 0.f;
        g.w = (yy.w > 0.f && yy.w < 6.f) ? g.w : 0.f;
        cs.x += g.x; cs.y += g.y; cs.z += g.z; cs.w += g.w;
        if (on) reinterpret_cast<float4 *>(dx + r * ldx)[q] = g;
    }
    if (on) {
        reinterpret_cast<float4 *>(col_partial + wave * (long long)C4 * 4)[q] = cs;
        if (OUTER) reinterpret_cast<float4 *>(col_partial_w + wave * (long long)C4 * 4)[q] = cw;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// rows_grad: dW[r, :] = sum over the (sample m, slot k) pairs with idx[m, k] == r of g[m, :]  -- the gradient of the sparse
// first layer's table (x^T g for the 0/1 count matrix x), for one or two tables that share idx (g is [M, ncol], ncol = H or 2H:
// actor columns first).  Deterministic: a STABLE radix sort of the pairs by row (rocPRIM) puts every row's samples in
// ascending order; chunks of kChunk sorted pairs are summed by one wavefront each (stage A); a row whose run lies inside one
// chunk is stored directly, runs that cross chunk borders leave per-chunk partials that stage B adds up in chunk order.
// Traffic: every pair gathers one g row (ncol * 4 bytes); ~9.8 M pairs x 1.6 KB = 15.7 GB per update at BASELINE config 3,
// served by the Infinity Cache / HBM at 5-6 TB/s (MI355X_MICROARCH.md, "Indexed rows") -- the bound of this kernel.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kChunk = 512;
constexpr uint32_t kNoRow = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void rows_keys_kernel(const long long *__restrict__ idx, long long n_pairs, int K, uint32_t n_rows,
                                                        uint32_t *__restrict__ keys, uint32_t *__restrict__ samp) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pairs) return;
    const long long r = idx[j];
    keys[j] = (r >= 0 && r < (long long)n_rows) ? (uint32_t)r : n_rows;     // "no row" sorts behind every row and is skipped
    samp[j] = (uint32_t)(j / K);
}

struct F4x2 { float4 a, b; };
__device__ __forceinline__ void add8(F4x2 &s, const F4x2 &v) {
    s.a.x += v.a.x; s.a.y += v.a.y; s.a.z += v.a.z; s.a.w += v.a.w;
    s.b.x += v.b.x; s.b.y += v.b.y; s.b.z += v.b.z; s.b.w += v.b.w;
}
// A g / dW / carry row is ncol4 float4 wide (<= 128): lane l owns float4 l and, if l + 64 < ncol4, float4 l + 64.
__device__ __forceinline__ F4x2 load_row(const float *base, int lane, int ncol4) {
    F4x2 v;
    const float4 *p = reinterpret_cast<const float4 *>(base);
    v.a = p[lane < ncol4 ? lane : 0];
    v.b = p[lane + 64 < ncol4 ? lane + 64 : 0];
    return v;
}
// Row r of the gradient tables: columns [0, H) -> dW0, [H, 2H) -> dW1 (ncol4 = H/4 or H/2).
__device__ __forceinline__ void store_out(float *dw0, float *dw1, int h4, uint32_t r, int lane, int ncol4, const F4x2 &v) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int q = lane + 64 * half;
        if (q < ncol4) {
            float *t = (q < h4) ? dw0 : dw1;
            const int c = (q < h4) ? q : q - h4;
            reinterpret_cast<float4 *>(t + (unsigned long long)r * (unsigned)h4 * 4u)[c] = half ? v.b : v.a;
        }
    }
}
__device__ __forceinline__ void store_carry(float *carry, long long slot, int lane, int ncol4, const F4x2 &v) {
    float4 *p = reinterpret_cast<float4 *>(carry + slot * (long long)ncol4 * 4);
    if (lane < ncol4) p[lane] = v.a;
    if (lane + 64 < ncol4) p[lane + 64] = v.b;
}

__global__ __launch_bounds__(256) void rows_sum_stage_a(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ samp,
                                                        long long n_pairs, uint32_t n_rows, const float *__restrict__ g, int ncol4,
                                                        int h4, float *__restrict__ dw0, float *__restrict__ dw1,
                                                        float *__restrict__ carry, uint32_t *__restrict__ head_row,
                                                        uint32_t *__restrict__ tail_row) {
    const int lane = threadIdx.x & 63;
    const long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long base = chunk * kChunk;
    if (base >= n_pairs) return;
    const long long end = (base + kChunk < n_pairs) ? base + kChunk : n_pairs;
    const uint32_t before = (base > 0) ? rows[base - 1] : kNoRow;          // a run touching the chunk start continues one from
    const uint32_t after = (end < n_pairs) ? rows[end] : kNoRow;           // the previous chunk iff it has the same row
    uint32_t hrow = kNoRow, trow = kNoRow;
    uint32_t cur = kNoRow;
    bool cur_at_start = false;
    F4x2 acc = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const unsigned long long row_bytes = (unsigned long long)ncol4 * 16ull;
    bool stop = false;
    for (long long b = base; b < end && !stop; b += 64) {
        const long long i = b + lane;
        const uint32_t r_l = (i < end) ? rows[i] : kNoRow;
        const uint32_t m_l = (i < end && r_l < n_rows) ? samp[i] : 0u;     // lanes without a pair gather row 0 and are never added
        const int n_here = (int)((end - b) < 64 ? (end - b) : 64);
        for (int k0 = 0; k0 < n_here && !stop; k0 += 8) {
            F4x2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {                                   // 8 gathers (16 loads) in flight
                const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)m_l, (k0 + j) & 63);
                v[j] = load_row(reinterpret_cast<const float *>(reinterpret_cast<const char *>(g) + (unsigned long long)m * row_bytes),
                                lane, ncol4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k0 + j >= n_here) break;
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)r_l, (k0 + j) & 63);
                if (r >= n_rows) { stop = true; break; }                    // the "no row" tail of the sorted list
                if (r != cur) {
                    if (cur != kNoRow) {                                    // the finished run ends inside the chunk
                        if (cur_at_start && before == cur) { store_carry(carry, chunk * 2, lane, ncol4, acc); hrow = cur; }
                        else store_out(dw0, dw1, h4, cur, lane, ncol4, acc);
                    }
                    cur_at_start = (b + k0 + j == base);
                    cur = r;
                    acc = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                }
                add8(acc, v[j]);
            }
        }
    }
    if (cur != kNoRow) {                                                    // the last run: does it go on in the next chunk?
        const bool is_head = cur_at_start && before == cur;
        const bool is_tail = !stop && after == cur;
        if (is_head) { store_carry(carry, chunk * 2, lane, ncol4, acc); hrow = cur; if (is_tail) trow = cur; }
        else if (is_tail) { store_carry(carry, chunk * 2 + 1, lane, ncol4, acc); trow = cur; }
        else store_out(dw0, dw1, h4, cur, lane, ncol4, acc);
    }
    if (lane == 0) { head_row[chunk] = hrow; tail_row[chunk] = trow; }
}

// Stage B: chunk c holds the START of a run that crosses chunk borders iff it has a tail that is not also its head.  That
// wavefront adds the partials of the chunks the run passes through, in chunk order, and stores the row.
__global__ __launch_bounds__(256) void rows_sum_stage_b(const uint32_t *__restrict__ head_row, const uint32_t *__restrict__ tail_row,
                                                        long long n_chunks, const float *__restrict__ carry, int ncol4, int h4,
                                                        float *__restrict__ dw0, float *__restrict__ dw1) {
    const int lane = threadIdx.x & 63;
    const long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const uint32_t row = tail_row[c];
    if (row == kNoRow || head_row[c] == row) return;           // no crossing run starts here (a through-chunk is not a start)
    // last chunk of the run: the through-chunks c+1 .. have head_row == tail_row == row; the final one has head_row == row only
    long long last = c + 1;
    while (last < n_chunks && head_row[last] == row && tail_row[last] == row) last += 64;      // gallop, then step back
    long long lo = (last - 64 > c + 1) ? last - 64 : c + 1;
    if (last >= n_chunks) last = n_chunks - 1;
    while (lo < last) {                                        // first chunk in [lo, last] that is NOT a through-chunk of `row`
        const long long mid = (lo + last) >> 1;
        if (head_row[mid] == row && tail_row[mid] == row) lo = mid + 1; else last = mid;
    }
    // `lo` is the chunk where the run ends (its head partial), guaranteed to exist: tail_row[c] == row means rows[end of c] == row
    const long long row_f = (long long)ncol4 * 4;
    F4x2 acc = load_row(carry + (c * 2 + 1) * row_f, lane, ncol4);
    long long k = c + 1;
    for (; k + 8 <= lo + 1; k += 8) {
        F4x2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = load_row(carry + (k + j) * 2 * row_f, lane, ncol4);
#pragma unroll
        for (int j = 0; j < 8; ++j) add8(acc, v[j]);
    }
    for (; k <= lo; ++k) { const F4x2 v = load_row(carry + k * 2 * row_f, lane, ncol4); add8(acc, v); }
    store_out(dw0, dw1, h4, row, lane, ncol4, acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// tf.train.RMSPropOptimizer(lr, decay = 0.9, momentum = 0, epsilon = 1e-10), TF1 kernel semantics (main.py:300-301):
//     ms <- decay * ms + (1 - decay) * g^2          (ms initialised to ONES by the caller)
//     w  <- w - lr * g / sqrt(ms + epsilon)         (epsilon inside the square root)
// One pass over flat [n] buffers; g may be pre-scaled (g * g_scale: the 1 / world_size of the gradient mean).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsprop_tf1_kernel(float *__restrict__ w, float *__restrict__ ms, const float *__restrict__ g,
                                                          long long n, float lr, float decay, float eps, float g_scale) {
    const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = i4 * 4;
    if (i + 3 < n) {
        float4 gg = reinterpret_cast<const float4 *>(g)[i4];
        float4 mm = reinterpret_cast<float4 *>(ms)[i4];
        float4 ww = reinterpret_cast<float4 *>(w)[i4];
        gg.x *= g_scale; gg.y *= g_scale; gg.z *= g_scale; gg.w *= g_scale;
        mm.x = decay * mm.x + (1.f - decay) * gg.x * gg.x; mm.y = decay * mm.y + (1.f - decay) * gg.y * gg.y;
        mm.z = decay * mm.z + (1.f - decay) * gg.z * gg.z; mm.w = decay * mm.w + (1.f - decay) * gg.w * gg.w;
        ww.x -= lr * gg.x / sqrtf(mm.x + eps); ww.y -= lr * gg.y / sqrtf(mm.y + eps);
        ww.z -= lr * gg.z / sqrtf(mm.z + eps); ww.w -= lr * gg.w / sqrtf(mm.w + eps);
        reinterpret_cast<float4 *>(ms)[i4] = mm;
        reinterpret_cast<float4 *>(w)[i4] = ww;
    } else {
        for (long long k = i; k < n; ++k) {
            const float gk = g[k] * g_scale;
            const float m = decay * ms[k] + (1.f - decay) * gk * gk;
            ms[k] = m;
            w[k] -= lr * gk / sqrtf(m + eps);
        }
    }
}

// v[m] = sum_c y[m, c] * w[c] + b[0]: the critic's value head forwards (main.py:155) -- a [M, C] x [C, 1] product, which rocBLAS
// runs as a gemv at 614 us per 65536 rows (tools/profile_a2c.py); this is one streaming pass.  One wavefront per row at a time.
__global__ __launch_bounds__(256) void rowdot_kernel(const float *__restrict__ y, const float *__restrict__ w, const float *__restrict__ b,
                                                     long long M, int C4, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    const bool on = lane < C4;
    const float4 ww = on ? reinterpret_cast<const float4 *>(w)[lane] : float4{0.f, 0.f, 0.f, 0.f};
    const float bias = (b != nullptr) ? b[0] : 0.f;
    for (long long r = wave; r < M; r += n_waves) {
        float s = 0.f;
        if (on) {
            const float4 yy = reinterpret_cast<const float4 *>(y + r * (long long)C4 * 4)[lane];
            s = yy.x * ww.x + yy.y * ww.y + yy.z * ww.z + yy.w * ww.w;
        }
        s = wave_sum_f(s);
        if (lane == 0) out[r] = s + bias;
    }
}

// n-step returns of a rollout (a2c_single_thread.py:176-183): run = bootstrap; for t = T-1 .. 0: run = r[t] + gamma * run; out[t] = run.
// One thread per env; reads and writes are coalesced across envs.  (PyTorch: 2 x T small launches, 0.3 ms per update.)
__global__ __launch_bounds__(256) void nstep_returns_kernel(const float *__restrict__ rew, const float *__restrict__ boot, long long N,
                                                            int T, float gamma, float *__restrict__ out) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float run = boot[n];
    for (int t = T - 1; t >= 0; --t) {
#pragma clang fp contract(off)   // two roundings per step, like the PyTorch form r[t] + gamma * run (hipcc would contract it into an FMA)
        run = rew[(long long)t * N + n] + gamma * run;
        out[(long long)t * N + n] = run;
    }
}

bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// fixed reduction grids (results depend on them: keep them constants of the library, not of the device)
constexpr int kLossBlocks = 1024;     // x 4 wavefronts
constexpr int kReluBlocks = 1024;

int per_lane_cols(int A) { return (A + 63) / 64; }

}  // namespace

namespace {
int launch_ok(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail2(UAVAGENT_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return UAVAGENT_OK;
}
}  // namespace

extern "C" int uavagent_obs_indices(const int16_t *ue_xy, const int32_t *bs_xy, const int8_t *serving, int64_t n_envs, int32_t n_ue,
                                    int32_t n_bs, int32_t grid, int64_t *idx_out, void *stream) {
    if (n_envs < 0 || n_ue < 1 || n_bs < 1 || grid < 1) return fail2(UAVAGENT_E_INVALID, "obs_indices: bad shape");
    if (n_envs == 0) return UAVAGENT_OK;
    if (!ue_xy || !bs_xy || !serving || !idx_out) return fail2(UAVAGENT_E_INVALID, "obs_indices: null pointer");
    const long long total = (long long)n_envs * (n_ue + n_bs);
    hipLaunchKernelGGL(obs_indices_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ue_xy, bs_xy,
                       serving, (long long)n_envs, (int)n_ue, (int)n_bs, (int)grid, reinterpret_cast<long long *>(idx_out));
    return launch_ok("obs_indices");
}

extern "C" int uavagent_sample_actions(const float *logits, int64_t ld_logits, const float *uniforms, int64_t n_rows, int32_t n_actions,
                                       int64_t *actions_out, float *prob_out, void *stream) {
    if (n_rows < 0 || n_actions < 1 || n_actions > 1024 || ld_logits < n_actions)
        return fail2(UAVAGENT_E_INVALID, "sample_actions: need 1 <= n_actions <= 1024 and ld_logits >= n_actions");
    if (n_rows == 0) return UAVAGENT_OK;
    if (!logits || !uniforms || !actions_out) return fail2(UAVAGENT_E_INVALID, "sample_actions: null pointer");
    const dim3 grid((unsigned)((n_rows + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    long long *ao = reinterpret_cast<long long *>(actions_out);
#define UAVAGENT_SAMPLE(P_) hipLaunchKernelGGL((sample_actions_kernel<P_>), grid, blk, 0, s, logits, (long long)ld_logits, uniforms, (long long)n_rows, (int)n_actions, ao, prob_out)
    switch (per_lane_cols(n_actions)) {
        case 1: UAVAGENT_SAMPLE(1); break;
        case 2: UAVAGENT_SAMPLE(2); break;
        case 3: case 4: UAVAGENT_SAMPLE(4); break;
        case 5: case 6: case 7: case 8: UAVAGENT_SAMPLE(8); break;
        case 9: case 10: UAVAGENT_SAMPLE(10); break;
        default: UAVAGENT_SAMPLE(16); break;
    }
#undef UAVAGENT_SAMPLE
    return launch_ok("sample_actions");
}

extern "C" size_t uavagent_loss_grad_workspace_bytes(int32_t n_actions) {
    const size_t waves = (size_t)kLossBlocks * 4;
    return up256(waves * (size_t)n_actions * sizeof(float)) + up256(waves * 3 * sizeof(double));
}

extern "C" int uavagent_a2c_loss_grad(float *logits_inout, int64_t ld_logits, const float *v, const float *v_target, const int64_t *actions,
                                      int64_t m_rows, int32_t n_actions, float beta, float *dv_out, float *dbias_out,
                                      double *loss_out, void *workspace, void *stream) {
    if (m_rows < 1 || n_actions < 1 || n_actions > 1024 || ld_logits < n_actions)
        return fail2(UAVAGENT_E_INVALID, "a2c_loss_grad: need m_rows >= 1, 1 <= n_actions <= 1024, ld_logits >= n_actions");
    if (!logits_inout || !v || !v_target || !actions || !dv_out || !dbias_out || !loss_out || !workspace)
        return fail2(UAVAGENT_E_INVALID, "a2c_loss_grad: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const long long waves = (long long)kLossBlocks * 4;
    float *colp = reinterpret_cast<float *>(workspace);
    double *lossp = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + up256((size_t)waves * n_actions * sizeof(float)));
    const float inv_m = 1.0f / (float)m_rows;
    const long long *ac = reinterpret_cast<const long long *>(actions);
#define UAVAGENT_LOSS(P_) hipLaunchKernelGGL((a2c_loss_grad_kernel<P_>), dim3(kLossBlocks), dim3(256), 0, s, logits_inout, v, v_target, ac, \
                                             (long long)m_rows, (int)n_actions, (long long)ld_logits, beta, inv_m, dv_out, colp, lossp)
    // rows that start 16-byte aligned (the learner's: ld 640): the float4 form; UAVAGENT_LOSS_SCALAR=1 keeps the first form (A/B runs)
    const char *force_scalar = getenv("UAVAGENT_LOSS_SCALAR");
    const bool vec = ((reinterpret_cast<uintptr_t>(logits_inout) & 15) == 0) && ((ld_logits & 3) == 0) && (ld_logits >= ((n_actions + 3) & ~3)) &&
                     !(force_scalar && force_scalar[0] == '1');
#define UAVAGENT_LOSSV(P_) hipLaunchKernelGGL((a2c_loss_grad_vec_kernel<P_>), dim3(kLossBlocks), dim3(256), 0, s, logits_inout, v, v_target, ac, \
                                              (long long)m_rows, (int)n_actions, (long long)ld_logits, beta, inv_m, dv_out, colp, lossp)
    if (vec) {
        const int nv = (n_actions + 3) / 4;
        if (nv <= 64) UAVAGENT_LOSSV(1); else if (nv <= 128) UAVAGENT_LOSSV(2); else if (nv <= 192) UAVAGENT_LOSSV(3); else UAVAGENT_LOSSV(4);
    } else
    switch (per_lane_cols(n_actions)) {
        case 1: UAVAGENT_LOSS(1); break;
        case 2: UAVAGENT_LOSS(2); break;
        case 3: case 4: UAVAGENT_LOSS(4); break;
        case 5: case 6: case 7: case 8: UAVAGENT_LOSS(8); break;
        case 9: case 10: UAVAGENT_LOSS(10); break;
        default: UAVAGENT_LOSS(16); break;
    }
#undef UAVAGENT_LOSS
#undef UAVAGENT_LOSSV
    if (int rc = launch_ok("a2c_loss_grad")) return rc;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n_actions + 63) / 64), dim3(1024), 0, s, colp, waves, (int)n_actions, dbias_out);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, s, lossp, waves, 1.0 / (double)m_rows, loss_out);
    return launch_ok("a2c_loss_grad reduce");
}

extern "C" size_t uavagent_relu6_bwd_workspace_bytes(int32_t n_cols) { return 2 * up256((size_t)kReluBlocks * 4 * (size_t)n_cols * sizeof(float)); }

extern "C" int uavagent_relu6_bwd(const float *dy, const float *y, const float *dv, const float *w3, int64_t m_rows, int32_t n_cols,
                                  float *dx_out, int64_t ldx, float *dbias_out, float *dw3_out, void *workspace, void *stream) {
    if (m_rows < 1 || n_cols < 4 || n_cols > 256 || (n_cols & 3)) return fail2(UAVAGENT_E_INVALID, "relu6_bwd: n_cols must be a multiple of 4 in [4, 256], m_rows >= 1");
    const bool outer = (dy == nullptr);
    if (!y || !dx_out || !dbias_out || !workspace || (outer && (!dv || !w3 || !dw3_out))) return fail2(UAVAGENT_E_INVALID, "relu6_bwd: null pointer");
    if (ldx < n_cols || (ldx & 3) || !al16(dy) || !al16(y) || !al16(dx_out) || !al16(w3))
        return fail2(UAVAGENT_E_INVALID, "relu6_bwd: buffers must be 16-byte aligned, ldx a multiple of 4 and >= n_cols");
    hipStream_t s = (hipStream_t)stream;
    const long long waves = (long long)kReluBlocks * 4;
    float *p0 = reinterpret_cast<float *>(workspace);
    float *p1 = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + up256((size_t)waves * n_cols * sizeof(float)));
    if (outer) hipLaunchKernelGGL((relu6_bwd_kernel<true>), dim3(kReluBlocks), dim3(256), 0, s, dy, y, dv, w3, (long long)m_rows, (int)(n_cols / 4), dx_out, (long long)ldx, p0, p1);
    else hipLaunchKernelGGL((relu6_bwd_kernel<false>), dim3(kReluBlocks), dim3(256), 0, s, dy, y, dv, w3, (long long)m_rows, (int)(n_cols / 4), dx_out, (long long)ldx, p0, p1);
    if (int rc = launch_ok("relu6_bwd")) return rc;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n_cols + 63) / 64), dim3(1024), 0, s, p0, waves, (int)n_cols, dbias_out);
    if (outer) hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n_cols + 63) / 64), dim3(1024), 0, s, p1, waves, (int)n_cols, dw3_out);
    return launch_ok("relu6_bwd reduce");
}

namespace {
struct RowsWs { size_t keys_in, keys_out, samp_in, samp_out, head, tail, carry, sort_tmp, sort_tmp_bytes, total; };
unsigned sort_bits(int64_t n_rows) { unsigned b = 1; while ((1ull << b) <= (unsigned long long)n_rows) ++b; return b; }   // keys in [0, n_rows]
int rows_ws(int64_t n_pairs, int32_t ncol, int64_t n_rows, RowsWs &w) {
    const size_t n = (size_t)n_pairs, chunks = (n + kChunk - 1) / kChunk;
    size_t tmp = 0;
    uint32_t *nk = nullptr;
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, nk, nk, nk, nk, n, 0u, sort_bits(n_rows), (hipStream_t)0);
    if (e != hipSuccess) return fail2(UAVAGENT_E_HIP, std::string("rows_grad: rocprim size query: ") + hipGetErrorString(e));
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += up256(bytes); return o; };
    w.keys_in = take(n * 4); w.keys_out = take(n * 4); w.samp_in = take(n * 4); w.samp_out = take(n * 4);
    w.head = take(chunks * 4); w.tail = take(chunks * 4); w.carry = take(chunks * 2 * (size_t)ncol * 4);
    w.sort_tmp = take(tmp); w.sort_tmp_bytes = tmp; w.total = off;
    return UAVAGENT_OK;
}
}  // namespace

extern "C" size_t uavagent_rows_grad_workspace_bytes(int64_t m_rows, int32_t k, int32_t n_cols_total, int64_t n_rows) {
    RowsWs w;
    if (m_rows < 1 || k < 1 || n_cols_total < 4 || n_rows < 1) return 0;
    if (rows_ws(m_rows * k, n_cols_total, n_rows, w)) return 0;
    return w.total;
}

// The two halves of uavagent_rows_grad_f32 (ABI 4).  The sort needs only idx, which a learner has long before g exists: it can run
// beside the backward pass (another stream), and serve several sums over the same samples (one per trunk).
extern "C" int uavagent_rows_grad_sort(const int64_t *idx, int64_t m_rows, int32_t k, int32_t n_cols_total, int64_t n_rows, void *workspace,
                                       size_t workspace_bytes, void *stream) {
    if (m_rows < 1 || k < 1 || k > 64 || n_cols_total < 4 || n_cols_total > 512 || (n_cols_total & 3) || n_rows < 1)
        return fail2(UAVAGENT_E_INVALID, "rows_grad: need m_rows >= 1, 1 <= k <= 64, h a multiple of 4 in [4, 256], 1 or 2 tables");
    if (!idx || !workspace) return fail2(UAVAGENT_E_INVALID, "rows_grad: null pointer");
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return fail2(UAVAGENT_E_INVALID, "rows_grad: g and the tables must be 16-byte aligned, the workspace 256-byte aligned");
    const long long n_pairs = (long long)m_rows * k;
    if (n_pairs > 0x7FFFFFFFll || (unsigned long long)n_rows >= 0xFFFFFFF0ull) return fail2(UAVAGENT_E_INVALID, "rows_grad: too many pairs / rows for 32-bit keys");
    RowsWs w;
    if (int rc = rows_ws(n_pairs, n_cols_total, n_rows, w)) return rc;
    if (workspace_bytes < w.total) return fail2(UAVAGENT_E_INVALID, "rows_grad: workspace smaller than uavagent_rows_grad_workspace_bytes()");
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    uint32_t *keys_in = reinterpret_cast<uint32_t *>(ws + w.keys_in), *keys_out = reinterpret_cast<uint32_t *>(ws + w.keys_out);
    uint32_t *samp_in = reinterpret_cast<uint32_t *>(ws + w.samp_in), *samp_out = reinterpret_cast<uint32_t *>(ws + w.samp_out);
    hipLaunchKernelGGL(rows_keys_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const long long *>(idx),
                       n_pairs, (int)k, (uint32_t)n_rows, keys_in, samp_in);
    if (int rc = launch_ok("rows_grad keys")) return rc;
    size_t tmp = w.sort_tmp_bytes;
    const hipError_t e = rocprim::radix_sort_pairs(ws + w.sort_tmp, tmp, keys_in, keys_out, samp_in, samp_out, (size_t)n_pairs, 0u, sort_bits(n_rows), s);
    if (e != hipSuccess) return fail2(UAVAGENT_E_HIP, std::string("rows_grad: rocprim radix_sort_pairs: ") + hipGetErrorString(e));
    return UAVAGENT_OK;
}

extern "C" int uavagent_rows_grad_sums_f32(const float *g, int64_t m_rows, int32_t k, int32_t h, int32_t n_tables, int64_t n_rows,
                                           float *dw0_out, float *dw1_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (m_rows < 1 || k < 1 || k > 64 || h < 4 || h > 256 || (h & 3) || (n_tables != 1 && n_tables != 2) || n_rows < 1)
        return fail2(UAVAGENT_E_INVALID, "rows_grad: need m_rows >= 1, 1 <= k <= 64, h a multiple of 4 in [4, 256], 1 or 2 tables");
    if (!g || !dw0_out || (n_tables == 2 && !dw1_out) || !workspace) return fail2(UAVAGENT_E_INVALID, "rows_grad: null pointer");
    if (!al16(g) || !al16(dw0_out) || !al16(dw1_out) || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail2(UAVAGENT_E_INVALID, "rows_grad: g and the tables must be 16-byte aligned, the workspace 256-byte aligned");
    const long long n_pairs = (long long)m_rows * k;
    if (n_pairs > 0x7FFFFFFFll || (unsigned long long)n_rows >= 0xFFFFFFF0ull) return fail2(UAVAGENT_E_INVALID, "rows_grad: too many pairs / rows for 32-bit keys");
    const int ncol = h * n_tables;
    RowsWs w;
    if (int rc = rows_ws(n_pairs, ncol, n_rows, w)) return rc;
    if (workspace_bytes < w.total) return fail2(UAVAGENT_E_INVALID, "rows_grad: workspace smaller than uavagent_rows_grad_workspace_bytes()");
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    const uint32_t *keys_out = reinterpret_cast<const uint32_t *>(ws + w.keys_out), *samp_out = reinterpret_cast<const uint32_t *>(ws + w.samp_out);
    uint32_t *head = reinterpret_cast<uint32_t *>(ws + w.head), *tail = reinterpret_cast<uint32_t *>(ws + w.tail);
    float *carry = reinterpret_cast<float *>(ws + w.carry);
    // rows nobody touches keep a zero gradient
    hipError_t e = hipMemsetAsync(dw0_out, 0, (size_t)n_rows * h * sizeof(float), s);
    if (e == hipSuccess && n_tables == 2) e = hipMemsetAsync(dw1_out, 0, (size_t)n_rows * h * sizeof(float), s);
    if (e != hipSuccess) return fail2(UAVAGENT_E_HIP, std::string("rows_grad: memset: ") + hipGetErrorString(e));
    const long long chunks = (n_pairs + kChunk - 1) / kChunk;
    const unsigned blocks = (unsigned)((chunks + 3) / 4);
    hipLaunchKernelGGL(rows_sum_stage_a, dim3(blocks), dim3(256), 0, s, keys_out, samp_out, n_pairs, (uint32_t)n_rows, g, (int)(ncol / 4),
                       (int)(h / 4), dw0_out, n_tables == 2 ? dw1_out : dw0_out, carry, head, tail);
    if (int rc = launch_ok("rows_grad stage A")) return rc;
    hipLaunchKernelGGL(rows_sum_stage_b, dim3(blocks), dim3(256), 0, s, head, tail, chunks, carry, (int)(ncol / 4), (int)(h / 4), dw0_out,
                       n_tables == 2 ? dw1_out : dw0_out);
    return launch_ok("rows_grad stage B");
}

extern "C" int uavagent_rows_grad_f32(const int64_t *idx, const float *g, int64_t m_rows, int32_t k, int32_t h, int32_t n_tables,
                                      int64_t n_rows, float *dw0_out, float *dw1_out, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    if (h < 4 || h > 256 || (h & 3) || (n_tables != 1 && n_tables != 2))
        return fail2(UAVAGENT_E_INVALID, "rows_grad: need m_rows >= 1, 1 <= k <= 64, h a multiple of 4 in [4, 256], 1 or 2 tables");
    if (!idx || !g || !dw0_out || (n_tables == 2 && !dw1_out) || !workspace) return fail2(UAVAGENT_E_INVALID, "rows_grad: null pointer");
    if (!al16(g) || !al16(dw0_out) || !al16(dw1_out) || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail2(UAVAGENT_E_INVALID, "rows_grad: g and the tables must be 16-byte aligned, the workspace 256-byte aligned");
    if (int rc = uavagent_rows_grad_sort(idx, m_rows, k, h * n_tables, n_rows, workspace, workspace_bytes, stream)) return rc;
    return uavagent_rows_grad_sums_f32(g, m_rows, k, h, n_tables, n_rows, dw0_out, dw1_out, workspace, workspace_bytes, stream);
}

extern "C" int uavagent_rowdot_f32(const float *y, const float *w, const float *bias, int64_t m_rows, int32_t n_cols, float *out,
                                   void *stream) {
    if (m_rows < 0 || n_cols < 4 || n_cols > 256 || (n_cols & 3)) return fail2(UAVAGENT_E_INVALID, "rowdot: n_cols must be a multiple of 4 in [4, 256]");
    if (m_rows == 0) return UAVAGENT_OK;
    if (!y || !w || !out || !al16(y) || !al16(w)) return fail2(UAVAGENT_E_INVALID, "rowdot: null or unaligned (16 B) pointer");
    const long long blocks = (m_rows + 3) / 4 < 2048 ? (m_rows + 3) / 4 : 2048;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, w, bias, (long long)m_rows, (int)(n_cols / 4), out);
    return launch_ok("rowdot");
}

extern "C" int uavagent_nstep_returns_f32(const float *rewards, const float *bootstrap, int64_t n_envs, int32_t n_steps, float gamma,
                                          float *out, void *stream) {
    if (n_envs < 0 || n_steps < 0) return fail2(UAVAGENT_E_INVALID, "nstep_returns: negative size");
    if (n_envs == 0 || n_steps == 0) return UAVAGENT_OK;
    if (!rewards || !bootstrap || !out) return fail2(UAVAGENT_E_INVALID, "nstep_returns: null pointer");
    hipLaunchKernelGGL(nstep_returns_kernel, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, bootstrap,
                       (long long)n_envs, (int)n_steps, gamma, out);
    return launch_ok("nstep_returns");
}

extern "C" int uavagent_rmsprop_tf1(float *w, float *ms, const float *g, int64_t n, float lr, float decay, float eps, float g_scale,
                                    void *stream) {
    if (n < 0) return fail2(UAVAGENT_E_INVALID, "rmsprop: negative n");
    if (n == 0) return UAVAGENT_OK;
    if (!w || !ms || !g || !al16(w) || !al16(ms) || !al16(g)) return fail2(UAVAGENT_E_INVALID, "rmsprop: null or unaligned (16 B) pointer");
    const long long n4 = (n + 3) / 4;
    hipLaunchKernelGGL(rmsprop_tf1_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ms, g, (long long)n, lr,
                       decay, eps, g_scale);
    return launch_ok("rmsprop");
}
