// libuavagent.so, part 3: float32 MFMA GEMMs for the dense layers of the MLP actor-critic (interface: include/uavagent.h, ABI 3).
//
// The reference's hidden layers are 200 wide (main.py:147-148,153) and its policy head 625 = 5^4 (mobile_env.py:104), so every
// dense GEMM of an A2C update on M = rollout x envs samples has a 200 in it:
//   forwards   [M, 200] x [200, 200]            (critic layer 2; the actor's come from the rollout)          main.py:148,153
//   dX         [M, 625] x [625, 200],  [M, 200] x [200, 200]   (backwards through a layer's weights)
//   dW         [200, M] x [M, 625],    [200, M] x [M, 200]     (weight gradients: a 409 600-deep reduction at BASELINE config 3)
// rocBLAS / hipBLASLt run them at 45-108 TFLOP/s of the 157 TFLOP/s f32-MFMA peak even with TunableOp's picks
// (profiles/r02f_gemm_tunableop.txt): 200 is 6.25 tiles of 32 and 12.5 of 16, and the dW shapes are all reduction.  These kernels
// are written for exactly those shapes: v_mfma_f32_16x16x4_f32 (exact float32: a k-ordered fmaf chain, MI355X_MICROARCH.md, Matrix
// cores), 200 padded to 13 blocks of 16 (4 % instead of 12-28 %), the dW reduction split over all CUs with a SECOND, ordered pass
// (no float atomics: bit-reproducible, which the checkpoint test relies on), relu6 / bias / relu6-mask epilogues fused, and the bias
// gradient taken from the dW kernel itself (a column of ones appended to the left operand in LDS).
//
// Fragment maps used below (cdna_hip_programming.md section 3): lane l = (r = l & 15, q = l >> 4);
//   A operand = A[row r][k = q],  B operand = B[k = q][col r],  accumulator register t = C[row 4 q + t][col r].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/uavagent.h"
#include "agent_common.h"

namespace {

int fail3(int code, const std::string &msg) { return uavagent_internal::fail(code, msg); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kNP = 208;        // 200 padded to 13 column blocks; 208 = 6 x 32 + 16 floats: rows of an [k][208] LDS tile start 16 banks
                                // apart, so the two k-rows a 32-lane group of a fragment read touches never share a bank
constexpr int kRB = 13;         // 16-row blocks of the 200-wide dimension

// =====================================================================================================================
// dW:  C[i, j] = sum_m A[m, i] * B[m, j]   (A [M, I <= 200], B [M, J]; C [I, J]), plus dbias[j] = sum_m B[m, j].
//
// Both operands are streamed once, in 32-row chunks that are contiguous in memory ([32, 200] floats = 25.6 KB), through a double
// buffered LDS image [k][208].  One workgroup = one (split of M, tile of J) and 4 wavefronts, one per SIMD (the accumulators alone are
// 172 / 132 VGPRs: this is a 512-register kernel), the 13 x NCB output blocks dealt out so that every SIMD gets the same number of
// MFMAs: wave w owns all 13 row blocks of NCB / 4 column blocks, and the NCB % 4 left-over column blocks are cut by rows (NCB = 13:
// 39 + 4 / 3 / 3 / 3 blocks; NCB = 10: 26 + 7 / 6 / 7 / 6).  13 is prime: any rectangular split leaves a SIMD with 49 of 169 blocks
// (16 % idle).  The accumulators live for the whole split; at the end every workgroup writes ONE slab in fragment order (coalesced
// 16-byte stores) and gemm_tn_reduce adds the slabs in split order and scatters to C.  A's padding column 200 holds 1.0 in LDS, so row 200 of the product is the column sum of B: the
// bias gradient of the layer, for free.
// =====================================================================================================================
template <int NCB> struct TnPlan {
    static constexpr int CW = NCB / 4;                  // whole column blocks per wave
    static constexpr int REM = NCB % 4;                 // left-over column blocks: cut by rows
    static_assert(REM == 1 || REM == 2, "TnPlan: 13 and 10 column blocks are the tile widths built");
    static constexpr int NX = (REM == 1) ? 4 : 7;       // most left-over blocks a wave gets
    static constexpr int NBLK = kRB * CW + NX;          // accumulator blocks per wave (the last ones unused on some waves)
    static constexpr int BJP = (NCB * 16) % 32 == 16 ? NCB * 16 : NCB * 16 + 16;   // LDS row stride of the B tile: == 16 (mod 32)
    __host__ __device__ static constexpr int xcol(int wq) { return 4 * CW + (REM == 1 ? 0 : (wq >> 1)); }
    __host__ __device__ static constexpr int xr0(int wq) { return REM == 1 ? (wq == 0 ? 0 : 1 + 3 * wq) : ((wq & 1) ? 7 : 0); }
    __host__ __device__ static constexpr int nx(int wq) { return REM == 1 ? (wq == 0 ? 4 : 3) : ((wq & 1) ? 6 : 7); }
};

constexpr int kTnBK = 32;       // rows of M per chunk = 8 k-steps of 4

template <int NCB, bool BVEC>
__global__ __launch_bounds__(256, 1) void gemm_tn_kernel(const float *__restrict__ A, int n_i, const float *__restrict__ B, long long ldb,
                                                          int n_j, long long M, long long rows_per_split, int n_jt, int n_split,
                                                          f32x4 *__restrict__ slabs) {
    using P = TnPlan<NCB>;
    constexpr int CW = P::CW, NX = P::NX, NBLK = P::NBLK, BJP = P::BJP;
    constexpr int A_TILE = kTnBK * kNP, B_TILE = kTnBK * BJP;
    __shared__ __attribute__((aligned(16))) float lds[2 * (A_TILE + B_TILE)];
    float *const sA = lds, *const sB = lds + 2 * A_TILE;

    const int tid = threadIdx.x, lane = tid & 63, wq = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches below)
    const int r = lane & 15, q = lane >> 4;
    // workgroup -> (split, J tile): the J tiles of one split read the same rows of A, so they sit on one XCD (ids = xcd mod 8) and
    // share its L2 (speed only; any mapping is correct)
    int jt, split;
    {
        const int id = blockIdx.x;
        if ((n_split & 7) == 0) { const int xcd = id & 7, w = id >> 3; jt = w % n_jt; split = (w / n_jt) * 8 + xcd; }
        else { jt = id % n_jt; split = id / n_jt; }
    }
    const int j0 = jt * NCB * 16;
    const long long m0 = (long long)split * rows_per_split;
    const long long m1 = (m0 + rows_per_split < M) ? m0 + rows_per_split : M;
    const int n_chunks = (m1 > m0) ? (int)((m1 - m0 + kTnBK - 1) / kTnBK) : 0;

    // ---- LDS image: zero everything once (padding columns must hold finite values), then the column of ones ----
    for (int i = tid; i < 2 * (A_TILE + B_TILE); i += 256) lds[i] = 0.0f;
    __syncthreads();
    if (tid < 2 * kTnBK) sA[(tid >> 5) * A_TILE + (tid & 31) * kNP + n_i] = 1.0f;     // n_i <= 200 (host)

    // ---- per-thread staging plan (the same every chunk).  A: a chunk is 32 x n_i CONTIGUOUS floats, float4 number idx of it goes to
    // LDS row idx / (n_i / 4).  B: 32 rows of this J tile, float4s (BVEC) or floats. ----
    constexpr int NAL = 7;                                               // 32 x 200 / 4 = 1600 float4 over 256 threads
    const int a4 = n_i >> 2;
    int a_lds[NAL];
#pragma unroll
    for (int i = 0; i < NAL; ++i) { const int idx = tid + 256 * i, row = idx / a4; a_lds[i] = row * kNP + (idx - row * a4) * 4; }
    const int bj = (n_j - j0 < NCB * 16) ? n_j - j0 : NCB * 16;          // valid columns of this J tile
    constexpr int NBL = BVEC ? 7 : (kTnBK * NCB * 16 + 255) / 256;        // staging loads per thread for B
    const int b4 = BVEC ? (bj >> 2) : NCB * 16;                          // items per row (float4s / floats)
    int b_row[BVEC ? NBL : 1], b_c[BVEC ? NBL : 1];
    if (BVEC) {
#pragma unroll
        for (int i = 0; i < NBL; ++i) { const int idx = tid + 256 * i; b_row[BVEC ? i : 0] = idx / b4; b_c[BVEC ? i : 0] = idx - (idx / b4) * b4; }
    }

    float4 ra[NAL];
    float4 rbv[BVEC ? NBL : 1];
    float rbs[BVEC ? 1 : NBL];
    auto load_chunk = [&](int c) {
        const long long mb = m0 + (long long)c * kTnBK;
        const int rows = (m1 - mb < kTnBK) ? (int)(m1 - mb) : kTnBK;     // rows of this chunk that exist
        const float *ga = A + mb * (long long)n_i;
#pragma unroll
        for (int i = 0; i < NAL; ++i) {
            const int idx = tid + 256 * i;
            ra[i] = (idx < rows * a4) ? *reinterpret_cast<const float4 *>(ga + idx * 4) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            if (BVEC) {
                const int br = b_row[BVEC ? i : 0];
                rbv[BVEC ? i : 0] = (br < rows) ? *reinterpret_cast<const float4 *>(B + (mb + br) * ldb + j0 + b_c[BVEC ? i : 0] * 4) : float4{0.f, 0.f, 0.f, 0.f};
            } else {
                const int idx = tid + 256 * i, br = idx / (NCB * 16), bc = idx - br * (NCB * 16);
                rbs[BVEC ? 0 : i] = (br < rows && bc < bj) ? B[(mb + br) * ldb + j0 + bc] : 0.0f;
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float *dA = sA + buf * A_TILE, *dB = sB + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < NAL; ++i)
            if (tid + 256 * i < kTnBK * a4) *reinterpret_cast<float4 *>(dA + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            if (BVEC) {
                if (b_row[BVEC ? i : 0] < kTnBK) *reinterpret_cast<float4 *>(dB + b_row[BVEC ? i : 0] * BJP + b_c[BVEC ? i : 0] * 4) = rbv[BVEC ? i : 0];
            } else {
                const int idx = tid + 256 * i, br = idx / (NCB * 16), bc = idx - br * (NCB * 16);
                if (br < kTnBK) dB[br * BJP + bc] = rbs[BVEC ? 0 : i];
            }
        }
    };

    f32x4 acc[kRB][CW], accx[NX];
#pragma unroll
    for (int i = 0; i < kRB; ++i)
#pragma unroll
        for (int c = 0; c < CW; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NX; ++i) accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int xcol = P::xcol(wq), xr0 = P::xr0(wq), nx = P::nx(wq);

    if (n_chunks > 0) { load_chunk(0); store_chunk(0); }
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) load_chunk(c + 1);                           // in flight behind this chunk's MFMAs
        const float *tA = sA + buf * A_TILE + q * kNP + r, *tB = sB + buf * B_TILE + q * BJP + r;
#pragma unroll
        for (int ks = 0; ks < kTnBK / 4; ++ks) {
            const float *pa = tA + ks * 4 * kNP, *pb = tB + ks * 4 * BJP;    // this lane's k row of the k-step
            float a[kRB], b[CW];
#pragma unroll
            for (int i = 0; i < kRB; ++i) a[i] = pa[i * 16];
#pragma unroll
            for (int cc = 0; cc < CW; ++cc) b[cc] = pb[(wq * CW + cc) * 16];
            const float bx = pb[xcol * 16];
            float ax[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) ax[i] = pa[((i < nx) ? xr0 + i : 0) * 16];
#pragma unroll
            for (int i = 0; i < kRB; ++i)
#pragma unroll
                for (int cc = 0; cc < CW; ++cc) acc[i][cc] = MFMA16(a[i], b[cc], acc[i][cc]);
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (i < nx) accx[i] = MFMA16(ax[i], bx, accx[i]);            // (wave-uniform branch)
        }
        if (c + 1 < n_chunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- one slab per workgroup, in fragment order (coalesced 16-byte stores) ----
    f32x4 *dst = slabs + (((long long)split * n_jt + jt) * 4 + wq) * (long long)(NBLK * 64) + lane;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) dst[blk * 64] = (blk < kRB * CW) ? acc[blk / CW][blk % CW] : accx[blk - kRB * CW];
}

// Second pass of dW: one thread per accumulator fragment (jt, wave, block, lane) adds the slabs in split order (fixed order: the
// result does not depend on scheduling) and scatters its 4 values; row n_i of the product (the ones column) goes to dbias.
template <int NCB>
__global__ __launch_bounds__(256) void gemm_tn_reduce(const f32x4 *__restrict__ slabs, int n_split, int n_jt, int n_i, int n_j, float *__restrict__ C,
                                                       long long ldc, float *__restrict__ dbias) {
    using P = TnPlan<NCB>;
    constexpr int CW = P::CW, NBLK = P::NBLK;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int per_slab = 4 * NBLK * 64;
    if (t >= n_jt * per_slab) return;
    const int jt = t / per_slab, rem = t - jt * per_slab;
    const int wq = rem / (NBLK * 64), rem2 = rem - wq * (NBLK * 64), blk = rem2 >> 6, lane = rem2 & 63;
    const f32x4 *src = slabs + (long long)jt * per_slab + rem;
    const long long stride = (long long)n_jt * per_slab;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= n_split; s += 8) {
        f32x4 w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = src[(s + i) * stride];
#pragma unroll
        for (int i = 0; i < 8; ++i) v += w[i];
    }
    for (; s < n_split; ++s) v += src[s * stride];
    int rb, cb;
    if (blk < kRB * CW) { rb = blk / CW; cb = wq * CW + blk % CW; }
    else { const int i = blk - kRB * CW; if (i >= P::nx(wq)) return; rb = P::xr0(wq) + i; cb = P::xcol(wq); }
    const int col = jt * NCB * 16 + cb * 16 + (lane & 15);
    if (col >= n_j) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = rb * 16 + 4 * (lane >> 4) + k;
        if (row < n_i) C[row * ldc + col] = v[k];
        else if (row == n_i && dbias != nullptr) dbias[col] = v[k];
    }
}

// =====================================================================================================================
// Rows GEMM:  C[m, n] = epilogue( sum_k A[m, k] * Wop[k, n] ),  n < N <= 208, K any.
//   NT = false:  Wop = W        (W [K, N] row-major: a forward layer,  x @ W)
//   NT = true:   Wop = W^T      (W [N, K] row-major: backwards through a layer,  dy @ W^T)
//   EPI 0 none;  1: + bias[n], then relu6 when asked (tf.nn.relu6);  2: relu6 backwards, C = (0 < H[m, n] < 6) ? sum : 0 with H the
//   layer's forward output (main.py:147-148,153).
// One workgroup = 128 rows x all N columns, 4 wavefronts of 32 rows x 13 column blocks (104 accumulator VGPRs), two workgroups per
// CU so that one computes while the other loads or stores.  K is walked in chunks of 20 (5 k-steps) through double-buffered LDS:
// the A tile as [row][22] (stride 22: the 16 rows x 2 k of a fragment read fall on 32 different banks), the W tile as [k][208] (NT
// false) or [n][22] (NT true).  VEC = false loads dword by dword (row strides that are not a multiple of 4 floats: the 625-wide
// policy head).
// =====================================================================================================================
constexpr int kRowsBM = 128, kRowsBK = 20, kRowsLD = 22;
constexpr int kRowsATile = kRowsBM * kRowsLD;                                   // 2816 floats
constexpr int kRowsWTile = (kRowsBK * kNP > kNP * kRowsLD) ? kRowsBK * kNP : kNP * kRowsLD;   // 4576 floats

template <bool NT, bool VEC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw,
                                                            int K, int N, long long M, const float *__restrict__ bias, int relu6,
                                                            const float *__restrict__ H, long long ldh, float *__restrict__ C, long long ldc) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (kRowsATile + kRowsWTile)];
    float *const sA = lds, *const sW = lds + 2 * kRowsATile;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const long long m0 = (long long)blockIdx.x * kRowsBM;

    for (int i = tid; i < 2 * (kRowsATile + kRowsWTile); i += 256) lds[i] = 0.0f;   // padding rows / columns must be finite
    __syncthreads();

    // staging plan.  VEC: A = 128 rows x 5 float4 (3 loads per thread), W = 1000 float4 (4 loads); else 10 + 16 dword loads.
    constexpr int NA = VEC ? 3 : 10, NW = VEC ? 4 : 17;
    constexpr int APR = VEC ? kRowsBK / 4 : kRowsBK;            // items per A row
    int a_row[NA], a_c[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { const int idx = tid + 256 * i; a_row[i] = idx / APR; a_c[i] = idx - a_row[i] * APR; }
    // W items: NT false: [k (20)][N / 4 or N];  NT true: [n (N)][5 or 20]
    const int wpr = NT ? APR : (VEC ? (N >> 2) : N);            // items per W-tile row
    const int w_rows = NT ? N : kRowsBK;
    int w_row[NW], w_c[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) { const int idx = tid + 256 * i; w_row[i] = idx / wpr; w_c[i] = idx - w_row[i] * wpr; }

    float4 va[VEC ? NA : 1], vw[VEC ? NW : 1];
    float fa[VEC ? 1 : NA], fw[VEC ? 1 : NW];
    auto load_chunk = [&](int c) {
        const int k0 = c * kRowsBK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const long long m = m0 + a_row[i];
            if (VEC) {
                const bool ok = (a_row[i] < kRowsBM) && (m < M) && (k0 + a_c[i] * 4 < K);    // K % 4 == 0 on this path
                va[VEC ? i : 0] = ok ? *reinterpret_cast<const float4 *>(A + m * lda + k0 + a_c[i] * 4) : float4{0.f, 0.f, 0.f, 0.f};
            } else {
                const bool ok = (a_row[i] < kRowsBM) && (m < M) && (k0 + a_c[i] < K);
                fa[VEC ? 0 : i] = ok ? A[m * lda + k0 + a_c[i]] : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (NT) {      // W [N, K]: row = n, column = k0 + ...
                if (VEC) {
                    const bool ok = (w_row[i] < w_rows) && (k0 + w_c[i] * 4 < K);
                    vw[VEC ? i : 0] = ok ? *reinterpret_cast<const float4 *>(W + (long long)w_row[i] * ldw + k0 + w_c[i] * 4) : float4{0.f, 0.f, 0.f, 0.f};
                } else {
                    const bool ok = (w_row[i] < w_rows) && (k0 + w_c[i] < K);
                    fw[VEC ? 0 : i] = ok ? W[(long long)w_row[i] * ldw + k0 + w_c[i]] : 0.0f;
                }
            } else {       // W [K, N]: row = k0 + ..., column = n
                if (VEC) {
                    const bool ok = (w_row[i] < w_rows) && (k0 + w_row[i] < K);
                    vw[VEC ? i : 0] = ok ? *reinterpret_cast<const float4 *>(W + (long long)(k0 + w_row[i]) * ldw + w_c[i] * 4) : float4{0.f, 0.f, 0.f, 0.f};
                } else {
                    const bool ok = (w_row[i] < w_rows) && (k0 + w_row[i] < K);
                    fw[VEC ? 0 : i] = ok ? W[(long long)(k0 + w_row[i]) * ldw + w_c[i]] : 0.0f;
                }
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float *dA = sA + buf * kRowsATile, *dW = sW + buf * kRowsWTile;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (a_row[i] < kRowsBM) {
                if (VEC) {      // row stride 22 floats = 88 B: 8-byte aligned, two 8-byte stores
                    float2 *d = reinterpret_cast<float2 *>(dA + a_row[i] * kRowsLD + a_c[i] * 4);
                    d[0] = float2{va[VEC ? i : 0].x, va[VEC ? i : 0].y}; d[1] = float2{va[VEC ? i : 0].z, va[VEC ? i : 0].w};
                } else dA[a_row[i] * kRowsLD + a_c[i]] = fa[VEC ? 0 : i];
            }
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (w_row[i] < w_rows) {
                if (NT) {
                    if (VEC) {
                        float2 *d = reinterpret_cast<float2 *>(dW + w_row[i] * kRowsLD + w_c[i] * 4);
                        d[0] = float2{vw[VEC ? i : 0].x, vw[VEC ? i : 0].y}; d[1] = float2{vw[VEC ? i : 0].z, vw[VEC ? i : 0].w};
                    } else dW[w_row[i] * kRowsLD + w_c[i]] = fw[VEC ? 0 : i];
                } else {
                    if (VEC) *reinterpret_cast<float4 *>(dW + w_row[i] * kNP + w_c[i] * 4) = vw[VEC ? i : 0];
                    else dW[w_row[i] * kNP + w_c[i]] = fw[VEC ? 0 : i];
                }
            }
    };

    f32x4 acc[2][kRB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < kRB; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int n_chunks = (K + kRowsBK - 1) / kRowsBK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) load_chunk(c + 1);
        const float *tA = sA + buf * kRowsATile + (wave * 32 + r) * kRowsLD + q;
        const float *tW = NT ? (sW + buf * kRowsWTile + r * kRowsLD + q) : (sW + buf * kRowsWTile + q * kNP + r);
        const int kleft = K - c * kRowsBK;
        const int nks = (kleft >= kRowsBK) ? kRowsBK / 4 : (kleft + 3) / 4;
#pragma unroll
        for (int ks = 0; ks < kRowsBK / 4; ++ks) {
            if (ks < nks) {
                const float a0 = tA[ks * 4], a1 = tA[16 * kRowsLD + ks * 4];
                float b[kRB];
#pragma unroll
                for (int cb = 0; cb < kRB; ++cb) b[cb] = NT ? tW[cb * 16 * kRowsLD + ks * 4] : tW[ks * 4 * kNP + cb * 16];
#pragma unroll
                for (int cb = 0; cb < kRB; ++cb) { acc[0][cb] = MFMA16(a0, b[cb], acc[0][cb]); acc[1][cb] = MFMA16(a1, b[cb], acc[1][cb]); }
            }
        }
        if (c + 1 < n_chunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulator register t of block (rb, cb) = C[m0 + wave * 32 + rb * 16 + 4 q + t][cb * 16 + r] ----
#pragma unroll
    for (int cb = 0; cb < kRB; ++cb) {
        const int col = cb * 16 + r;
        if (col < N) {
            const float bv = (EPI == 1 && bias != nullptr) ? bias[col] : 0.0f;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long long m = m0 + wave * 32 + rb * 16 + 4 * q + t;
                    if (m < M) {
                        float v = acc[rb][cb][t];
                        if (EPI == 1) { v += bv; if (relu6) v = fminf(fmaxf(v, 0.0f), 6.0f); }
                        if (EPI == 2) { const float h = H[m * ldh + col]; v = (h > 0.0f && h < 6.0f) ? v : 0.0f; }
                        C[m * ldc + col] = v;
                    }
                }
        }
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct TnShape { int ncb, n_jt, n_split; long long rows_per_split; size_t ws_bytes; };
// The split of M: one workgroup per CU (256 on MI355X) in total; every split a whole number of 32-row chunks.
TnShape tn_shape(long long M, int n_j) {
    TnShape s;
    s.ncb = (n_j <= 208) ? 13 : 10;
    s.n_jt = (n_j <= 208) ? 1 : (n_j + 159) / 160;
    int want = 256 / s.n_jt;
    if (want < 8) want = 8;
    const long long chunks = (M + kTnBK - 1) / kTnBK;
    long long per = (chunks + want - 1) / want;
    if (per < 1) per = 1;
    s.rows_per_split = per * kTnBK;
    s.n_split = (int)((M + s.rows_per_split - 1) / s.rows_per_split);
    if (s.n_split < 1) s.n_split = 1;
    const int nblk = (s.ncb == 13) ? TnPlan<13>::NBLK : TnPlan<10>::NBLK;
    s.ws_bytes = (size_t)s.n_split * s.n_jt * 4 * nblk * 64 * sizeof(f32x4);
    return s;
}
}  // namespace

extern "C" size_t uavagent_gemm_tn_workspace_bytes(int64_t m_rows, int32_t n_j) {
    if (m_rows < 0 || n_j < 1) return 0;
    return tn_shape(m_rows, n_j).ws_bytes;
}

extern "C" int uavagent_gemm_tn_f32(const float *a, const float *b, int64_t m_rows, int32_t n_i, int32_t n_j, int64_t ldb, float *c, int64_t ldc,
                                    float *dbias_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!a || !b || !c || !workspace) return fail3(UAVAGENT_E_INVALID, "gemm_tn: null pointer");
    if (m_rows < 1 || n_i < 4 || n_i > 200 || (n_i & 3) || n_j < 1 || n_j > 640 || ldb < n_j || ldc < n_j)
        return fail3(UAVAGENT_E_INVALID, "gemm_tn: need m_rows >= 1, n_i % 4 == 0 in [4, 200], 1 <= n_j <= 640, ldb >= n_j, ldc >= n_j");
    if (!aligned16(a) || !aligned16(workspace)) return fail3(UAVAGENT_E_INVALID, "gemm_tn: a and workspace must be 16-byte aligned");
    const TnShape s = tn_shape(m_rows, n_j);
    if (workspace_bytes < s.ws_bytes) return fail3(UAVAGENT_E_INVALID, "gemm_tn: workspace smaller than uavagent_gemm_tn_workspace_bytes()");
    hipStream_t st = (hipStream_t)stream;
    f32x4 *slabs = reinterpret_cast<f32x4 *>(workspace);
    const bool bvec = aligned16(b) && (ldb % 4 == 0) && (n_j % 4 == 0);
    const dim3 grid((unsigned)(s.n_split * s.n_jt)), blk(256);
    if (s.ncb == 13) {
        if (bvec) hipLaunchKernelGGL((gemm_tn_kernel<13, true>), grid, blk, 0, st, a, n_i, b, (long long)ldb, n_j, (long long)m_rows, s.rows_per_split, s.n_jt, s.n_split, slabs);
        else hipLaunchKernelGGL((gemm_tn_kernel<13, false>), grid, blk, 0, st, a, n_i, b, (long long)ldb, n_j, (long long)m_rows, s.rows_per_split, s.n_jt, s.n_split, slabs);
        const int n_thr = s.n_jt * 4 * TnPlan<13>::NBLK * 64;
        hipLaunchKernelGGL((gemm_tn_reduce<13>), dim3((n_thr + 255) / 256), dim3(256), 0, st, slabs, s.n_split, s.n_jt, n_i, n_j, c, (long long)ldc, dbias_out);
    } else {
        if (bvec) hipLaunchKernelGGL((gemm_tn_kernel<10, true>), grid, blk, 0, st, a, n_i, b, (long long)ldb, n_j, (long long)m_rows, s.rows_per_split, s.n_jt, s.n_split, slabs);
        else hipLaunchKernelGGL((gemm_tn_kernel<10, false>), grid, blk, 0, st, a, n_i, b, (long long)ldb, n_j, (long long)m_rows, s.rows_per_split, s.n_jt, s.n_split, slabs);
        const int n_thr = s.n_jt * 4 * TnPlan<10>::NBLK * 64;
        hipLaunchKernelGGL((gemm_tn_reduce<10>), dim3((n_thr + 255) / 256), dim3(256), 0, st, slabs, s.n_split, s.n_jt, n_i, n_j, c, (long long)ldc, dbias_out);
    }
    if (hipGetLastError() != hipSuccess) return fail3(UAVAGENT_E_HIP, "gemm_tn: launch failed");
    return UAVAGENT_OK;
}

extern "C" int uavagent_gemm_rows_f32(const float *a, int64_t lda, const float *w, int64_t ldw, int32_t w_transposed, int64_t m_rows, int32_t k,
                                      int32_t n, const float *bias, int32_t relu6, const float *relu6_mask_h, int64_t ldh, float *c, int64_t ldc,
                                      void *stream) {
    if (!a || !w || !c) return fail3(UAVAGENT_E_INVALID, "gemm_rows: null pointer");
    if (m_rows < 1 || k < 1 || n < 1 || n > 208 || lda < k || ldc < n || ldw < (w_transposed ? k : n))
        return fail3(UAVAGENT_E_INVALID, "gemm_rows: need m_rows, k >= 1, 1 <= n <= 208, lda >= k, ldc >= n, ldw >= the row length of w");
    if (relu6_mask_h && (bias || relu6)) return fail3(UAVAGENT_E_INVALID, "gemm_rows: the relu6-mask epilogue excludes bias / relu6");
    if (relu6_mask_h && ldh < n) return fail3(UAVAGENT_E_INVALID, "gemm_rows: ldh < n");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = aligned16(a) && aligned16(w) && (lda % 4 == 0) && (ldw % 4 == 0) && (k % 4 == 0) && (n % 4 == 0);
    const dim3 grid((unsigned)((m_rows + kRowsBM - 1) / kRowsBM)), blk(256);
    const int epi = relu6_mask_h ? 2 : ((bias || relu6) ? 1 : 0);
#define UAV_ROWS(NT_, VEC_, EPI_) hipLaunchKernelGGL((gemm_rows_kernel<NT_, VEC_, EPI_>), grid, blk, 0, st, a, (long long)lda, w, (long long)ldw, \
        (int)k, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc)
#define UAV_ROWS_E(NT_, VEC_) do { if (epi == 0) UAV_ROWS(NT_, VEC_, 0); else if (epi == 1) UAV_ROWS(NT_, VEC_, 1); else UAV_ROWS(NT_, VEC_, 2); } while (0)
    if (w_transposed) { if (vec) UAV_ROWS_E(true, true); else UAV_ROWS_E(true, false); }
    else { if (vec) UAV_ROWS_E(false, true); else UAV_ROWS_E(false, false); }
#undef UAV_ROWS_E
#undef UAV_ROWS
    if (hipGetLastError() != hipSuccess) return fail3(UAVAGENT_E_HIP, "gemm_rows: launch failed");
    return UAVAGENT_OK;
}
