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
#include <atomic>
#include <stdlib.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <type_traits>

#include "../../include/uavagent.h"
#include "agent_common.h"

namespace {

int fail3(int code, const std::string &msg) { return uavagent_internal::fail(code, msg); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kNP = 208;        // 200 padded to 13 column blocks; 208 = 6 x 32 + 16 floats: rows of an [k][208] LDS tile start 16 banks
                                // apart, so the two k-rows a 32-lane group of a fragment read touches never share a bank
constexpr int kRB = 13;         // 16-row blocks of the 200-wide dimension

// Branch-free guarded staging.  LOAD: the address is made safe by a select and the raw value kept in registers (so the load stays in
// flight behind the MFMAs of the current chunk); STORE: the value is zeroed by a BIT MASK just before it goes to LDS.  (Written as
// `ok ? *p : 0`, or with a select on the loaded value, hipcc puts every load of the staging pass in a basic block of its own behind an
// exec branch -- CodeGenPrepare turns a select with a load operand back into a branch; masked right after the load, it waits for
// every load before the first MFMA.)
__device__ __forceinline__ float mask_f(float v, bool ok) { return __uint_as_float(__float_as_uint(v) & (ok ? 0xFFFFFFFFu : 0u)); }
__device__ __forceinline__ float4 mask_f4(float4 v, bool ok) { return float4{mask_f(v.x, ok), mask_f(v.y, ok), mask_f(v.z, ok), mask_f(v.w, ok)}; }
// "This value has arrived": waits for every outstanding load and hands the registers back as the asm's own outputs, so that hipcc no
// longer attaches a pending load to them.  Without it a value loaded ahead of a store loop (bias, the relu6 mask's H rows) drags an
// s_waitcnt vmcnt(0) into every iteration behind a branch -- which also waits for the PREVIOUS iteration's store: 16 serialised store
// round trips per workgroup (the epilogue of the first versions: 15 000 cycles of a 124 000-cycle workgroup, profiles/r03m_*).
__device__ __forceinline__ void arrived4(float4 &v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ float4 ldraw4(const float *p, bool ok, const float *safe) { return *reinterpret_cast<const float4 *>(ok ? p : safe); }
__device__ __forceinline__ float ldraw1(const float *p, bool ok, const float *safe) { return *(ok ? p : safe); }

// =====================================================================================================================
// dW:  C[i, j] = sum_m A[m, i] * B[m, j]   (A [M, I <= 200], B [M, J]; C [I, J]), plus dbias[j] = sum_m B[m, j].
//
// Both operands are streamed once, in 32-row chunks (a chunk of A is 32 x 200 CONTIGUOUS floats = 25.6 KB), through a double buffered
// LDS image [k][208] / [k][BJP].  One workgroup = one (split of M, tile of J) and 8 wavefronts, two per SIMD, every one running all 8
// k-steps of a chunk on its own share of the 13 x NCB output blocks.  The shares are built so that every wavefront issues the same
// number of MFMAs (13 is prime: any rectangular split of 13 x 13 leaves one wave with 49 of 169 blocks, 16 % idle):
//   plan 13 (J <= 208, 13 x 13 blocks): waves 0-3 rows 0-6 x 3 columns (21) + block (12, 12) on wave 0; waves 4-7 rows 7-12 x 3 columns
//            (18) + three blocks of column 12 each (rows 0-11): 22 / 21 MFMAs per k-step;
//   plan 20 (a 320-wide tile of a wider J, 13 x 20 blocks): wave pair (w, w + 4) shares 5 columns: two whole columns each and column
//            5 w + 2 cut by rows (0-6 / 7-12): 33 / 32.
// Fragments for k-step s + 1 are read from LDS while the MFMAs of k-step s issue.  The accumulators (<= 33 blocks = 132 VGPRs) live
// for the whole split; at the end every workgroup writes ONE slab in fragment order (coalesced 16-byte stores) and gemm_tn_reduce adds
// the slabs in split order and scatters to C.  A's padding column 200 holds 1.0 in LDS, so row 200 of the product is the column sum of
// B: the bias gradient of the layer, for free.
// =====================================================================================================================
template <int PLAN> struct TnPlan;
// Every wavefront of a plan has the SAME shape (NR x NC block rectangle + NX single blocks), only the origins differ, so the kernel has
// one code path (two shapes behind a wave-uniform branch cost hipcc twice the registers: it spilled).  The price: a few duplicate
// blocks (marked invalid, computed and ignored) on the waves that have fewer real ones.
template <> struct TnPlan<13> {          // J <= 208: 13 x 13 blocks, 169 real of 8 x 22 issued
    static constexpr int NCB = 13, BJP = 208, NR = 6, NC = 3, NX = 4, NBLKW = NR * NC + NX;
    static constexpr bool XONECOL = false;
    // waves 0-3: rows 0-5 x columns 3 w .. 3 w + 2, extras (6, 3 w + i) and, on wave 0, (12, 12)
    // waves 4-7: rows 7-12 x the same columns, extras (3 w + i, 12)
    __host__ __device__ static constexpr int r0(int w) { return (w >> 2) ? 7 : 0; }
    __host__ __device__ static constexpr int c0(int w) { return 3 * (w & 3); }
    __host__ __device__ static constexpr int xr(int w, int i) { return (w >> 2) ? 3 * (w & 3) + (i < 3 ? i : 0) : (i < 3 ? 6 : 12); }
    __host__ __device__ static constexpr int xc(int w, int i) { return (w >> 2) ? 12 : (i < 3 ? 3 * (w & 3) + i : 12); }
    __host__ __device__ static constexpr bool xvalid(int w, int i) { return i < 3 || w == 0; }
};
template <> struct TnPlan<20> {          // a 320-wide tile of a wider J: 13 x 20 blocks, 260 real of 8 x 33 issued
    static constexpr int NCB = 20, BJP = 336, NR = 13, NC = 2, NX = 7, NBLKW = NR * NC + NX;
    static constexpr bool XONECOL = true;      // all single blocks of a wave lie in one column: one B fragment serves them
    // wave pair (w, w + 4) shares columns 5 w .. 5 w + 4: two whole columns each, column 5 w + 2 cut by rows (0-6 / 7-12)
    __host__ __device__ static constexpr int r0(int) { return 0; }
    __host__ __device__ static constexpr int c0(int w) { return 5 * (w & 3) + ((w >> 2) ? 3 : 0); }
    __host__ __device__ static constexpr int xr(int w, int i) { return (w >> 2) ? 7 + (i < 6 ? i : 0) : i; }
    __host__ __device__ static constexpr int xc(int w, int) { return 5 * (w & 3) + 2; }
    __host__ __device__ static constexpr bool xvalid(int w, int i) { return (w >> 2) ? i < 6 : true; }
};

constexpr int kTnBK = 32;       // rows of M per chunk = 8 k-steps of 4

// Diagnostic build (-DUAVGEMM_STAMPS, tools/r03_gemm_stamps.sh): s_memtime at the phase boundaries of gemm_tn_kernel, summed per wavefront
// into 4 counters behind the slabs (total, prologue, MFMA phases, chunk boundaries).  Values go to that buffer only, never to an output.
#ifdef UAVGEMM_STAMPS
#define GEMM_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GEMM_STAMP(var) do { } while (0)
#endif

// The 8 k-steps of one chunk for one wavefront: an NR x NC block rectangle at (pa, pb) + NX single blocks at (pxa[i], pxb[i]).  The
// pointers are this lane's element of k row q of the tiles (row strides kNP / BJP); every k-step offset is an immediate of the ds_read.
// Fragments of k-step s + 1 are read while the MFMAs of k-step s issue, ONE LDS read ahead of each MFMA, in exactly the order written
// here: a sched_barrier after every MFMA keeps hipcc from (a) bunching the 15-17 reads and their address arithmetic in front of a
// k-step, where they cost the in-order wave ~40 % more cycles per MFMA (s_memtime stamps, profiles/r03h_*; the partner wave of the SIMD
// cannot fill such gaps: MI355X_MICROARCH.md, Two waves per SIMD, item 1), and (b) collecting the single-block MFMAs of several k-steps
// into runs on ONE accumulator (what sched_group_barrier patterns produced: dependent MFMAs back to back, 40 cycles each).
template <int NR, int NC, int NX, int BJP, bool XONECOL>
__device__ __forceinline__ void tn_ksteps(const float *pa, const float *pb, const float *const (&pxa)[NX], const float *const (&pxb)[NX],
                                          f32x4 (&acc)[NR][NC], f32x4 (&accx)[NX]) {
    float a[2][NR], b[2][NC], xa[2][NX], xb[2][XONECOL ? 1 : NX];
    constexpr int NXB = XONECOL ? 1 : NX;
    constexpr int NREADS = NR + NC + NX + NXB, NMFMA = NR * NC + NX;
    static_assert(NREADS <= NMFMA, "one LDS read per MFMA gap");
    // read number l of a k-step: A fragments of the rectangle, its B fragments, then the single blocks' A and B fragments
#define TN_READ(KS, S, L)                                                                                                  \
    do {                                                                                                                   \
        if ((L) < NR) a[S][(L) < NR ? (L) : 0] = pa[(KS) * 4 * kNP + (L) * 16];                                           \
        else if ((L) < NR + NC) b[S][(L) - NR < NC && (L) >= NR ? (L) - NR : 0] = pb[(KS) * 4 * BJP + ((L) - NR) * 16];   \
        else if ((L) < NR + NC + NX) xa[S][(L) - NR - NC < NX && (L) >= NR + NC ? (L) - NR - NC : 0] = pxa[(L) - NR - NC < NX && (L) >= NR + NC ? (L) - NR - NC : 0][(KS) * 4 * kNP]; \
        else xb[S][(L) - NR - NC - NX < NXB && (L) >= NR + NC + NX ? (L) - NR - NC - NX : 0] = pxb[(L) - NR - NC - NX < NXB && (L) >= NR + NC + NX ? (L) - NR - NC - NX : 0][(KS) * 4 * BJP]; \
    } while (0)
#pragma unroll
    for (int l = 0; l < NREADS; ++l) TN_READ(0, 0, l);
#pragma unroll
    for (int ks = 0; ks < kTnBK / 4; ++ks) {
        const int cur = ks & 1;
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) {
            if (ks + 1 < kTnBK / 4 && m < NREADS) TN_READ(ks + 1, cur ^ 1, m);
            if (m < NR * NC) acc[m / NC][m % NC] = MFMA16(a[cur][m / NC], b[cur][m % NC], acc[m / NC][m % NC]);
            else accx[m - NR * NC] = MFMA16(xa[cur][m - NR * NC], xb[cur][XONECOL ? 0 : m - NR * NC], accx[m - NR * NC]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef TN_READ
}

template <int PLAN, bool BVEC>
__global__ __launch_bounds__(512, 1) void gemm_tn_kernel(const float *__restrict__ A, int n_i, const float *__restrict__ B, long long ldb,
                                                          int n_j, long long M, long long rows_per_split, int n_jt, int n_split,
                                                          f32x4 *__restrict__ slabs, int dbg) {
    using P = TnPlan<PLAN>;
    constexpr int NCB = P::NCB, BJP = P::BJP, NR = P::NR, NC = P::NC, NX = P::NX, NBLKW = P::NBLKW;
    constexpr int A_TILE = kTnBK * kNP, B_TILE = kTnBK * BJP;
    __shared__ __attribute__((aligned(16))) float lds[2 * (A_TILE + B_TILE)];
    float *const sA = lds, *const sB = lds + 2 * A_TILE;

    unsigned long long ts0 = 0, ts1 = 0, tsa = 0, tsb = 0, tsc = 0, t_mfma = 0, t_bound = 0;
    (void)ts0; (void)ts1; (void)tsa; (void)tsb; (void)tsc; (void)t_mfma; (void)t_bound;
    GEMM_STAMP(ts0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches)
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
    for (int i = tid; i < 2 * (A_TILE + B_TILE); i += 512) lds[i] = 0.0f;
    __syncthreads();
    if (tid < 2 * kTnBK) sA[(tid >> 5) * A_TILE + (tid & 31) * kNP + n_i] = 1.0f;     // n_i <= 200 (host)

    // ---- per-thread staging plan (the same every chunk).  A: float4 number idx of the contiguous chunk goes to LDS row idx / (n_i / 4).
    // B: 32 rows of this J tile, float4s (BVEC: the row stride and the tile origin are multiples of 4 floats and the columns up to the
    // next multiple of 4 past n_j are readable and finite -- they only feed output columns >= n_j, which are never stored) or floats. ----
    constexpr int NAL = 4;                                               // 32 x 200 / 4 = 1600 float4 over 512 threads
    const int a4 = n_i >> 2;
    int a_lds[NAL];
#pragma unroll
    for (int i = 0; i < NAL; ++i) { const int idx = tid + 512 * i, row = idx / a4; a_lds[i] = row * kNP + (idx - row * a4) * 4; }
    const int n_j4 = (n_j + 3) & ~3;
    const int bj = BVEC ? ((n_j4 - j0 < NCB * 16) ? n_j4 - j0 : NCB * 16) : ((n_j - j0 < NCB * 16) ? n_j - j0 : NCB * 16);   // columns staged
    constexpr int NBL = BVEC ? (kTnBK * NCB * 4 + 511) / 512 : (kTnBK * NCB * 16 + 511) / 512;
    int b_lds[BVEC ? NBL : 1], b_g[BVEC ? NBL : 1];     // (row < rows  <=>  b_g < rows * ldb;  row < 32  <=>  b_lds < 32 * BJP)
    if (BVEC) {
        const int b4 = bj >> 2;
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            const int idx = tid + 512 * i, row = idx / b4, c4 = idx - row * b4;
            b_lds[BVEC ? i : 0] = row * BJP + c4 * 4; b_g[BVEC ? i : 0] = row * (int)ldb + c4 * 4;
        }
    }

    float4 ra[NAL];
    float4 rbv[BVEC ? NBL : 1];
    float rbs[BVEC ? 1 : NBL];
    auto chunk_rows = [&](int c) { const long long mb = m0 + (long long)c * kTnBK; return (m1 - mb < kTnBK) ? (int)(m1 - mb) : kTnBK; };   // >= 1
    auto load_chunk = [&](int c) {                       // raw loads from safe addresses; store_chunk zeroes what does not exist
        const long long mb = m0 + (long long)c * kTnBK;
        const int rows = chunk_rows(c);
        const float *ga = A + mb * (long long)n_i;
        const float *gb = B + mb * ldb + j0;
#pragma unroll
        for (int i = 0; i < NAL; ++i) { const int idx = tid + 512 * i; ra[i] = ldraw4(ga + idx * 4, idx < rows * a4, ga); }
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            if (BVEC) rbv[BVEC ? i : 0] = ldraw4(gb + b_g[BVEC ? i : 0], b_g[BVEC ? i : 0] < rows * (int)ldb, gb);
            else {
                const int idx = tid + 512 * i, br = idx / (NCB * 16), bc = idx - br * (NCB * 16);
                rbs[BVEC ? 0 : i] = ldraw1(gb + (long long)br * ldb + bc, br < rows && bc < bj, gb);
            }
        }
    };
    auto store_chunk = [&](int buf, int c) {
        const int rows = chunk_rows(c);
        float *dA = sA + buf * A_TILE, *dB = sB + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < NAL; ++i) {
            const int idx = tid + 512 * i;
            if (idx < kTnBK * a4) *reinterpret_cast<float4 *>(dA + a_lds[i]) = mask_f4(ra[i], idx < rows * a4);
        }
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            if (BVEC) {
                if (b_lds[BVEC ? i : 0] < kTnBK * BJP) *reinterpret_cast<float4 *>(dB + b_lds[BVEC ? i : 0]) = mask_f4(rbv[BVEC ? i : 0], b_g[BVEC ? i : 0] < rows * (int)ldb);
            } else {
                const int idx = tid + 512 * i, br = idx / (NCB * 16), bc = idx - br * (NCB * 16);
                if (br < kTnBK) dB[br * BJP + bc] = mask_f(rbs[BVEC ? 0 : i], br < rows && bc < bj);
            }
        }
    };

    f32x4 acc[NR][NC], accx[NX];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NX; ++i) accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fa = q * kNP + r, fb = q * BJP + r;                      // this lane's element of k row q
    const int oa = fa + P::r0(wave) * 16, ob = fb + P::c0(wave) * 16;
    int oxa[NX], oxb[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { oxa[i] = fa + P::xr(wave, i) * 16; oxb[i] = fb + P::xc(wave, i) * 16; }
    const float *pxa[NX], *pxb[NX];

    if (n_chunks > 0) { load_chunk(0); store_chunk(0, 0); }
    __syncthreads();
    GEMM_STAMP(ts1);
    tsc = ts1;
#ifdef UAVGEMM_STAMPS
    if ((dbg & 1) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if ((dbg & 8) && wave < 4) __builtin_amdgcn_s_setprio(1);
    const bool skip = ((dbg & 2) && wave >= 4) || ((dbg & 4) && wave < 4);
#else
    (void)dbg;
    constexpr bool skip = false;
#endif
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        tsa = tsc;
        if (c + 1 < n_chunks) load_chunk(c + 1);                           // in flight behind this chunk's MFMAs
        const float *tA = sA + buf * A_TILE, *tB = sB + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < NX; ++i) { pxa[i] = tA + oxa[i]; pxb[i] = tB + oxb[i]; }
        if (!skip) tn_ksteps<NR, NC, NX, BJP, P::XONECOL>(tA + oa, tB + ob, pxa, pxb, acc, accx);
        GEMM_STAMP(tsb);
        if (c + 1 < n_chunks) store_chunk(buf ^ 1, c + 1);
        __syncthreads();
        GEMM_STAMP(tsc);
        t_mfma += tsb - tsa; t_bound += tsc - tsb;
    }

    // ---- one slab per workgroup, in fragment order (coalesced 16-byte stores): [wave][block][lane], blocks of a wave = its rectangle
    // row by row, then its single blocks ----
    f32x4 *dst = slabs + (((long long)split * n_jt + jt) * 8 + wave) * (long long)(NBLKW * 64) + lane;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) dst[(i * NC + c) * 64] = acc[i][c];
#pragma unroll
    for (int i = 0; i < NX; ++i) dst[(NR * NC + i) * 64] = accx[i];
#ifdef UAVGEMM_STAMPS
    unsigned long long ts2;
    GEMM_STAMP(ts2);
    if (lane == 0) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(slabs + (long long)n_split * n_jt * 8 * NBLKW * 64) + ((long long)blockIdx.x * 8 + wave) * 4;
        d[0] = ts2 - ts0; d[1] = ts1 - ts0; d[2] = t_mfma; d[3] = t_bound;
    }
#endif
}

// Second pass of dW: 64 accumulator fragments (J tile, wave, block, lane) per workgroup, each summed over the splits by four threads (a
// quarter of the splits each, ascending) whose partial sums are then added in quarter order: a fixed association, so the result does
// not depend on scheduling.  Row n_i of the product (the ones column) goes to dbias.
template <int PLAN>
__global__ __launch_bounds__(256) void gemm_tn_reduce(const f32x4 *__restrict__ slabs, int n_split, int n_jt, int n_i, int n_j, float *__restrict__ C,
                                                       long long ldc, float *__restrict__ dbias) {
    using P = TnPlan<PLAN>;
    constexpr int NBLKW = P::NBLKW, NC = P::NC, NMAIN = P::NR * P::NC;
    __shared__ f32x4 part[4][64];
    const int it = threadIdx.x & 63, sq = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + it;
    const int per_slab = 8 * NBLKW * 64;
    const bool live = t < n_jt * per_slab;
    const int tt = live ? t : 0;
    const int jt = tt / per_slab, rem = tt - jt * per_slab;
    const int wave = rem / (NBLKW * 64), rem2 = rem - wave * (NBLKW * 64), blk = rem2 >> 6, lane = rem2 & 63;
    const f32x4 *src = slabs + (long long)jt * per_slab + rem;
    const long long stride = (long long)n_jt * per_slab;
    const int per_q = (n_split + 3) / 4;
    int s = sq * per_q;
    const int s1 = (s + per_q < n_split) ? s + per_q : n_split;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (; s + 8 <= s1; s += 8) {
            f32x4 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = src[(s + i) * stride];
#pragma unroll
            for (int i = 0; i < 8; ++i) v += w[i];
        }
        for (; s < s1; ++s) v += src[s * stride];
    }
    part[sq][it] = v;
    __syncthreads();
    if (sq != 0 || !live) return;
    int rb, cb;
    if (blk < NMAIN) { rb = P::r0(wave) + blk / NC; cb = P::c0(wave) + blk % NC; }
    else { const int i = blk - NMAIN; if (!P::xvalid(wave, i)) return; rb = P::xr(wave, i); cb = P::xc(wave, i); }
    v = part[0][it];
    v += part[1][it]; v += part[2][it]; v += part[3][it];
    const int col = jt * P::NCB * 16 + cb * 16 + (lane & 15);
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
// Four kernels, chosen by uavagent_gemm_rows_f32:
//   gemm_rows_glds_kernel   dy @ W^T / x @ (W^T)^T, aligned operands, K % 40 == 0: LDS-DMA ring (the learner's shapes: all of them)
//   gemm_rows_nt_kernel     the same product for any K % 4 == 0: register-staged, k-contiguous tiles, wide fragment reads
//   gemm_rows_vec_kernel    x @ W with W [K, N] as stored (n-contiguous W tile [k][208], A tile [row][22]: stride 22 puts the 16 rows x
//                           2 k of a fragment read on 32 different banks), aligned operands: 128 rows x <= 208 columns per workgroup,
//                           float4 staging without branches, epilogue through LDS
//   gemm_rows_kernel        the general form (dword loads, direct epilogue) for everything else
// =====================================================================================================================
constexpr int kRowsBM = 128, kRowsBK = 20, kRowsLD = 22;
constexpr int kRowsATile = kRowsBM * kRowsLD;                                   // 2816 floats
constexpr int kRowsWTile = (kRowsBK * kNP > kNP * kRowsLD) ? kRowsBK * kNP : kNP * kRowsLD;   // 4576 floats
constexpr int kRowsLDC = 212;                                                   // epilogue staging row stride (848 B = 53 x 16)
static_assert(4 * 16 * kRowsLDC <= 2 * (kRowsATile + kRowsWTile), "epilogue staging must fit the tile buffers");

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_rows_vec_kernel(const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw,
                                                                int K, int N, long long M, const float *__restrict__ bias, int relu6,
                                                                const float *__restrict__ H, long long ldh, float *__restrict__ C, long long ldc,
                                                                float *__restrict__ col_partial) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (kRowsATile + kRowsWTile)];
    float *const sA = lds, *const sW = lds + 2 * kRowsATile;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const long long m0 = (long long)blockIdx.x * kRowsBM;

    for (int i = tid; i < 2 * (kRowsATile + kRowsWTile); i += 256) lds[i] = 0.0f;   // padding rows / columns must be finite
    __syncthreads();

    // staging plan: A = 128 rows x 5 float4 (3 loads per thread), W = 1000 float4 at N = 200 (<= 1040: 5 loads cover N <= 208)
    constexpr int NA = 3, NW = 5;
    int a_g[NA], a_l[NA], a_k[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + 256 * i, row = idx / 5, c4 = idx - row * 5;
        a_g[i] = row * (int)lda + c4 * 4; a_l[i] = row * kRowsLD + c4 * 4; a_k[i] = c4 * 4;
        a_ok[i] = (row < kRowsBM) && (m0 + row < M);
    }
    const int n4 = N >> 2;
    int w_g[NW], w_l[NW], w_k[NW];
    bool w_ok[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / n4, c4 = idx - row * n4;       // W [K, N]: tile row = k (20), N / 4 float4 per row
        w_g[i] = row * (int)ldw + c4 * 4; w_l[i] = row * kNP + c4 * 4; w_k[i] = row; w_ok[i] = row < kRowsBK;
    }
    float4 va[NA], vw[NW];
    const float *gA = A + m0 * lda;
    auto load_chunk = [&](int c) {                       // raw loads from safe addresses; store_chunk zeroes what does not exist
        const int k0 = c * kRowsBK;
#pragma unroll
        for (int i = 0; i < NA; ++i) va[i] = ldraw4(gA + k0 + a_g[i], a_ok[i] && (k0 + a_k[i] < K), A);
#pragma unroll
        for (int i = 0; i < NW; ++i) vw[i] = ldraw4(W + (long long)k0 * ldw + w_g[i], w_ok[i] && (k0 + w_k[i] < K), W);
    };
    auto store_chunk = [&](int buf, int c) {
        const int k0 = c * kRowsBK;
        float *dA = sA + buf * kRowsATile, *dW = sW + buf * kRowsWTile;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (tid + 256 * i < kRowsBM * 5) {      // row stride 22 floats = 88 B: 8-byte aligned, two 8-byte stores
                const float4 v = mask_f4(va[i], a_ok[i] && (k0 + a_k[i] < K));
                float2 *d = reinterpret_cast<float2 *>(dA + a_l[i]);
                d[0] = float2{v.x, v.y}; d[1] = float2{v.z, v.w};
            }
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (w_ok[i]) {
                *reinterpret_cast<float4 *>(dW + w_l[i]) = mask_f4(vw[i], k0 + w_k[i] < K);
            }
    };

    f32x4 acc[2][kRB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < kRB; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int n_chunks = (K + kRowsBK - 1) / kRowsBK;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    const int fa = (wave * 32 + r) * kRowsLD + q, fw = q * kNP + r;
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) load_chunk(c + 1);
        const float *tA = sA + buf * kRowsATile + fa, *tW = sW + buf * kRowsWTile + fw;
        // all 5 k-steps, always: k >= K was staged as zeros in BOTH tiles.  Fragments of k-step s + 1 are read while k-step s issues, one
        // LDS read ahead of each MFMA, in source order (sched_barrier after every MFMA: see tn_ksteps).
        float a[2][2], b[2][kRB];
#define ROWS_READ(KS, S, L)                                                                                                  \
        do {                                                                                                                \
            if ((L) < 2) a[S][(L) < 2 ? (L) : 0] = tA[(L) * 16 * kRowsLD + (KS) * 4];                                       \
            else b[S][(L) >= 2 ? (L) - 2 : 0] = tW[(KS) * 4 * kNP + ((L) - 2) * 16];                                               \
        } while (0)
#pragma unroll
        for (int l = 0; l < kRB + 2; ++l) ROWS_READ(0, 0, l);
#pragma unroll
        for (int ks = 0; ks < kRowsBK / 4; ++ks) {
            const int cur = ks & 1;
#pragma unroll
            for (int m = 0; m < 2 * kRB; ++m) {
                if (ks + 1 < kRowsBK / 4 && m < kRB + 2) ROWS_READ(ks + 1, cur ^ 1, m);
                acc[m & 1][m >> 1] = MFMA16(a[cur][m & 1], b[cur][m >> 1], acc[m & 1][m >> 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef ROWS_READ
        if (c + 1 < n_chunks) store_chunk(buf ^ 1, c + 1);
        __syncthreads();
    }

    // ---- epilogue through LDS: wave w stages 16 rows x 208 columns at a time (accumulator register t of block cb = row 4 q + t, column
    // cb * 16 + r), then lane l < N / 4 moves float4 number l of one whole row per instruction: 800 contiguous bytes per wave-instruction
    // for C (and for H) instead of dword accesses that touch 64 bytes per row.  Each lane thereby owns 4 fixed columns, so the column
    // sums of what is stored (the bias gradient of the layer below, col_partial != nullptr) are a running float4 per lane: rows ascending
    // inside the wave, then the 4 waves in order, then the workgroups in order (rows_colsum_reduce): bit-reproducible. ----
    float *E = lds + wave * (16 * kRowsLDC);
    float4 cs = float4{0.f, 0.f, 0.f, 0.f};
    const bool col_ok = lane < n4;
    float4 bv = float4{0.f, 0.f, 0.f, 0.f};
    if (EPI == 1 && bias != nullptr && col_ok) bv = *reinterpret_cast<const float4 *>(bias + lane * 4);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int cb = 0; cb < kRB; ++cb)
#pragma unroll
            for (int t = 0; t < 4; ++t) E[(4 * q + t) * kRowsLDC + cb * 16 + r] = acc[rb][cb][t];
        __syncthreads();                       // (a wave reads only what it wrote itself; the barrier orders its lanes' writes and reads)
#pragma unroll 4
        for (int row = 0; row < 16; ++row) {
            const long long m = m0 + wave * 32 + rb * 16 + row;
            if (col_ok && m < M) {
                float4 v = *reinterpret_cast<const float4 *>(E + row * kRowsLDC + lane * 4);
                if (EPI == 1) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (relu6) { v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f); v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f); }
                }
                if (EPI == 2) {
                    const float4 h = *reinterpret_cast<const float4 *>(H + m * ldh + lane * 4);
                    v.x = (h.x > 0.f && h.x < 6.f) ? v.x : 0.f; v.y = (h.y > 0.f && h.y < 6.f) ? v.y : 0.f;
                    v.z = (h.z > 0.f && h.z < 6.f) ? v.z : 0.f; v.w = (h.w > 0.f && h.w < 6.f) ? v.w : 0.f;
                }
                *reinterpret_cast<float4 *>(C + m * ldc + lane * 4) = v;
                cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
            }
        }
        __syncthreads();
    }
    if (col_partial != nullptr) {
        float4 *S = reinterpret_cast<float4 *>(lds);                       // [4 waves][52 float4]
        if (lane < 52) S[wave * 52 + lane] = cs;
        __syncthreads();
        if (wave == 0 && lane < 52) {
            float4 t = S[lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) { const float4 u = S[w * 52 + lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            *reinterpret_cast<float4 *>(col_partial + (long long)blockIdx.x * kNP + lane * 4) = t;
        }
    }
}

// out[c] = sum over workgroups of col_partial[wg][c], in a fixed order: a block owns 64 columns, 16 slab-threads per column each add a
// contiguous slab of the partial rows in ascending order, then the 16 slab sums are added in slab order.
__global__ __launch_bounds__(1024) void rows_colsum_reduce(const float *__restrict__ partial, long long n_part, int n_cols, float *__restrict__ out) {
    __shared__ float slab_sum[16][64];
    const int cl = threadIdx.x & 63, slab = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long long per = (n_part + 15) / 16;
    const long long w0 = slab * per, w1 = (w0 + per < n_part) ? w0 + per : n_part;
    float sum = 0.f;
    if (c < n_cols) {
        long long w = w0;
        for (; w + 4 <= w1; w += 4) {
            const float a0 = partial[w * kNP + c], a1 = partial[(w + 1) * kNP + c], a2 = partial[(w + 2) * kNP + c], a3 = partial[(w + 3) * kNP + c];
            sum += a0; sum += a1; sum += a2; sum += a3;
        }
        for (; w < w1; ++w) sum += partial[w * kNP + c];
    }
    slab_sum[slab][cl] = sum;
    __syncthreads();
    if (slab == 0 && c < n_cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += slab_sum[k][cl];
        out[c] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dy @ W^T (and x @ W with W handed over transposed), version 3: BOTH tiles k-contiguous in LDS, rows of 24 floats (20 k of a chunk + 4
// padding: with this stride the sixteen 16-byte pieces a ds_read_b128 lane group touches fall on 16 different bank quads).  A lane
// (r, q) then takes its operands of FOUR k-steps with one ds_read_b128 (k = 4 q + s for k-step s: any bijection of k serves as long as
// both operands use it) and the fifth with a ds_read_b32 (k = 16 + q): 2 LDS reads per 5 MFMAs of a block instead of 5.  An in-order
// wave pays ~4 cycles of MFMA issue for every LDS read between two MFMAs (s_memtime stamps on the dW kernel: 37 cycles per MFMA with one
// read per gap, against the pipe's 32), and the partner wave of the SIMD cannot use those gaps.
//   BM = 128: 4 waves x 32 rows x NB column blocks, 2 workgroups per CU (the update: M = rollout x envs rows);
//   BM =  64: 4 waves x 16 rows x NB column blocks, N cut into slices of NB blocks across blockIdx.y (the rollout's M = envs rows: 8192 rows
//             give 256 / 512 workgroups at NB = 7 / 10).
// Epilogues as gemm_rows_vec_kernel; the column sums (col_partial) exist for single-slice launches only.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kNtBK = 20, kNtLD = 24;

template <int BM, int NB, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_rows_nt_kernel(const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw,
                                                               int K, int N, long long M, const float *__restrict__ bias, int relu6,
                                                               const float *__restrict__ H, long long ldh, float *__restrict__ C, long long ldc,
                                                               float *__restrict__ col_partial) {
    static_assert(BM == 128 || BM == 64, "wave tile = 32 or 16 rows");
    constexpr int RBW = BM / 64;                                   // 16-row blocks per wave
    constexpr int A_TILE = BM * kNtLD, W_TILE = NB * 16 * kNtLD;
    constexpr int LDC = NB * 16 + 4;                               // epilogue staging row stride (16-byte multiple)
    constexpr int LDS_FLOATS = (2 * (A_TILE + W_TILE) > 4 * 16 * LDC) ? 2 * (A_TILE + W_TILE) : 4 * 16 * LDC;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float *const sA = lds, *const sW = lds + 2 * A_TILE;
    unsigned long long ts0 = 0, ts1 = 0, tsa = 0, tsb = 0, tsc = 0, t_mfma = 0, t_bound = 0, ts2 = 0, ts3 = 0;
    (void)ts0; (void)ts1; (void)tsa; (void)tsb; (void)tsc; (void)t_mfma; (void)t_bound; (void)ts2; (void)ts3;
    GEMM_STAMP(ts0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * NB * 16;                           // this workgroup's slice of the output columns
    const int nloc = (N - n0 < NB * 16) ? N - n0 : NB * 16;

    // staging plan: float4 number c4 (k = 4 c4 ..) of tile row `row`; raw loads from safe addresses, zeroed at the LDS store
    constexpr int NA = (BM * 5 + 255) / 256, NW = (NB * 16 * 5 + 255) / 256;
    int a_g[NA], a_l[NA], w_g[NW], w_l[NW];
    bool a_ok[NA], w_ok[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + 256 * i, row = idx / 5, c4 = idx - row * 5;
        a_g[i] = row * (int)lda + c4 * 4; a_l[i] = (idx < BM * 5) ? row * kNtLD + c4 * 4 : -1;
        a_ok[i] = (idx < BM * 5) && (m0 + row < M);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int idx = tid + 256 * i, row = idx / 5, c4 = idx - row * 5;
        w_g[i] = row * (int)ldw + c4 * 4; w_l[i] = (idx < NB * 16 * 5) ? row * kNtLD + c4 * 4 : -1;
        w_ok[i] = (idx < NB * 16 * 5) && (row < nloc);
    }
    float4 va[NA], vw[NW];
    const float *gA = A + m0 * lda, *gW = W + (long long)n0 * ldw;
    auto load_chunk = [&](int c) {
        const int k0 = c * kNtBK;
#pragma unroll
        for (int i = 0; i < NA; ++i) va[i] = ldraw4(gA + k0 + a_g[i], a_ok[i] && (k0 + (a_l[i] % kNtLD) < K), A);
#pragma unroll
        for (int i = 0; i < NW; ++i) vw[i] = ldraw4(gW + k0 + w_g[i], w_ok[i] && (k0 + (w_l[i] % kNtLD) < K), W);
    };
    auto store_chunk = [&](int buf, int c) {
        const int k0 = c * kNtBK;
        float *dA = sA + buf * A_TILE, *dW = sW + buf * W_TILE;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (a_l[i] >= 0) *reinterpret_cast<float4 *>(dA + a_l[i]) = mask_f4(va[i], a_ok[i] && (k0 + (a_l[i] % kNtLD) < K));
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (w_l[i] >= 0) *reinterpret_cast<float4 *>(dW + w_l[i]) = mask_f4(vw[i], w_ok[i] && (k0 + (w_l[i] % kNtLD) < K));
    };

    f32x4 acc[RBW][NB];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int c = 0; c < NB; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int n_chunks = (K + kNtBK - 1) / kNtBK;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    const int fa = (wave * (16 * RBW) + r) * kNtLD + 4 * q, fw = r * kNtLD + 4 * q;    // this lane's 16-byte piece of the first 16 k
    const int fa5 = (wave * (16 * RBW) + r) * kNtLD + 16 + q, fw5 = r * kNtLD + 16 + q;  // and its element of the fifth k-step
    GEMM_STAMP(ts1);
    tsc = ts1;
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        tsa = tsc;
        if (c + 1 < n_chunks) load_chunk(c + 1);
        const float *tA = sA + buf * A_TILE, *tW = sW + buf * W_TILE;
        // reads in issue order: a128[*], w128[0 .. AHEAD-1]; then one read ahead of each column block (sched_barrier pins the order, see
        // tn_ksteps): the b128 of the block AHEAD blocks on during k-step 0, the b32 operands of k-step 4 during k-steps 1 and 2
        constexpr int AHEAD = (NB < 4) ? NB : 4;
        f32x4 a4[RBW], w4[NB];
        float a1[RBW], w1[NB];
#pragma unroll
        for (int i = 0; i < RBW; ++i) a4[i] = *reinterpret_cast<const f32x4 *>(tA + fa + i * 16 * kNtLD);
#pragma unroll
        for (int cb = 0; cb < AHEAD; ++cb) w4[cb] = *reinterpret_cast<const f32x4 *>(tW + fw + cb * 16 * kNtLD);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                // reads ahead: during k-step 0 the next column block's b128; during k-steps 1-3 the b32 operands of k-step 4
                if (ks == 0 && cb + AHEAD < NB) w4[cb + AHEAD < NB ? cb + AHEAD : 0] = *reinterpret_cast<const f32x4 *>(tW + fw + (cb + AHEAD) * 16 * kNtLD);
                if (ks == 1 && cb < RBW) a1[cb < RBW ? cb : 0] = tA[fa5 + cb * 16 * kNtLD];
                if (ks == 2) w1[cb] = tW[fw5 + cb * 16 * kNtLD];
#pragma unroll
                for (int i = 0; i < RBW; ++i) {
                    acc[i][cb] = MFMA16(ks < 4 ? a4[i][ks < 4 ? ks : 0] : a1[i], ks < 4 ? w4[cb][ks < 4 ? ks : 0] : w1[cb], acc[i][cb]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        GEMM_STAMP(tsb);
        if (c + 1 < n_chunks) store_chunk(buf ^ 1, c + 1);
        __syncthreads();
        GEMM_STAMP(tsc);
        t_mfma += tsb - tsa; t_bound += tsc - tsb;
    }
    GEMM_STAMP(ts2);

    // ---- epilogue through LDS (see gemm_rows_vec_kernel): 16 rows x NB * 16 columns per wave and round, one whole row per instruction ----
    float *E = lds + wave * (16 * LDC);
    float4 cs = float4{0.f, 0.f, 0.f, 0.f};
    const bool col_ok = lane * 4 < nloc;
    float4 bv = float4{0.f, 0.f, 0.f, 0.f};
    if (EPI == 1 && bias != nullptr && col_ok) bv = *reinterpret_cast<const float4 *>(bias + n0 + lane * 4);
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        // the relu6 mask's H rows are requested BEFORE the staging pass and the barrier, all 16 at once: loaded inside the store loop,
        // each batch of them exposed a full HBM round trip per workgroup (the masked dX ran 0.10-0.14 ms behind the plain one)
        float4 hv[16];
        if (EPI == 2) {
#pragma unroll
            for (int row = 0; row < 16; ++row) {
                const long long m = m0 + wave * (16 * RBW) + rb * 16 + row;
                hv[row] = ldraw4(H + m * ldh + n0 + lane * 4, col_ok && m < M, H);
            }
        }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int t = 0; t < 4; ++t) E[(4 * q + t) * LDC + cb * 16 + r] = acc[rb][cb][t];
        __syncthreads();
        if (EPI == 1) arrived4(bv);
        if (EPI == 2) {
#pragma unroll
            for (int row = 0; row < 16; ++row) arrived4(hv[row]);
        }
#pragma unroll
        for (int row = 0; row < 16; ++row) {
            const long long m = m0 + wave * (16 * RBW) + rb * 16 + row;
            if (col_ok && m < M) {
                float4 v = *reinterpret_cast<const float4 *>(E + row * LDC + lane * 4);
                if (EPI == 1) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (relu6) { v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f); v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f); }
                }
                if (EPI == 2) {
                    const float4 h = hv[row];
                    v.x = (h.x > 0.f && h.x < 6.f) ? v.x : 0.f; v.y = (h.y > 0.f && h.y < 6.f) ? v.y : 0.f;
                    v.z = (h.z > 0.f && h.z < 6.f) ? v.z : 0.f; v.w = (h.w > 0.f && h.w < 6.f) ? v.w : 0.f;
                }
                *reinterpret_cast<float4 *>(C + m * ldc + n0 + lane * 4) = v;
                cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
            }
        }
        __syncthreads();
    }
    if (col_partial != nullptr) {                                          // (host: N <= 208; a slice owns its columns of the partial row)
        float4 *S = reinterpret_cast<float4 *>(lds);                       // [4 waves][52 float4]
        if (lane < 52) S[wave * 52 + lane] = cs;
        __syncthreads();
        if (wave == 0 && col_ok) {
            float4 t = S[lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) { const float4 u = S[w * 52 + lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            *reinterpret_cast<float4 *>(col_partial + (long long)blockIdx.x * kNP + n0 + lane * 4) = t;
        }
    }
#ifdef UAVGEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_STAMP(ts3);
    if (col_partial != nullptr && lane == 0 && blockIdx.y == 0) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(col_partial + (long long)gridDim.x * kNP) + ((long long)blockIdx.x * 4 + wave) * 6;
        d[0] = ts3 - ts0; d[1] = ts1 - ts0; d[2] = t_mfma; d[3] = t_bound; d[4] = ts3 - ts2; d[5] = ts0;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// dy @ W^T, version 4 (K a multiple of 40): tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write pass) into a ring of NSTAGE buffers, chunks of 40 k, the loads of chunk c + NSTAGE - 1 issued before chunk c is computed
// and retired by a COUNTED s_waitcnt vmcnt + a raw s_barrier (cdna_hip_programming.md section 5, Pipelining across barriers).  What
// the register-staged kernels above showed in s_memtime stamps (profiles/r03m_*): with a prefetch distance of one 20-k chunk a
// workgroup spent 22 % of its life at chunk boundaries waiting for loads (80-byte pieces of 128 rows) and 9 % in its prologue -- and a
// rollout-sized launch (64 rows per workgroup, 35 MFMAs per chunk) nearly all of it.
// LDS image of a stage: [BM A rows | NB * 16 W rows] x 40 floats, unpadded -- lane-linear for the DMA (a wave-instruction fills 1 KiB),
// and with a 160-byte row stride the sixteen 16-byte pieces of a ds_read_b128 lane group still fall on 16 different bank quads.  A lane
// (r, q) reads its operands of k-steps 0-3 / 4-7 with one ds_read_b128 each (k = 4 q + s, 16 + 4 q + s) and of k-steps 8-9 with a
// ds_read_b64 (k = 32 + 2 q + s).  One wavefront = 16 rows x NB column blocks (BM / 16 wavefronts); rows beyond M or N and the
// padding of a stage are loaded from a 16-byte zero block.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float g_zero16[4];          // (zero-initialised: device globals are)
constexpr int kGlBK = 40;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int NB, int NSTAGE, int EPI>
__global__ __launch_bounds__(BM * 4) void gemm_rows_glds_kernel(const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw,
                                                                 int K, int N, long long M, const float *__restrict__ bias, int relu6,
                                                                 const float *__restrict__ H, long long ldh, float *__restrict__ C, long long ldc,
                                                                 float *__restrict__ col_partial) {
    constexpr int NWAVE = BM / 16, NTHR = NWAVE * 64;
    constexpr int ROWS = BM + NB * 16;                      // tile rows of a stage: A rows, then W rows
    constexpr int F4 = ROWS * (kGlBK / 4);                  // float4s of a stage
    constexpr int NL = (F4 + NTHR - 1) / NTHR;              // LDS-DMA instructions per wave and chunk
    constexpr int STAGE_F = NL * NTHR * 4;                  // floats of a stage, padded to whole passes of the workgroup
    constexpr int LDC = NB * 16 + 4;
    constexpr int LDS_F = (NSTAGE * STAGE_F > NWAVE * 16 * LDC) ? NSTAGE * STAGE_F : NWAVE * 16 * LDC;
    static_assert(NSTAGE == 2 || NSTAGE == 3, "ring of 2 or 3 stages");
    static_assert((NSTAGE - 1) * NL <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) float lds[LDS_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * NB * 16;
    const int nloc = (N - n0 < NB * 16) ? N - n0 : NB * 16;

    // ---- per-pass DMA sources.  TWO register sets, for even and odd chunks, each advanced by two chunks (80 floats) just before it is
    // used again: a set is then never written while an LDS-DMA that read it can still be in flight (hipcc guards such a write with
    // s_waitcnt vmcnt(0), which would drain the ring; a temporary address register per instruction gets the same treatment as soon as
    // it is reused) ----
    const float *src[2][NL];
    unsigned real = 0u;
#pragma unroll
    for (int p = 0; p < NL; ++p) {
        const int idx = (p * NWAVE + wave) * 64 + lane, row = idx / (kGlBK / 4), c4 = idx - row * (kGlBK / 4);
        const float *s = g_zero16;
        if (row < BM) { if (m0 + row < M) { s = A + (m0 + row) * lda + c4 * 4; real |= 1u << p; } }
        else if (row < ROWS) { if (row - BM < nloc) { s = W + (long long)(n0 + row - BM) * ldw + c4 * 4; real |= 1u << p; } }
        src[0][p] = s;
        src[1][p] = s + (((real >> p) & 1u) ? kGlBK : 0);
    }
    auto issue_chunk = [&](auto set_c, int c) {               // chunk c (parity = set) -> stage c % NSTAGE
        constexpr int SET = decltype(set_c)::value;
        float *stage = lds + (c % NSTAGE) * STAGE_F;
#pragma unroll
        for (int p = 0; p < NL; ++p) {
            if (c >= 2) src[SET][p] += (((real >> p) & 1u) ? 2 * kGlBK : 0);
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)src[SET][p], (lds_void_t *)(stage + (p * NWAVE + wave) * 256), 16, 0, 0);
        }
    };
    using set0_t = std::integral_constant<int, 0>;
    using set1_t = std::integral_constant<int, 1>;

    f32x4 acc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n_chunks = K / kGlBK;                          // (host: K % 40 == 0, K >= 40)
    issue_chunk(set0_t{}, 0);
    if (NSTAGE == 3 && n_chunks > 1) issue_chunk(set1_t{}, 1);
    const int fa = (wave * 16 + r) * kGlBK, fw = (BM + r) * kGlBK;       // this lane's rows of the A / W part of a stage
    auto chunk = [&](auto set_c, int c) {                     // SET = parity of chunk c + NSTAGE - 1, the one issued here
        // chunk c has landed once at most the younger chunks' loads are outstanding; then the barrier makes every wave's part visible
        const int younger = (n_chunks - 1 - c < NSTAGE - 2) ? n_chunks - 1 - c : NSTAGE - 2;
        if (NSTAGE == 3 && younger == 1) wait_vmcnt<NL>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (c + NSTAGE - 1 < n_chunks) issue_chunk(set_c, c + NSTAGE - 1);   // its stage was read last in chunk c - 1: every wave is past that
        const float *st = lds + (c % NSTAGE) * STAGE_F;
        // group g of k: 0 = k 0-15, 1 = k 16-31 (one b128 per 16-row block), 2 = k 32-39 (one b64); group g + 1 is read while the first
        // k-step of group g issues, one read ahead of each MFMA (sched_barrier pins the order, see tn_ksteps)
        f32x4 a01[2], w01[2][NB];
        f32x2 a2, w2[NB];
        a01[0] = *reinterpret_cast<const f32x4 *>(st + fa + 4 * q);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) w01[0][cb] = *reinterpret_cast<const f32x4 *>(st + fw + cb * 16 * kGlBK + 4 * q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < kGlBK / 4; ++ks) {
            const int g = ks >> 2, e = ks & 3;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                if (ks == 0) {                                              // reads of group 1
                    if (cb == 0) a01[1] = *reinterpret_cast<const f32x4 *>(st + fa + 16 + 4 * q);
                    w01[1][cb] = *reinterpret_cast<const f32x4 *>(st + fw + cb * 16 * kGlBK + 16 + 4 * q);
                }
                if (ks == 4) {                                              // reads of group 2
                    if (cb == 0) a2 = *reinterpret_cast<const f32x2 *>(st + fa + 32 + 2 * q);
                    w2[cb] = *reinterpret_cast<const f32x2 *>(st + fw + cb * 16 * kGlBK + 32 + 2 * q);
                }
                const float av = (g < 2) ? a01[g < 2 ? g : 0][e] : a2[e & 1];
                const float wv = (g < 2) ? w01[g < 2 ? g : 0][cb][e] : w2[cb][e & 1];
                acc[cb] = MFMA16(av, wv, acc[cb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // chunk c issues chunk c + NSTAGE - 1: with 2 stages that one has the other parity, with 3 the same
    for (int c = 0; c < n_chunks; c += 2) {
        if (NSTAGE == 2) { chunk(set1_t{}, c); if (c + 1 < n_chunks) chunk(set0_t{}, c + 1); }
        else { chunk(set0_t{}, c); if (c + 1 < n_chunks) chunk(set1_t{}, c + 1); }
    }
    __syncthreads();                                          // every wave is done with the ring: the epilogue stages through it

    // ---- epilogue through LDS (see gemm_rows_vec_kernel): 16 rows x NB * 16 columns per wave, one whole row per instruction ----
    float *E = lds + wave * (16 * LDC);
    float4 cs = float4{0.f, 0.f, 0.f, 0.f};
    const bool col_ok = lane * 4 < nloc;
    float4 bv = float4{0.f, 0.f, 0.f, 0.f};
    if (EPI == 1 && bias != nullptr && col_ok) bv = *reinterpret_cast<const float4 *>(bias + n0 + lane * 4);
    float4 hv[16];
    if (EPI == 2) {
#pragma unroll
        for (int row = 0; row < 16; ++row) {
            const long long m = m0 + wave * 16 + row;
            hv[row] = ldraw4(H + m * ldh + n0 + lane * 4, col_ok && m < M, H);
        }
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int t = 0; t < 4; ++t) E[(4 * q + t) * LDC + cb * 16 + r] = acc[cb][t];
    __syncthreads();
    if (EPI == 1) arrived4(bv);
    if (EPI == 2) {
#pragma unroll
        for (int row = 0; row < 16; ++row) arrived4(hv[row]);
    }
#pragma unroll
    for (int row = 0; row < 16; ++row) {
        const long long m = m0 + wave * 16 + row;
        if (col_ok && m < M) {
            float4 v = *reinterpret_cast<const float4 *>(E + row * LDC + lane * 4);
            if (EPI == 1) {
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (relu6) { v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f); v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f); }
            }
            if (EPI == 2) {
                const float4 h = hv[row];
                v.x = (h.x > 0.f && h.x < 6.f) ? v.x : 0.f; v.y = (h.y > 0.f && h.y < 6.f) ? v.y : 0.f;
                v.z = (h.z > 0.f && h.z < 6.f) ? v.z : 0.f; v.w = (h.w > 0.f && h.w < 6.f) ? v.w : 0.f;
            }
            *reinterpret_cast<float4 *>(C + m * ldc + n0 + lane * 4) = v;
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        }
    }
    if (col_partial != nullptr) {          // (host: N <= 208) rows ascending inside a wave, then the waves in order, then the workgroups
        __syncthreads();
        float4 *S = reinterpret_cast<float4 *>(lds);                       // [NWAVE][52 float4]
        if (lane < 52) S[wave * 52 + lane] = cs;
        __syncthreads();
        if (wave == 0 && col_ok) {
            float4 t = S[lane];
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) { const float4 u = S[w * 52 + lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            *reinterpret_cast<float4 *>(col_partial + (long long)blockIdx.x * kNP + n0 + lane * 4) = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dy @ W^T for K = 200, many rows (the update's 200-wide layers), version 6: W^T RESIDENT in LDS, persistent workgroups.
// Measured on version 4 with its LDS-DMA re-issue compiled out (wrong results, timing only: profiles/r03y_*): a 128-row workgroup at
// K = 200 lives 62 000 cycles of which 41 600 are its MFMAs -- a third of its life is the prologue (first tiles' round trip), the
// epilogue and the chunk barriers, with nothing beside it on the CU.  Here none of those recur per tile:
//   * the grid is one workgroup per CU; workgroups [0, nA) own output columns [0, 112) (7 blocks), the others [112, 208) (6 blocks);
//     a workgroup loads its 112 x 200 (96 x 200) slice of W^T into LDS ONCE (89.6 / 76.8 KB) and keeps it;
//   * its 8 wavefronts then walk 16-row tiles of A (tile t of the type's wave list, stride = number of such waves: both types sweep the
//     rows front to back at the same pace -- 8 nA / 7 = 8 nB / 6 tiles per unit of MFMA work -- so the second read of an A row comes out
//     of L2 / Infinity Cache); the A fragments go global -> registers (a wave's rows are its own; with the k bijection k = 16 g + 4 q + s
//     a lane's four k-steps of operands are one 16-byte load), next tile's loads in flight under this tile's 350 (300) MFMAs;
//   * no barrier after the prologue: W is read-only, a wave's epilogue patch in LDS is its own, and its stores drain under the next
//     tile's MFMAs (the wave waits for its loads ONCE per tile, before the stores are issued: see arrived4).
// Column sums (bias gradient of the layer below): per lane over the wave's tiles in ascending order, then the 8 waves, then the
// workgroups of the type, in order: deterministic.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kRsK = 200, kRsThr = 512, kRsWaves = 8, kRsNBA = 7, kRsNBB = 6, kRsG = kRsK / 16;                 // 12 groups of 16 k + 8
constexpr int kRsLdsF = kRsNBA * 16 * kRsK + kRsWaves * 16 * (kRsNBA * 16 + 4);                                  // 37 248 floats = 148 992 B
static_assert(kRsK % 16 == 8 && kRsLdsF * 4 <= 160 * 1024, "the resident kernel's LDS plan");

template <int NB, int EPI>
__device__ __forceinline__ void rows_resident_body(const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw, int N,
                                                   long long M, const float *__restrict__ bias, int relu6, const float *__restrict__ H,
                                                   long long ldh, float *__restrict__ C, long long ldc, float *__restrict__ col_partial,
                                                   float *lds, const int n0, const int wg, const int nwg) {
    constexpr int NC = NB * 16, LDC = NC + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    float *const sW = lds;                                        // [NC][200]
    float *const E = lds + NC * kRsK + wave * (16 * LDC);         // this wave's epilogue patch [16][LDC]
    const int nloc = (N - n0 < NC) ? N - n0 : NC;

    // ---- W^T slice -> LDS, once (lane-linear: float4 idx = row * 50 + c4) ----
    constexpr int WF4 = NC * (kRsK / 4), WSLOTS = (WF4 + 63) / 64;
#pragma unroll
    for (int p = 0; p < (WSLOTS + kRsWaves - 1) / kRsWaves; ++p) {
        const int slot = p * kRsWaves + wave;
        if (slot < WSLOTS) {
            const int idx = slot * 64 + lane, row = idx / (kRsK / 4), c4 = idx - row * (kRsK / 4);
            const float *g = (idx < WF4 && row < nloc) ? W + (long long)(n0 + row) * ldw + c4 * 4 : g_zero16;
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)g, (lds_void_t *)(sW + slot * 256), 16, 0, 0);
        }
    }
    // ---- tiles of this wave: t = wg * 8 + wave, + nwg * 8, ... ----
    const long long n_tiles = (M + 15) / 16, stride = (long long)nwg * kRsWaves;
    long long t = (long long)wg * kRsWaves + wave;
    f32x4 a4[kRsG], a4n[kRsG];
    f32x2 a2, a2n;
    auto load_a = [&](f32x4 (&d4)[kRsG], f32x2 &d2, long long tile) {
        long long m = (tile < n_tiles ? tile : n_tiles - 1) * 16 + r;                 // (no tile left: any valid rows, never used)
        m = (m < M) ? m : M - 1;                                                      // rows past M: a valid row, never stored
        const float *pa = A + m * lda + 4 * q;
#pragma unroll
        for (int g = 0; g < kRsG; ++g) d4[g] = *reinterpret_cast<const f32x4 *>(pa + 16 * g);
        d2 = *reinterpret_cast<const f32x2 *>(A + m * lda + 16 * kRsG + 2 * q);
    };
    load_a(a4n, a2n, t);
    wait_vmcnt<0>();
    __syncthreads();                                              // W^T is in LDS; the only barrier of the kernel's main part

    const int half = lane >> 5, L = lane & 31;                    // epilogue: lanes 0-31 take even rows, 32-63 odd rows; L = float4 column
    const bool col_ok = L * 4 < nloc;
    float4 bv = float4{0.f, 0.f, 0.f, 0.f};
    if (EPI == 1 && bias != nullptr && col_ok) bv = *reinterpret_cast<const float4 *>(bias + n0 + L * 4);
    if (EPI == 1) arrived4(bv);
    float4 cs = float4{0.f, 0.f, 0.f, 0.f};
    const float *pw = sW + r * kRsK + 4 * q;
    for (; t < n_tiles; t += stride) {
#pragma unroll
        for (int g = 0; g < kRsG; ++g) a4[g] = a4n[g];
        a2 = a2n;
        load_a(a4n, a2n, t + stride);                              // next tile's A fragments, in flight under this tile's MFMAs
        f32x4 acc[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 w4[2][NB];
        f32x2 w2[NB];
        float4 hv[8];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) w4[0][cb] = *reinterpret_cast<const f32x4 *>(pw + cb * 16 * kRsK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < kRsG; ++g) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    if (ks == 0) {                                  // operands of the next group, one read ahead of each MFMA
                        if (g + 1 < kRsG) w4[(g + 1) & 1][cb] = *reinterpret_cast<const f32x4 *>(pw + cb * 16 * kRsK + 16 * (g + 1));
                        else w2[cb] = *reinterpret_cast<const f32x2 *>(sW + (cb * 16 + r) * kRsK + 16 * kRsG + 2 * q);
                    }
                    acc[cb] = MFMA16(a4[g][ks], w4[g & 1][cb][ks], acc[cb]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (EPI == 2 && g == kRsG - 3) {                       // the relu6 mask's H rows: late enough to find registers, early enough to land
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const long long m = t * 16 + 2 * i + half;
                    hv[i] = ldraw4(H + m * ldh + n0 + L * 4, col_ok && m < M, H);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                acc[cb] = MFMA16(a2[ks], w2[cb][ks], acc[cb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        // ---- epilogue of the tile through the wave's own LDS patch: rows leave as 16-byte pieces of contiguous rows ----
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) E[(4 * q + tt) * LDC + cb * 16 + r] = acc[cb][tt];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ONE wait per tile, before the stores are issued: the next tile's A fragments (issued 350 MFMAs ago) and the H rows have arrived
#pragma unroll
        for (int g = 0; g < kRsG; g += 1) asm volatile("" : "+v"(a4n[g]));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a2n));
#pragma unroll
        for (int g = 0; g < kRsG; g += 1) asm volatile("" : "+v"(a4n[g]));
        if (EPI == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) arrived4(hv[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 2 * i + half;
            const long long m = t * 16 + row;
            if (col_ok && m < M) {
                float4 v = *reinterpret_cast<const float4 *>(E + row * LDC + L * 4);
                if (EPI == 1) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (relu6) { v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f); v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f); }
                }
                if (EPI == 2) {
                    const float4 hh = hv[i];
                    v.x = (hh.x > 0.f && hh.x < 6.f) ? v.x : 0.f; v.y = (hh.y > 0.f && hh.y < 6.f) ? v.y : 0.f;
                    v.z = (hh.z > 0.f && hh.z < 6.f) ? v.z : 0.f; v.w = (hh.w > 0.f && hh.w < 6.f) ? v.w : 0.f;
                }
                *reinterpret_cast<float4 *>(C + m * ldc + n0 + L * 4) = v;
                cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                          // the next tile's patch writes come after these reads
    }
    if (col_partial != nullptr) {           // per column: even rows + odd rows of a wave, the waves in order; then the workgroups (rows_colsum_reduce)
        __syncthreads();
        float4 *S = reinterpret_cast<float4 *>(lds + NC * kRsK);                 // [8][2][32] float4 over the patches
        S[(wave * 2 + half) * 32 + L] = cs;
        __syncthreads();
        if (wave == 0 && half == 0 && col_ok) {
            float4 tsum = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < kRsWaves; ++w) {
                const float4 u0 = S[(w * 2) * 32 + L], u1 = S[(w * 2 + 1) * 32 + L];
                tsum.x += u0.x + u1.x; tsum.y += u0.y + u1.y; tsum.z += u0.z + u1.z; tsum.w += u0.w + u1.w;
            }
            *reinterpret_cast<float4 *>(col_partial + (long long)wg * kNP + n0 + L * 4) = tsum;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(kRsThr, 1) void gemm_rows_resident_kernel(const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw,
                                                                         int N, long long M, const float *__restrict__ bias, int relu6,
                                                                         const float *__restrict__ H, long long ldh, float *__restrict__ C, long long ldc,
                                                                         float *__restrict__ col_partial, int nA) {
    __shared__ __attribute__((aligned(16))) float lds[kRsLdsF];
    if ((int)blockIdx.x < nA) rows_resident_body<kRsNBA, EPI>(A, lda, W, ldw, N, M, bias, relu6, H, ldh, C, ldc, col_partial, lds, 0, (int)blockIdx.x, nA);
    else rows_resident_body<kRsNBB, EPI>(A, lda, W, ldw, N, M, bias, relu6, H, ldh, C, ldc, col_partial, lds, kRsNBA * 16, (int)blockIdx.x - nA, (int)gridDim.x - nA);
}

// ---------------------------------------------------------------------------------------------------------------------
// The actor's head of one rollout step in ONE kernel (choose_action, main.py:165-169 with the network of main.py:147-150):
//   h2 = relu6(h1 @ W2 + b2);  logits = h2 @ W3 + b3;  action ~ softmax(logits) by the inverse-CDF draw of uavagent_sample_actions.
// As three launches (two gemm_rows_glds launches + the sampling kernel) a step spends 12 + 26 + 9 us of which about a third is launch,
// prologue (the first tiles' round trip), epilogue and the re-read of the logits; here a workgroup owns 32 rows from h1 to the action:
// h1 tile -> LDS (DMA), W2^T streamed through a 2-stage LDS-DMA ring (5 chunks of 40 k), h2 tile kept in LDS (and stored for the update),
// W3^T streamed through a 3-stage ring in three parts of 256 / 256 / 128 policy columns (15 chunks), all 640 logits of the 32 rows held in
// the accumulators of the 8 waves, then written to an LDS tile from which the rows are stored (coalesced) and sampled.  Same k order per output
// element as gemm_rows_glds_kernel, same arithmetic as sample_actions_kernel: results are bit-identical to the three-launch form.
// 8 waves, two per SIMD: wave w = row block w & 1, column quarter w >> 1 (4 of 16 column blocks in layer 2 -- 13 are real --, 4 / 4 / 2 of the
// 16 / 16 / 8 of the policy parts).  (With 4 waves each wave issued 13 LDS-DMA instructions per 100 MFMAs and a lone workgroup took 41 us.
// Two stages of 320 policy columns against three of 256: 33.5 against 34 us for a lone workgroup, 38.9 us at 8192 rows either way -- the
// prefetch distance was not what the boundaries cost; a workgroup issues 84 LDS-DMA instructions per wave for its 672 KB of weights.)
// ---------------------------------------------------------------------------------------------------------------------
// RB = 16-row blocks per workgroup: 2 (32 rows, wave = row block w & 1 x column quarter w >> 1) for a whole rollout batch -- 8192 rows are one
// workgroup per CU --, 1 (16 rows, wave = column eighth w) for HALF a batch, so that it still spreads over every CU (A2CRunner(pipeline_halves)):
// the time of a workgroup is mostly its stream of W2^T + W3^T (672 KB) and does not shrink with the number of workgroups.
constexpr int kHdH = 200, kHdNP = 640, kHdWaves = 8, kHdThr = kHdWaves * 64;
#ifdef UAVGEMM_STAMPS
__device__ unsigned long long *g_head_dbg;   // [workgroups][8 waves][10]: s_memtime stamps of actor_head_kernel (tools/head_stamps.py sets it)
#endif
__device__ int g_head_no_stagger;      // A/B switch (UAVAGENT_HEAD_STAGGER=0 sets it to 1 through uavagent_actor_head_f32): every wave issues at the chunk's start
constexpr int kHdW2Rows = 256, kHdW2Pass = kHdW2Rows * 10 / kHdThr;                              // 2560 float4: 5 passes
constexpr int kHdW3Rows = 320;                                                                  // (rows of a phase-1 ring stage: 256 used)
constexpr int kHdStageF = kHdW3Rows * kGlBK;                                                    // 12 800 floats per ring stage (phase 1's two)
constexpr int kHdPartRows = 256, kHdP3StageF = kHdPartRows * kGlBK, kHdP3Pass = kHdPartRows * 10 / kHdThr;   // phase 2: 10 240 floats per stage, 5 passes
// LDS: [h2 tile | h1 tile | ring]; phase 2 reuses [h1 tile | ring] as a three-stage ring and then as the logits tile
template <int RB> struct HdPlan {
    static constexpr int Rows = 16 * RB, H1F4 = Rows * kHdH / 4, H1Pass = (H1F4 + kHdThr - 1) / kHdThr;
    static constexpr int RingF = (2 * kHdStageF > 3 * kHdP3StageF - Rows * kHdH) ? 2 * kHdStageF : 3 * kHdP3StageF - Rows * kHdH;
    static constexpr int LdsF = 2 * Rows * kHdH + RingF;                                   // RB = 2: 38 400 floats = 153 600 B; RB = 1: 33 920 = 135 680 B
    static constexpr int CQ = 8 / RB, NCB1 = 16 / CQ, NCBC = 8 / CQ;                       // column groups; column blocks per wave in layer 2 / parts A, B and in part C
    static_assert(kHdW2Rows * kGlBK <= kHdStageF && 3 * kHdP3StageF <= Rows * kHdH + RingF && Rows * kHdNP <= Rows * kHdH + RingF &&
                  kHdPartRows * 10 % kHdThr == 0 && (kHdNP - 2 * kHdPartRows) * 10 % 64 == 0 && kHdNP - 2 * kHdPartRows == 128 && LdsF * 4 <= 160 * 1024 &&
                  Rows % kHdWaves == 0 && (Rows * kHdNP / 4) % kHdThr == 0, "the head's LDS plan");
};

__device__ __forceinline__ float wave_max_g(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// 10 k-steps of one chunk: NCB column blocks of one 16-row block.  pa = this lane's A row at the chunk's first k, pw = its W row of
// column block 0 in the stage (row stride 40); group g + 1 is read while the first k-step of group g issues (see gemm_rows_glds_kernel).
// mid(): called once, between k-steps 4 and 5 -- where waves 4-7 of the head issue the NEXT chunk's LDS-DMA loads (see actor_head_kernel:
// the stagger)
template <int NCB, class Mid>
__device__ __forceinline__ void head_chunk(const float *pa, const float *pw, int q, f32x4 (&acc)[NCB], Mid mid) {
    f32x4 a01[2], w01[2][NCB];
    f32x2 a2, w2[NCB];
    a01[0] = *reinterpret_cast<const f32x4 *>(pa + 4 * q);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) w01[0][cb] = *reinterpret_cast<const f32x4 *>(pw + cb * 16 * kGlBK + 4 * q);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < kGlBK / 4; ++ks) {
        const int g = ks >> 2, e = ks & 3;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if (ks == 0) {
                if (cb == 0) a01[1] = *reinterpret_cast<const f32x4 *>(pa + 16 + 4 * q);
                w01[1][cb] = *reinterpret_cast<const f32x4 *>(pw + cb * 16 * kGlBK + 16 + 4 * q);
            }
            if (ks == 4) {
                if (cb == 0) a2 = *reinterpret_cast<const f32x2 *>(pa + 32 + 2 * q);
                w2[cb] = *reinterpret_cast<const f32x2 *>(pw + cb * 16 * kGlBK + 32 + 2 * q);
            }
            if (ks == 5 && cb == 0) { mid(); __builtin_amdgcn_sched_barrier(0); }
            const float av = (g < 2) ? a01[g < 2 ? g : 0][e] : a2[e & 1];
            const float wv = (g < 2) ? w01[g < 2 ? g : 0][cb][e] : w2[cb][e & 1];
            acc[cb] = MFMA16(av, wv, acc[cb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// One tile (16 RB rows from h1 to the action) of the head; `lds` = the workgroup's HdPlan<RB>::LdsF floats, m0 = the tile's first row.
// COH (the gated kernel below): h1 was written by ANOTHER kernel while this one was running and the actions are read by it -- the h1 tile is
// loaded past this CU's L1 (LDS-DMA with sc1) and the actions are stored through (agent-scope atomics).
template <int RB, bool COH>
__device__ __forceinline__ void actor_head_tile(float *lds, const long long m0, const float *__restrict__ h1, const float *__restrict__ w2t,
                                                const float *__restrict__ b2, const float *__restrict__ w3t, const float *__restrict__ b3p,
                                                const float *__restrict__ uni, long long n_rows, int n_act,
                                                float *__restrict__ h2_out, float *__restrict__ logits, long long ldl,
                                                long long *__restrict__ action) {
    using P = HdPlan<RB>;
    constexpr int kHdRows = P::Rows, kHdH1F4 = P::H1F4, kHdH1Pass = P::H1Pass, NCB1 = P::NCB1, NCBC = P::NCBC;
    // [h2 tile | h1 tile | ring]: the policy head (phase 2) no longer needs h1 and runs a THREE-stage ring over [h1 tile | ring]
    float *const sH2 = lds, *const sH1 = lds + kHdRows * kHdH, *const ring = lds + 2 * kHdRows * kHdH;
    float *const ring3 = sH1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4, rb = (RB == 2) ? (wave & 1) : 0, qt = (RB == 2) ? (wave >> 1) : wave;   // row block, column group
    const bool stagger_first = (wave < 4) || g_head_no_stagger;      // (waves w and w + 4 share a SIMD)
    unsigned long long hs0 = 0, hs1 = 0, hs2 = 0, hs3 = 0, hs4 = 0, hs5 = 0, hs6 = 0, hs7 = 0, hs_b = 0, hs_t = 0, hs_wait = 0, hs_first = 0;
    (void)hs0; (void)hs1; (void)hs2; (void)hs3; (void)hs4; (void)hs5; (void)hs6; (void)hs7; (void)hs_b; (void)hs_t; (void)hs_wait; (void)hs_first;
    GEMM_STAMP(hs0);
    // the uniforms of this wave's rows, loaded NOW: read where the draw needs them, each was a global round trip (~2 000 cycles by the stamps)
    // in the middle of a dependent chain, once per row
    float u_row[kHdRows / kHdWaves];
#pragma unroll
    for (int i = 0; i < kHdRows / kHdWaves; ++i) {
        const long long m = m0 + wave * (kHdRows / kHdWaves) + i;
        u_row[i] = uni[m < n_rows ? m : 0];
    }
    using set0_t = std::integral_constant<int, 0>;
    using set1_t = std::integral_constant<int, 1>;
    // (every wait below is vmcnt(0), so the waves need not issue the same number of LDS-DMA instructions: passes are cut at whole waves)

    // ---- phase 0: the h1 tile (32 x 200 contiguous floats) and chunk 0 of W2^T ----
    {
        const long long lim = (n_rows - m0 < kHdRows ? n_rows - m0 : kHdRows) * (kHdH / 4);      // float4s of the tile that exist
        const float *g0 = h1 + m0 * kHdH;
#pragma unroll
        for (int p = 0; p < kHdH1Pass; ++p) {
            const int slot = p * kHdWaves + wave;
            if (slot * 64 < kHdH1F4) {
                const int idx = slot * 64 + lane;
                const float *g = (idx < lim) ? g0 + idx * 4 : g_zero16;
                if (kHdH1F4 % 64 == 0 || idx < kHdH1F4)          // (RB = 1: the tile ends inside the last slot; the lanes beyond it would write into the ring)
                    __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)g, (lds_void_t *)(sH1 + slot * 256), 16, 0, COH ? 16 : 0);   // (aux 16 = sc1)
            }
        }
    }
    // W2^T [200, 200] as 256 tile rows (rows >= 200: zeros) x 40 k per chunk; two pointer sets (even / odd chunks), see gemm_rows_glds_kernel
    const float *s2[2][kHdW2Pass];
    unsigned real2 = 0u;
#pragma unroll
    for (int p = 0; p < kHdW2Pass; ++p) {
        const int idx = (p * kHdWaves + wave) * 64 + lane, row = idx / 10, c4 = idx - row * 10;
        const float *g = g_zero16;
        if (row < kHdH) { g = w2t + row * kHdH + c4 * 4; real2 |= 1u << p; }
        s2[0][p] = g;
        s2[1][p] = g + (((real2 >> p) & 1u) ? kGlBK : 0);
    }
    auto issue2 = [&](auto set_c, int c) {
        constexpr int SET = decltype(set_c)::value;
        float *stage = ring + (c & 1) * kHdStageF;
#pragma unroll
        for (int p = 0; p < kHdW2Pass; ++p) {
            if (c >= 2) s2[SET][p] += (((real2 >> p) & 1u) ? 2 * kGlBK : 0);
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)s2[SET][p], (lds_void_t *)(stage + (p * kHdWaves + wave) * 256), 16, 0, 0);
        }
    };
    issue2(set0_t{}, 0);

    // ---- phase 1: h2 = relu6(h1 @ W2 + b2): wave = 16 rows x 4 of 16 column blocks (13 real) ----
    f32x4 acc1[NCB1];
#pragma unroll
    for (int c = 0; c < NCB1; ++c) acc1[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *pa1 = sH1 + (rb * 16 + r) * kHdH;
    auto chunk2 = [&](auto set_c, int c) {
        GEMM_STAMP(hs_b);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GEMM_STAMP(hs_t);
        hs_wait += hs_t - hs_b;
        if (c == 0) hs_first = hs_t;
        // The stagger: after the barrier both waves of a SIMD would issue their 5 LDS-DMA instructions at once (60-185 cycles of issue each,
        // MI355X_MICROARCH.md) and leave the matrix pipe idle meanwhile; waves 0-3 issue at the start of the chunk, waves 4-7 in its middle
        // (under their partner's MFMAs and vice versa).  Per wave the order of its loads is unchanged, so the counted waits still hold.
        if (stagger_first && c + 1 < 5) issue2(set_c, c + 1);
        head_chunk<NCB1>(pa1 + c * kGlBK, ring + (c & 1) * kHdStageF + (qt * NCB1 * 16 + r) * kGlBK, q, acc1,
                         [&] { if (!stagger_first && c + 1 < 5) issue2(set_c, c + 1); });
    };
    chunk2(set1_t{}, 0); chunk2(set0_t{}, 1); chunk2(set1_t{}, 2); chunk2(set0_t{}, 3); chunk2(set1_t{}, 4);
    // bias + relu6 -> the h2 tile in LDS (row = m, k-contiguous: the A operand of the policy head)
#pragma unroll
    for (int cb = 0; cb < NCB1; ++cb) {
        const int col = (qt * NCB1 + cb) * 16 + r;
        if (col < kHdH) {
            const float bv = b2[col];
#pragma unroll
            for (int t = 0; t < 4; ++t) sH2[(rb * 16 + 4 * q + t) * kHdH + col] = fminf(fmaxf(acc1[cb][t] + bv, 0.0f), 6.0f);
        }
    }
    GEMM_STAMP(hs1);                                   // phase 1's MFMAs done, h2 written to LDS
    __syncthreads();                                   // (every wave is past the ring's last read too: phase 2 may refill it)
    GEMM_STAMP(hs2);

    // ---- phase 2: logits = h2 @ W3 + b3 in three PARTS of 256 / 256 / 128 policy columns x 5 chunks of 40 k; wave = 16 rows x 4 / 4 / 2 of a
    // part's 16 / 16 / 8 column blocks.  Stages of 256 rows x 40 k (40 KB): three of them fit [h1 tile | ring], so chunk p + 2 is issued when
    // chunk p starts and has two chunk times to land -- with two stages of 320 rows a lone workgroup spent 14 of its 34 us at the 15 chunk
    // boundaries waiting for loads issued one chunk earlier.  Same k order per output element as before (chunks of 40, same k bijection).
    const float *s3[2][kHdP3Pass];
#pragma unroll
    for (int p = 0; p < kHdP3Pass; ++p) {
        const int idx = (p * kHdWaves + wave) * 64 + lane, row = idx / 10, c4 = idx - row * 10;     // row < 256: five passes fill a stage exactly
        s3[0][p] = w3t + row * kHdH + c4 * 4;
        s3[1][p] = s3[0][p] + kGlBK;
    }
    // chunk index p3 = 0..14: part p3 / 5, k chunk p3 % 5; a set is advanced from chunk p3 - 2 to chunk p3.  Part 2 has 128 rows = 20 wave
    // slots: passes 0 and 1 whole, pass 2 for waves 0-3 only (w3t has 640 rows: nothing beyond them is read).
    auto off3 = [](int p3) { return (p3 / 5) * (kHdPartRows * kHdH) + (p3 % 5) * kGlBK; };
    auto issue3 = [&](auto set_c, int p3) {
        constexpr int SET = decltype(set_c)::value;
        float *stage = ring3 + (p3 % 3) * kHdP3StageF;
        const int delta = (p3 >= 2) ? off3(p3) - off3(p3 - 2) : 0;
#pragma unroll
        for (int p = 0; p < kHdP3Pass; ++p) {
            const int slot = p * kHdWaves + wave;
            if (p3 < 10 || slot < (kHdNP - 2 * kHdPartRows) * 10 / 64) {
                if (p3 >= 2) s3[SET][p] += delta;
                __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)s3[SET][p], (lds_void_t *)(stage + slot * 256), 16, 0, 0);
            }
        }
    };
    issue3(set0_t{}, 0);
    // store the h2 tile for the update while the first chunk is on its way: 1600 float4, coalesced rows
    {
        const long long lim = (n_rows - m0 < kHdRows ? n_rows - m0 : kHdRows) * (kHdH / 4);
#pragma unroll
        for (int i = 0; i < kHdH1Pass; ++i) {
            const int idx = tid + kHdThr * i;
            if (idx < lim) *reinterpret_cast<float4 *>(h2_out + m0 * kHdH + idx * 4) = *reinterpret_cast<const float4 *>(sH2 + idx * 4);
        }
    }
    issue3(set1_t{}, 1);
    f32x4 accA[NCB1], accB[NCB1], accC[NCBC];
#pragma unroll
    for (int c = 0; c < NCB1; ++c) { accA[c] = f32x4{0.f, 0.f, 0.f, 0.f}; accB[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int c = 0; c < NCBC; ++c) accC[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *pa2 = sH2 + (rb * 16 + r) * kHdH;
    // chunk p3 has landed once at most the loads of chunk p3 + 1 are outstanding (5 per wave; in part 2: 3 for waves 0-3, 2 for the others)
    auto land3 = [&](int p3) {
        GEMM_STAMP(hs_b);
        if (p3 + 1 >= 15) wait_vmcnt<0>();
        else if (p3 + 1 < 10) wait_vmcnt<kHdP3Pass>();
        else if (wave < 4) wait_vmcnt<3>();
        else wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GEMM_STAMP(hs_t);
        hs_wait += hs_t - hs_b;
        if (p3 == 0) hs3 = hs_t;                       // phase 2's first chunk has landed (h2 store + two chunk issues lie before)
    };
    auto chunk3 = [&](auto set_c, auto ncb_c, int p3, f32x4 (&acc)[decltype(ncb_c)::value]) {       // SET = parity of p3: chunk p3 + 2 is issued here
        constexpr int NCB = decltype(ncb_c)::value;
        land3(p3);
        if (stagger_first && p3 + 2 < 15) issue3(set_c, p3 + 2);       // its stage was read last in chunk p3 - 1: every wave is past that
        head_chunk<NCB>(pa2 + (p3 % 5) * kGlBK, ring3 + (p3 % 3) * kHdP3StageF + (qt * NCB * 16 + r) * kGlBK, q, acc,
                        [&] { if (!stagger_first && p3 + 2 < 15) issue3(set_c, p3 + 2); });
    };
    using n4_t = std::integral_constant<int, NCB1>;      // column blocks of a wave in parts A and B ...
    using n2_t = std::integral_constant<int, NCBC>;      // ... and in part C
    chunk3(set0_t{}, n4_t{}, 0, accA); chunk3(set1_t{}, n4_t{}, 1, accA); chunk3(set0_t{}, n4_t{}, 2, accA); chunk3(set1_t{}, n4_t{}, 3, accA); chunk3(set0_t{}, n4_t{}, 4, accA);
    chunk3(set1_t{}, n4_t{}, 5, accB); chunk3(set0_t{}, n4_t{}, 6, accB); chunk3(set1_t{}, n4_t{}, 7, accB); chunk3(set0_t{}, n4_t{}, 8, accB); chunk3(set1_t{}, n4_t{}, 9, accB);
    chunk3(set0_t{}, n2_t{}, 10, accC); chunk3(set1_t{}, n2_t{}, 11, accC); chunk3(set0_t{}, n2_t{}, 12, accC); chunk3(set1_t{}, n2_t{}, 13, accC); chunk3(set0_t{}, n2_t{}, 14, accC);
    GEMM_STAMP(hs4);                                   // phase 2's MFMAs done
    __syncthreads();                                   // the ring is free: it becomes the [32][640] logits tile

    // ---- phase 3: + bias -> logits tile in LDS; coalesced store; one wave samples 4 rows ----
    float *sL = ring3;
#pragma unroll
    for (int cb = 0; cb < NCB1; ++cb) {
        const int colA = (qt * NCB1 + cb) * 16 + r, colB = kHdPartRows + colA;
        const float bA = b3p[colA], bB = b3p[colB];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sL[(rb * 16 + 4 * q + t) * kHdNP + colA] = accA[cb][t] + bA;
            sL[(rb * 16 + 4 * q + t) * kHdNP + colB] = accB[cb][t] + bB;
        }
    }
#pragma unroll
    for (int cb = 0; cb < NCBC; ++cb) {
        const int colC = 2 * kHdPartRows + (qt * NCBC + cb) * 16 + r;
        const float bC = b3p[colC];
#pragma unroll
        for (int t = 0; t < 4; ++t) sL[(rb * 16 + 4 * q + t) * kHdNP + colC] = accC[cb][t] + bC;
    }
    __syncthreads();
    GEMM_STAMP(hs5);                                   // logits tile in LDS
    {
        const int rows = (int)(n_rows - m0 < kHdRows ? n_rows - m0 : kHdRows);
#pragma unroll 5
        for (int i = 0; i < kHdRows * kHdNP / 4 / kHdThr; ++i) {                  // 5120 (2560) float4 over 512 threads
            const int idx = tid + kHdThr * i, row = idx / (kHdNP / 4), c4 = idx - row * (kHdNP / 4);
            if (row < rows) *reinterpret_cast<float4 *>(logits + (m0 + row) * ldl + c4 * 4) = *reinterpret_cast<const float4 *>(sL + row * kHdNP + c4 * 4);
        }
    }
    GEMM_STAMP(hs6);                                   // logits stores issued
    constexpr int PER = 10;                            // sample_actions_kernel<10>: lane l owns columns 10 l .. 10 l + 9
#pragma unroll
    for (int i = 0; i < kHdRows / kHdWaves; ++i) {
        const int lr = wave * (kHdRows / kHdWaves) + i;
        const long long m = m0 + lr;
        if (m >= n_rows) break;
        const float *row = sL + lr * kHdNP;
        float v[PER];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane * PER + k;
            v[k] = (c < n_act) ? row[c] : -3.0e38f;
            mx = fmaxf(mx, v[k]);
        }
        mx = wave_max_g(mx);
        float loc = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane * PER + k;
            v[k] = (c < n_act) ? draw_exp(v[k] - mx) : 0.f;
            loc += v[k];
        }
        float incl = loc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        const float total = __shfl(incl, 63, 64);
        const float target = u_row[i] * total;
        const unsigned long long over = __ballot(incl > target);
        int a = n_act - 1;
        if (over != 0ull) {
            // every lane walks its OWN PER values from its own exclusive prefix; the walk of the lane that holds the crossing is the answer
            // (the same sums in the same order as walking that lane's values by broadcast, without ten dependent cross-lane reads)
            const int first = __ffsll((long long)over) - 1;
            float c = incl - loc;
            int fk = -1;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                c += v[k];
                if (fk < 0 && c > target) fk = k;
            }
            fk = __shfl(fk, first, 64);
            a = fk < 0 ? first * PER + PER - 1 : first * PER + fk;
            if (a > n_act - 1) a = n_act - 1;
        }
        if (lane == 0) {
            if (COH) __hip_atomic_store(action + m, (long long)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else action[m] = a;
        }
    }
#ifdef UAVGEMM_STAMPS
    GEMM_STAMP(hs7);
    if (g_head_dbg != nullptr && lane == 0) {
        unsigned long long *d = g_head_dbg + ((long long)blockIdx.x * kHdWaves + wave) * 10;
        d[0] = hs0; d[1] = hs_first - hs0; d[2] = hs1 - hs0; d[3] = hs2 - hs0; d[4] = hs3 - hs0; d[5] = hs4 - hs0; d[6] = hs5 - hs0; d[7] = hs6 - hs0; d[8] = hs7 - hs0;
        d[9] = hs_wait;
    }
#endif
}

template <int RB>
__global__ __launch_bounds__(kHdThr, 1) void actor_head_kernel(const float *__restrict__ h1, const float *__restrict__ w2t, const float *__restrict__ b2,
                                                                const float *__restrict__ w3t, const float *__restrict__ b3p,
                                                                const float *__restrict__ uni, long long n_rows, int n_act,
                                                                float *__restrict__ h2_out, float *__restrict__ logits, long long ldl,
                                                                long long *__restrict__ action) {
    __shared__ __attribute__((aligned(16))) float lds[HdPlan<RB>::LdsF];
    actor_head_tile<RB, false>(lds, (long long)blockIdx.x * HdPlan<RB>::Rows, h1, w2t, b2, w3t, b3p, uni, n_rows, n_act, h2_out, logits, ldl, action);
}

// ---------------------------------------------------------------------------------------------------------------------
// The head of a WHOLE rollout as one persistent launch (uavagent_actor_head_gated_f32) beside the env library's persistent rollout kernel
// (uavenv_rollout_gated, include/uavenv.h, which also states the protocol): a workgroup owns a PAIR of 16-row blocks for all T steps and
// alternates between them -- while the env kernel steps and encodes one block, this kernel runs the head of the other.  Per block and step:
// wait until gate_obs[b] >= t + 1 (h1[t] of the block's rows is in memory), one 16-row tile exactly as actor_head_kernel<1> computes it
// (same bits), actions stored through, s_waitcnt vmcnt(0), gate_act[b] = t + 1.  No kernel boundary, graph node or host call between a
// step's kernels any more (a2c_single_thread.py:113-118 is a loop over independent workers).  Every wait is bounded: after spin_us the
// wave stores kGateErr in the library's host-mapped error word and leaves (uavagent_device_error).  At most UAVAGENT_GATE_VGPRS = 112
// VGPRs (amdgpu_num_vgpr): two of this kernel's waves and two of the env kernel's (<= 144 VGPRs each) share a SIMD's 512.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kGateErr = 0x47415445u;      // "GATE"
#ifdef UAVAGENT_GATE_STAMPS     /* diagnostic build (tools/gated_timeline.py): s_memrealtime of pair 0's events, same buffer as the env kernel's stamps */
__device__ unsigned long long *g_hgate_dbg;   // [T][2 halves][8]: slot 0 = h1 seen, 1 = actions published
#define HGATE_STAMP(t, half, k) do { if (g_hgate_dbg != nullptr && pair == 0 && threadIdx.x == 0) g_hgate_dbg[((t) * 2 + (half)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HGATE_STAMP(t, half, k) do { } while (0)
#endif
__device__ __forceinline__ bool head_gate_wait(uint32_t *word, uint32_t need, uint32_t *err, uint32_t spin_us) {
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long budget = (unsigned long long)spin_us * 100ull;           // s_memrealtime ticks at 100 MHz
    bool ok = false;
    for (;;) {
        uint32_t v = 0u;
        if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)v) >= need) { ok = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t_start > budget) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (!ok && (threadIdx.x & 63) == 0) __hip_atomic_store(err, kGateErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("" ::: "memory");
    return ok;
}
#ifdef UAVAGENT_GATE_NOCAP      /* timing experiment: the kernel without its register cap (such a build cannot run beside its partner) */
#define UAVAGENT_GATE_CAP
#else
#ifndef UAVAGENT_GATE_VGPRS
#define UAVAGENT_GATE_VGPRS 112     /* this kernel's share of a SIMD lane's 512 registers is 2 x this; the env kernel has the rest: 144
                                     * (uavenv_gated_kernel.h).  112 + 144 against 128 + 128: the pair 3.75-3.78 against 3.86 ms per rollout, same box
                                     * (profiles/r04gz_gated_pair_vgpr_split_sweep.txt): the env side is the one the rollout waits for */
#endif
#define UAVAGENT_GATE_CAP __attribute__((amdgpu_num_vgpr(UAVAGENT_GATE_VGPRS / 2)))   /* (gfx90a and later: the attribute counts VGPR + AGPR pairs) */
#endif
__global__ __launch_bounds__(kHdThr) UAVAGENT_GATE_CAP void actor_head_gated_kernel(const float *__restrict__ h1, const float *__restrict__ w2t, const float *__restrict__ b2,
                                                                      const float *__restrict__ w3t, const float *__restrict__ b3p,
                                                                      const float *__restrict__ uni, long long n_rows, int n_steps, int n_act,
                                                                      float *__restrict__ h2_out, float *__restrict__ logits, long long ldl,
                                                                      long long *__restrict__ action, uint32_t *gate_obs, uint32_t *gate_act,
                                                                      uint32_t *claim, uint32_t *err, uint32_t spin_us) {
    // (DYNAMIC LDS: with the 135 KB declared statically hipcc reasons that only two waves per SIMD can ever be resident, ignores
    //  its register cap and takes 143 VGPRs -- and then this kernel and its partner no longer fit one CU: 2 x 112 + 2 x 144)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n_blocks = (int)((n_rows + 15) / 16), n_pairs = (n_blocks + 1) >> 1;
    __shared__ int s_pair;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_pair = (int)atomicAdd(claim, 1u);      // pairs are claimed in arrival order (uavenv_gated_kernel.h: "Residency")
        __syncthreads();
        const int pair = s_pair;
        if (pair >= n_pairs) break;
        for (int t = 0; t < n_steps; ++t) {
            const long long row0 = (long long)t * n_rows;
            for (int half = 0; half < 2; ++half) {
                const int blk = 2 * pair + half;
                if (blk >= n_blocks) continue;
                if (!head_gate_wait(gate_obs + blk, (uint32_t)t + 1u, err, spin_us)) return;
                HGATE_STAMP(t, half, 0);
                __syncthreads();                             // every wave has left the previous tile's LDS
                // (the weights' base pointers pass through an empty asm: otherwise hipcc hoists the per-lane LDS-DMA source pointers of every
                //  tile -- 40 VGPRs -- out of the step loop and, capped at 96 VGPRs, spills them)
                const float *w2t_i = w2t, *w3t_i = w3t, *b2_i = b2, *b3p_i = b3p;
                asm volatile("" : "+s"(w2t_i), "+s"(w3t_i), "+s"(b2_i), "+s"(b3p_i));
                actor_head_tile<1, true>(lds, 16ll * blk, h1 + row0 * kHdH, w2t_i, b2_i, w3t_i, b3p_i, uni + row0, n_rows, n_act, h2_out + row0 * kHdH,
                                         logits + row0 * ldl, ldl, action + row0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the actions have left
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(gate_act + blk, (uint32_t)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                HGATE_STAMP(t, half, 1);
            }
        }
    }
}

#ifdef UAVGEMM_STAMPS
extern "C" int uavagent_debug_set_head_stamps(void *dev_ptr) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_head_dbg), &dev_ptr, sizeof(void *)) == hipSuccess ? 0 : -1;
}
#endif

template <bool NT, bool VEC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw,
                                                            int K, int N, long long M, const float *__restrict__ bias, int relu6,
                                                            const float *__restrict__ H, long long ldh, float *__restrict__ C, long long ldc) {
    static_assert(!VEC, "the aligned shapes run gemm_rows_vec_kernel");
    __shared__ __attribute__((aligned(16))) float lds[2 * (kRowsATile + kRowsWTile)];
    float *const sA = lds, *const sW = lds + 2 * kRowsATile;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const long long m0 = (long long)blockIdx.x * kRowsBM;

    for (int i = tid; i < 2 * (kRowsATile + kRowsWTile); i += 256) lds[i] = 0.0f;   // padding rows / columns must be finite
    __syncthreads();

    constexpr int NA = 10, NW = 17;                              // 128 x 20 and <= 208 x 20 floats over 256 threads
    const int wpr = NT ? kRowsBK : N;                            // items per W-tile row
    const int w_rows = NT ? N : kRowsBK;
    float fa[NA], fw[NW];
    auto load_chunk = [&](int c) {
        const int k0 = c * kRowsBK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i, row = idx / kRowsBK, kk = idx - row * kRowsBK;
            const long long m = m0 + row;
            fa[i] = ldraw1(A + m * lda + k0 + kk, (row < kRowsBM) && (m < M) && (k0 + kk < K), A);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = tid + 256 * i, row = idx / wpr, cc = idx - row * wpr;
            if (NT) fw[i] = ldraw1(W + (long long)row * ldw + k0 + cc, (row < w_rows) && (k0 + cc < K), W);       // W [N, K]
            else fw[i] = ldraw1(W + (long long)(k0 + row) * ldw + cc, (row < w_rows) && (k0 + row < K), W);       // W [K, N]
        }
    };
    auto store_chunk = [&](int buf, int c) {
        const int k0 = c * kRowsBK;
        float *dA = sA + buf * kRowsATile, *dW = sW + buf * kRowsWTile;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i, row = idx / kRowsBK, kk = idx - row * kRowsBK;
            if (row < kRowsBM) dA[row * kRowsLD + kk] = mask_f(fa[i], (m0 + row < M) && (k0 + kk < K));
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = tid + 256 * i, row = idx / wpr, cc = idx - row * wpr;
            if (row < w_rows) dW[row * (NT ? kRowsLD : kNP) + cc] = mask_f(fw[i], NT ? (k0 + cc < K) : (k0 + row < K));
        }
    };

    f32x4 acc[2][kRB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < kRB; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int n_chunks = (K + kRowsBK - 1) / kRowsBK;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) load_chunk(c + 1);
        const float *tA = sA + buf * kRowsATile + (wave * 32 + r) * kRowsLD + q;
        const float *tW = NT ? (sW + buf * kRowsWTile + r * kRowsLD + q) : (sW + buf * kRowsWTile + q * kNP + r);
#pragma unroll
        for (int ks = 0; ks < kRowsBK / 4; ++ks) {               // k >= K was staged as zeros in both tiles
            const float a0 = tA[ks * 4], a1 = tA[16 * kRowsLD + ks * 4];
            float b[kRB];
#pragma unroll
            for (int cb = 0; cb < kRB; ++cb) b[cb] = NT ? tW[cb * 16 * kRowsLD + ks * 4] : tW[ks * 4 * kNP + cb * 16];
#pragma unroll
            for (int cb = 0; cb < kRB; ++cb) { acc[0][cb] = MFMA16(a0, b[cb], acc[0][cb]); acc[1][cb] = MFMA16(a1, b[cb], acc[1][cb]); }
        }
        if (c + 1 < n_chunks) store_chunk(buf ^ 1, c + 1);
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
struct TnShape { int plan, n_jt, n_split; long long rows_per_split; size_t ws_bytes; };
// The split of M: one workgroup per CU (256 on MI355X) in total; every split a whole number of 32-row chunks.
TnShape tn_shape(long long M, int n_j) {
    TnShape s;
    s.plan = (n_j <= 208) ? 13 : 20;
    s.n_jt = (n_j <= 208) ? 1 : (n_j + 319) / 320;
    int want = 256 / s.n_jt;
    if (want < 8) want = 8;
    const long long chunks = (M + kTnBK - 1) / kTnBK;
    long long per = (chunks + want - 1) / want;
    if (per < 1) per = 1;
    s.rows_per_split = per * kTnBK;
    s.n_split = (int)((M + s.rows_per_split - 1) / s.rows_per_split);
    if (s.n_split < 1) s.n_split = 1;
    const int nblkw = (s.plan == 13) ? TnPlan<13>::NBLKW : TnPlan<20>::NBLKW;
    s.ws_bytes = (size_t)s.n_split * s.n_jt * 8 * nblkw * 64 * sizeof(f32x4);
#ifdef UAVGEMM_STAMPS
    s.ws_bytes += (size_t)s.n_split * s.n_jt * 8 * 4 * sizeof(unsigned long long);
#endif
    return s;
}
}  // namespace

namespace {
template <int PLAN> int tn_plan_check() {
    using P = TnPlan<PLAN>;
    int owners[kRB][P::NCB] = {};
    for (int w = 0; w < 8; ++w) {
        for (int blk = 0; blk < P::NR * P::NC; ++blk) {
            const int rb = P::r0(w) + blk / P::NC, cb = P::c0(w) + blk % P::NC;
            if (rb < 0 || rb >= kRB || cb < 0 || cb >= P::NCB) return -1;
            ++owners[rb][cb];
        }
        for (int i = 0; i < P::NX; ++i) {
            const int rb = P::xr(w, i), cb = P::xc(w, i);
            if (rb < 0 || rb >= kRB || cb < 0 || cb >= P::NCB) return -2;     // (duplicates too must read inside the tile)
            if (P::xvalid(w, i)) ++owners[rb][cb];
        }
    }
    for (int rb = 0; rb < kRB; ++rb)
        for (int cb = 0; cb < P::NCB; ++cb)
            if (owners[rb][cb] != 1) return -3;
    return 8 * P::NBLKW;                                                     // MFMAs issued per k-step and workgroup
}
}  // namespace

// Test hook (host only, no GPU): every 16 x 16 output block of a dW tile is owned by exactly one wavefront of the plan; returns the
// number of MFMAs a workgroup issues per k-step (176 for 169 real blocks of plan 13, 264 for 260 of plan 20), negative on a hole / overlap.
extern "C" int uavagent_debug_tn_plan_check(int32_t plan) {
    if (plan == 13) return tn_plan_check<13>();
    if (plan == 20) return tn_plan_check<20>();
    return fail3(UAVAGENT_E_INVALID, "tn_plan_check: plans 13 and 20 exist");
}

extern "C" size_t uavagent_gemm_tn_workspace_bytes(int64_t m_rows, int32_t n_j) {
    if (m_rows < 0 || n_j < 1) return 0;
    return tn_shape(m_rows, n_j).ws_bytes;
}

extern "C" int uavagent_gemm_tn_f32(const float *a, const float *b, int64_t m_rows, int32_t n_i, int32_t n_j, int64_t ldb, float *c, int64_t ldc,
                                    float *dbias_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!a || !b || !c || !workspace) return fail3(UAVAGENT_E_INVALID, "gemm_tn: null pointer");
    if (m_rows < 1 || n_i < 4 || n_i > 200 || (n_i & 3) || n_j < 1 || n_j > 640 || ldb < n_j || ldc < n_j || ldb > 65536)
        return fail3(UAVAGENT_E_INVALID, "gemm_tn: need m_rows >= 1, n_i % 4 == 0 in [4, 200], 1 <= n_j <= 640, n_j <= ldb <= 65536, ldc >= n_j");
    if (!aligned16(a) || !aligned16(workspace)) return fail3(UAVAGENT_E_INVALID, "gemm_tn: a and workspace must be 16-byte aligned");
    const TnShape s = tn_shape(m_rows, n_j);
    if (workspace_bytes < s.ws_bytes) return fail3(UAVAGENT_E_INVALID, "gemm_tn: workspace smaller than uavagent_gemm_tn_workspace_bytes()");
    hipStream_t st = (hipStream_t)stream;
    f32x4 *slabs = reinterpret_cast<f32x4 *>(workspace);
    // float4 staging of b: rows start on 16-byte boundaries and the columns up to the next multiple of 4 past n_j exist (inside the row
    // stride) -- they must hold finite values (the learner keeps its [M, 625] logits in rows of 640 with a zero tail)
    const bool bvec = aligned16(b) && (ldb % 4 == 0) && (ldb >= ((n_j + 3) & ~3));
    const dim3 grid((unsigned)(s.n_split * s.n_jt)), blk(512);
    int dbg = 0;
#ifdef UAVGEMM_STAMPS
    if (const char *e = std::getenv("UAVGEMM_DBG")) dbg = std::atoi(e);     // diagnostic build only: read per call
#endif
#define UAV_TN(PLAN_)                                                                                                                        \
    do {                                                                                                                                     \
        if (bvec) hipLaunchKernelGGL((gemm_tn_kernel<PLAN_, true>), grid, blk, 0, st, a, n_i, b, (long long)ldb, n_j, (long long)m_rows, s.rows_per_split, s.n_jt, s.n_split, slabs, dbg); \
        else hipLaunchKernelGGL((gemm_tn_kernel<PLAN_, false>), grid, blk, 0, st, a, n_i, b, (long long)ldb, n_j, (long long)m_rows, s.rows_per_split, s.n_jt, s.n_split, slabs, dbg);    \
        const int n_items = s.n_jt * 8 * TnPlan<PLAN_>::NBLKW * 64;                                                                          \
        hipLaunchKernelGGL((gemm_tn_reduce<PLAN_>), dim3((n_items + 63) / 64), dim3(256), 0, st, slabs, s.n_split, s.n_jt, n_i, n_j, c, (long long)ldc, dbias_out); \
    } while (0)
    if (s.plan == 13) UAV_TN(13); else UAV_TN(20);
#undef UAV_TN
    if (hipGetLastError() != hipSuccess) return fail3(UAVAGENT_E_HIP, "gemm_tn: launch failed");
    return UAVAGENT_OK;
}

// CU count of the CURRENT device (the caller has made the operands' device current), read once per device ordinal.
static int cu_count_of_current_device() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 2) v = 256;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

extern "C" int uavagent_actor_head_f32(const float *h1, const float *w2t, const float *b2, const float *w3t_padded, const float *b3_padded,
                                       const float *uniforms, int64_t n_rows, int32_t n_hidden, int32_t n_actions, float *h2_out,
                                       float *logits_out, int64_t ld_logits, int64_t *actions_out, void *stream) {
    if (!h1 || !w2t || !b2 || !w3t_padded || !b3_padded || !uniforms || !h2_out || !logits_out || !actions_out)
        return fail3(UAVAGENT_E_INVALID, "actor_head: null pointer");
    if (n_hidden != kHdH || n_actions <= 576 || n_actions > kHdNP || ld_logits < kHdNP || (ld_logits & 3) || n_rows < 0)
        return fail3(UAVAGENT_E_INVALID, "actor_head: built for 200 hidden units and 577..640 actions (the reference's 625 = 5^4; 10 policy columns "
                                         "per lane, like uavagent_sample_actions at that width), ld_logits >= 640 and a multiple of 4");
    if (!aligned16(h1) || !aligned16(w2t) || !aligned16(w3t_padded) || !aligned16(h2_out) || !aligned16(logits_out))
        return fail3(UAVAGENT_E_INVALID, "actor_head: matrices must be 16-byte aligned");
    if (n_rows == 0) return UAVAGENT_OK;
    // 32-row workgroups while they give every CU one (a whole rollout batch of 8192 rows on 256 CUs); 16-row workgroups for fewer rows (half a
    // batch on its own stream): the time of a workgroup is its weight stream, so fewer, larger workgroups would only leave CUs idle
    {   // A/B switch of the DMA stagger, read once per process
        static const int no_stagger = [] { const char *e = getenv("UAVAGENT_HEAD_STAGGER"); return (e && e[0] == '0') ? 1 : 0; }();
        static bool pushed = false;
        if (!pushed && no_stagger) {
            if (hipMemcpyToSymbol(HIP_SYMBOL(g_head_no_stagger), &no_stagger, sizeof(int)) != hipSuccess) return fail3(UAVAGENT_E_HIP, "actor_head: hipMemcpyToSymbol failed");
        }
        pushed = true;
    }
    const char *force_env = getenv("UAVAGENT_HEAD_RB");          // tests / A-B runs: 1 or 2 forces the tile height (read per call)
    const int force_rb = force_env ? atoi(force_env) : 0;
    const int n_cu = cu_count_of_current_device();
    const bool small = force_rb ? (force_rb == 1) : (n_rows <= 24ll * n_cu);
    if (small)
        hipLaunchKernelGGL(actor_head_kernel<1>, dim3((unsigned)((n_rows + 15) / 16)), dim3(kHdThr), 0, (hipStream_t)stream, h1, w2t, b2, w3t_padded,
                           b3_padded, uniforms, (long long)n_rows, (int)n_actions, h2_out, logits_out, (long long)ld_logits,
                           reinterpret_cast<long long *>(actions_out));
    else
        hipLaunchKernelGGL(actor_head_kernel<2>, dim3((unsigned)((n_rows + 31) / 32)), dim3(kHdThr), 0, (hipStream_t)stream, h1, w2t, b2, w3t_padded,
                           b3_padded, uniforms, (long long)n_rows, (int)n_actions, h2_out, logits_out, (long long)ld_logits,
                           reinterpret_cast<long long *>(actions_out));
    if (hipGetLastError() != hipSuccess) return fail3(UAVAGENT_E_HIP, "actor_head: launch failed");
    return UAVAGENT_OK;
}

#ifdef UAVAGENT_GATE_STAMPS
extern "C" int uavagent_debug_set_gate_stamps(void *dev_ptr) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_hgate_dbg), &dev_ptr, sizeof(void *)) == hipSuccess ? 0 : -1;
}
#endif

// ---- the gated head's error word: host-mapped, one per process (allocated by uavagent_gate_prepare, never inside a launch) ----
static uint32_t *g_gate_err_host = nullptr, *g_gate_err_dev = nullptr;
extern "C" int uavagent_gate_prepare(void) {
    if (g_gate_err_host != nullptr) return UAVAGENT_OK;
    void *hp = nullptr, *dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return fail3(UAVAGENT_E_HIP, "gate_prepare: hipHostMalloc failed");
    std::memset(hp, 0, 64);
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipHostFree(hp); return fail3(UAVAGENT_E_HIP, "gate_prepare: hipHostGetDevicePointer failed"); }
    g_gate_err_host = static_cast<uint32_t *>(hp); g_gate_err_dev = static_cast<uint32_t *>(dp);
    return UAVAGENT_OK;
}
extern "C" int uavagent_device_error(uint32_t *code) {
    if (!code) return fail3(UAVAGENT_E_INVALID, "device_error: null pointer");
    *code = g_gate_err_host ? *(volatile uint32_t *)g_gate_err_host : 0u;
    return UAVAGENT_OK;
}
extern "C" int uavagent_device_error_clear(void) {
    if (g_gate_err_host) *(volatile uint32_t *)g_gate_err_host = 0u;
    return UAVAGENT_OK;
}

extern "C" int uavagent_actor_head_gated_f32(const float *h1, const float *w2t, const float *b2, const float *w3t_padded, const float *b3_padded,
                                             const float *uniforms, int64_t n_rows, int32_t n_steps, int32_t n_hidden, int32_t n_actions,
                                             float *h2_out, float *logits_out, int64_t ld_logits, int64_t *actions_out, uint32_t *gate_obs,
                                             uint32_t *gate_actions, uint32_t *claim, uint32_t spin_us, void *stream) {
    if (!h1 || !w2t || !b2 || !w3t_padded || !b3_padded || !uniforms || !h2_out || !logits_out || !actions_out || !gate_obs || !gate_actions || !claim)
        return fail3(UAVAGENT_E_INVALID, "actor_head_gated: null pointer");
    if (n_hidden != kHdH || n_actions <= 576 || n_actions > kHdNP || ld_logits < kHdNP || (ld_logits & 3) || n_rows < 1 || n_steps < 1)
        return fail3(UAVAGENT_E_INVALID, "actor_head_gated: built for 200 hidden units and 577..640 actions, ld_logits >= 640 and a multiple of 4, "
                                         "n_rows >= 1, n_steps >= 1");
    if (!aligned16(h1) || !aligned16(w2t) || !aligned16(w3t_padded) || !aligned16(h2_out) || !aligned16(logits_out) || (n_rows & 3))
        return fail3(UAVAGENT_E_INVALID, "actor_head_gated: matrices must be 16-byte aligned and n_rows a multiple of 4 (every step's block of rows "
                                         "starts 16-byte aligned)");
    if (g_gate_err_dev == nullptr) return fail3(UAVAGENT_E_INVALID, "actor_head_gated: call uavagent_gate_prepare() first (it allocates the error word)");
    if (*(volatile uint32_t *)g_gate_err_host != 0u)
        return fail3(UAVAGENT_E_DEVICE, "actor_head_gated: an earlier gated launch timed out on the device (uavagent_device_error); clear it first");
    const long long pairs = ((n_rows + 15) / 16 + 1) / 2;
    const unsigned grid = (unsigned)std::min<long long>(pairs, cu_count_of_current_device());
    constexpr size_t lds_bytes = (size_t)HdPlan<1>::LdsF * sizeof(float);
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&actor_head_gated_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (attr_rc != hipSuccess) return fail3(UAVAGENT_E_HIP, "actor_head_gated: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    hipLaunchKernelGGL(actor_head_gated_kernel, dim3(grid), dim3(kHdThr), lds_bytes, (hipStream_t)stream, h1, w2t, b2, w3t_padded, b3_padded, uniforms,
                       (long long)n_rows, (int)n_steps, (int)n_actions, h2_out, logits_out, (long long)ld_logits,
                       reinterpret_cast<long long *>(actions_out), gate_obs, gate_actions, claim, g_gate_err_dev, spin_us ? spin_us : 2000000u);
    if (hipGetLastError() != hipSuccess) return fail3(UAVAGENT_E_HIP, "actor_head_gated: launch failed");
    return UAVAGENT_OK;
}

extern "C" size_t uavagent_gemm_rows_workspace_bytes(int64_t m_rows) {
    if (m_rows < 1) return 0;
    size_t b = (size_t)((m_rows + 63) / 64) * kNP * sizeof(float);          // one partial row per workgroup of the 64-row kernels (the most)
#ifdef UAVGEMM_STAMPS
    b += (size_t)((m_rows + 63) / 64) * 4 * 6 * sizeof(unsigned long long);
#endif
    return b;
}

// UAVAGENT_ROWS_RESIDENT=0 (read once): the update's K = 200 GEMMs on version 4's 128-row workgroups instead of the resident-W kernel (A/B runs)
static const bool g_rows_resident = [] { const char *e = getenv("UAVAGENT_ROWS_RESIDENT"); return e ? e[0] != '0' : true; }();

extern "C" int uavagent_gemm_rows_f32(const float *a, int64_t lda, const float *w, int64_t ldw, int32_t w_transposed, int64_t m_rows, int32_t k,
                                      int32_t n, const float *bias, int32_t relu6, const float *relu6_mask_h, int64_t ldh, float *c, int64_t ldc,
                                      float *colsum_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!a || !w || !c) return fail3(UAVAGENT_E_INVALID, "gemm_rows: null pointer");
    if (m_rows < 1 || k < 1 || n < 1 || n > 1024 || lda < k || ldc < n || ldw < (w_transposed ? k : n) || lda > 65536 || ldw > 65536)
        return fail3(UAVAGENT_E_INVALID, "gemm_rows: need m_rows, k >= 1, 1 <= n <= 1024, k <= lda <= 65536, ldc >= n, row length of w <= ldw <= 65536");
    if (relu6_mask_h && (bias || relu6)) return fail3(UAVAGENT_E_INVALID, "gemm_rows: the relu6-mask epilogue excludes bias / relu6");
    if (relu6_mask_h && ldh < n) return fail3(UAVAGENT_E_INVALID, "gemm_rows: ldh < n");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = aligned16(a) && aligned16(w) && aligned16(c) && aligned16(bias) && aligned16(relu6_mask_h) && (lda % 4 == 0) && (ldw % 4 == 0) &&
                     (ldc % 4 == 0) && (ldh % 4 == 0) && (k % 4 == 0) && (n % 4 == 0);
    if (n > 208 && !(vec && w_transposed))
        return fail3(UAVAGENT_E_INVALID, "gemm_rows: n > 208 needs the aligned w_transposed form (N is then cut into slices)");
    const dim3 grid((unsigned)((m_rows + kRowsBM - 1) / kRowsBM)), blk(256);
    const int epi = relu6_mask_h ? 2 : ((bias || relu6) ? 1 : 0);
    long long n_part_resident = 0;
    float *colp = nullptr;
    if (colsum_out) {
        if (n > 208) return fail3(UAVAGENT_E_INVALID, "gemm_rows: column sums exist for n <= 208");
        if (!vec) return fail3(UAVAGENT_E_INVALID, "gemm_rows: column sums need the aligned path (16-byte aligned operands, strides / k / n multiples of 4)");
        if (!workspace || !aligned16(workspace) || workspace_bytes < uavagent_gemm_rows_workspace_bytes(m_rows))
            return fail3(UAVAGENT_E_INVALID, "gemm_rows: column sums need a 16-byte aligned workspace of uavagent_gemm_rows_workspace_bytes()");
        colp = reinterpret_cast<float *>(workspace);
    }
#define UAV_ARGS grid, blk, 0, st, a, (long long)lda, w, (long long)ldw, (int)k, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc
#define UAV_NT(BM_, NB_)                                                                                             \
    do {                                                                                                             \
        const dim3 g2((unsigned)((m_rows + BM_ - 1) / BM_), (unsigned)((n + NB_ * 16 - 1) / (NB_ * 16)));               \
        if (epi == 0) hipLaunchKernelGGL((gemm_rows_nt_kernel<BM_, NB_, 0>), g2, blk, 0, st, a, (long long)lda, w, (long long)ldw, (int)k, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp); \
        else if (epi == 1) hipLaunchKernelGGL((gemm_rows_nt_kernel<BM_, NB_, 1>), g2, blk, 0, st, a, (long long)lda, w, (long long)ldw, (int)k, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp); \
        else hipLaunchKernelGGL((gemm_rows_nt_kernel<BM_, NB_, 2>), g2, blk, 0, st, a, (long long)lda, w, (long long)ldw, (int)k, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp); \
    } while (0)
#define UAV_ROWS_E(NT_)        /* the general form: dword loads */                                                    \
    do {                                                                                                             \
        if (epi == 0) hipLaunchKernelGGL((gemm_rows_kernel<NT_, false, 0>), UAV_ARGS);                                \
        else if (epi == 1) hipLaunchKernelGGL((gemm_rows_kernel<NT_, false, 1>), UAV_ARGS);                           \
        else hipLaunchKernelGGL((gemm_rows_kernel<NT_, false, 2>), UAV_ARGS);                                         \
    } while (0)
#define UAV_GL(BM_, NB_, NS_)                                                                                         \
    do {                                                                                                             \
        const dim3 g2((unsigned)((m_rows + BM_ - 1) / BM_), (unsigned)((n + NB_ * 16 - 1) / (NB_ * 16))), b2(BM_ * 4);  \
        if (epi == 0) hipLaunchKernelGGL((gemm_rows_glds_kernel<BM_, NB_, NS_, 0>), g2, b2, 0, st, a, (long long)lda, w, (long long)ldw, (int)k, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp); \
        else if (epi == 1) hipLaunchKernelGGL((gemm_rows_glds_kernel<BM_, NB_, NS_, 1>), g2, b2, 0, st, a, (long long)lda, w, (long long)ldw, (int)k, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp); \
        else hipLaunchKernelGGL((gemm_rows_glds_kernel<BM_, NB_, NS_, 2>), g2, b2, 0, st, a, (long long)lda, w, (long long)ldw, (int)k, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp); \
    } while (0)
    if (w_transposed && vec && k == kRsK && n > kRsNBA * 16 && n <= 208 && m_rows > 32768 && g_rows_resident) {
        // W^T resident in LDS, persistent workgroups (gemm_rows_resident_kernel): the update's 200-wide layers
        const int n_cu = cu_count_of_current_device();          // per device: a process may drive several
        const int nA = (n_cu * kRsNBA + (kRsNBA + kRsNBB) / 2) / (kRsNBA + kRsNBB) < n_cu ? (n_cu * kRsNBA + (kRsNBA + kRsNBB) / 2) / (kRsNBA + kRsNBB) : n_cu - 1;
        if (colp && hipMemsetAsync(colp, 0, (size_t)nA * kNP * sizeof(float), st) != hipSuccess) return fail3(UAVAGENT_E_HIP, "gemm_rows: memset failed");
        const dim3 g2((unsigned)n_cu), b2(kRsThr);
        if (epi == 0) hipLaunchKernelGGL((gemm_rows_resident_kernel<0>), g2, b2, 0, st, a, (long long)lda, w, (long long)ldw, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp, nA);
        else if (epi == 1) hipLaunchKernelGGL((gemm_rows_resident_kernel<1>), g2, b2, 0, st, a, (long long)lda, w, (long long)ldw, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp, nA);
        else hipLaunchKernelGGL((gemm_rows_resident_kernel<2>), g2, b2, 0, st, a, (long long)lda, w, (long long)ldw, (int)n, (long long)m_rows, bias, (int)relu6, relu6_mask_h, (long long)ldh, c, (long long)ldc, colp, nA);
        n_part_resident = nA;
    } else if (w_transposed && vec && (k % kGlBK == 0)) {
        // LDS-DMA ring (gemm_rows_glds_kernel).  Few rows (a rollout step): 64-row workgroups, N in slices; many rows: 128-row workgroups.
        // (64-row workgroups with a 2-stage ring: two of them fit a CU's LDS and cover each other's prologue / epilogue; measured at 8192
        //  rows against a 3-stage ring with one workgroup per CU: 12.3 vs 13.3 us at N = 200, 26.2 vs 31.2 us at N = 640)
        if (m_rows <= 32768) { if (n <= 224) UAV_GL(64, 7, 2); else UAV_GL(64, 10, 2); }
        else if (n <= 208) UAV_GL(128, 13, 2);
        else UAV_GL(128, 10, 2);
    } else if (w_transposed && vec) {
        // k-contiguous tiles, wide fragment reads (gemm_rows_nt_kernel).  Few rows (a rollout step): 64-row workgroups and N in slices,
        // so that 8192 rows still give >= 256 workgroups; many rows (the update): 128-row workgroups over all <= 208 columns.
        if (m_rows <= 32768) { if (n <= 224) UAV_NT(64, 7); else UAV_NT(64, 10); }
        else if (n <= 208) UAV_NT(128, 13);
        else UAV_NT(128, 10);
    } else if (w_transposed) UAV_ROWS_E(true);
    else if (vec) {                                     // x @ W with W [K, N] as stored: n-contiguous W tile, register-staged
        if (epi == 0) hipLaunchKernelGGL((gemm_rows_vec_kernel<0>), UAV_ARGS, colp);
        else if (epi == 1) hipLaunchKernelGGL((gemm_rows_vec_kernel<1>), UAV_ARGS, colp);
        else hipLaunchKernelGGL((gemm_rows_vec_kernel<2>), UAV_ARGS, colp);
    } else UAV_ROWS_E(false);
#undef UAV_NT
#undef UAV_GL
#undef UAV_ROWS_E
#undef UAV_ARGS
    if (colp) {
        const long long n_part = n_part_resident ? n_part_resident : (w_transposed && m_rows <= 32768) ? (m_rows + 63) / 64 : (long long)grid.x;     // partial rows the kernel that ran wrote
        hipLaunchKernelGGL(rows_colsum_reduce, dim3((n + 63) / 64), dim3(1024), 0, st, colp, n_part, (int)n, colsum_out);
    }
    if (hipGetLastError() != hipSuccess) return fail3(UAVAGENT_E_HIP, "gemm_rows: launch failed");
    return UAVAGENT_OK;
}
