// Issue cost per wave-instruction of the opcodes the env kernels are made of, measured with s_memtime on gfx950.
// MI355X_MICROARCH.md's cycle table covers f32 VALU and MFMA only; this fills in float64, integer multiply and
// cross-lane ops so that an opcode histogram of the step kernel can be weighted by cycles instead of counted.
//
//   build (here):   mkdir -p ab_build && hipcc -O3 --offload-arch=gfx950 -o ab_build/issue_cost tools/issue_cost.hip   (ab_build/ is git-ignored scratch; delete it afterwards)
//   run (GPU box):  ./ab_build/issue_cost > gpurun_out/issue_cost.txt
//
// Each kernel issues ITERS x 32 instances of ONE opcode on 8 independent registers (no dependent chain shorter than
// 8 instructions) between two s_memtime reads.  No global memory traffic except one result store per wave; every loop
// bound is a compile-time constant.  Launched as one block of 64 threads (one wave alone on a SIMD) and of 512 threads
// (8 waves on one CU = two per SIMD): if the per-wave cycles double, the SIMD was already saturated by one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 64
#define OPS_PER_ITER 32

#define X8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define TIMED(STMT)                                                        \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
    _Pragma("unroll 1")                                                  \
    for (int it = 0; it < ITERS; ++it) { X8(STMT) X8(STMT) X8(STMT) X8(STMT) } \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#define FINISH(SINK)                                                                     \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    if (seed == 0x7fffffff) sink[threadIdx.x] = (double)(SINK);   /* never true: keeps the results alive */

#define PROLOGUE                                                                                     \
    double d[8]; unsigned long long q[8]; unsigned u[8]; float f[8]; unsigned long long cy[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)cy; \
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + 1e-9 * (seed + i + threadIdx.x); q[i] = seed * 77ull + i + threadIdx.x; \
                                  u[i] = seed * 13u + i + threadIdx.x; f[i] = 1.0f + 1e-6f * (seed + i); } \
    const double a = 1.0 + 1e-12 * seed, b = 1e-13 * seed; const unsigned x = 0x9E3779B9u + seed, y = 0xD2511F53u; \
    const float fa = 1.0f + 1e-7f * seed, fb = 1e-8f * seed; const int e1 = (seed >> 20);   /* 0 at run time, unknown at compile time */                \
    const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)seed; const unsigned xs = 0x85EBCA6Bu + seed; /* f64 constant built with scalar integer ops only (1.0 + a few ulps): a VALU-produced value is handed to the asm in VGPRs even under an "s" constraint */ \
    const unsigned long long as_bits = 0x3FF0000000000000ull | ((unsigned long long)(unsigned)seed << 4); \
    (void)a; (void)b; (void)x; (void)y; (void)fa; (void)fb; (void)e1; (void)mask; (void)xs; (void)as_bits;

#define KERNEL(NAME, STMT_MACRO, SINKEXPR)                                                          \
    __global__ __launch_bounds__(512) void k_##NAME(unsigned long long *out, double *sink, int seed) { \
        PROLOGUE TIMED(STMT_MACRO) FINISH(SINKEXPR) }

#define DSUM (d[0] + d[1] + d[2] + d[3] + d[4] + d[5] + d[6] + d[7])
#define QSUM (q[0] + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + q[7])
#define USUM (u[0] + u[1] + u[2] + u[3] + u[4] + u[5] + u[6] + u[7])
#define CYSUM (cy[0] + cy[1] + cy[2] + cy[3] + cy[4] + cy[5] + cy[6] + cy[7])
#define FSUM (f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7])

// Every destination is declared read-write ("+") even where the opcode only writes it: with "=" the dead results all land in
// one scratch register and hipcc pads the back-to-back writes with s_nop 0, which would be billed to the opcode.
#define S_FMA64(i)  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b));
#define S_MUL64(i)  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(a));
#define S_ADD64(i)  asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(b));
#define S_RCP64(i)  asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define S_RSQ64(i)  asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
#define S_LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e1));
#define S_FREXPM64(i) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[i]));
#define S_CVT64U(i) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d[i]) : "v"(u[i]));
#define S_MAD64(i)  asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(q[i]), "+s"(cy[i]) : "v"(x), "v"(y));
#define S_MULLO(i)  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(y));
#define S_MULHI(i)  asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(y));
#define S_XOR(i)    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(x));
#define S_ADDU(i)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(x));
#define S_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(x));
#define S_MOV32(i)  asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(x));
#define S_MOV64(i)  asm volatile("v_mov_b64 %0, %1" : "+v"(d[i]) : "v"(a));
#define S_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %1, 3, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
#define S_CMP(i)    asm volatile("v_cmp_eq_u32 %0, %1, %2" : "+s"(cy[i]) : "v"(u[i]), "v"(x));
#define S_FMA32(i)  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(fa), "v"(fb));
#define S_BPERM(i)  asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(u[i]) : "v"(x) : "memory");
#define S_CNDMASK64(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(x), "s"(mask));          /* lane mask in an ordinary SGPR pair */
#define S_XOR_S(i)  asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "s"(xs));                                /* 32-bit SGPR data operand */
#define S_FMA64_S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "s"(as_bits), "v"(b));                 /* 64-bit SGPR data operand, as the unpinned kernel reads its constants */
#define S_SNOP(i)   asm volatile("s_nop 0");
#define S_EMPTY(i)  asm volatile("" ::: "memory");

KERNEL(empty_loop, S_EMPTY, DSUM)
KERNEL(s_nop_0, S_SNOP, DSUM)
KERNEL(v_fma_f32, S_FMA32, FSUM)
KERNEL(v_fma_f64, S_FMA64, DSUM)
KERNEL(v_mul_f64, S_MUL64, DSUM)
KERNEL(v_add_f64, S_ADD64, DSUM)
KERNEL(v_rcp_f64, S_RCP64, DSUM)
KERNEL(v_rsq_f64, S_RSQ64, DSUM)
KERNEL(v_ldexp_f64, S_LDEXP64, DSUM)
KERNEL(v_frexp_mant_f64, S_FREXPM64, DSUM)
KERNEL(v_cvt_f64_u32, S_CVT64U, DSUM + USUM)
KERNEL(v_mad_u64_u32, S_MAD64, QSUM + CYSUM)
KERNEL(v_mul_lo_u32, S_MULLO, USUM)
KERNEL(v_mul_hi_u32, S_MULHI, USUM)
KERNEL(v_xor_b32, S_XOR, USUM)
KERNEL(v_add_u32, S_ADDU, USUM)
KERNEL(v_cndmask_b32, S_CNDMASK, USUM)
KERNEL(v_cndmask_b32_sgprmask, S_CNDMASK64, USUM)
KERNEL(v_xor_b32_sgprsrc, S_XOR_S, USUM)
KERNEL(v_fma_f64_sgprsrc, S_FMA64_S, DSUM)
// same loop as v_cndmask_b32, but vcc is written by a VALU compare in this wave just before the loop
__global__ __launch_bounds__(512) void k_v_cndmask_b32_vccfresh(unsigned long long *out, double *sink, int seed) {
    PROLOGUE
    asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\ts_nop 4" : : "v"(u[0]), "v"(x) : "vcc");
    TIMED(S_CNDMASK) FINISH(USUM)
}
// Mixed streams: does the CU-wide cap on vcc-masked selects bite at realistic densities?  One select per 8 or per 32
// instructions among f64 FMAs and 32-bit xors (the env step kernel: 100 vcc-form selects in 2041 instructions = 1 in 20).
#define S_FILL_0(i) S_FMA64(i)
#define S_FILL_1(i) S_XOR(i)
#define FILL7 S_FILL_0(1) S_FILL_1(2) S_FILL_0(3) S_FILL_1(4) S_FILL_0(5) S_FILL_1(6) S_FILL_0(7)
#define FILL8 S_FILL_1(0) FILL7
#define MIXKERNEL(NAME, SELECT, REST)                                                                    \
    __global__ __launch_bounds__(512) void k_##NAME(unsigned long long *out, double *sink, int seed) {   \
        PROLOGUE                                                                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        _Pragma("unroll 1")                                                                              \
        for (int it = 0; it < ITERS; ++it) { SELECT(0) FILL7 REST REST REST }                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        FINISH(DSUM + USUM) }
#define SEL_VCC_GROUP  S_CNDMASK(0) FILL7
#define SEL_SGPR_GROUP S_CNDMASK64(0) FILL7
MIXKERNEL(mix_1in8_select_vcc, S_CNDMASK, SEL_VCC_GROUP)
MIXKERNEL(mix_1in8_select_sgprmask, S_CNDMASK64, SEL_SGPR_GROUP)
MIXKERNEL(mix_1in32_select_vcc, S_CNDMASK, FILL8)
MIXKERNEL(mix_1in32_select_sgprmask, S_CNDMASK64, FILL8)
MIXKERNEL(mix_no_select, S_FILL_1, FILL8)
KERNEL(v_mov_b32, S_MOV32, USUM)
KERNEL(v_mov_b64, S_MOV64, DSUM)
KERNEL(v_lshl_add_u64, S_LSHLADD64, QSUM)
KERNEL(v_cmp_eq_u32, S_CMP, USUM + CYSUM)
KERNEL(ds_bpermute_b32, S_BPERM, USUM)

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef void (*kern_t)(unsigned long long *, double *, int);
struct Entry { const char *name; kern_t fn; };

static double run(kern_t fn, int threads, unsigned long long *d_out, double *d_sink, double *max_out) {
    const int waves = threads / 64;
    std::vector<unsigned long long> h(waves);
    double best_mean = 1e30, best_max = 1e30;
    for (int rep = 0; rep < 5; ++rep) {                     // first repetition warms the instruction cache
        hipLaunchKernelGGL(fn, dim3(1), dim3(threads), 0, 0, d_out, d_sink, rep + 1);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s = 0, m = 0;
        for (int w = 0; w < waves; ++w) { s += (double)h[w]; if ((double)h[w] > m) m = (double)h[w]; }
        s /= waves;
        if (rep > 0 && s < best_mean) { best_mean = s; best_max = m; }
    }
    *max_out = best_max / (ITERS * OPS_PER_ITER);
    return best_mean / (ITERS * OPS_PER_ITER);
}

int main() {
    unsigned long long *d_out; double *d_sink;
    CK(hipMalloc(&d_out, 64 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_sink, 512 * sizeof(double)));
#define E(N) {#N, k_##N}
    const Entry table[] = {E(empty_loop), E(s_nop_0), E(v_fma_f32), E(v_fma_f64), E(v_mul_f64), E(v_add_f64), E(v_rcp_f64), E(v_rsq_f64),
                           E(v_ldexp_f64), E(v_frexp_mant_f64), E(v_cvt_f64_u32), E(v_mad_u64_u32), E(v_mul_lo_u32), E(v_mul_hi_u32),
                           E(v_xor_b32), E(v_add_u32), E(v_cndmask_b32), E(v_cndmask_b32_vccfresh), E(v_cndmask_b32_sgprmask), E(v_xor_b32_sgprsrc), E(v_fma_f64_sgprsrc), E(mix_no_select), E(mix_1in32_select_vcc), E(mix_1in32_select_sgprmask), E(mix_1in8_select_vcc), E(mix_1in8_select_sgprmask), E(v_mov_b32), E(v_mov_b64), E(v_lshl_add_u64), E(v_cmp_eq_u32),
                           E(ds_bpermute_b32)};
    std::printf("# s_memtime ticks per wave-instruction, %d instances, best of 4 warm launches (mean over waves; max over waves)\n",
                ITERS * OPS_PER_ITER);
    std::printf("%-26s %22s %30s %30s\n", "opcode", "1 wave (64 threads)", "4 waves, 1/SIMD (256 thr)", "8 waves, 2/SIMD (512 thr)");
    for (const Entry &e : table) {
        double m1, m4, m8;
        const double c1 = run(e.fn, 64, d_out, d_sink, &m1);
        const double c4 = run(e.fn, 256, d_out, d_sink, &m4);
        const double c8 = run(e.fn, 512, d_out, d_sink, &m8);
        std::printf("%-26s %10.2f (max %6.2f) %16.2f (max %6.2f) %16.2f (max %6.2f)\n", e.name, c1, m1, c4, m4, c8, m8);
    }
    CK(hipFree(d_out)); CK(hipFree(d_sink));
    return 0;
}
