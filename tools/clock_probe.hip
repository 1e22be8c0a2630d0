// Shader clock under load: s_memtime (shader-clock cycles) against s_memrealtime (constant 100 MHz) around a loop of float32 MFMAs
// (v_mfma_f32_16x16x4_f32, the instruction of csrc/agent_gemm.hip), on every SIMD of the chip or on one wavefront.
//   hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int n_acc>
__global__ __launch_bounds__(256) void probe(uint64_t *out, float *sink, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = (float)threadIdx.x * 1e-6f, b = (float)blockIdx.x * 1e-6f;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < n_acc; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = c1 - c0; out[2 * w + 1] = r1 - r0;
    }
}
template <int n_acc>
static void run(const char *what, int blocks, int threads, int iters) {
    const int waves = blocks * threads / 64;
    uint64_t *d; float *sink;
    hipMalloc(&d, waves * 16); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<n_acc>, dim3(blocks), dim3(threads), 0, 0, d, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(2 * waves);
        hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
        std::vector<double> mhz(waves);
        double cyc = 0;
        for (int w = 0; w < waves; ++w) { mhz[w] = 100.0 * (double)h[2 * w] / (double)h[2 * w + 1]; cyc += (double)h[2 * w]; }
        std::sort(mhz.begin(), mhz.end());
        const double mfma_per_wave = (double)iters * n_acc;
        printf("{\"case\": \"%s\", \"rep\": %d, \"waves\": %d, \"kernel_ms\": %.3f, \"shader_mhz_min_med_max\": [%.0f, %.0f, %.0f], \"cycles_per_mfma_per_wave\": %.2f, \"tflops\": %.1f}\n",
               what, rep, waves, ms, mhz[0], mhz[waves / 2], mhz[waves - 1], cyc / waves / mfma_per_wave,
               2.0 * 16 * 16 * 4 * mfma_per_wave * waves / (ms * 1e-3) / 1e12);
    }
    hipFree(d); hipFree(sink);
}
int main() {
    run<1>("one wave, ONE accumulator (dependent chain)", 1, 64, 400000);
    run<4>("one wave, 4 independent accumulators", 1, 64, 400000);
    run<8>("one wave, 8 independent accumulators", 1, 64, 200000);
    run<4>("1 wave per SIMD (256 x 256 threads), 4 accumulators", 256, 256, 400000);
    run<8>("1 wave per SIMD, 8 accumulators", 256, 256, 200000);
    run<4>("2 waves per SIMD (512 x 256 threads), 4 accumulators", 512, 256, 200000);
    run<8>("2 waves per SIMD, 8 accumulators", 512, 256, 100000);
    return 0;
}
