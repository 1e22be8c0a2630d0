/* A plain C client of include/uavenv.h (TEST INFRASTRUCTURE): no Python, no torch -- hipMalloc'd buffers, the null stream.
 * Build (tests/test_capi_c_client_gpu.py does it):  gcc -D__HIP_PLATFORM_AMD__ capi_client.c -I$ROCM/include -Iinclude -L<lib dir> -luavenv -L$ROCM/lib -lamdhip64
 * Prints one line per step: step_n of env 0, sum of rewards, sum of serving indices, first UE cell -- the Python test compares
 * them with BatchedMobiEnv on the same seed.  Mirrors what a maintainer's binding does: default_config -> create -> init ->
 * warmup(200) -> reset -> step* -> step_many -> destroy. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uavenv.h"

#define CHECK(x) do { int _rc = (x); if (_rc != 0) { fprintf(stderr, "%s failed: %d %s\n", #x, _rc, uavenv_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)

int main(int argc, char **argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 300;
    const int B = 4, U = 20, G = 100, T = 5;
    UavEnvConfig cfg;
    CHECK(uavenv_default_config(&cfg, B, U, G));
    uavenv_t *h = NULL;
    CHECK(uavenv_create(&cfg, N, 0, 0x5EEDull, 0u, &h));
    UavEnvOut out;
    memset(&out, 0, sizeof(out));
    /* T blocks of every output: block 0 doubles as the single-step output set */
    HIP(hipMalloc((void **)&out.reward_dev, sizeof(float) * N * T));
    HIP(hipMalloc((void **)&out.done_dev, N * T));
    HIP(hipMalloc((void **)&out.mean_sinr_dev, sizeof(float) * N * T));
    HIP(hipMalloc((void **)&out.n_out_dev, sizeof(int32_t) * N * T));
    HIP(hipMalloc((void **)&out.ue_xy_dev, sizeof(int16_t) * N * U * 2 * T));
    HIP(hipMalloc((void **)&out.bs_xy_dev, sizeof(int32_t) * N * B * 2 * T));
    HIP(hipMalloc((void **)&out.serving_dev, N * U * T));
    HIP(hipMalloc((void **)&out.cur_sinr_dev, sizeof(float) * N * U * T));
    HIP(hipMalloc((void **)&out.step_n_dev, sizeof(int32_t) * N * T));
    int64_t *act_host = (int64_t *)malloc(sizeof(int64_t) * N * T), *act_dev = NULL;
    for (long long i = 0; i < N * T; ++i) act_host[i] = (i * 7919 + 13) % 625;     /* the Python test builds the same table */
    HIP(hipMalloc((void **)&act_dev, sizeof(int64_t) * N * T));
    HIP(hipMemcpy(act_dev, act_host, sizeof(int64_t) * N * T, hipMemcpyHostToDevice));

    CHECK(uavenv_init(h, NULL, NULL));
    CHECK(uavenv_warmup(h, 200, NULL, NULL));
    CHECK(uavenv_reset(h, NULL, NULL, &out, NULL));
    float *rew = (float *)malloc(sizeof(float) * N * T);
    int8_t *srv = (int8_t *)malloc(N * U * T);
    int16_t *ue = (int16_t *)malloc(sizeof(int16_t) * N * U * 2 * T);
    int32_t *sn = (int32_t *)malloc(sizeof(int32_t) * N * T);
    for (int t = 0; t < 3; ++t) {                                            /* three single steps */
        CHECK(uavenv_step(h, act_dev + (long long)t * N, NULL, &out, NULL));
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(rew, out.reward_dev, sizeof(float) * N, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(srv, out.serving_dev, N * U, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(ue, out.ue_xy_dev, sizeof(int16_t) * N * U * 2, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(sn, out.step_n_dev, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
        double rs = 0; long long ss = 0;
        for (long long i = 0; i < N; ++i) rs += rew[i];
        for (long long i = 0; i < N * U; ++i) ss += srv[i];
        printf("step %d %.9g %lld %d %d\n", sn[0], rs, ss, ue[0], ue[1]);
    }
    CHECK(uavenv_step_many(h, act_dev, T, &out, NULL));                     /* then T steps in one launch, blocks [T, ...] */
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(rew, out.reward_dev, sizeof(float) * N * T, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(srv, out.serving_dev, N * U * T, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ue, out.ue_xy_dev, sizeof(int16_t) * N * U * 2 * T, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(sn, out.step_n_dev, sizeof(int32_t) * N * T, hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t) {
        double rs = 0; long long ss = 0;
        for (long long i = 0; i < N; ++i) rs += rew[(long long)t * N + i];
        for (long long i = 0; i < N * U; ++i) ss += srv[(long long)t * N * U + i];
        printf("step %d %.9g %lld %d %d\n", sn[(long long)t * N], rs, ss, ue[(long long)t * N * U * 2], ue[(long long)t * N * U * 2 + 1]);
    }
    if (uavenv_step(h, NULL, NULL, &out, NULL) == 0) { fprintf(stderr, "null actions accepted\n"); return 1; }   /* error path */
    printf("error %s\n", uavenv_last_error());
    uavenv_destroy(h);
    return 0;
}
