/* C ABI of the one hand-written kernel on the learner side (libuavagent.so, gfx950).
 *
 * NOT part of the env drop-in boundary (that is uavenv.h).  The reference's actor and critic (main.py:143-156) start with a dense
 * layer on the raveled (nBS+1, G, G) state, 50 000 inputs of which nBS + nUE are non-zero (main.py:190,202).  On the batched path
 * that layer is a sum of nBS + nUE rows of a [N_S, H] table per sample; agent.py keeps the plain PyTorch form
 * (F.embedding_bag(idx, W, mode="sum") + b) as the reference implementation and the CPU path, and uses this kernel on the GPU.
 */
#ifndef UAVAGENT_H
#define UAVAGENT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVAGENT_OK 0
#define UAVAGENT_E_INVALID (-1)
#define UAVAGENT_E_HIP (-3)

int uavagent_abi_version(void);
const char *uavagent_last_error(void);

/* out_a[m, :] = sum_k w_a[idx[m, k], :] + bias_a   (k ascending, fp32; bias added last, like embedding_bag(...) + b)
 * and, when w_c != NULL, the same for (w_c, bias_c, out_c) with the SAME indices: actor and critic read one index list.
 * All pointers are device pointers on the current device; rows are contiguous (row stride = h floats).
 *   idx   int64 [m_rows, k]   each in [0, n_rows), or outside that range (by convention -1) = "no row": the entry adds nothing and
 *                             is never dereferenced (a faulting kernel can reset a shared GPU host).  An all -1 list yields the
 *                             bias: the reference's all-zero first state (a2c_single_thread.py:155)
 *   w_*   f32   [n_rows, h]   h % 4 == 0, 4 <= h <= 256, 16-byte aligned, n_rows * h * 4 < 4 GiB
 *   bias_* f32  [h] or NULL;  out_* f32 [m_rows, h], 16-byte aligned
 *   1 <= k <= 64.  `stream` is a hipStream_t (0 = the null stream).  Asynchronous. */
int uavagent_sparse_rows_sum_f32(const float *w_a, const float *bias_a, float *out_a, const float *w_c, const float *bias_c,
                                 float *out_c, const int64_t *idx, int64_t m_rows, int32_t k, int32_t h, int64_t n_rows,
                                 void *stream);

#ifdef __cplusplus
}
#endif
#endif
