/* C ABI of the hand-written kernels on the learner side (libuavagent.so, gfx950).
 *
 * NOT part of the env drop-in boundary (that is uavenv.h).  The reference's actor and critic (main.py:143-156) start with a dense
 * layer on the raveled (nBS+1, G, G) state, 50 000 inputs of which nBS + nUE are non-zero (main.py:190,202).  On the batched path
 * that layer is a sum of nBS + nUE rows of a [N_S, H] table per sample; agent.py keeps the plain PyTorch form
 * (F.embedding_bag(idx, W, mode="sum") + b) as the reference implementation and the CPU path, and uses these kernels on the GPU.
 * ABI 2 added the kernels around the dense layers: index construction and action sampling for the rollout, and for the update the
 * fused loss gradient, relu6 backward with bias gradients, the table gradient (sorted, deterministic) and the TF1 RMSProp step.  ABI 3
 * adds the dense layers themselves for the reference's widths (200 hidden units, 625 actions): float32 MFMA GEMMs with the relu6 / bias
 * / relu6-mask / bias-gradient epilogues fused, and the actor's head of a rollout step as one kernel; agent.py keeps torch.mm as the
 * alternative (hip_gemms=False) and the tests compare the two.  ABI 4 adds uavagent_first_layer_from_obs_f32 (index construction folded
 * into the first layer's gather: one launch less per rollout step).  Every entry point is asynchronous on `stream` (a hipStream_t,
 * 0 = the null stream), allocates nothing (workspaces are caller-owned; *_workspace_bytes give their sizes), returns 0 or a
 * negative UAVAGENT_E_* code and never throws; all pointers are device pointers on the current device.
 */
#ifndef UAVAGENT_H
#define UAVAGENT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVAGENT_OK 0
#define UAVAGENT_E_INVALID (-1)
#define UAVAGENT_E_HIP (-3)
#define UAVAGENT_E_DEVICE (-5)   /* a gated launch gave up on the device (uavagent_device_error) */

int uavagent_abi_version(void);   /* 5 (5: + uavagent_actor_head_gated_f32, uavagent_gate_prepare, uavagent_device_error(_clear)) */
const char *uavagent_last_error(void);

/* out_a[m, :] = sum_k w_a[idx[m, k], :] + bias_a   (k ascending, fp32; bias added last, like embedding_bag(...) + b)
 * and, when w_c != NULL, the same for (w_c, bias_c, out_c) with the SAME indices: actor and critic read one index list.
 * Rows are contiguous (row stride = h floats).
 *   idx   int64 [m_rows, k]   each in [0, n_rows), or outside that range (by convention -1) = "no row": the entry adds nothing and
 *                             is never dereferenced (a faulting kernel can reset a shared GPU host).  An all -1 list yields the
 *                             bias: the reference's all-zero first state (a2c_single_thread.py:155)
 *   w_*   f32   [n_rows, h]   h % 4 == 0, 4 <= h <= 256, 16-byte aligned, n_rows * h * 4 < 4 GiB
 *   bias_* f32  [h] or NULL;  out_* f32 [m_rows, h], 16-byte aligned
 *   1 <= k <= 64. */
int uavagent_sparse_rows_sum_f32(const float *w_a, const float *bias_a, float *out_a, const float *w_c, const float *bias_c,
                                 float *out_c, const int64_t *idx, int64_t m_rows, int32_t k, int32_t h, int64_t n_rows,
                                 void *stream);
/* The same with the layer's activation fused when relu6 != 0: out = relu6(sum + bias)  (tf.nn.relu6, main.py:147-148,153). */
int uavagent_first_layer_f32(const float *w_a, const float *bias_a, float *out_a, const float *w_c, const float *bias_c,
                             float *out_c, const int64_t *idx, int64_t m_rows, int32_t k, int32_t h, int64_t n_rows,
                             int32_t relu6, void *stream);

/* Compact observation of the env (UavEnvOut: ue_xy i16 [N,U,2], bs_xy i32 [N,B,2], serving i8 [N,U]) -> flat indices of the
 * non-zero cells of the reference's raveled (nBS+1, G, G) state (main.py:190,202): idx_out int64 [N, B+U], UAV cells (plane 0)
 * first, then every UE in the plane of its serving UAV; a cell outside the grid becomes -1.  = agent.obs_to_indices. */
int uavagent_obs_indices(const int16_t *ue_xy, const int32_t *bs_xy, const int8_t *serving, int64_t n_envs, int32_t n_ue,
                         int32_t n_bs, int32_t grid, int64_t *idx_out, void *stream);

/* ABI 4: uavagent_obs_indices + uavagent_first_layer_f32 in ONE launch, for the rollout loop (a2c_single_thread.py:153-158: the state
 * env.step returned is raveled and fed to the networks).  Sample m = env m; its index list is built from the compact observation
 * exactly as uavagent_obs_indices builds it (node k < n_bs: UAV k, plane 0; else UE k - n_bs in plane 1 + serving; -1 off the grid),
 * stored to idx_out [n_envs, n_bs + n_ue] unless idx_out is NULL, and summed as uavagent_first_layer_f32 sums it (same order, same
 * bits).  n_bs + n_ue <= 64; n_rows >= (n_bs + 1) * grid^2; ue_xy 4-byte and bs_xy 8-byte aligned. */
int uavagent_first_layer_from_obs_f32(const float *w_a, const float *bias_a, float *out_a, const float *w_c, const float *bias_c,
                                      float *out_c, const int16_t *ue_xy, const int32_t *bs_xy, const int8_t *serving, int64_t n_envs,
                                      int32_t n_ue, int32_t n_bs, int32_t grid, int32_t h, int64_t n_rows, int32_t relu6,
                                      int64_t *idx_out, void *stream);

/* choose_action (main.py:165-169): p = softmax(logits[n, :]); np.random.choice(n_actions, p=p) with the uniform u[n] supplied by
 * the caller: the first a with cumsum(p)[a] > u[n] * cumsum(p)[-1].  logits f32 [n_rows, n_actions] with row stride ld_logits floats
 * (ABI 3: the learner keeps its 625 logits in rows of 640), uniforms f32 [n_rows] in [0, 1), actions_out int64 [n_rows]; prob_out f32
 * [n_rows, n_actions] contiguous or NULL.  n_actions <= 1024. */
int uavagent_sample_actions(const float *logits, int64_t ld_logits, const float *uniforms, int64_t n_rows, int32_t n_actions,
                            int64_t *actions_out, float *prob_out, void *stream);

/* Loss of main.py:64-74 and its gradient, one pass.  IN: logits [m_rows, n_actions], row stride ld_logits (actor output before the softmax), v and
 * v_target f32 [m_rows], actions int64 [m_rows].  OUT: logits_inout overwritten with d(a_loss)/d(logits); dv_out [m_rows] =
 * d(c_loss)/dv; dbias_out [n_actions] = column sums of the logits gradient; loss_out double[3] = {a_loss, c_loss, sum(dv)}.
 *   td = v_target - v;  c_loss = mean(td^2);  a_loss = mean(-(beta * H + log(p[a] + 1e-5) * td)),  H = -sum p log(p + 1e-5),
 *   td enters a_loss as a constant (tf.stop_gradient, main.py:70).
 * Rows that start 16-byte aligned with ld_logits a multiple of 4 (the learner's: 625 logits in rows of 640) take a float4 kernel that uses
 * the hardware's exp2 / log2 / reciprocal (about 1 ulp each); it also writes zeros to the columns [n_actions, n_actions rounded up to 4) of
 * every row (the learner keeps that tail zero anyway).  Other layouts take the dword kernel with the library's expf / logf. */
size_t uavagent_loss_grad_workspace_bytes(int32_t n_actions);
int uavagent_a2c_loss_grad(float *logits_inout, int64_t ld_logits, const float *v, const float *v_target, const int64_t *actions,
                           int64_t m_rows, int32_t n_actions, float beta, float *dv_out, float *dbias_out, double *loss_out,
                           void *workspace, void *stream);

/* relu6 backwards with the bias gradient: dx[m, c] = dy[m, c] * (0 < y[m, c] < 6), dbias_out[c] = sum_m dx[m, c].
 * y, dy f32 [m_rows, n_cols] contiguous; dx_out has row stride ldx floats (>= n_cols; lets two results share one [M, 2H] buffer).
 * dy == NULL selects the critic's value head (v = h @ w3 + b3, main.py:155): dy[m, c] = dv[m] * w3[c] is formed on the fly and
 * dw3_out[c] = sum_m y[m, c] * dv[m] is produced in the same pass.  n_cols % 4 == 0, <= 256. */
size_t uavagent_relu6_bwd_workspace_bytes(int32_t n_cols);
int uavagent_relu6_bwd(const float *dy, const float *y, const float *dv, const float *w3, int64_t m_rows, int32_t n_cols,
                       float *dx_out, int64_t ldx, float *dbias_out, float *dw3_out, void *workspace, void *stream);

/* out[m] = sum_c y[m, c] * w[c] + bias[0]: the critic's value head forwards (v = h @ w3 + b3, main.py:155) as one streaming pass.
 * y f32 [m_rows, n_cols], w f32 [n_cols], bias f32 [1] or NULL, out f32 [m_rows]; n_cols % 4 == 0, <= 256. */
int uavagent_rowdot_f32(const float *y, const float *w, const float *bias, int64_t m_rows, int32_t n_cols, float *out, void *stream);

/* Gradient of the first-layer tables: dw[r, :] = sum of g[m, :] over all (m, k) with idx[m, k] == r; rows nobody references
 * get 0, entries outside [0, n_rows) are skipped.  g f32 [m_rows, n_tables * h]: columns [0, h) belong to dw0_out, [h, 2h) to
 * dw1_out (n_tables = 2: actor and critic share idx).  Stable sort by row + segmented sums in ascending sample order: the result
 * is bit-reproducible (no float atomics).  Workspace: 256-byte aligned, uavagent_rows_grad_workspace_bytes(...) bytes. */
size_t uavagent_rows_grad_workspace_bytes(int64_t m_rows, int32_t k, int32_t n_cols_total, int64_t n_rows);
int uavagent_rows_grad_f32(const int64_t *idx, const float *g, int64_t m_rows, int32_t k, int32_t h, int32_t n_tables,
                           int64_t n_rows, float *dw0_out, float *dw1_out, void *workspace, size_t workspace_bytes, void *stream);
/* ABI 4: the two halves of uavagent_rows_grad_f32 as calls of their own.  _sort (keys + stable radix sort of the (row, sample) pairs into
 * the workspace) needs only idx, which exists before the backward pass starts: it may run beside it on another stream, and one sort serves
 * any number of _sums calls over the same samples (one per trunk when their gradients are exchanged separately).  _sums must see the
 * workspace a _sort of the same (idx, m_rows, k, n_rows) wrote, after it in stream order (or behind an event); the workspace must hold
 * uavagent_rows_grad_workspace_bytes for the larger of the two n_cols_total (= h * n_tables). */
int uavagent_rows_grad_sort(const int64_t *idx, int64_t m_rows, int32_t k, int32_t n_cols_total, int64_t n_rows, void *workspace,
                            size_t workspace_bytes, void *stream);
int uavagent_rows_grad_sums_f32(const float *g, int64_t m_rows, int32_t k, int32_t h, int32_t n_tables, int64_t n_rows, float *dw0_out,
                                float *dw1_out, void *workspace, size_t workspace_bytes, void *stream);

/* n-step value targets of a rollout (a2c_single_thread.py:176-183): out[t, n] = r[t, n] + gamma * out[t+1, n], out[T, n] := bootstrap[n]
 * (v(s_T), or 0 where the episode ended).  rewards / out f32 [n_steps, n_envs], bootstrap f32 [n_envs]. */
int uavagent_nstep_returns_f32(const float *rewards, const float *bootstrap, int64_t n_envs, int32_t n_steps, float gamma, float *out,
                               void *stream);

/* tf.train.RMSPropOptimizer(lr, decay, momentum = 0, epsilon) as TF1 applies it (main.py:300-301), on flat buffers of n floats:
 *   gs = g * g_scale;  ms <- decay * ms + (1 - decay) * gs^2;  w <- w - lr * gs / sqrt(ms + epsilon). */
int uavagent_rmsprop_tf1(float *w, float *ms, const float *g, int64_t n, float lr, float decay, float eps, float g_scale,
                         void *stream);

/* ---- ABI 3: float32 MFMA GEMMs for the 200-wide layers (csrc/agent_gemm.hip; v_mfma_f32_16x16x4_f32, exact float32 products summed
 * in k order inside a tile).  They replace torch.mm / torch.addmm in the update for the shapes the reference's network has
 * (main.py:143-156: 200 hidden units, 625 actions); anything else stays with the BLAS library. ---- */

/* Rows GEMM:  c[m, j] = epilogue( sum_k a[m, k] * wop[k, j] ),  0 <= j < n <= 208, any k.
 *   w_transposed == 0:  wop = w,    w f32 [k, n] (row stride ldw): a layer forwards, x @ W
 *   w_transposed != 0:  wop = w^T,  w f32 [n, k] (row stride ldw): backwards through a layer, dy @ W^T
 *   epilogue: bias f32 [n] or NULL is added, then relu6 != 0 clamps to [0, 6] (tf.nn.relu6, main.py:147-148,153);  OR
 *             relu6_mask_h f32 [m_rows, n] (row stride ldh) != NULL: c = (0 < h < 6) ? sum : 0 -- relu6 backwards, h the layer's output
 *             (excludes bias / relu6).
 *   a f32 [m_rows, k] (row stride lda), c f32 [m_rows, n] (row stride ldc).  16-byte aligned pointers with lda, ldw, ldc, ldh, k, n all
 *   multiples of 4 take the fast path (float4 staging, coalesced epilogue); anything else is loaded dword by dword.
 *   colsum_out f32 [n] or NULL: column sums of c as stored (the bias gradient of the layer below when c is a masked dx); fast path only,
 *   needs `workspace` (16-byte aligned, uavagent_gemm_rows_workspace_bytes(m_rows) bytes); fixed summation order, bit-reproducible. */
size_t uavagent_gemm_rows_workspace_bytes(int64_t m_rows);
int uavagent_gemm_rows_f32(const float *a, int64_t lda, const float *w, int64_t ldw, int32_t w_transposed, int64_t m_rows, int32_t k,
                           int32_t n, const float *bias, int32_t relu6, const float *relu6_mask_h, int64_t ldh, float *c, int64_t ldc,
                           float *colsum_out, void *workspace, size_t workspace_bytes, void *stream);

/* The actor's head of one rollout step in ONE launch (choose_action, main.py:165-169, on the network of main.py:147-150):
 *   h2 = relu6(h1 @ W2 + b2);  logits = h2 @ W3 + b3;  actions_out[n] = the inverse-CDF draw of uavagent_sample_actions with uniforms[n].
 * Bit-identical to uavagent_gemm_rows_f32 (twice) + uavagent_sample_actions.  Built for the reference's widths: n_hidden = 200 and
 * 577..640 actions (625 = 5^4).  h1 f32 [n_rows, 200] contiguous; w2t f32 [200, 200] = W2 TRANSPOSED; b2 f32 [200]; w3t_padded f32
 * [640, 200] = W3 transposed, rows >= n_actions zero; b3_padded f32 [640], entries >= n_actions zero; h2_out f32 [n_rows, 200];
 * logits_out f32 [n_rows, 640 of ld_logits] (the tail comes out zero); actions_out int64 [n_rows].  16-byte aligned matrices. */
int uavagent_actor_head_f32(const float *h1, const float *w2t, const float *b2, const float *w3t_padded, const float *b3_padded,
                            const float *uniforms, int64_t n_rows, int32_t n_hidden, int32_t n_actions, float *h2_out, float *logits_out,
                            int64_t ld_logits, int64_t *actions_out, void *stream);

/* The actor's head of a WHOLE rollout (n_steps x uavagent_actor_head_f32) as ONE persistent launch that runs beside the env library's
 * persistent rollout kernel (uavenv_rollout_gated, include/uavenv.h -- which states the protocol): per block b of 16 consecutive rows and
 * step t the kernel waits until gate_obs[b] >= t + 1 (h1[t] of the block is in memory), computes the block exactly as
 * uavagent_actor_head_f32 does (bit-identical h2, logits, actions), stores the actions through and sets gate_actions[b] = t + 1.  All
 * per-step arrays are [n_steps][n_rows][...] contiguous: h1, h2_out [T][n_rows][200], uniforms and actions_out [T][n_rows], logits_out
 * [T][n_rows][ld_logits].  n_rows a multiple of 4.  The caller zeroes gate_actions and presets gate_obs (1 where h1[0] is ready) before the
 * launch, zeroes `claim` (one word: pairs of blocks are handed out in arrival order, like the env kernel's) and launches the two kernels on
 * DIFFERENT streams (or parallel graph branches).  min(blocks / 2, CUs) workgroups of 8 wavefronts, <= 112 VGPRs, 135 KB of LDS: one per CU,
 * beside one workgroup of the env kernel.  Every wait is bounded (spin_us of the 100 MHz clock; 0 = 2 s):
 * a partner that never arrives leaves 0x47415445 "GATE" in the library's error word (uavagent_device_error; uavagent_device_error_clear)
 * and the kernel exits.  uavagent_gate_prepare() allocates that host-mapped word once per process -- call it outside any stream capture. */
int uavagent_gate_prepare(void);
int uavagent_device_error(uint32_t *code);
int uavagent_device_error_clear(void);
int uavagent_actor_head_gated_f32(const float *h1, const float *w2t, const float *b2, const float *w3t_padded, const float *b3_padded,
                                  const float *uniforms, int64_t n_rows, int32_t n_steps, int32_t n_hidden, int32_t n_actions, float *h2_out,
                                  float *logits_out, int64_t ld_logits, int64_t *actions_out, uint32_t *gate_obs, uint32_t *gate_actions,
                                  uint32_t *claim, uint32_t spin_us, void *stream);

/* Weight gradient:  c[i, j] = sum_m a[m, i] * b[m, j]  and, when dbias_out != NULL, dbias_out[j] = sum_m b[m, j]  (the bias gradient of
 * the same layer: a column of ones rides along in the kernel).  a f32 [m_rows, n_i] CONTIGUOUS (n_i % 4 == 0, <= 200, 16-byte aligned),
 * b f32 [m_rows, n_j] (row stride ldb, n_j <= 640), c f32 [n_i, n_j] (row stride ldc).  The sum over m is split over the CUs; partial
 * sums go to `workspace` (16-byte aligned, uavagent_gemm_tn_workspace_bytes(m_rows, n_j) bytes) and a second pass adds them in a fixed
 * order: bit-reproducible, no float atomics. */
size_t uavagent_gemm_tn_workspace_bytes(int64_t m_rows, int32_t n_j);
/* Test hook (host only): the dW kernel deals the 13 x 13 (plan 13) or 13 x 20 (plan 20) output blocks of a tile out to 8 wavefronts of equal
 * shape; > 0 = MFMAs issued per k-step when every block has exactly one owner, negative on a hole or an overlap. */
int uavagent_debug_tn_plan_check(int32_t plan);
int uavagent_gemm_tn_f32(const float *a, const float *b, int64_t m_rows, int32_t n_i, int32_t n_j, int64_t ldb, float *c, int64_t ldc,
                         float *dbias_out, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
