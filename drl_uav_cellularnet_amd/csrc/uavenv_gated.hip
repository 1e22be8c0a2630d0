// libuavenv: uavenv_rollout_gated (include/uavenv.h) -- the persistent gated rollout kernel of uavenv_kernels.h (env_kernel_gated) and its
// launch.  A translation unit of its own: the kernel instantiations here and the ~190 of uavenv_capi.hip build independently.
#include <algorithm>

#include "uavenv_handle.h"
#include "uavenv_gated_kernel.h"

using namespace uavk;
using uavenv_internal::fail;
using uavenv_internal::poisoned;
using uavenv_internal::fill_call;
using uavenv_internal::call_is_fast;

extern "C" int uavenv_rollout_gated(uavenv_t *h, const UavEnvGatedRollout *r, const UavEnvOut *out, void *stream) {
    if (!h || !r || !out) return fail(UAVENV_E_INVALID, "rollout_gated: null handle, description or outputs");
    if (r->n_steps < 1 || !r->actions_dev || !r->gate_actions_dev || !r->gate_obs_dev || !r->claim_dev || !r->enc_table_a_dev || !r->enc_out_a_dev)
        return fail(UAVENV_E_INVALID, "rollout_gated: n_steps >= 1, the action tape, both gate arrays, the claim word, table a and its output are required");
    if ((r->enc_table_c_dev != nullptr) != (r->enc_out_c_dev != nullptr)) return fail(UAVENV_E_INVALID, "rollout_gated: table c and its output come together");
    if (!h->packed || h->bt != 4 || h->kp.B != 4 || h->N > 0x7FFFFFFFll)
        return fail(UAVENV_E_INVALID, "rollout_gated: built for n_ue <= 64 and n_bs == 4");
    if (r->enc_hidden < 4 || r->enc_hidden % 4 != 0 || r->enc_hidden > 256 || r->enc_rows < 1 ||
        (unsigned long long)r->enc_rows * (unsigned long long)r->enc_hidden * 4ull >= (1ull << 32))
        return fail(UAVENV_E_INVALID, "rollout_gated: hidden must be a multiple of 4 up to 256 and a table smaller than 4 GiB");
    auto mis16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) != 0; };
    if (mis16(r->enc_table_a_dev) || mis16(r->enc_out_a_dev) || mis16(r->enc_bias_a_dev) || mis16(r->enc_table_c_dev) || mis16(r->enc_out_c_dev) ||
        mis16(r->enc_bias_c_dev))
        return fail(UAVENV_E_INVALID, "rollout_gated: tables, biases and encoder outputs must be 16-byte aligned");
    DeviceGuard guard(h->device);
    if (int rc_dev = poisoned(h, "rollout_gated")) return rc_dev;
    KParams p = h->kp;
    fill_call(p, nullptr, out);
    if (!call_is_fast(p)) return fail(UAVENV_E_INVALID, "rollout_gated: all nine standard outputs, no float64 copies");
    p.n_ticks = 1;
    GatedParams g;
    g.T = r->n_steps; g.n_blocks = (int)((h->N + kGateRows - 1) / kGateRows);
    g.actions = (const long long *)r->actions_dev; g.gate_act = r->gate_actions_dev; g.gate_obs = r->gate_obs_dev; g.claim = r->claim_dev; g.reward = r->reward_dev;
    g.wa = r->enc_table_a_dev; g.ba = r->enc_bias_a_dev; g.wc = r->enc_table_c_dev; g.bc = r->enc_bias_c_dev;
    g.oa = r->enc_out_a_dev; g.oc = r->enc_out_c_dev; g.idx_out = (long long *)r->idx_out_dev;
    g.n_rows = r->enc_rows; g.H4 = r->enc_hidden / 4; g.relu6 = r->enc_relu6; g.G = p.G;
    const int pairs = (g.n_blocks + 1) / 2;
    const unsigned grid = (unsigned)std::min<long long>(pairs, h->n_simd / 4);
    const dim3 blk(64 * kGateWaves);
    hipStream_t s = (hipStream_t)stream;
    const int K = p.U + p.B;
    const bool two = g.wc != nullptr;
#define GATED_ARGS h->blob, p.gid_of_u, p.N, p.U, p.epw, p.Gr, p.B, (int)uavk::lane_div_magic((uint32_t)p.U), g, p
#define GATED_LAUNCH(PLC_, KT_)                                                                                          \
    do {                                                                                                                 \
        if (two) hipLaunchKernelGGL((env_kernel_gated<4, PLC_, KT_, true>), dim3(grid), blk, 0, s, GATED_ARGS);          \
        else hipLaunchKernelGGL((env_kernel_gated<4, PLC_, KT_, false>), dim3(grid), blk, 0, s, GATED_ARGS);             \
    } while (0)
    // (the node count as a template parameter for the reference's two shapes, 4 UAVs + 20 / 40 UEs: the row loop unrolls around v_readlane)
    if (h->plc) { if (K == 24) GATED_LAUNCH(true, 24); else if (K == 44) GATED_LAUNCH(true, 44); else GATED_LAUNCH(true, 0); }
    else { if (K == 24) GATED_LAUNCH(false, 24); else if (K == 44) GATED_LAUNCH(false, 44); else GATED_LAUNCH(false, 0); }
#undef GATED_LAUNCH
#undef GATED_ARGS
    HIP_TRY(hipGetLastError());
    return UAVENV_OK;
}


#ifdef UAVENV_GATE_STAMPS
extern "C" int uavenv_debug_set_gate_stamps(void *dev_ptr) {
    return hipMemcpyToSymbol(HIP_SYMBOL(uavk::g_gate_dbg), &dev_ptr, sizeof(void *)) == hipSuccess ? 0 : -1;
}
#endif
