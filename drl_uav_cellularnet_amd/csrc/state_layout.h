// Byte layout of the per-handle state slab: ONE definition for the host (uavenv_create / UavEnvStateLayout) and the
// device (the packed env kernel derives every field address from the slab base with scalar arithmetic instead of
// fetching ~19 pointers from the kernarg block; see the kernarg notes in uavenv_kernels.h).
// Struct-of-arrays [field][env][...], every field starts on a 256-byte boundary.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "philox.h"  // UAVENV_HD

namespace uavk {

struct StateOffsets {
    size_t total;
    size_t ue_x, ue_y, ue_hu;                  // f64 [N,U]
    size_t g_x, g_y, g_fl, g_v, g_cos, g_sin;  // f64 [N,Gr]
    size_t agg, deagg, tick;                   // i32 / i32 / u32 [N]
    size_t bs_xy;                              // i32 [N,B,2]
    size_t serving;                            // i8  [N,U]
    size_t fifo;                               // i8  [N,3,U]
    size_t fifo_depth;                         // i32 [N]
    size_t out_bits;                           // u64 [N,ceil(U/64)]
    size_t step_n;                             // i32 [N]
    size_t ue_xy;                              // i16 [N,U,2]
};

UAVENV_HD StateOffsets compute_layout(long long n_envs, int n_ue, int n_bs, int n_groups) {
    const size_t N = (size_t)n_envs, U = (size_t)n_ue, B = (size_t)n_bs, Gr = (size_t)n_groups;
    const size_t W64 = (U + 63) / 64;
    StateOffsets L;
    size_t off = 0;
#define UAV_PUT(field, bytes) do { L.field = off; off = (off + (size_t)(bytes) + 255) & ~(size_t)255; } while (0)
    UAV_PUT(ue_x, N * U * 8); UAV_PUT(ue_y, N * U * 8); UAV_PUT(ue_hu, N * U * 8);
    UAV_PUT(g_x, N * Gr * 8); UAV_PUT(g_y, N * Gr * 8); UAV_PUT(g_fl, N * Gr * 8);
    UAV_PUT(g_v, N * Gr * 8); UAV_PUT(g_cos, N * Gr * 8); UAV_PUT(g_sin, N * Gr * 8);
    UAV_PUT(agg, N * 4); UAV_PUT(deagg, N * 4); UAV_PUT(tick, N * 4);
    UAV_PUT(bs_xy, N * B * 2 * 4); UAV_PUT(serving, N * U); UAV_PUT(fifo, N * 3 * U); UAV_PUT(fifo_depth, N * 4);
    UAV_PUT(out_bits, N * W64 * 8); UAV_PUT(step_n, N * 4); UAV_PUT(ue_xy, N * U * 2 * 2);
#undef UAV_PUT
    L.total = off;
    return L;
}

}  // namespace uavk
