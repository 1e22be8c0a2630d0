// Byte layout of the per-handle state slab: ONE definition for the host (uavenv_create / UavEnvStateLayout) and the
// device (the packed env kernel derives every field address from the slab base with scalar arithmetic instead of
// fetching pointers from the kernarg block; see the kernarg notes in uavenv_kernels.h).
//
// Arrays of RECORDS, [field][env][...], every array on a 256-byte boundary.  A record is a multiple of 16 bytes, so a lane
// moves it with whole global_load/store_dwordx4 and a wavefront still touches contiguous memory.  Why records: the step
// kernel was bound by memory-INSTRUCTION issue, not bytes (profiles/r01_v18_phase_stamps.txt: 30 stores at ~53 ticks each =
// 15 % of a wavefront's life, the load phase 22 %); with one array per scalar field the four lane classes (walkers, group
// owners, UAV owners, head lane) each add their own instructions.  Same bytes, 23 -> 12 loads and 30 -> 18 stores per step.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "philox.h"  // UAVENV_HD

namespace uavk {

struct alignas(16) UePos {      // walker position, ue_mobility.py:434-435 (float64 cells)
    double x, y;
};
struct alignas(16) UeAux {      // the rest of a walker's state
    double hu;                  // heading uniform drawn last tick (theta = 2*pi*hu, :508-510)
    int16_t ix, iy;             // (int)x, (int)y as handed to the channel (mobile_env.py:154-155)
    int8_t serving;             // current_BS (channel.py:113-124,162-167)
    int8_t r0, r1, r2;          // bestBS_buf rows, oldest first (channel.py:148-153)
};
struct alignas(16) GrpRec {     // one RPGM group, ue_mobility.py:442-446
    double x, y, fl, v, c, s;   // centre, remaining flight length, speed, cos / sin of the heading
};
struct alignas(16) EnvRec {     // per-env scalars
    uint32_t tick;              // Philox time
    int32_t agg, deagg;         // aggregation phase counters (ue_mobility.py:461-487)
    int32_t fifo_depth;         // rows of bestBS_buf in use, 1..3
    int32_t step_n;             // mobile_env.py:181
    int32_t pad0, pad1, pad2;
};
static_assert(sizeof(UePos) == 16 && sizeof(UeAux) == 16 && sizeof(GrpRec) == 48 && sizeof(EnvRec) == 32, "record sizes are part of the state ABI");

struct StateOffsets {
    size_t total;
    size_t ue_pos;     // UePos  [N,U]
    size_t ue_aux;     // UeAux  [N,U]
    size_t grp;        // GrpRec [N,Gr]
    size_t env;        // EnvRec [N]
    size_t bs_xy;      // i32    [N,B,2]
    size_t out_bits;   // u64    [N,ceil(U/64)]  previous outage set
};

UAVENV_HD StateOffsets compute_layout(long long n_envs, int n_ue, int n_bs, int n_groups) {
    const size_t N = (size_t)n_envs, U = (size_t)n_ue, B = (size_t)n_bs, Gr = (size_t)n_groups;
    const size_t W64 = (U + 63) / 64;
    StateOffsets L;
    size_t off = 0;
#define UAV_PUT(field, bytes) do { L.field = off; off = (off + (size_t)(bytes) + 255) & ~(size_t)255; } while (0)
    UAV_PUT(ue_pos, N * U * sizeof(UePos)); UAV_PUT(ue_aux, N * U * sizeof(UeAux));
    UAV_PUT(grp, N * Gr * sizeof(GrpRec)); UAV_PUT(env, N * sizeof(EnvRec));
    UAV_PUT(bs_xy, N * B * 2 * 4); UAV_PUT(out_bits, N * W64 * 8);
#undef UAV_PUT
    L.total = off;
    return L;
}

}  // namespace uavk
