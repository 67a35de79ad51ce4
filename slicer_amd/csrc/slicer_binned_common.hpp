// slicer_binned_common.hpp -- helpers shared by the kernels of SLICER_ALGO_BINNED (slicer_project_bin.hip: K1;
// slicer_binned.hip: K2-K4).  gfx950 only.
#pragma once
#include "slicer_kernels.hpp"

namespace slicer {

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

template <typename T>
__device__ __forceinline__ T dmin(T a, T b) { return a < b ? a : b; }

#ifndef SLICER_LDS_BARRIER
#define SLICER_LDS_BARRIER 1
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the global-memory queue
// (s_waitcnt vmcnt(0)), which stalls every wave on loads and stores that nothing behind the barrier depends on.
// Use where the barrier protects LDS contents; register dependences on loaded values are tracked by the compiler.
__device__ __forceinline__ void lds_barrier()
{
#if SLICER_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// LDS traffic of this wave is complete and the compiler may not move memory operations across
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Sorted record of the per-particle-mass (hydro) path: (xs, ys, m) in one 12-byte store / load.  The constant-mass path
// keeps 8-byte float2 records; PendingList::sxy points at one or the other (sm != nullptr tells which).
struct __attribute__((packed, aligned(4))) Rec3 {
    float x, y, m;
};

// (unit, tile-in-unit) of a map cell under the tile geometry G.  A unit is a plane, or a band of rows_per_unit tile
// rows of a plane on large maps (BinGeom).
__device__ __forceinline__ void cell_to_tile(int gx, int gy, int plane, const BinGeom &G, unsigned &unit,
                                             unsigned &tile_in_unit)
{
    const unsigned ty = (unsigned)(gy >> G.th_log2), tx = (unsigned)(gx >> G.tw_log2);
    unsigned band = 0, trow = ty;
    if (G.units_per_plane > 1) {
        band = ty / (unsigned)G.rows_per_unit;
        trow = ty - band * (unsigned)G.rows_per_unit;
    }
    unit = (unsigned)plane * (unsigned)G.units_per_plane + band;
    tile_in_unit = trow * (unsigned)G.ntx + tx;
}

}  // namespace slicer
