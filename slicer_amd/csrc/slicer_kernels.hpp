// slicer_kernels.hpp -- host-callable launchers of the HIP kernels (internal; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slicer_device.hpp"

namespace slicer {

enum Mas { kTSC = 0, kNGP = 1 };
enum Acc { kF32 = 0, kF64 = 1, kFixed64 = 2, kCountU32 = 3 };  // kCountU32: NGP constant-mass counts

// Where the deposits of the current (file, type) go.
struct Targets {
    void *acc[kMaxPlanes];                 // accumulator map of each plane (type-specific or shared)
    unsigned long long *nsel[kMaxPlanes];  // selected-entry counter of each plane for this type
    int *neg_flag;                         // set when any transformed coordinate is < 0
};

struct LaunchCfg {
    int mas;
    int acc;
    bool has_mass;
};

// fused transform + select + project + global-atomic deposit (SLICER_ALGO_DIRECT)
hipError_t launch_direct(const LaunchCfg &cfg, const float *d_pos, const float *d_mass, uint64_t n,
                         const PassParams &P, const Targets &T, hipStream_t s);

// tot = sum over types / accumulator -> f32 conversion, one plane
struct FinalizeArgs {
    const void *acc[6];  // per-type accumulators (nullptr = type never appeared)
    float *toti[6];      // per-type f32 outputs (may alias acc for kF32; nullptr = not wanted)
    const void *acc_shared;  // want_type_maps == 0: single accumulator of all types (else nullptr)
    float *tot;
    double inv_scale[6];  // kFixed64: 2^-k per type
    double inv_scale_shared;
    uint64_t npix2;
};
hipError_t launch_finalize_tsc(int acc, const FinalizeArgs &A, hipStream_t s);

// NGP per-file fold (densitymaps.cpp:511-513 with exact sequential f32 sums rebuilt from counts)
struct FoldArgs {
    void *scratch[6];  // per-type per-file scratch: u32 counts (mode 1) or f32 sums (mode 2); zeroed after
    int mode[6];       // 0 inactive in this file, 1 counts * mconst, 2 f32 scratch
    float mconst[6];
    float *toti[6];
    float *tot;
    uint64_t npix2;
};
hipError_t launch_fold_ngp(const FoldArgs &A, hipStream_t s);

hipError_t launch_synth(float *d_pos, uint64_t first, uint64_t count, double box, uint64_t seed, int clustered,
                        hipStream_t s);

hipError_t launch_debug_project(const float *d_pos, uint64_t n, const PassParams &P, float *d_xs, float *d_ys,
                                int32_t *d_plane, uint64_t *d_src, uint64_t capacity, unsigned long long *d_count,
                                int *neg_flag, hipStream_t s);

}  // namespace slicer
