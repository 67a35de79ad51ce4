// slicer_kernels.hpp -- host-callable launchers of the HIP kernels (internal; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slicer_device.hpp"

namespace slicer {

enum Mas { kTSC = 0, kNGP = 1 };
enum Acc { kF32 = 0, kF64 = 1, kFixed64 = 2, kCountU32 = 3,  // kCountU32: NGP constant-mass counts
           // tile kernel only: f32 / f64 global accumulators behind INTEGER tile cells in LDS (exact tile sums at
           // 2^-49 of the mass scale; ds_add_u64 runs at twice the rate of ds_add_f64)
           kF32I = 4, kF64I = 5 };

// Where the deposits of the current (file, type) go.
struct Targets {
    void *acc[kMaxPlanes];                 // accumulator map of each plane (type-specific or shared)
    unsigned long long *nsel[kMaxPlanes];  // selected-entry counter of each plane for this type
    int *neg_flag;                         // set when any transformed coordinate is < 0
    unsigned *max_mass;                    // bits of the largest selected per-particle mass of this species (sort kernel)
};

struct LaunchCfg {
    int mas;
    int acc;
    bool has_mass;
};

// fused transform + select + project + global-atomic deposit (SLICER_ALGO_DIRECT)
hipError_t launch_direct(const LaunchCfg &cfg, const float *d_pos, const float *d_mass, uint64_t n,
                         const PassParams &P, const Targets &T, hipStream_t s);

// shot-noise thinning path (snopt > 0): ordered selection ranks + libc deviates (slicer_kernels.hip)
hipError_t launch_thin_count(const float *d_pos, uint64_t n, const PassParams &P, unsigned *counts,
                             unsigned long long *base, int *neg_flag, hipStream_t s);
hipError_t launch_thin_deposit(const LaunchCfg &cfg, const float *d_pos, const float *d_mass, uint64_t n,
                               const PassParams &P, const Targets &T, const unsigned long long *base,
                               const float *urand, double thr, double mfac, hipStream_t s);

// glibc's rand() stream continued on the device (slicer_rand.hip): the process-global TYPE_3 state in and out, the
// jump tables, and the generator -- deviates [0, *d_nsel) as rand() / float(RAND_MAX), *d_nsel <= max_draws read on the
// device; d_state (31 words, oldest first) is advanced by *d_nsel draws in place
constexpr int kRandPow2 = 48;
bool libc_rand_grab(uint32_t *v31);
bool libc_rand_put(const uint32_t *v31);
void libc_rand_model_fill(uint32_t *v31, float *out, unsigned long long n);
size_t rand_tables_bytes();
hipError_t rand_tables_upload(void *d_tables, hipStream_t s);
size_t rand_wave_states_bytes(unsigned long long max_draws);
hipError_t launch_rand_deviates(const unsigned long long *d_nsel, uint32_t *d_state, uint32_t *d_wave_states,
                                const void *d_tables, float *d_urand, unsigned long long max_draws, hipStream_t s);

// Zero-fill of several buffers with ONE launch (the accumulator maps of a pass: one dispatch instead of one
// hipMemsetAsync per map).  Sizes in 4-byte words; the buffers come from hipMalloc (16-byte aligned at least).
constexpr int kZeroMax = 16;
struct ZeroList {
    void *p[kZeroMax];
    unsigned long long quad0[kZeroMax + 1];  // exclusive prefix of the buffers' sizes in 16-byte quads (rounded up)
    unsigned long long words[kZeroMax];
    int n;
};
hipError_t launch_zero_many(const ZeroList &Z, hipStream_t s);

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

hipError_t launch_debug_math(int op, const double *d_a, const double *d_b, double *d_out, uint64_t n, hipStream_t s);
hipError_t launch_debug_project(const float *d_pos, uint64_t n, const PassParams &P, float *d_xs, float *d_ys,
                                int32_t *d_plane, uint64_t *d_src, uint64_t capacity, unsigned long long *d_count,
                                int *neg_flag, hipStream_t s);

// ---- SLICER_ALGO_BINNED (slicer_binned.hip) ----
struct BinGeom {
    int tw_log2, th_log2;  // tile width / height in pixels (powers of two)
    int ntx, nty;          // tiles per map row / column
    // A *unit* is what one scatter workgroup group and one compact region serve: a whole plane while a plane has
    // <= 8192 tiles in total with the others, otherwise a band of rows_per_unit tile rows of one plane (large maps).
    int tiles_per_unit;    // rows_per_unit * ntx
    int units_per_plane;   // 1, or ceil(nty / rows_per_unit)
    int rows_per_unit;     // tile rows per unit
    int n_units;           // n_planes * units_per_plane (<= kMaxUnits)
    int nbins;             // n_units * tiles_per_unit; bin = unit * tiles_per_unit + tile_in_unit
    int batch;             // particles per K1 workgroup
    int region;            // records a K1 workgroup may emit per unit = size of its region in the compact buffer:
                           // batch * (2 nrepperp + 1)^2 (lateral replication, densitymaps.cpp:377-399), <= 65535
};
constexpr int kMaxUnits = 32;
constexpr int kMaxBins = 32768;  // all (unit, tile) bins of a pass: K1's packed u16 histogram is <= 64 KiB of LDS

struct BinWorkspace {
    // two-level sort
    float2 *c1;
    unsigned *sb_off, *sb_n;
    unsigned short *sb_start;
    unsigned *ptab;    // [n_units * ngroups][tiles_per_unit + 1] start of every tile's run inside sxy, item by item
    unsigned *item_tot;  // [ngroups][n_units] records per item of the sort kernel (zero before the project+bin launch)
    unsigned *tot;     // [nbins] records per bin, summed over the pending chunks of the group
    float2 *cxy;       // [n_units][nblocks][region] compact (xs, ys) of (unit, K1 workgroup), bcount[unit][b] valid
    unsigned short *cbin;  // same shape: tile-in-unit of each compact record (K3 sorts by it)
    float *cm;         // same shape: per-particle mass (hydro) or nullptr
    float2 *sxy;       // [max_chunk] records grouped by bin
    float *sm;         // non-null with per-particle masses: sxy then holds 12-byte (xs, ys, m) records
    unsigned *hist16;  // [nblocks][ceil(nbins/2)] per-workgroup histogram, two u16 counters per word
    unsigned *hist;    // [nblocks][nbins] exclusive prefix over workgroups (write cursors)
    unsigned *total;   // [nbins] exclusive prefix of the bin totals inside each group of kScanBins bins | [ngroups] group sums
    unsigned *base;    // [nbins + 1] start of every bin's run in sxy (written by the sort kernel for the tile kernel)
    unsigned *bcount;  // [n_units][nblocks]
};

// Kernel-argument block of k_project_bin_fast (slicer_project_bin.hip): only what its hot loop reads, so that the
// uniform operands stay in SGPRs; the full PassParams rides along as a separate kernel argument for the exact paths.
struct K1Args {
    const float *pos;
    const float *mass;         // per-particle masses or nullptr
    uint64_t n;
    float2 *cxy;
    unsigned short *cbin;
    float *cm;
    unsigned *hist16;
    unsigned *bcount;
    int *neg_flag;
    unsigned long long *nsel;  // counter of plane 0 for this type; plane p at nsel + 6 p
    // transform (gadget2io.cpp:204-270), per OUTPUT axis a (source axis perm[a]; the permutation itself is a template
    // argument of the kernel, `face` selects the instantiation)
    float rb, boxf;            // RN32(1 / box), (float)box
    float ws[3], wo[3];        // first wrap as fma(q, ws, wo): (+1, 0) for sgn = +1, (-1, 1) for sgn = -1
    float c0f[3];              // Random.x0 / y0 / z0: f32 values, as the reference draws them
    float rcase;
    int face;                  // Random.face - 1
    // slabs (densitymaps.cpp:346-347,374) of consecutive planes: zlo[p] as in PassParams, +inf for p >= n_planes;
    // zlast = zhi[n_planes - 1]
    float zlo[4], zlast;
    int n_planes, vec, ngp;
    int stack;                 // compact survivors through the wave stack before projecting (few survivors) or not
    // conservative FOV pre-test
    float k_ra, eps_ra, k_dec, eps_dec;
    // projection + grid
    double series_max, lim, inv_fov;
    double nn_d;               // (double)nn
    float nn_f;
    int nn;
    int pow2;                  // nn is a power of two: the cell index is floorf(xs * nn_f); else floor(xs * nn_d) with
                               // exact cell boundaries left to the exact epilogue (grid_tie, slicer_device.hpp)
    // tile geometry (BinGeom)
    int tw_log2, th_log2, ntx, tiles_per_unit, units_per_plane, rows_per_unit, n_units, nbins, batch;
    int ntx_log2, tpu_log2;    // log2 of ntx / tiles_per_unit where both are powers of two (power-of-two maps), else -1
    // two-level sort (sort2 != 0): the kernel sorts its records by unit (= coarse bin: a band of 2^crow_log2 tile rows of
    // one plane) in LDS, sub-batch by sub-batch, and writes each sub-batch contiguously into its region of c1
    int sort2, crow_log2;
    float2 *c1;               // [nblocks][batch] records, sub-batch after sub-batch
    unsigned *sb_off;         // [nblocks * kSubBatches] first record of every sub-batch (index into c1)
    unsigned short *sb_start; // [nblocks * kSubBatches][kSubRow] start of every unit's run inside the sub-batch
    unsigned *sb_n;           // [nblocks] sub-batches the workgroup wrote
    unsigned *item_tot;       // [ngroups][n_units] records of every item of the sort kernel (zero before the launch)
};
// two-level sort: LDS staging of the project+bin kernel (records per workgroup between two flushes) and the most
// sub-batches a workgroup can write: every flush but the last writes more than kStageCap - one round's records
constexpr int kStageCap = 5632;
constexpr int kSubBatches = 16;
constexpr int kSort2Blocks = 16;  // project+bin workgroups whose sub-batches form one group (item) of the sort kernel
constexpr int kMaxCoarse = 256;  // units of a pass of the two-level sort (8-bit ids in the staging area)
constexpr int kSubRow = kMaxCoarse + 2;  // entries per row of the sub-batch table (unit starts + the end; even)
constexpr int kMaxCoarseTiles = 256;  // tiles per unit of the two-level sort (8-bit ids in the sort kernel)

// exhaustive device check of the f32 form of r / box used by k_project_bin_fast (all 2^31 non-negative floats)
// exhaustive device check of quot_dl3 (slicer_device.hpp) for one map size; d_out9 zeroed by the caller
hipError_t launch_check_dl_quotient(double dl, unsigned *d_out9, hipStream_t s);
hipError_t launch_check_box_quotient(double box, unsigned *d_mismatches, hipStream_t s);

size_t project_bin_lds_bytes(const BinGeom &G, bool has_mass);
size_t project_bin_sort2_lds_bytes();
size_t tile_lds_bytes(const BinGeom &G, int acc, bool runs);
size_t scatter_lds_bytes(const BinGeom &G, bool has_mass);
hipError_t launch_project_bin(const LaunchCfg &cfg, bool fast, const float *d_pos, const float *d_mass, uint64_t n,
                              const PassParams &P, const K1Args &A, const BinGeom &G, const BinWorkspace &W,
                              const Targets &T, hipStream_t s);
hipError_t launch_bin_scan(const LaunchCfg &cfg, int nblocks, int n_planes, const BinGeom &G, const BinWorkspace &W,
                           const Targets &T, hipStream_t s);
hipError_t launch_bin_scatter(const LaunchCfg &cfg, int nblocks, int n_planes, int max_workgroups, const BinGeom &G,
                              const BinWorkspace &W, const Targets &T, hipStream_t s);
// two-level sort, second level: every item (unit, group of `slots_per_group` sub-batch slots) gathers the unit's runs
// out of its sub-batches, sorts them by tile in LDS and writes them to a contiguous piece of sxy + its row of ptab
size_t sort2_lds_bytes(int slots_per_group, int tiles_per_unit);
hipError_t launch_sort2(int nblocks, int slots_per_group, int ngroups, int max_workgroups, const PassParams &P,
                        const BinGeom &G, const BinWorkspace &W, hipStream_t s);
// Chunks whose records are binned but not yet deposited: the tile kernel walks all of them, so one
// LDS tile zero + flush is amortised over up to kMaxPending chunks (e.g. the sub-files of a snapshot).
// The host flushes at pending_limit() chunks: 8 where a chunk brings a tile ~2048 records (the headline case: more per
// launch measured no better), up to kMaxPending where a chunk brings few (16384^2 maps: zeroing and flushing a 68 KB
// tile for ~500 records is most of the kernel); the two-level sort's run table holds kMaxPendingRuns chunks.
#ifndef SLICER_MAX_PENDING
#define SLICER_MAX_PENDING 32
#endif
constexpr int kMaxPending = SLICER_MAX_PENDING;
constexpr int kMaxPendingRuns = 8;
constexpr int kMaxSortGroups = 32;  // two-level sort: items (groups of project+bin workgroups) per coarse bin and chunk
struct PendingList {
    int n;
    // Runs of a (unit, tile) bin over the pending chunks: chunk c contributes the runs [run0[c], run0[c + 1]) -- one
    // (its range base[c][bin] .. base[c][bin + 1] of the one-level sort), or one per group of the two-level sort
    // (ptab[c] != nullptr: ptab[c][(unit * ngroups[c] + g) * (tiles_per_unit + 1) + tile .. + 1]).
    int run0[kMaxPending + 1];
    const unsigned *ptab[kMaxPending];
    int ngroups[kMaxPending];
    unsigned *tot;  // two-level sort: records of every bin summed over the pending chunks (k_sort2 adds, k_build_items
                    // reads and zeroes), or nullptr: totals from base[]
    const float2 *sxy[kMaxPending];
    const float *sm[kMaxPending];  // non-null: the chunk's sxy holds 12-byte (xs, ys, m) records (Rec3)
    const unsigned *base[kMaxPending];
    float mconst[kMaxPending];    // (float)massarr[t] of the chunk's file
    float sm_const[kMaxPending];  // sqrtf(mconst)
    // NGP counts only (see NgpFold): chunks of one sub-file carry the same file_id and sit next to each other; fold = the
    // file's counts may be folded into the f32 maps while they are in LDS (its records are complete in this list)
    unsigned short file_id[kMaxPending];
    unsigned char fold[kMaxPending];
    unsigned char done[kMaxPending];  // host bookkeeping: slicer_file_end has closed the chunk's file
};
size_t tile_items_bytes(const BinGeom &G, uint64_t total_particles);
// NGP counts with one constant mass per species: the reference's pixel value is, file by file, the k-fold sequential
// f32 sum s <- fl(s + m) of the file's k entries in the pixel (utilities.cpp:75), added to tot and toti
// (densitymaps.cpp:511-513).  For a sub-file that holds one species only and whose records are complete in the pending
// list (PendingList.fold), the tile kernel does exactly that while the counts are in LDS -- one file after the other,
// in order, each tile owned by one workgroup (no split parts in such a launch) -- instead of adding them to a global
// count map that a map-wide pass folds afterwards; sub-files wait in the pending list, so a snapshot's files share one
// launch.  Files that do not qualify (several species, a chunk through the fused kernel, a flush in mid-file) go
// through the count map and k_fold_ngp as before.
struct NgpFold {
    int on;                    // some pending chunk has fold set
    float *tot[kMaxPlanes];    // all-types map of each plane of the group
    float *toti[kMaxPlanes];   // this species' map, or null
};
hipError_t launch_tile_deposit(const LaunchCfg &cfg, const PassParams &P, const BinGeom &G, const PendingList &L,
                               const Targets &T, const NgpFold &F, void *items_ws, unsigned epoch,
                               uint64_t total_particles, int int_mode, bool *int_cells_used, hipStream_t s);

}  // namespace slicer
