// slicer_binned.hip -- SLICER_ALGO_BINNED: project -> per-tile bins -> LDS-privatised tile deposit.
//
// Why: a TSC deposit is 9 read-modify-writes on a random pixel; as global float atomics that is
// ~0.08 TB/s of added bytes on MI355X (64 lanes in 64 rows; MI355X_MICROARCH.md "Global float
// atomics"), i.e. ~2e9 particles/s.  Here the scatter is done in LDS instead:
//
//   K1 k_project_bin_* : (slicer_project_bin.hip) stream raw POS (12 B/particle, dwordx4 loads), bit-faithful
//                        transform, slab select, fp64 projection; emits (xs, ys) records + their tile bin, a
//                        per-workgroup histogram row and the workgroup's record counts.  No global atomics on the
//                        data path.
//   K2 k_scan_blocks   : exclusive prefix over (bin-major, workgroup-minor) -> every K1 workgroup's write cursor for
//                        every bin (radix-partition style; no atomics), plus in-group bin prefixes and group sums
//                        from which K3 derives the bin bases itself.
//   K3 k_bin_scatter   : moves each record to its bin's contiguous run: persistent workgroups counting-sort
//                        the records of one (plane, K1 workgroup) region by tile in LDS and store them in
//                        tile order (coalesced runs).
//   K4 k_tile_deposit  : one workgroup per (plane, tile) (more for heavy tiles): tile + 1-pixel halo
//                        privatised in LDS as 8-byte cells (f64 or integer) or 4-byte NGP counts, one software-pipelined
//                        walk over all pending chunks, then one shaped (row-contiguous) flush of the non-zero cells;
//                        NGP counts of whole sub-files are folded into the f32 maps at the file boundaries of the walk.
//
// Replaces the CPU loops of gadget2io.cpp:195-274, densitymaps.cpp:355-401 and utilities.cpp:66-95.
#include "slicer_binned_common.hpp"

#pragma clang fp contract(off)

namespace slicer {

// K1 (project + bin) lives in slicer_project_bin.hip.

// ---------------------------------------------------------------------------------------------
// K2a: per bin, exclusive prefix over workgroups + total: segment sums, a 32-way scan in LDS, prefix write.
// ---------------------------------------------------------------------------------------------
constexpr int kScanBins = 32;               // bins per workgroup of k_scan_blocks
constexpr int kScanSegs = 1024 / kScanBins;  // segments of the K1-workgroup axis

__global__ __launch_bounds__(1024) void k_scan_blocks(const unsigned short *__restrict__ hist16,
                                                      unsigned *__restrict__ prefix, unsigned *__restrict__ lex,
                                                      unsigned *__restrict__ gsum, int nblocks, int nbins)
{
    // 32 bins (a half-wave reads 64 contiguous bytes of a histogram row) x 32 segments of the workgroup axis:
    // 256 workgroups for 8192 bins, 16 rows per lane at 512 K1 workgroups.
    // Besides the per-(workgroup, bin) write cursors it leaves, for the bins of its group, the exclusive prefix of the
    // bin totals inside the group (lex) and the group's sum (gsum): the sort kernel turns those into bin bases itself
    // (a 256-entry scan per workgroup), which saves the single-workgroup scan launch that used to sit in between.
    __shared__ unsigned s_seg[kScanSegs][kScanBins];
    const int bl = threadIdx.x % kScanBins, seg = threadIdx.x / kScanBins;
    const int bin = blockIdx.x * kScanBins + bl;
    const size_t stride16 = (size_t)((nbins + 1) >> 1) * 2;
    const int per = (nblocks + kScanSegs - 1) / kScanSegs;
    const int lo = seg * per, hi = lo + per < nblocks ? lo + per : nblocks;
    // (up to kKeep rows per lane -- 512 K1 workgroups give 16 -- stay in registers between the two passes: one read)
    constexpr int kKeep = 16;
    unsigned short keep[kKeep];
    const bool kept = per <= kKeep;
    unsigned sum = 0;
    if (bin < nbins) {
        if (kept) {
#pragma unroll
            for (int j = 0; j < kKeep; j++) {
                keep[j] = lo + j < hi ? hist16[(size_t)(lo + j) * stride16 + bin] : (unsigned short)0;
                sum += keep[j];
            }
        } else {
            for (int b = lo; b < hi; b++)
                sum += hist16[(size_t)b * stride16 + bin];
        }
    }
    s_seg[seg][bl] = sum;
    __syncthreads();
    unsigned run = 0;
    for (int k = 0; k < seg; k++)
        run += s_seg[k][bl];
    if (bin < nbins) {
        if (kept) {
#pragma unroll
            for (int j = 0; j < kKeep; j++)
                if (lo + j < hi) {
                    prefix[(size_t)(lo + j) * nbins + bin] = run;
                    run += keep[j];
                }
        } else {
            for (int b = lo; b < hi; b++) {
                unsigned v = hist16[(size_t)b * stride16 + bin];
                prefix[(size_t)b * nbins + bin] = run;
                run += v;
            }
        }
    }
    if (seg == kScanSegs - 1) {  // lanes 992..1023: one half-wave holds the totals of the group's 32 bins
        const unsigned tot = bin < nbins ? run : 0u;
        unsigned x = tot;
#pragma unroll
        for (int d = 1; d < kScanBins; d <<= 1) {
            const unsigned y = (unsigned)__shfl_up((int)x, d, kScanBins);
            if (bl >= d)
                x += y;
        }
        if (bin < nbins)
            lex[bin] = x - tot;
        if (bl == kScanBins - 1)
            gsum[blockIdx.x] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// K3: scatter records into their bin runs
// ---------------------------------------------------------------------------------------------
// One work item per (unit, K1 workgroup) pair (a unit is a plane, or a band of tile rows of a plane on large maps),
// taken by persistent 512-thread workgroups, two per CU.  An item's records are counting-sorted by tile in LDS
// (sub-batches of kSortBatch records, exchanged kSortStage at a time), so that the records of one (unit, tile) run
// are stored by adjacent lanes: a plain scatter issues one 32-byte sector write per 8-byte record (measured write
// amplification 4.2x), runs of 3-8 records cut that to 1-2 sectors per run.  The phases of an item (loads, scan,
// LDS exchange, stores) are serial inside a workgroup; the second workgroup of the CU fills the gaps (81 -> 70 us).
#ifndef SLICER_K3_BLOCK
#define SLICER_K3_BLOCK 512
#endif
constexpr int kSortBlock = SLICER_K3_BLOCK;
constexpr int kSortBatch = 8192;
#ifndef SLICER_K3_STAGE
#define SLICER_K3_STAGE 4096
#endif
#ifndef SLICER_K3_WAVES
#define SLICER_K3_WAVES 4  // waves per SIMD the register budget allows: two 512-thread workgroups per CU
#endif
constexpr int kSortStage = SLICER_K3_STAGE;  // sorted records staged in LDS at a time

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned *s_wave /*[kSortBlock/64]*/)
{
    // inclusive scan inside the wave, wave totals through LDS
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = (unsigned)__shfl_up((int)x, d);
        if ((int)lane_id() >= d)
            x += y;
    }
    const int w = threadIdx.x >> 6;
    if (lane_id() == 63)
        s_wave[w] = x;
    lds_barrier();
    unsigned off = 0;
    for (int k = 0; k < w; k++)
        off += s_wave[k];
    lds_barrier();
    return off + x - v;
}

template <bool HAS_MASS>
__global__ __launch_bounds__(kSortBlock, SLICER_K3_WAVES) void k_bin_scatter(const float2 *__restrict__ cxy,
                                                            const unsigned short *__restrict__ cbin,
                                                            const float *__restrict__ cm,
                                                            const unsigned *__restrict__ hist16w,
                                                            const unsigned *__restrict__ prefix,
                                                            const unsigned *__restrict__ lex,
                                                            const unsigned *__restrict__ gsum,
                                                            unsigned *__restrict__ base,
                                                            const unsigned *__restrict__ bcount, int nblocks,
                                                            BinGeom G, float2 *__restrict__ sxy,
                                                            float *__restrict__ sm, int count_planes, Targets T)
{
    extern __shared__ unsigned smem_sc[];
    const int tpp = G.tiles_per_unit;
    const int tw = (tpp + 1) >> 1;    // words of a packed u16 table
    unsigned *cnt = smem_sc;          // [tpp] u16 x2 per word: records of the sub-batch per tile (<= kSortBatch)
    unsigned *pos = cnt + tw;         // [tpp] u32 running sorted position of each tile (ends at start + cnt)
    unsigned *cur = pos + tpp;        // [tpp] u32 global write cursor of this unit minus the tile's sorted start
    float2 *sorted_xy = reinterpret_cast<float2 *>(cur + tpp + (tw & 1));
    unsigned short *sorted_tile = reinterpret_cast<unsigned short *>(sorted_xy + kSortStage);
    float *sorted_m = reinterpret_cast<float *>(sorted_tile + kSortStage);  // HAS_MASS only
    __shared__ unsigned s_wave[kSortBlock / 64];
    __shared__ unsigned s_gp[kMaxBins / kScanBins + 1];  // exclusive prefix of the scan kernel's group sums

    const int tid = threadIdx.x;
    // bin bases: base[bin] = s_gp[bin / 32] + lex[bin].  Every workgroup scans the (<= 1024) group sums for itself;
    // workgroup 0 also writes base[] out for the tile kernel and adds every plane's record count (= its selected
    // entries on the TSC path) to the counters.
    {
        const int ngroups = (G.nbins + kScanBins - 1) / kScanBins;
        constexpr int kPerLane = (kMaxBins / kScanBins + kSortBlock - 1) / kSortBlock;  // 2 at 512 threads
        unsigned v[kPerLane], sum = 0;
#pragma unroll
        for (int j = 0; j < kPerLane; j++) {
            const int g = tid * kPerLane + j;
            v[j] = g < ngroups ? gsum[g] : 0u;
            sum += v[j];
        }
        unsigned e = block_exclusive_scan(sum, s_wave);
#pragma unroll
        for (int j = 0; j < kPerLane; j++) {
            const int g = tid * kPerLane + j;
            if (g < ngroups)
                s_gp[g] = e;
            e += v[j];
        }
        if (tid == kSortBlock - 1)
            s_gp[ngroups] = e;  // all records
        lds_barrier();
        if (blockIdx.x == 0) {
            for (int i = tid; i <= G.nbins; i += kSortBlock)
                base[i] = i < G.nbins ? s_gp[i / kScanBins] + lex[i] : s_gp[ngroups];
            if (tid < count_planes) {
                const int bpp = G.units_per_plane * G.tiles_per_unit;  // bins per plane
                auto at = [&](int i) { return i < G.nbins ? s_gp[i / kScanBins] + lex[i] : s_gp[ngroups]; };
                const unsigned c = at((tid + 1) * bpp) - at(tid * bpp);
                if (c)
                    atomicAdd(T.nsel[tid], (unsigned long long)c);
            }
        }
    }
    // Persistent workgroups: each takes the (unit, K1 workgroup) items b, b + gridDim.x, ...  item -> (unit, K1
    // workgroup) keeps an XCD's items on a contiguous range of K1 workgroups (gridDim.x is a multiple of 8, so
    // item & 7 is this workgroup's XCD): runs of one tile written by neighbouring K1 workgroups then meet in the
    // same L2.
    const int per_unit = 8 * ((nblocks + 7) / 8);
    const int per_xcd = per_unit / 8;
    const int per = (tpp + kSortBlock - 1) / kSortBlock;  // tiles per lane in the scan (<= 8)
    auto get16 = [](const unsigned *tab, unsigned t) { return (tab[t >> 1] >> ((t & 1u) * 16u)) & 0xFFFFu; };
    constexpr int R = kSortBatch / kSortBlock;  // records per lane and sub-batch
    // largest selected mass of the species (as the deposit sees it: above MAX_M counts as 0) -> the quantum of integer
    // tile cells (TileQuantum); masses are non-negative, so their bits order like the values.  One atomic per wave and
    // launch, and only if it can raise the maximum.
    float mass_max = 0.0f;
    for (int item = blockIdx.x; item < G.n_units * per_unit; item += gridDim.x) {
        const int plane = item / per_unit;  // the unit index (a whole plane unless the map is large)
        const int u = item % per_unit;
        const int lb = (u & 7) * per_xcd + (u >> 3);
        if (lb >= nblocks)
            continue;
        const unsigned count = bcount[(size_t)plane * nblocks + lb];
        if (count == 0)
            continue;
        lds_barrier();  // the previous item's tables are no longer read
        const unsigned *row = prefix + (size_t)lb * G.nbins + (size_t)plane * tpp;
        const unsigned *lrow = lex + (size_t)plane * tpp;
        const unsigned bin0 = (unsigned)plane * (unsigned)tpp;
        for (int i = tid; i < tpp; i += kSortBlock)
            cur[i] = s_gp[(bin0 + (unsigned)i) / kScanBins] + lrow[i] + row[i];
        const uint64_t r0 = ((uint64_t)plane * nblocks + lb) * (uint64_t)G.region;

        // A region that fits one sub-batch (the usual case) needs no counting pass: its per-tile counts are this K1
        // workgroup's histogram row, already in the packed layout of cnt (needs the unit's first bin word-aligned).
        const bool single = count <= (unsigned)kSortBatch && (((unsigned)plane * (unsigned)tpp) & 1u) == 0u;
        for (unsigned s0 = 0; s0 < count; s0 += kSortBatch) {
            const unsigned nsub = count - s0 < (unsigned)kSortBatch ? count - s0 : (unsigned)kSortBatch;
            if (!single) {
                for (int i = tid; i < tw; i += kSortBlock)
                    cnt[i] = 0;
                lds_barrier();
            }
            unsigned tile[R];
            float2 xy[R];
            float m[R];
#pragma unroll
            for (int k = 0; k < R; k++) {
                const unsigned i = (unsigned)k * kSortBlock + tid;
                if (i < nsub) {
                    tile[k] = cbin[r0 + s0 + i];
                    xy[k] = cxy[r0 + s0 + i];
                    if (HAS_MASS)
                        m[k] = cm[r0 + s0 + i];
                    if (!single)
                        atomicAdd(&cnt[tile[k] >> 1], 1u << ((tile[k] & 1u) * 16u));
                }
            }
            if (single) {
                const unsigned *hrow =
                    hist16w + (size_t)lb * (size_t)((G.nbins + 1) >> 1) + (((size_t)plane * tpp) >> 1);
                for (int i = tid; i < tw; i += kSortBlock)
                    cnt[i] = (i == tw - 1 && (tpp & 1)) ? (hrow[i] & 0xFFFFu) : hrow[i];
            }
            lds_barrier();
            // exclusive scan of cnt -> pos; lane handles tiles [tid*per, tid*per + per).  The cursor is stored minus
            // the tile's sorted start, so that the write-out needs a single table: dst = cur[t] + sorted position.
            {
                unsigned sum = 0;
                for (int j = 0; j < per; j++) {
                    const unsigned t = (unsigned)(tid * per + j);
                    if ((int)t < tpp)
                        sum += get16(cnt, t);
                }
                unsigned e = block_exclusive_scan(sum, s_wave);
                for (int j = 0; j < per; j++) {
                    const unsigned t = (unsigned)(tid * per + j);
                    if ((int)t < tpp) {
                        pos[t] = e;
                        cur[t] -= e;
                        e += get16(cnt, t);
                    }
                }
            }
            lds_barrier();
            // sorted position of every record (one returning LDS add), kept in the upper half of tile[]
#pragma unroll
            for (int k = 0; k < R; k++) {
                const unsigned i = (unsigned)k * kSortBlock + tid;
                if (i < nsub)
                    tile[k] |= atomicAdd(&pos[tile[k]], 1u) << 16;
                else
                    tile[k] = 0xFFFF0000u;  // position 65535: outside every staging round
            }
            if (HAS_MASS) {
#pragma unroll
                for (int k = 0; k < R; k++)
                    if ((unsigned)k * kSortBlock + tid < nsub)
                        mass_max = fmaxf(mass_max, cap_mass(m[k]));
            }
            // exchange through LDS and write out, kSortStage sorted positions at a time (the staging area is what
            // limits the workgroups per CU)
            for (unsigned lo = 0; lo < nsub; lo += kSortStage) {
                lds_barrier();  // positions final (first round) / previous round's staging consumed
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const unsigned q = (tile[k] >> 16) - lo;
                    if (q < (unsigned)kSortStage) {
                        sorted_xy[q] = xy[k];
                        sorted_tile[q] = (unsigned short)(tile[k] & 0xFFFFu);
                        if (HAS_MASS)
                            sorted_m[q] = m[k];
                    }
                }
                lds_barrier();
                const unsigned hi = nsub - lo < (unsigned)kSortStage ? nsub - lo : (unsigned)kSortStage;
                for (unsigned q = tid; q < hi; q += kSortBlock) {
                    const unsigned dst = cur[sorted_tile[q]] + lo + q;
                    if (HAS_MASS) {  // one 12-byte record instead of an 8-byte and a 4-byte stream
                        const float2 v = sorted_xy[q];
                        reinterpret_cast<Rec3 *>(sxy)[dst] = Rec3{v.x, v.y, sorted_m[q]};
                    } else {
                        sxy[dst] = sorted_xy[q];
                    }
                }
            }
            if (!single) {
                lds_barrier();
                // next sub-batch: cursor = old cursor + count = (cursor - start) + (start + count) = cur + pos
                for (int i = tid; i < tpp; i += kSortBlock)
                    cur[i] += pos[i];
            }
        }
    }
    if (HAS_MASS) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
            mass_max = fmaxf(mass_max, __shfl_xor(mass_max, d));
        if (lane_id() == 0 && __float_as_uint(mass_max) > *T.max_mass)
            atomicMax(T.max_mass, __float_as_uint(mass_max));
    }
}

// ---------------------------------------------------------------------------------------------
// K3': second level of the two-level sort
// ---------------------------------------------------------------------------------------------
// The project+bin kernel (SORT2) leaves, per workgroup, sub-batches of records sorted by unit (coarse bin) with a table
// of where each unit's run starts.  One work item here = (unit, group of slots_per_group sub-batch slots = 16
// project+bin workgroups): it gathers the unit's runs of those sub-batches (contiguous pieces of ~0.5 KB), sorts the
// ~6000 records by tile-in-unit in LDS (counting sort over <= 256 tiles) and writes them as ONE contiguous piece of
// sxy, tile after tile, plus the item's row of ptab (where each tile's run starts).  Pieces are allocated with one
// atomic add per item; the tile kernel finds a tile's runs through ptab (build_run_table).  No global histogram, no
// prefix matrix, every store a run of >= ~0.4 KB.
constexpr int kS2Block = 512;
constexpr int kS2Cap = 8192;  // records sorted at a time (the LDS staging area); larger items take several windows
constexpr int kS2R = kS2Cap / kS2Block;
constexpr int kS2MaxMine = 64;  // items one persistent workgroup may have to take (host: nitems <= 64 * workgroups)

template <bool POW2>
__device__ __forceinline__ unsigned sort2_tile_of(float2 r, const PassParams &P, const BinGeom &G)
{
    const int nn = P.nn;
    int gx = grid_index<POW2>(r.x, P), gy = grid_index<POW2>(r.y, P);
    gx = min(max(gx, 0), nn - 1);  // (border-ring entries were binned with the clamped cell)
    gy = min(max(gy, 0), nn - 1);
    const unsigned ty = (unsigned)(gy >> G.th_log2), tx = (unsigned)(gx >> G.tw_log2);
    return (ty % (unsigned)G.rows_per_unit) * (unsigned)G.ntx + tx;
}

template <bool POW2>
__global__ __launch_bounds__(kS2Block, 4) void k_sort2(const float2 *__restrict__ c1, const unsigned *__restrict__ sb_off,
                                                       const unsigned short *__restrict__ sb_start,
                                                       const unsigned *__restrict__ sb_n, int nblocks, int slots_per_group,
                                                       int ngroups, BinGeom G, PassParams P, float2 *__restrict__ sxy,
                                                       unsigned *__restrict__ ptab, const unsigned *__restrict__ item_tot,
                                                       unsigned *tile_tot)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s2[];
    const int S = slots_per_group, ntc = G.tiles_per_unit;
    float2 *stage = reinterpret_cast<float2 *>(smem_s2);                      // [kS2Cap] records in sorted order
    unsigned char *scol = reinterpret_cast<unsigned char *>(stage + kS2Cap);  // [kS2Cap] tile-in-unit of stage[i]
    unsigned *r_src = reinterpret_cast<unsigned *>(scol + kS2Cap);            // [S] first record of run r in c1
    unsigned *r_off = r_src + S;                                              // [S + 1] runs concatenated: start of run r
    unsigned *cnt = r_off + S + 1;                                            // [ntc] records per tile (this window)
    unsigned *pos = cnt + ntc;                                                // [ntc] running sorted position (window)
    unsigned *adj = pos + ntc;  // [ntc] global position of the tile's first record of this window minus its sorted start
    unsigned *cur = adj + ntc;  // [ntc] records of the tile already written (+ its start inside the item)
    unsigned short *first = reinterpret_cast<unsigned short *>(cur + ntc);  // [kS2Cap / 64] run of the window's record 64 c
    __shared__ unsigned s_wave[kS2Block / 64];
    __shared__ unsigned s_mybase[kS2MaxMine];
    const int tid = threadIdx.x;
    const int nslots = nblocks * kSubBatches;
    const int nitems = G.n_units * ngroups;
    // Where this workgroup's items go in sxy: the exclusive prefix of the item totals (summed by the project+bin kernel),
    // in item order -- every workgroup scans the (<= 8192) totals once for itself; no allocation atomics, and the layout
    // of sxy does not depend on the order in which the items are processed.
    {
        const int per = (nitems + kS2Block - 1) / kS2Block;
        unsigned sum = 0;
        for (int j = 0; j < per; j++) {
            const int idx = tid * per + j;
            sum += idx < nitems ? item_tot[idx] : 0u;
        }
        unsigned e = block_exclusive_scan(sum, s_wave);
        for (int j = 0; j < per; j++) {
            const int idx = tid * per + j;
            if (idx < nitems) {
                const int rel = idx - (int)blockIdx.x;
                if (rel >= 0 && rel % (int)gridDim.x == 0)
                    s_mybase[rel / (int)gridDim.x] = e;
                e += item_tot[idx];
            }
        }
        lds_barrier();
    }
    // this thread's run of item `it`: (length, first record in c1); requested one item ahead, so that the two dependent
    // table reads of the next item travel while the current one is sorted
    auto item_run = [&](int it, unsigned &len, unsigned &src) {
        len = 0;
        src = 0;
        if (it >= nitems || tid >= S)
            return;
        const int g = it / G.n_units, b = it % G.n_units;
        const int sl = g * S + tid;
        if (sl < nslots) {
            const int w = sl / kSubBatches, f = sl % kSubBatches;
            if ((unsigned)f < sb_n[w]) {
                const unsigned a = sb_start[(size_t)sl * kSubRow + b], e = sb_start[(size_t)sl * kSubRow + b + 1];
                len = e - a;
                src = sb_off[sl] + a;
            }
        }
    };
    unsigned len_next, src_next;
    item_run(blockIdx.x, len_next, src_next);
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int g = it / G.n_units, b = it % G.n_units;  // (neighbouring items read neighbouring runs of the same sub-batches)
        const unsigned len = len_next, src = src_next;
        item_run(it + (int)gridDim.x, len_next, src_next);
        lds_barrier();  // the previous item's tables and staging are no longer read
        const unsigned off = block_exclusive_scan(len, s_wave);
        if (tid < S) {
            r_src[tid] = src;
            r_off[tid] = off;
        }
        if (tid == S - 1)
            r_off[S] = off + len;
        for (int i = tid; i < ntc; i += kS2Block)
            cnt[i] = 0;
        lds_barrier();
        const unsigned total = r_off[S];
        unsigned *prow = ptab + ((size_t)b * (size_t)ngroups + (size_t)g) * (size_t)(ntc + 1);
        // (an item whose runs do not add up to the total the project+bin kernel reported is dropped rather than
        // written over its neighbours: cannot happen unless the two kernels disagree, and then the parity tests see it)
        if (total == 0 || total != item_tot[it]) {  // (uniform)
            for (int i = tid; i <= ntc; i += kS2Block)
                prow[i] = 0;
            continue;
        }
        const unsigned s_base = s_mybase[(it - (int)blockIdx.x) / (int)gridDim.x];
        const bool multi = total > (unsigned)kS2Cap;
        // windows of kS2Cap records of the concatenated runs; pass 0 of a multi-window item only counts (the item's
        // tile starts must be known before its first record is placed)
        for (int pass = multi ? 0 : 1; pass < 2; pass++) {
            for (unsigned p0 = 0; p0 < total; p0 += kS2Cap) {
                const unsigned p1 = p0 + kS2Cap < total ? p0 + kS2Cap : total, nsub = p1 - p0;
                // run that holds the first record of every piece of 64 (binary search over the run starts)
                for (unsigned pc = tid; pc < (nsub + 63) >> 6; pc += kS2Block) {
                    const unsigned q = p0 + (pc << 6);
                    int lo = 0, hi = S - 1;  // largest r with r_off[r] <= q
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (r_off[mid] <= q)
                            lo = mid;
                        else
                            hi = mid - 1;
                    }
                    first[pc] = (unsigned short)lo;
                }
                lds_barrier();
                // gather: thread tid takes the window's records tid, tid + 512, ...: sixteen independent loads
                float2 rec[kS2R];
                unsigned sp[kS2R];
#pragma unroll
                for (int k = 0; k < kS2R; k++) {
                    const unsigned i = (unsigned)k * kS2Block + tid;
                    const unsigned q = p0 + (i < nsub ? i : nsub - 1);  // clamped: unconditional loads
                    int r = first[(q - p0) >> 6];
                    while (q >= r_off[r + 1])
                        r++;
                    rec[k] = c1[r_src[r] + (q - r_off[r])];
                }
#pragma unroll
                for (int k = 0; k < kS2R; k++) {
                    const unsigned i = (unsigned)k * kS2Block + tid;
                    sp[k] = 0xFFFFFFFFu;
                    if (i < nsub) {
                        sp[k] = sort2_tile_of<POW2>(rec[k], P, G);
                        atomicAdd(&cnt[sp[k]], 1u);
                    }
                }
                lds_barrier();
                const unsigned c = tid < ntc ? cnt[tid] : 0u;
                if (pass == 0) {  // counting pass: leave the counts, the item's tile starts follow after the last window
                    if (p1 < total)
                        continue;
                    const unsigned e = block_exclusive_scan(c, s_wave);
                    if (tid < ntc) {
                        cur[tid] = e;  // start of the tile's records inside the item
                        prow[tid] = s_base + e;
                        cnt[tid] = 0;
                    }
                    if (tid == 0)
                        prow[ntc] = s_base + total;
                    lds_barrier();
                    continue;
                }
                const unsigned e = block_exclusive_scan(c, s_wave);
                if (tid < ntc) {
                    unsigned at;  // start of this window's records of the tile inside the item
                    if (multi) {
                        at = cur[tid];
                    } else {
                        at = e;
                        prow[tid] = s_base + e;
                    }
                    pos[tid] = e;
                    adj[tid] = s_base + at - e;
                    cur[tid] = at + c;
                    cnt[tid] = 0;
                    if (c)
                        atomicAdd(&tile_tot[(size_t)b * ntc + tid], c);
                }
                if (!multi && tid == 0)
                    prow[ntc] = s_base + total;
                lds_barrier();
#pragma unroll
                for (int k = 0; k < kS2R; k++)
                    if (sp[k] != 0xFFFFFFFFu) {
                        const unsigned t = sp[k], at = atomicAdd(&pos[t], 1u);
                        stage[at] = rec[k];
                        scol[at] = (unsigned char)t;
                    }
                lds_barrier();
                for (unsigned i = tid; i < nsub; i += kS2Block)
                    sxy[adj[scol[i]] + i] = stage[i];
                if (p1 < total)
                    lds_barrier();  // the next window overwrites the staging area
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4: LDS-privatised tile deposit
// ---------------------------------------------------------------------------------------------
// Global accumulator type of each mode, and the type of the LDS tile cell.  Measured on MI355X
// (tools/lds_atomic_bench.hip, 3x3 random cells of a 130x130 tile, 512-thread workgroups):
//   ds_add_f32 0.20 T lane-ops/s   ds_add_f64 1.81 T   ds_add_u64 3.51 T   ds_add_u32 6.61 T
// ds_add_f32 is 9x slower than ds_add_f64, so no mode keeps f32 cells in LDS: f32/f64 modes sum the
// tile in f64 (and round once per flush), fixed point in u64, NGP counts in u32.
template <int ACC> struct AccT { using type = float; using lds = double; };
template <> struct AccT<kF64> { using type = double; using lds = double; };
template <> struct AccT<kFixed64> { using type = unsigned long long; using lds = unsigned long long; };
template <> struct AccT<kCountU32> { using type = unsigned; using lds = unsigned; };
template <> struct AccT<kF32I> { using type = float; using lds = unsigned long long; };
template <> struct AccT<kF64I> { using type = double; using lds = unsigned long long; };
template <int ACC> constexpr bool kIntCells = ACC == kF32I || ACC == kF64I;

// Integer tile cells of the F32 / F64 modes.  A contribution c is exact in units of 2^-49 of the mass scale iff
// c * tile_scale is an integer -- true for every c >= tile_cmin = 2^-25 scales (24-bit mantissa), i.e. for all but the
// ~0.1 % of contributions that are vanishing TSC weights.  Those are not rounded into the tile (a pixel holding nothing
// else must come out exact: T-TSC with k = 1) but added straight to the global map with a float atomic.  The kernel
// tests one number per record -- the smallest of its nine products -- and notes the rare records that fail in an LDS
// list, treated after the loop (slow_record); the loop itself stays branch-free.
constexpr unsigned kSlowCap = 512;  // noted records per work item (<= 16384 records: 1 % are ~160)

// The quantum of an integer-cell tile: 2^(le - 49) with 2^le above the (largest) particle mass of the launch.  With one
// constant mass it comes from the pass parameters; with per-particle masses from the largest selected mass the sort
// kernel has seen for the species (Targets::max_mass), read when the tile kernel starts.
struct TileQuantum {
    double scale, inv_scale;  // 2^(49 - le), 2^(le - 49)
    float cmin;               // 2^(le - 25): every contribution >= this is an exact multiple of the quantum
};

template <int ACC>
__device__ __forceinline__ void lds_add(typename AccT<ACC>::lds *cell, float c, const PassParams &P, const TileQuantum &Q)
{
    if (ACC == kF32 || ACC == kF64)
        atomicAdd(reinterpret_cast<double *>(cell), (double)c);  // ds_add_f64
    else if (ACC == kFixed64)
        atomicAdd(reinterpret_cast<unsigned long long *>(cell), rn_scaled_u64(c, P.fixed_scale));  // ds_add_u64
    else if (kIntCells<ACC>)
        atomicAdd(reinterpret_cast<unsigned long long *>(cell), rn_scaled_u64(c, Q.scale));
}

// One record of an integer-cell tile with the representability test per contribution: exact ones into the tile, the
// others straight to the global accumulator map (gmap, acc_t = float or double).
template <int ACC, bool POW2>
__device__ __forceinline__ void slow_record(float xs, float ys, float sq, const PassParams &P, const TileQuantum &Q,
                                            typename AccT<ACC>::lds *tile, typename AccT<ACC>::type *gmap, int x0, int y0,
                                            int W)
{
    using acc_t = typename AccT<ACC>::type;
    const int nn = P.nn;
    const int gx = grid_index<POW2>(xs, P), gy = grid_index<POW2>(ys, P);
    float wx[3], wy[3];
    tsc_axis<POW2>(xs, gx, P, wx);
    tsc_axis<POW2>(ys, gy, P, wy);
    for (int a = 0; a < 3; a++) {
        wx[a] = sq * wx[a];
        wy[a] = sq * wy[a];
    }
    for (int b = 0; b < 3; b++)
        for (int a = 0; a < 3; a++) {
            const int px = gx + a - 1, py = gy + b - 1;
            if (px < 0 || px >= nn || py < 0 || py >= nn)
                continue;
            const float c = wx[a] * wy[b];
            const double t = (double)c * Q.scale;
            if (t == rint(t))
                atomicAdd(reinterpret_cast<unsigned long long *>(tile + (gy - y0 + b) * W + (gx - x0 + a)),
                          (unsigned long long)t);
            else
                atomicAdd(gmap + (size_t)px + (size_t)nn * (size_t)py, (acc_t)c);
        }
}

// Work items of the tile kernel: a (plane, tile) bin with many records (a halo core can put 10^5..10^7
// particles into one tile) is split into parts of <= kItemRecs records, each deposited by its own workgroup
// into its own LDS tile and flushed atomically, so that one heavy tile neither serialises on one CU nor
// stretches the kernel's tail.  Empty bins get no item.
#ifndef SLICER_ITEM_RECS
#define SLICER_ITEM_RECS 16384
#endif
constexpr unsigned kItemRecs = SLICER_ITEM_RECS;
#ifndef SLICER_WHOLE_RECS
#define SLICER_WHOLE_RECS 65536
#endif
// ... but a bin of up to kWholeRecs records stays whole: every part pays for zeroing and flushing a tile of its own
// (536 us against 554 us for the uniform headline case), while only a bin far beyond the usual load needs many hands
constexpr unsigned kWholeRecs = SLICER_WHOLE_RECS;
#ifndef SLICER_MERGE_RECS
#define SLICER_MERGE_RECS 131072
#endif
// bins beyond about SLICER_MERGE_RECS records (a halo core) go through the wave-level pre-reduction
constexpr unsigned kMergeParts = SLICER_MERGE_RECS / kItemRecs > 2 ? SLICER_MERGE_RECS / kItemRecs : 2;

struct TileItems {
    unsigned *nparts;   // [nbins] parts of every bin (0 = empty: no work item)
    uint2 *extra;       // {bin, part} of the parts >= 1 of heavy bins, in no particular order
    unsigned *n_extra;  // entries of extra[]: zero before k_build_items, which also zeroes next_n_extra
    unsigned *next_n_extra;
};

// Work items of the tile kernel: part 0 of bin b is workgroup b; the further parts of heavy bins are appended to a
// list with one atomic add per heavy bin (their order does not matter), so that the builder is a plain parallel
// kernel instead of a single-workgroup scan.
__global__ __launch_bounds__(256) void k_build_items(PendingList L, int nbins, TileItems I, int whole)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b == 0)
        *I.next_n_extra = 0;  // the counter of the next launch (stream order makes this safe)
    if (b >= nbins)
        return;
    unsigned tot = 0;
    if (L.tot) {  // two-level sort: the sort kernel summed every pending chunk's records of the bin; zeroed for the next list
        tot = L.tot[b];
        L.tot[b] = 0;
    } else {
        for (int c = 0; c < L.n; c++)
            tot += L.base[c][b + 1] - L.base[c][b];
    }
    // (whole: the launch folds NGP counts file by file inside the tile kernel, which needs every tile in one workgroup)
    const unsigned np = (whole || tot <= kWholeRecs) ? (tot != 0) : (tot + kItemRecs - 1) / kItemRecs;
    I.nparts[b] = np;
    if (np > 1) {
        const unsigned at = atomicAdd(I.n_extra, np - 1);
        for (unsigned q = 1; q < np; q++)
            I.extra[at + q - 1] = make_uint2((unsigned)b, q);
    }
}

constexpr int kTileBlock = 1024;

// One-level sort (one run per pending chunk: L.base): deposit this work item's share of every pending chunk into the
// LDS tile, the whole workgroup walking chunk by chunk.  CHECK = false for tiles whose halo
// lies inside the map (all but the border tiles): the per-cell map-edge tests and their exec-mask bookkeeping go.
struct NoBoundary {
    __device__ __forceinline__ void operator()(int, int) const {}
};

// boundary(c, c_next) is called when the walk leaves chunk c for chunk c_next (c_end at the very end), after the
// records of c_next's first round have been requested: the NGP path folds a finished sub-file there.
template <int MAS, int ACC, bool POW2, bool HAS_MASS, bool CHECK, typename Boundary = NoBoundary>
__device__ __forceinline__ void tile_accumulate_chunks(const PendingList &L, const PassParams &P,
                                                typename AccT<ACC>::lds *tile, unsigned bin, unsigned part,
                                                unsigned nparts, int x0, int y0, int W, unsigned *s_nslow,
                                                uint2 *s_slow, typename AccT<ACC>::type *gmap, const TileQuantum &Q,
                                                int c_begin, int c_end, Boundary &&boundary = Boundary())
{
    using lds_t = typename AccT<ACC>::lds;
    const int tid = threadIdx.x;
    const int nn = P.nn;
#ifndef SLICER_K4_U
#define SLICER_K4_U 2
#endif
    // records in flight per lane and round.  A chunk brings a tile of the headline case ~1600 records: with 2 x 1024
    // slots per round one round per chunk, no slot group that only loads clamped duplicates (A/B in one call, round 3:
    // 450-454 us with 2 against 467-472 with 4; clustered and 2048^2 (6400 records per chunk and tile): equal)
    constexpr int U = SLICER_K4_U;
    // (dealing the waves to the pending chunks, so that all runs stream in at once, measured 699 us against 665 us for
    // this chunk-by-chunk walk: the kernel is bound by the LDS atomic pipe, not by the loads)
    // This part's share of each chunk's run, [len*part/nparts, len*(part+1)/nparts): lane c of every wave fetches
    // chunk c's bounds, so the (<= 8) dependent loads cost one latency instead of one per chunk.
    unsigned my_start = 0, my_end = 0;
    {
        const int c = c_begin + (int)lane_id();
        if (c < c_end) {
            const unsigned run0 = L.base[c][bin], len = L.base[c][bin + 1] - run0;
            my_start = run0 + (unsigned)(((unsigned long long)len * part) / nparts);
            my_end = run0 + (unsigned)(((unsigned long long)len * (part + 1)) / nparts);
        }
    }
    auto bounds = [&](int c, unsigned &a, unsigned &b) {
        a = (unsigned)__builtin_amdgcn_readlane((int)my_start, c - c_begin);
        b = (unsigned)__builtin_amdgcn_readlane((int)my_end, c - c_begin);
    };
    // The walk over (chunk, round of U * kTileBlock records) pairs is software-pipelined: the records of the next
    // round -- of the next chunk, if this one is exhausted -- are requested before the current round is deposited.
    auto advance = [&](int &c, unsigned &i0, unsigned &end) {  // -> false when the walk is over
        i0 += U * kTileBlock;
        while (i0 >= end) {
            if (++c >= c_end)
                return false;
            bounds(c, i0, end);
        }
        return true;
    };
    auto fetch = [&](int c, unsigned i0, unsigned end, float2 (&r)[U], float (&mr)[U]) {
        const float2 *__restrict__ sxy = L.sxy[c];
#pragma unroll
        for (int u = 0; u < U; u++) {
            // clamped, unconditional loads: a load under a branch makes the compiler drain the memory queue
            // before each one (s_waitcnt vmcnt(0)), which serialises the U loads
            const unsigned i = i0 + u * kTileBlock + tid;
            const unsigned ic = i < end ? i : end - 1;
            if (HAS_MASS) {
                const Rec3 v = reinterpret_cast<const Rec3 *>(sxy)[ic];
                r[u] = make_float2(v.x, v.y);
                mr[u] = v.m;
            } else {
                r[u] = sxy[ic];
            }
        }
    };
    int c = c_begin - 1;
    unsigned i0 = 0, end = 0;
    bool live = advance(c, i0, end);
    float2 r[U], rn[U];
    float mr[U], mn[U];
    if (live)
        fetch(c, i0, end, r, mr);
    while (live) {
        int cn = c;
        unsigned in0 = i0, endn = end;
        const bool more = advance(cn, in0, endn);
        if (more)
            fetch(cn, in0, endn, rn, mn);
        const float mconst = L.mconst[c], smc = L.sm_const[c];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned i = i0 + u * kTileBlock + tid;
            if (i >= end)
                continue;
            const float xs = r[u].x, ys = r[u].y;
            float m = mconst, sq = smc;
            if (HAS_MASS) {
                m = cap_mass(mr[u]);
                sq = __fsqrt_rn(m);
            }
            const int gx = grid_index<POW2>(xs, P);
            const int gy = grid_index<POW2>(ys, P);
            if (MAS == kNGP) {
                lds_t *cell = tile + (gy - y0 + 1) * W + (gx - x0 + 1);
                if (ACC == kCountU32)
                    atomicAdd(reinterpret_cast<unsigned *>(cell), 1u);
                else
                    atomicAdd(reinterpret_cast<double *>(cell), (double)m);
            } else {
                float wx[3], wy[3];
                tsc_axis<POW2>(xs, gx, P, wx);
                tsc_axis<POW2>(ys, gy, P, wy);
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    wx[a] = sq * wx[a];
                    wy[a] = sq * wy[a];
                }
                if (kIntCells<ACC>) {
                    // smallest of the nine products (weights are >= 0; the centre cell holds the largest)
                    if (HAS_MASS && m == 0.0f)
                        continue;  // (a mass above MAX_M counts as 0: nine additions of +0)
                    const float cmin = fminf(wx[0], wx[2]) * fminf(wy[0], wy[2]);
                    if (cmin < Q.cmin) {  // rare: a contribution that is not a multiple of the tile's quantum
                        const unsigned k = atomicAdd(s_nslow, 1u);
                        if (k < kSlowCap)
                            s_slow[k] = make_uint2((unsigned)c, i);
                        else
                            slow_record<ACC, POW2>(xs, ys, sq, P, Q, tile, gmap, x0, y0, W);
                        continue;
                    }
                }
                lds_t *cell0 = tile + (gy - y0) * W + (gx - x0);  // cell (gx - 1, gy - 1)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const int py = gy + b - 1;
                    if (CHECK && (py < 0 || py >= nn))
                        continue;
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        const int px = gx + a - 1;
                        if (CHECK && (px < 0 || px >= nn))
                            continue;
                        lds_add<ACC>(cell0 + b * W + a, wx[a] * wy[b], P, Q);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            r[u] = rn[u];
            if (HAS_MASS)
                mr[u] = mn[u];
        }
        if (!more || cn != c)
            boundary(c, more ? cn : c_end);
        c = cn;
        i0 = in0;
        end = endn;
        live = more;
    }
}

// The records of one work item: a table of runs in LDS.  A run is a contiguous range of one pending chunk's sorted
// records: the chunk's whole (plane, tile) range with the one-level sort (k_bin_scatter: one run per chunk), or the
// tile's range inside one item of the two-level sort (k_sort2: one run per chunk and group, ~100 records each).  A part
// of a split bin takes the same fraction of every run.  The runs are walked as ONE sequence of records (s_pre: where
// each run starts in it), 64 consecutive records per wave instruction whatever the run lengths are.
struct RunRef {
    unsigned chunk, start;  // records [start, start + length) of L.sxy[chunk]; length = s_pre[k + 1] - s_pre[k]
};
constexpr int kMaxRuns = kMaxPendingRuns * kMaxSortGroups;
constexpr unsigned kWalkWindow = 65536;  // records walked per s_first table (kWalkWindow / 64 entries)

struct RunTable {
    RunRef *run;            // [kMaxRuns]
    unsigned *pre;          // [kMaxRuns + 1] exclusive prefix of the run lengths
    unsigned short *first;  // [kWalkWindow / 64] run that holds record 64 c of the current window
    const float2 **sxy;     // [kMaxPending] sorted records of every pending chunk
    float2 *mc;             // [kMaxPending] {mconst, sqrtf(mconst)} of every pending chunk
    int n;
};
constexpr size_t kRunTableBytes = sizeof(RunRef) * kMaxRuns + 4 * (kMaxRuns + 1) + 4 + 2 * (kWalkWindow / 64) +
                                  16 * kMaxPending;

__device__ __forceinline__ RunTable run_table_at(unsigned char *p)  // p: 8-byte aligned
{
    RunTable R;
    R.sxy = reinterpret_cast<const float2 **>(p);
    R.mc = reinterpret_cast<float2 *>(p + 8 * kMaxPending);
    R.run = reinterpret_cast<RunRef *>(p + 16 * kMaxPending);
    R.pre = reinterpret_cast<unsigned *>(R.run + kMaxRuns);
    R.first = reinterpret_cast<unsigned short *>(R.pre + kMaxRuns + 2);
    R.n = 0;
    return R;
}

// Fill the run table of (bin, part); every thread of the workgroup calls it (it holds barriers).
__device__ __forceinline__ void build_run_table(const PendingList &L, const BinGeom &G, unsigned bin, unsigned part,
                                                unsigned nparts, RunTable &R)
{
    const int tid = threadIdx.x;
    const int n_runs = L.run0[L.n];
    R.n = n_runs;
    if (tid < L.n) {
        R.sxy[tid] = L.sxy[tid];
        R.mc[tid] = make_float2(L.mconst[tid], L.sm_const[tid]);
    }
    for (int k = tid; k < n_runs; k += kTileBlock) {
        int c = 0;
        while (c + 1 < L.n && k >= L.run0[c + 1])
            c++;
        unsigned a, e;
        if (L.ptab[c]) {  // two-level sort: item (unit, group) of chunk c, entry tile-in-unit
            const unsigned unit = bin / (unsigned)G.tiles_per_unit, t = bin % (unsigned)G.tiles_per_unit;
            const unsigned g = (unsigned)(k - L.run0[c]);
            const unsigned *tab = L.ptab[c] + ((size_t)unit * (size_t)L.ngroups[c] + g) * (size_t)(G.tiles_per_unit + 1);
            a = tab[t];
            e = tab[t + 1];
        } else {
            a = L.base[c][bin];
            e = L.base[c][bin + 1];
        }
        const unsigned len = e - a;
        const unsigned lo = (unsigned)(((unsigned long long)len * part) / nparts);
        const unsigned hi = (unsigned)(((unsigned long long)len * (part + 1)) / nparts);
        RunRef r;
        r.chunk = (unsigned)c;
        r.start = a + lo;
        R.run[k] = r;
        R.pre[k + 1] = hi - lo;  // (lengths first; the prefix follows)
    }
    __syncthreads();
    if (tid < 64) {  // exclusive prefix over <= 256 lengths: four per lane of wave 0
        constexpr int PER = kMaxRuns / 64;
        unsigned v[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int k = tid * PER + j;
            v[j] = k < n_runs ? R.pre[k + 1] : 0u;
            sum += v[j];
        }
        unsigned x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = (unsigned)__shfl_up((int)x, d);
            if (tid >= d)
                x += y;
        }
        unsigned e = x - sum;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int k = tid * PER + j;
            e += v[j];
            if (k < n_runs)
                R.pre[k + 1] = e;
        }
        if (tid == 0)
            R.pre[0] = 0;
    }
    __syncthreads();
}

// Deposit the runs [k_begin, k_end) of the table into the LDS tile.  Every thread of the workgroup calls it (barriers
// around the s_first table); inside, every WAVE walks on its own: the records of the runs form one sequence, cut into
// pieces of 64; wave w takes pieces w, w + 16, ... -- two in flight, the next two requested before the current ones are
// deposited.  A lane finds its record's run from the piece's first run (s_first) and the run boundaries that follow.
// CHECK = false for tiles whose halo lies inside the map (all but the border tiles): the per-cell map-edge tests and
// their exec-mask bookkeeping go.
template <int MAS, int ACC, bool POW2, bool HAS_MASS, bool CHECK>
__device__ __forceinline__ void tile_accumulate(const PassParams &P, typename AccT<ACC>::lds *tile, const RunTable &R,
                                                int k_begin, int k_end, int x0, int y0, int W, unsigned *s_nslow,
                                                uint2 *s_slow, typename AccT<ACC>::type *gmap, const TileQuantum &Q)
{
    using lds_t = typename AccT<ACC>::lds;
    const unsigned lane = lane_id();
    const int wave = threadIdx.x >> 6;
    constexpr int NW = kTileBlock / 64;
    const int nn = P.nn;
#ifndef SLICER_K4_UR
#define SLICER_K4_UR 2
#endif
    constexpr int U = SLICER_K4_UR;  // pieces (of 64 records) in flight per wave
    const unsigned q_begin = R.pre[k_begin], q_end = R.pre[k_end];
    for (unsigned w0 = q_begin; w0 < q_end; w0 += kWalkWindow) {  // (one window unless a tile holds > 65536 records)
        const unsigned w1 = q_end - w0 > kWalkWindow ? w0 + kWalkWindow : q_end;
        const unsigned npiece = (w1 - w0 + 63) >> 6;
        if (w0 != q_begin)
            __syncthreads();  // the previous window's table is no longer read
        for (unsigned pc = threadIdx.x; pc < npiece; pc += kTileBlock) {  // run that holds the piece's first record
            const unsigned q = w0 + (pc << 6);
            int lo = k_begin, hi = k_end - 1;  // largest k with pre[k] <= q
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (R.pre[mid] <= q)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            R.first[pc] = (unsigned short)lo;
        }
        __syncthreads();
        struct Rec {
            float2 r;
            float m;
            unsigned i, c;  // record index inside its chunk, chunk (c = ~0: no record)
        };
        auto fetch = [&](unsigned pc, Rec &o) {
            o.c = 0xFFFFFFFFu;
            o.i = 0;
            o.r = make_float2(0.f, 0.f);
            o.m = 0.f;
            if (pc >= npiece)
                return;
            const unsigned q = w0 + (pc << 6) + lane;
            const unsigned qc = q < w1 ? q : w1 - 1;  // clamped: an unconditional load (a load under a lane-dependent
                                                        // branch makes the compiler drain the memory queue first)
            int k = R.first[pc];
            while (qc >= R.pre[k + 1])
                k++;
            const RunRef rr = R.run[k];
            const unsigned i = rr.start + (qc - R.pre[k]);
            const float2 *__restrict__ sxy = R.sxy[rr.chunk];
            if (HAS_MASS) {
                const Rec3 v = reinterpret_cast<const Rec3 *>(sxy)[i];
                o.r = make_float2(v.x, v.y);
                o.m = v.m;
            } else {
                o.r = sxy[i];
            }
            o.i = i;
            o.c = q < w1 ? rr.chunk : 0xFFFFFFFFu;
        };
        Rec cur[U], nxt[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            fetch((unsigned)wave + (unsigned)(u * NW), cur[u]);
        for (unsigned pc = (unsigned)wave; pc < npiece; pc += U * NW) {
#pragma unroll
            for (int u = 0; u < U; u++)
                fetch(pc + (unsigned)((U + u) * NW), nxt[u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned c = cur[u].c, i = cur[u].i;
                if (c == 0xFFFFFFFFu)
                    continue;
                const float xs = cur[u].r.x, ys = cur[u].r.y;
                const float2 mc = R.mc[c];
                float m = mc.x, sq = mc.y;
                if (HAS_MASS) {
                    m = cap_mass(cur[u].m);
                    sq = __fsqrt_rn(m);
                }
                const int gx = grid_index<POW2>(xs, P);
                const int gy = grid_index<POW2>(ys, P);
                if (MAS == kNGP) {
                    lds_t *cell = tile + (gy - y0 + 1) * W + (gx - x0 + 1);
                    if (ACC == kCountU32)
                        atomicAdd(reinterpret_cast<unsigned *>(cell), 1u);
                    else
                        atomicAdd(reinterpret_cast<double *>(cell), (double)m);
                } else {
                    float wx[3], wy[3];
                    tsc_axis<POW2>(xs, gx, P, wx);
                    tsc_axis<POW2>(ys, gy, P, wy);
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        wx[a] = sq * wx[a];
                        wy[a] = sq * wy[a];
                    }
                    if (kIntCells<ACC>) {
                        // smallest of the nine products (weights are >= 0; the centre cell holds the largest)
                        if (HAS_MASS && m == 0.0f)
                            continue;  // (a mass above MAX_M counts as 0: nine additions of +0)
                        const float cmin = fminf(wx[0], wx[2]) * fminf(wy[0], wy[2]);
                        if (cmin < Q.cmin) {  // rare: a contribution that is not a multiple of the tile's quantum
                            const unsigned kq = atomicAdd(s_nslow, 1u);
                            if (kq < kSlowCap)
                                s_slow[kq] = make_uint2(c, i);
                            else
                                slow_record<ACC, POW2>(xs, ys, sq, P, Q, tile, gmap, x0, y0, W);
                            continue;
                        }
                    }
                    lds_t *cell0 = tile + (gy - y0) * W + (gx - x0);  // cell (gx - 1, gy - 1)
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        const int py = gy + b - 1;
                        if (CHECK && (py < 0 || py >= nn))
                            continue;
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            const int px = gx + a - 1;
                            if (CHECK && (px < 0 || px >= nn))
                                continue;
                            lds_add<ACC>(cell0 + b * W + a, wx[a] * wy[b], P, Q);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                cur[u] = nxt[u];
        }
    }
}

// the k-fold sequential f32 sum s <- fl(s + m) of utilities.cpp:75: what a pixel of the reference's per-file NGP map holds
__device__ __forceinline__ float ngp_seq_sum(unsigned k, float m)
{
    float s = 0.0f;
    for (unsigned j = 0; j < k; j++)
        s = s + m;
    return s;
}

// ---- heavy bins: wave-level pre-reduction before the LDS atomic ---------------------------------------------------
// A bin that k_build_items split into parts holds a halo core: many records in a few pixels.  64 lanes adding to the
// same LDS cell serialise (the 9 ds_add of a wave cost ~64x their usual LDS time), so the parts of such bins first ask
// whether every active lane of the wave targets the same cell; if so the nine contributions are summed across the wave
// (xor butterfly) and one lane issues the nine atomics.  Sums are reordered (allowed in the F32 / F64 modes, exact in
// FIXED64: integers); waves that straddle several cells fall back to per-lane atomics.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += (unsigned long long)__shfl_xor((long long)v, off);
    return v;
}

template <int MAS, int ACC, bool POW2, bool HAS_MASS>
__device__ __forceinline__ void tile_accumulate_merged_chunks(const PendingList &L, const PassParams &P,
                                                       typename AccT<ACC>::lds *tile, unsigned bin, unsigned part,
                                                       unsigned nparts, int x0, int y0, int W,
                                                       typename AccT<ACC>::type *gmap, const TileQuantum &Q)
{
    using lds_t = typename AccT<ACC>::lds;
    const int tid = threadIdx.x;
    const int nn = P.nn;
    for (int c = 0; c < L.n; c++) {
        const unsigned run0 = L.base[c][bin], len = L.base[c][bin + 1] - run0;
        const unsigned start = run0 + (unsigned)(((unsigned long long)len * part) / nparts);
        const unsigned end = run0 + (unsigned)(((unsigned long long)len * (part + 1)) / nparts);
        const float2 *__restrict__ sxy = L.sxy[c];
        // every lane of the workgroup runs every iteration (no divergent exit: the butterfly needs all lanes)
        for (unsigned i0 = start; i0 < end; i0 += kTileBlock) {
            const unsigned i = i0 + tid;
            const bool act = i < end;
            float2 r = make_float2(0.f, 0.f);
            float m = L.mconst[c], sq = L.sm_const[c];
            if (HAS_MASS) {
                Rec3 v{0.f, 0.f, 0.f};
                if (act)
                    v = reinterpret_cast<const Rec3 *>(sxy)[i];
                r = make_float2(v.x, v.y);
                m = cap_mass(v.m);
                sq = __fsqrt_rn(m);
            } else if (act) {
                r = sxy[i];
            }
            const int gx = grid_index<POW2>(r.x, P), gy = grid_index<POW2>(r.y, P);
            const int cell = act ? (gy - y0) * W + (gx - x0) : -1;  // cell (gx - 1, gy - 1) of the halo'd tile
            const unsigned long long am = __ballot(act);
            if (am == 0ull)
                continue;
            const int lead = (int)__builtin_ctzll(am);
            const int cell0 = __shfl(cell, lead);
            const bool uniform = __ballot(act && cell != cell0) == 0ull;
            if (MAS == kNGP) {
                if (uniform) {
                    if ((int)lane_id() == lead) {
                        lds_t *cc = tile + cell0 + W + 1;
                        if (ACC == kCountU32)
                            atomicAdd(reinterpret_cast<unsigned *>(cc), (unsigned)__popcll(am));
                    }
                    if (ACC != kCountU32) {
                        const double tot = wave_sum(act ? (double)m : 0.0);
                        if ((int)lane_id() == lead)
                            atomicAdd(reinterpret_cast<double *>(tile + cell0 + W + 1), tot);
                    }
                } else if (act) {
                    lds_t *cc = tile + cell + W + 1;
                    if (ACC == kCountU32)
                        atomicAdd(reinterpret_cast<unsigned *>(cc), 1u);
                    else
                        atomicAdd(reinterpret_cast<double *>(cc), (double)m);
                }
                continue;
            }
            float wx[3], wy[3];
            tsc_axis<POW2>(r.x, gx, P, wx);
            tsc_axis<POW2>(r.y, gy, P, wy);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                wx[a] = sq * wx[a];
                wy[a] = sq * wy[a];
            }
#pragma unroll
            for (int b = 0; b < 3; b++) {
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    // map-edge tests as in the CHECK variant (heavy border tiles are rare enough not to specialise)
                    const int px = gx + a - 1, py = gy + b - 1;
                    bool in = act && px >= 0 && px < nn && py >= 0 && py < nn;
                    const float cf = wx[a] * wy[b];
                    if (kIntCells<ACC>) {  // contributions that are no multiple of the tile's quantum bypass the tile
                        const double t = (double)cf * Q.scale;
                        if (in && t != rint(t)) {
                            atomicAdd(gmap + (size_t)px + (size_t)nn * (size_t)py, (typename AccT<ACC>::type)cf);
                            in = false;
                        }
                    }
                    if (uniform) {  // wave-uniform branch
                        if (kIntCells<ACC>) {
                            const unsigned long long tot = wave_sum(in ? rn_scaled_u64(cf, Q.scale) : 0ull);
                            if ((int)lane_id() == lead && tot)
                                atomicAdd(reinterpret_cast<unsigned long long *>(tile + cell0 + b * W + a), tot);
                        } else if (ACC == kFixed64) {
                            const unsigned long long tot = wave_sum(in ? rn_scaled_u64(cf, P.fixed_scale) : 0ull);
                            if ((int)lane_id() == lead && tot)
                                atomicAdd(reinterpret_cast<unsigned long long *>(tile + cell0 + b * W + a), tot);
                        } else {
                            const double tot = wave_sum(in ? (double)cf : 0.0);
                            if ((int)lane_id() == lead && tot != 0.0)
                                atomicAdd(reinterpret_cast<double *>(tile + cell0 + b * W + a), tot);
                        }
                    } else if (in) {
                        lds_add<ACC>(tile + cell + b * W + a, cf, P, Q);
                    }
                }
            }
        }
    }
}

template <int MAS, int ACC, bool POW2, bool HAS_MASS>
__device__ __forceinline__ void tile_accumulate_merged(const PendingList &L, const PassParams &P,
                                                       typename AccT<ACC>::lds *tile, const RunTable &R, int x0, int y0,
                                                       int W, typename AccT<ACC>::type *gmap, const TileQuantum &Q)
{
    using lds_t = typename AccT<ACC>::lds;
    const int tid = threadIdx.x;
    const int nn = P.nn;
    for (int kk = 0; kk < R.n; kk++) {
        const RunRef rr = R.run[kk];
        const int c = __builtin_amdgcn_readfirstlane((int)rr.chunk);
        const unsigned start = (unsigned)__builtin_amdgcn_readfirstlane((int)rr.start);
        const unsigned end = start + (unsigned)__builtin_amdgcn_readfirstlane((int)(R.pre[kk + 1] - R.pre[kk]));
        const float2 *__restrict__ sxy = L.sxy[c];
        // every lane of a wave runs every iteration of its wave (no divergent exit: the butterfly needs all lanes)
        for (unsigned i0 = start; i0 < end; i0 += kTileBlock) {
            const unsigned i = i0 + tid;
            const bool act = i < end;
            float2 r = make_float2(0.f, 0.f);
            float m = L.mconst[c], sq = L.sm_const[c];
            if (HAS_MASS) {
                Rec3 v{0.f, 0.f, 0.f};
                if (act)
                    v = reinterpret_cast<const Rec3 *>(sxy)[i];
                r = make_float2(v.x, v.y);
                m = cap_mass(v.m);
                sq = __fsqrt_rn(m);
            } else if (act) {
                r = sxy[i];
            }
            const int gx = grid_index<POW2>(r.x, P), gy = grid_index<POW2>(r.y, P);
            const int cell = act ? (gy - y0) * W + (gx - x0) : -1;  // cell (gx - 1, gy - 1) of the halo'd tile
            const unsigned long long am = __ballot(act);
            if (am == 0ull)
                continue;
            const int lead = (int)__builtin_ctzll(am);
            const int cell0 = __shfl(cell, lead);
            const bool uniform = __ballot(act && cell != cell0) == 0ull;
            if (MAS == kNGP) {
                if (uniform) {
                    if ((int)lane_id() == lead) {
                        lds_t *cc = tile + cell0 + W + 1;
                        if (ACC == kCountU32)
                            atomicAdd(reinterpret_cast<unsigned *>(cc), (unsigned)__popcll(am));
                    }
                    if (ACC != kCountU32) {
                        const double tot = wave_sum(act ? (double)m : 0.0);
                        if ((int)lane_id() == lead)
                            atomicAdd(reinterpret_cast<double *>(tile + cell0 + W + 1), tot);
                    }
                } else if (act) {
                    lds_t *cc = tile + cell + W + 1;
                    if (ACC == kCountU32)
                        atomicAdd(reinterpret_cast<unsigned *>(cc), 1u);
                    else
                        atomicAdd(reinterpret_cast<double *>(cc), (double)m);
                }
                continue;
            }
            float wx[3], wy[3];
            tsc_axis<POW2>(r.x, gx, P, wx);
            tsc_axis<POW2>(r.y, gy, P, wy);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                wx[a] = sq * wx[a];
                wy[a] = sq * wy[a];
            }
#pragma unroll
            for (int b = 0; b < 3; b++) {
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    // map-edge tests as in the CHECK variant (heavy border tiles are rare enough not to specialise)
                    const int px = gx + a - 1, py = gy + b - 1;
                    bool in = act && px >= 0 && px < nn && py >= 0 && py < nn;
                    const float cf = wx[a] * wy[b];
                    if (kIntCells<ACC>) {  // contributions that are no multiple of the tile's quantum bypass the tile
                        const double t = (double)cf * Q.scale;
                        if (in && t != rint(t)) {
                            atomicAdd(gmap + (size_t)px + (size_t)nn * (size_t)py, (typename AccT<ACC>::type)cf);
                            in = false;
                        }
                    }
                    if (uniform) {  // wave-uniform branch
                        if (kIntCells<ACC>) {
                            const unsigned long long tot = wave_sum(in ? rn_scaled_u64(cf, Q.scale) : 0ull);
                            if ((int)lane_id() == lead && tot)
                                atomicAdd(reinterpret_cast<unsigned long long *>(tile + cell0 + b * W + a), tot);
                        } else if (ACC == kFixed64) {
                            const unsigned long long tot = wave_sum(in ? rn_scaled_u64(cf, P.fixed_scale) : 0ull);
                            if ((int)lane_id() == lead && tot)
                                atomicAdd(reinterpret_cast<unsigned long long *>(tile + cell0 + b * W + a), tot);
                        } else {
                            const double tot = wave_sum(in ? (double)cf : 0.0);
                            if ((int)lane_id() == lead && tot != 0.0)
                                atomicAdd(reinterpret_cast<double *>(tile + cell0 + b * W + a), tot);
                        }
                    } else if (in) {
                        lds_add<ACC>(tile + cell + b * W + a, cf, P, Q);
                    }
                }
            }
        }
    }
}

// Visit every cell of a W x H LDS tile (W = 64 k + 2): a wave per row with its lanes along the row for the first W - 2
// columns, then the two halo columns on the right as one dense range -- no division by the run-time row length and no
// nearly empty trip for the two cells beyond a multiple of 64.
template <typename Fn>
__device__ __forceinline__ void for_each_tile_cell(int W, int H, Fn &&fn)
{
    const int tid = threadIdx.x;
    for (int row = tid >> 6; row < H; row += kTileBlock / 64)
        for (int col = tid & 63; col < W - 2; col += 64)
            fn(row, col);
    for (int i = tid; i < 2 * H; i += kTileBlock)
        fn(i >> 1, W - 2 + (i & 1));
}

// RUNS: the pending chunks come from the two-level sort (a table of runs per tile, walked wave by wave); otherwise one
// run per chunk (L.base), walked by the whole workgroup.
// Registers: the TSC variants need ~54 and run two workgroups per CU.  The NGP count kernel keeps 16 pixel values per
// lane in registers for its in-tile fold; without the species' own map (HAS_MASS slot = false) it is held to 64 registers
// (28 bytes of scratch, touched at the file boundaries only) for the second workgroup per CU: tile kernel 615 -> 490 us,
// --mas ngp 2.05 -> 1.92 ms per snapshot (A/B in one call, profiles/r03_k4_stage_costs.log).  With the second map (16
// more values) the same limit spills 120 bytes and doubles the kernel's time (755 -> 1470 us): that variant stays at
// one workgroup per CU.
#ifndef SLICER_K4_NGP_WAVES
#define SLICER_K4_NGP_WAVES 8
#endif
template <int MAS, int ACC, bool POW2, bool HAS_MASS, bool RUNS>
__global__ __launch_bounds__(kTileBlock, (MAS == kNGP && ACC == kCountU32 && !HAS_MASS) ? SLICER_K4_NGP_WAVES : 4) void
k_tile_deposit(PendingList L, PassParams P, BinGeom G, Targets T, TileItems I, NgpFold F)
{
    using acc_t = typename AccT<ACC>::type;
    using lds_t = typename AccT<ACC>::lds;
    // 16-byte aligned by declaration: ds_add_u64 / ds_add_f64 on a cell that is only 4-byte aligned FAULTS (round 2: a
    // static __shared__ array in front of an unaligned dynamic array did exactly that).  The attribute makes the
    // compiler pad whatever static LDS precedes the dynamic segment; the small tables of this kernel live behind the tile.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_t *tile = reinterpret_cast<lds_t *>(smem_raw);
    static_assert(alignof(lds_t) <= 16 && sizeof(lds_t) <= 8, "tile cells are 4- or 8-byte scalars");

    unsigned bin = blockIdx.x, part = 0;
    if (blockIdx.x >= (unsigned)G.nbins) {
        const unsigned j = blockIdx.x - (unsigned)G.nbins;
        if (j >= *I.n_extra)
            return;
        const uint2 item = I.extra[j];
        bin = item.x, part = item.y;
    }
    const unsigned nparts = I.nparts[bin];
    if (nparts == 0)
        return;
    const int unit = bin / G.tiles_per_unit;
    const int t = bin % G.tiles_per_unit;
    const int plane = unit / G.units_per_plane;
    const int band = unit % G.units_per_plane;
    const int x0 = (t % G.ntx) << G.tw_log2;
    const int y0 = (band * G.rows_per_unit + t / G.ntx) << G.th_log2;
    const int W = (1 << G.tw_log2) + 2, H = (1 << G.th_log2) + 2;
    const int cells = W * H;
    const int tid = threadIdx.x;
    const int nn = P.nn;

    // behind the tile (no static __shared__ in this kernel: it would sit in front of the dynamic array and leave the
    // 8-byte cells 4-byte aligned -- misaligned 64-bit LDS atomics fault): a counter, then the list of noted records
    unsigned char *behind = smem_raw + ((sizeof(lds_t) * (size_t)cells + 7) & ~(size_t)7);
    unsigned &s_nslow = *reinterpret_cast<unsigned *>(behind);
    uint2 *s_slow = reinterpret_cast<uint2 *>(behind) + 1;  // [kSlowCap] {chunk, record}: integer-cell modes only
    RunTable R = run_table_at(behind + 8 + (kIntCells<ACC> ? kSlowCap * sizeof(uint2) : 0));
    acc_t *gmap = reinterpret_cast<acc_t *>(T.acc[plane]);
    TileQuantum Q{P.tile_scale, P.tile_inv_scale, P.tile_cmin};
    if (kIntCells<ACC> && HAS_MASS) {
        // per-particle masses: the scale follows the largest selected mass of the species (bits of a non-negative f32)
        const unsigned mm = *T.max_mass;
        const int le = mm ? (int)((mm >> 23) & 0xFFu) - 126 : 0;  // 2^le > m for every m <= that maximum
        Q.scale = __builtin_ldexp(1.0, 49 - le);
        Q.inv_scale = __builtin_ldexp(1.0, le - 49);
        Q.cmin = __builtin_ldexpf(1.0f, le - 25);
    }
    for (int i = tid; i < cells; i += kTileBlock)
        tile[i] = (lds_t)0;
    if (tid == 0)
        s_nslow = 0;
    if (RUNS)
        build_run_table(L, G, bin, part, nparts, R);  // (ends in a barrier: tile zeroed, table complete)
    else
        __syncthreads();

    // the halo [x0 - 1, x0 + W - 2] x [y0 - 1, y0 + H - 2] inside the map: no cell of this tile needs the edge test
    const bool interior = x0 >= 1 && y0 >= 1 && x0 + W - 2 < nn && y0 + H - 2 < nn;
    if (MAS == kNGP && ACC == kCountU32) {
        // NGP counts: one sub-file after the other (chunks of a file are neighbours in the list).  A file marked `fold`
        // is folded into the f32 maps right here -- this workgroup is the only one that touches these pixels in this
        // launch (NGP records hit cells of their own tile only, and no tile is split when F.on), so the pixel values
        // travel in registers from the first file to the last: lane `tid` owns the tile's cells tid, tid + 1024, ...
        // Any other file's counts go to the global count map.
        constexpr int CPT = 16;  // 128 x 128 cells / 1024 lanes
        // (the NGP-count instantiations have no per-particle masses; their HAS_MASS flag says instead whether the
        // species' own map is kept next to the all-types map: 16 more registers per lane, one workgroup per CU less)
        constexpr bool TYPE_MAP = HAS_MASS;
        float rt[CPT], ri[CPT];
        float *tot = F.tot[plane], *toti = F.toti[plane];
        const int tw = 1 << G.tw_log2, ncell = tw << G.th_log2;
        // lane's j-th cell: tile cell i = j * 1024 + tid, i.e. LDS cell cell0 + j * cstride, pixel idx0 + j * pstride
        const int row0 = tid >> G.tw_log2, col0 = tid & (tw - 1);
        const int rows_per_j = kTileBlock >> G.tw_log2;  // (tile widths are <= 1024)
        const int cell0 = (row0 + 1) * W + col0 + 1, cstride = rows_per_j * W;
        const size_t idx0 = (size_t)(x0 + col0) + (size_t)nn * (size_t)(y0 + row0), pstride = (size_t)nn * (size_t)rows_per_j;
        unsigned inmask = 0;  // bit j: that cell exists and lies inside the map
#pragma unroll
        for (int j = 0; j < CPT; j++)
            if (j * kTileBlock + tid < ncell && x0 + col0 < nn && y0 + row0 + j * rows_per_j < nn)
                inmask |= 1u << j;
        if (F.on) {
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const bool in = inmask >> j & 1u;
                rt[j] = in ? tot[idx0 + j * pstride] : 0.0f;
                ri[j] = (TYPE_MAP && in && toti) ? toti[idx0 + j * pstride] : 0.0f;
            }
        }
        unsigned touched = 0;
        // one sub-file after the other: the runs of its chunks are deposited (every wave walks its share), then the
        // file's counts are folded, or flushed to the count map
        auto boundary = [&](int c) {
            __syncthreads();
            if (F.on && L.fold[c]) {
                const float m = L.mconst[c];
                unsigned kk[CPT];
#pragma unroll
                for (int j = 0; j < CPT; j++)  // (all LDS reads first: one latency)
                    kk[j] = (inmask >> j & 1u) ? (unsigned)tile[cell0 + j * cstride] : 0u;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    if (kk[j] == 0)
                        continue;
                    tile[cell0 + j * cstride] = (lds_t)0;
                    const float v = ngp_seq_sum(kk[j], m);
                    rt[j] = rt[j] + v;  // tot += mapxyi, toti += mapxyi   densitymaps.cpp:511-513
                    if (TYPE_MAP)
                        ri[j] = ri[j] + v;
                    touched |= 1u << j;
                }
            } else {
                auto flush = [&](int row, int col) {
                    const unsigned k = (unsigned)tile[row * W + col];
                    if (k == 0)
                        return;
                    tile[row * W + col] = (lds_t)0;
                    atomicAdd(gmap + (size_t)(x0 - 1 + col) + (size_t)nn * (size_t)(y0 - 1 + row), (acc_t)k);
                };
                for_each_tile_cell(W, H, flush);
            }
            __syncthreads();
        };
        if (RUNS) {
            for (int c0 = 0; c0 < L.n;) {
                int c1 = c0 + 1;
                while (c1 < L.n && L.file_id[c1] == L.file_id[c0])
                    c1++;
                tile_accumulate<MAS, ACC, POW2, false, false>(P, tile, R, L.run0[c0], L.run0[c1], x0, y0, W, &s_nslow,
                                                              s_slow, gmap, Q);
                boundary(c0);
                c0 = c1;
            }
        } else {
            // one walk over all pending chunks; when it leaves the last chunk of a sub-file (the next round's records
            // are already on their way) the file's counts are folded, or flushed to the count map
            auto leave = [&](int c, int c_next) {
                if (c_next < L.n && L.file_id[c_next] == L.file_id[c])
                    return;
                boundary(c);
            };
            tile_accumulate_chunks<MAS, ACC, POW2, false, false>(L, P, tile, bin, part, nparts, x0, y0, W, &s_nslow, s_slow,
                                                                 gmap, Q, 0, L.n, leave);
        }
#pragma unroll
        for (int j = 0; j < CPT; j++)
            if (touched >> j & 1u) {
                tot[idx0 + j * pstride] = rt[j];
                if (TYPE_MAP && toti)
                    toti[idx0 + j * pstride] = ri[j];
            }
        return;
    }
    // pre-reduction only for bins far beyond a tile's usual load (>= 8 parts = 131072 records: a halo core); a bin that
    // is merely split in two or three is faster through the plain loop (--clustered: 810 us with, 700 us without)
    if (RUNS) {
        if (nparts >= kMergeParts)
            tile_accumulate_merged<MAS, ACC, POW2, HAS_MASS>(L, P, tile, R, x0, y0, W, gmap, Q);
        else if (MAS == kNGP || interior)
            tile_accumulate<MAS, ACC, POW2, HAS_MASS, false>(P, tile, R, 0, R.n, x0, y0, W, &s_nslow, s_slow, gmap, Q);
        else
            tile_accumulate<MAS, ACC, POW2, HAS_MASS, true>(P, tile, R, 0, R.n, x0, y0, W, &s_nslow, s_slow, gmap, Q);
    } else {
        if (nparts >= kMergeParts)
            tile_accumulate_merged_chunks<MAS, ACC, POW2, HAS_MASS>(L, P, tile, bin, part, nparts, x0, y0, W, gmap, Q);
        else if (MAS == kNGP || interior)
            tile_accumulate_chunks<MAS, ACC, POW2, HAS_MASS, false>(L, P, tile, bin, part, nparts, x0, y0, W, &s_nslow,
                                                                    s_slow, gmap, Q, 0, L.n);
        else
            tile_accumulate_chunks<MAS, ACC, POW2, HAS_MASS, true>(L, P, tile, bin, part, nparts, x0, y0, W, &s_nslow,
                                                                   s_slow, gmap, Q, 0, L.n);
    }
    __syncthreads();
    if (kIntCells<ACC>) {
        // the records noted in the loop: those of their contributions that are exact multiples of the quantum go into
        // the tile like all others, the vanishing ones straight to the global map
        const unsigned ns = s_nslow < kSlowCap ? s_nslow : kSlowCap;
        for (unsigned e = tid; e < ns; e += kTileBlock) {
            const uint2 w = s_slow[e];
            if (HAS_MASS) {
                const Rec3 r = reinterpret_cast<const Rec3 *>(L.sxy[w.x])[w.y];
                slow_record<ACC, POW2>(r.x, r.y, __fsqrt_rn(cap_mass(r.m)), P, Q, tile, gmap, x0, y0, W);
            } else {
                const float2 r = L.sxy[w.x][w.y];
                slow_record<ACC, POW2>(r.x, r.y, L.sm_const[w.x], P, Q, tile, gmap, x0, y0, W);
            }
        }
        __syncthreads();
    }

    // flush: consecutive lanes -> consecutive pixels of one map row (shaped atomics).  (Round 3 measured the alternative
    // for the cells only this workgroup adds to -- plain load + add + store, stores running at ~6 TB/s against ~1.3 TB/s
    // of added bytes for memory-side float atomics: tile kernel 471 -> 873 us.  The atomics are fire-and-forget, the
    // read-modify-write puts an HBM round trip per cell row on the flushing wave.)
    auto flush = [&](int row, int col) {
        const lds_t v = tile[row * W + col];
        const int px = x0 - 1 + col, py = y0 - 1 + row;
        if (v == (lds_t)0 || px < 0 || px >= nn || py < 0 || py >= nn)
            return;
        acc_t *cell = gmap + (size_t)px + (size_t)nn * (size_t)py;
        if (kIntCells<ACC>)  // exact tile sum -> one rounding to the accumulator type
            atomicAdd(cell, (acc_t)((double)v * Q.inv_scale));
        else
            atomicAdd(cell, (acc_t)v);
    };
    for_each_tile_cell(W, H, flush);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_bin_scan(const LaunchCfg &cfg, int nblocks, int n_planes, const BinGeom &G, const BinWorkspace &W,
                           const Targets &T, hipStream_t s)
{
    (void)cfg, (void)n_planes, (void)T;
    const int ngroups = (G.nbins + kScanBins - 1) / kScanBins;
    k_scan_blocks<<<ngroups, 1024, 0, s>>>(reinterpret_cast<const unsigned short *>(W.hist16), W.hist, W.total,
                                           W.total + kMaxBins, nblocks, G.nbins);
    return hipGetLastError();
}

size_t scatter_lds_bytes(const BinGeom &G, bool has_mass)
{
    const size_t tpp = (size_t)G.tiles_per_unit, tw = (tpp + 1) >> 1;
    return 4 * (tw + 2 * tpp + (tw & 1)) + (size_t)kSortStage * (8 + 2 + (has_mass ? 4 : 0));
}

hipError_t launch_bin_scatter(const LaunchCfg &cfg, int nblocks, int n_planes, int max_workgroups, const BinGeom &G,
                              const BinWorkspace &W, const Targets &T, hipStream_t s)
{
    const bool has_mass = cfg.has_mass;
    const size_t lds = scatter_lds_bytes(G, has_mass);
    const int items = G.n_units * 8 * ((nblocks + 7) / 8);
    const int nwg = std::min(items, std::max(8, max_workgroups / 8 * 8));
    const int count_planes = n_planes;  // records per plane -> selected-entry counters (NGP adds its dropped ones in K1)
    hipError_t e;
    if (has_mass) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_scatter<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        k_bin_scatter<true><<<nwg, kSortBlock, lds, s>>>(W.cxy, W.cbin, W.cm, W.hist16, W.hist, W.total, W.total + kMaxBins,
                                                          W.base, W.bcount, nblocks, G, W.sxy, W.sm, count_planes, T);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_scatter<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        k_bin_scatter<false><<<nwg, kSortBlock, lds, s>>>(W.cxy, W.cbin, W.cm, W.hist16, W.hist, W.total, W.total + kMaxBins,
                                                           W.base, W.bcount, nblocks, G, W.sxy, W.sm, count_planes, T);
    }
    return hipGetLastError();
}

size_t sort2_lds_bytes(int slots_per_group, int tiles_per_unit)
{
    return (size_t)kS2Cap * 9 + 4 * ((size_t)slots_per_group * 2 + 1 + 4 * (size_t)tiles_per_unit) + 2 * (kS2Cap / 64);
}

hipError_t launch_sort2(int nblocks, int slots_per_group, int ngroups, int max_workgroups, const PassParams &P,
                        const BinGeom &G, const BinWorkspace &W, hipStream_t s)
{
    const size_t lds = sort2_lds_bytes(slots_per_group, G.tiles_per_unit);
    const int nitems = G.n_units * ngroups;
    const int nwg = std::min(nitems, std::max(std::max(8, max_workgroups), (nitems + kS2MaxMine - 1) / kS2MaxMine));
    hipError_t e;
#define S2(P2_)                                                                                                       \
    do {                                                                                                              \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort2<P2_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                            \
        if (e != hipSuccess)                                                                                          \
            return e;                                                                                                 \
        k_sort2<P2_><<<nwg, kS2Block, lds, s>>>(W.c1, W.sb_off, W.sb_start, W.sb_n, nblocks, slots_per_group, ngroups, G, P, \
                                                W.sxy, W.ptab, W.item_tot, W.tot);                                      \
    } while (0)
    if (P.pow2)
        S2(true);
    else
        S2(false);
#undef S2
    return hipGetLastError();
}

size_t tile_lds_bytes(const BinGeom &G, int acc, bool runs)
{
    const size_t elem = acc == kCountU32 ? 4 : 8;
    const size_t cells = (size_t)((1 << G.tw_log2) + 2) * (size_t)((1 << G.th_log2) + 2);
    return ((elem * cells + 7) & ~(size_t)7) + 8 + ((acc == kF32I || acc == kF64I) ? kSlowCap * sizeof(uint2) : 0) +
           (runs ? kRunTableBytes : 0);  // (the run table of the two-level sort's walk)
}

template <int MAS, int ACC>
static hipError_t launch_k4(bool pow2, bool has_mass, const PassParams &P, const BinGeom &G, const PendingList &L,
                            const Targets &T, const TileItems &I, const NgpFold &F, unsigned max_items, hipStream_t s)
{
    const size_t lds = tile_lds_bytes(G, ACC, L.tot != nullptr);
#define K4_(P2_, HM_, RN_)                                                                                       \
    do {                                                                                                         \
        auto kern = k_tile_deposit<MAS, ACC, P2_, HM_, RN_>;                                                     \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            if (e != hipSuccess)                                                                                 \
                return e;                                                                                        \
        }                                                                                                        \
        kern<<<max_items, kTileBlock, lds, s>>>(L, P, G, T, I, F);                                      \
    } while (0)
#define K4(P2_, HM_)                                                                                             \
    do {                                                                                                         \
        if (L.tot)                                                                                               \
            K4_(P2_, HM_, true);                                                                                 \
        else                                                                                                     \
            K4_(P2_, HM_, false);                                                                                \
    } while (0)
    if (pow2) {
        if (has_mass) K4(true, true); else K4(true, false);
    } else {
        if (has_mass) K4(false, true); else K4(false, false);
    }
#undef K4
#undef K4_
    return hipGetLastError();
}

size_t tile_items_bytes(const BinGeom &G, uint64_t total_particles)
{
    const uint64_t max_extra = total_particles / kItemRecs + 1;
    return 16 + (size_t)(G.nbins + (G.nbins & 1)) * 4 + max_extra * sizeof(uint2);  // counters | nparts | extra
}

hipError_t launch_tile_deposit(const LaunchCfg &cfg, const PassParams &P, const BinGeom &G, const PendingList &L,
                               const Targets &T, const NgpFold &F, void *items_ws, unsigned epoch,
                               uint64_t total_particles, int int_mode, bool *int_cells_used, hipStream_t s)
{
    *int_cells_used = false;
    // total_particles bounds the number of records (each particle emits at most one on this path).  Workspace:
    // two counters (used alternately: launch `epoch` reads [epoch & 1] and zeroes the other one) | nparts | extra
    const unsigned max_items = (unsigned)((uint64_t)G.nbins + total_particles / kItemRecs + 1);
    TileItems I;
    unsigned *counters = reinterpret_cast<unsigned *>(items_ws);
    I.n_extra = counters + (epoch & 1u);
    I.next_n_extra = counters + ((epoch + 1u) & 1u);
    I.nparts = counters + 4;
    I.extra = reinterpret_cast<uint2 *>(I.nparts + G.nbins + (G.nbins & 1));
    k_build_items<<<(G.nbins + 255) / 256, 256, 0, s>>>(L, G.nbins, I, (cfg.mas == kNGP && cfg.acc == kCountU32 && F.on) ? 1 : 0);
    const bool pow2 = P.pow2 != 0;
    if (cfg.mas == kNGP) {
        if (cfg.acc == kCountU32) {
            // (has_mass slot of the count kernels: keep the species' own map in the in-tile fold)
            return launch_k4<kNGP, kCountU32>(pow2, F.on && F.toti[0] != nullptr, P, G, L, T, I, F, max_items, s);
        }
        return launch_k4<kNGP, kF32>(pow2, cfg.has_mass, P, G, L, T, I, F, max_items, s);
    }
    // constant-mass TSC in the F32 / F64 modes: integer tile cells (int_mode, the handle's option k4_int: 0 keeps the
    // f64 cells, 2 forces the integer ones).  They pay where the records dominate (2048^2 x 4 planes, 65536 particles per bin: 370 against 622 us);
    // a launch with few records per tile is mostly tile zeroing and flushing, where the u64 -> float conversion of every
    // cell costs what the cheaper LDS atomic saves (8192^2 x 4 planes, 4096 per bin: 1242 against 1205 us; 2048 per bin:
    // equal) -- below 2048 particles per bin the f64 cells stay.
    const bool int_cells = int_mode == 2 || (int_mode == 1 && total_particles / (uint64_t)G.nbins >= 2048);
    if (int_cells && (cfg.acc == kF32 || cfg.acc == kF64)) {
        if (cfg.has_mass) {  // the quantum follows the largest mass the sort kernel saw (TileQuantum)
            *int_cells_used = true;
            if (cfg.acc == kF32)
                return launch_k4<kTSC, kF32I>(pow2, true, P, G, L, T, I, F, max_items, s);
            return launch_k4<kTSC, kF64I>(pow2, true, P, G, L, T, I, F, max_items, s);
        }
        bool same_mass = true;  // one quantum per launch: all pending chunks carry the same constant mass
        for (int c = 1; c < L.n; c++)
            same_mass = same_mass && L.mconst[c] == L.mconst[0];
        if (same_mass && L.mconst[0] == P.mconst) {
            *int_cells_used = true;
            if (cfg.acc == kF32)
                return launch_k4<kTSC, kF32I>(pow2, false, P, G, L, T, I, F, max_items, s);
            return launch_k4<kTSC, kF64I>(pow2, false, P, G, L, T, I, F, max_items, s);
        }
    }
    switch (cfg.acc) {
    case kF32: return launch_k4<kTSC, kF32>(pow2, cfg.has_mass, P, G, L, T, I, F, max_items, s);
    case kF64: return launch_k4<kTSC, kF64>(pow2, cfg.has_mass, P, G, L, T, I, F, max_items, s);
    case kFixed64: return launch_k4<kTSC, kFixed64>(pow2, cfg.has_mass, P, G, L, T, I, F, max_items, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace slicer
