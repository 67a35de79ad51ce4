// slicer_project_bin.hip -- K1 of SLICER_ALGO_BINNED: stream raw POS, transform, slab select, project, emit records.
//
// Replaces the CPU loops of gadget2io.cpp:195-274 (readPos) and densitymaps.cpp:355-401 (mapParticles' selection and
// getPolar projection).  Every wave runs on its own: it streams 256 particles per round (4 per lane, three dwordx4
// loads, the next round's loads issued before the current round is processed), transforms them, and pushes the
// survivors of the slab test and of a conservative f32 FOV pre-test onto a wave-private LDS stack (positions from
// ballot + popcount, no atomics).  Whenever the stack holds >= 64 entries the wave pops 64 and runs the fp64 projection
// on a full wave.  There is no workgroup barrier in the loop; waves only share the histogram and the output cursors.
//
// Two variants of the kernel:
//   k_project_bin_fast     the common case (<= 4 planes per pass, no lateral replication, any map size, box size and
//                          random centre that passed the host's exactness checks).  The kernel is bound by VALU issue, so this
//                          variant is written for instruction count:
//                            * r/box as a correctly rounded f32 division (reciprocal + two FMAs) instead of an fp64
//                              product with a rounding-tie test -- valid for a box size iff an exhaustive device sweep
//                              over all 2^31 non-negative floats found it equal to (float)((double)r/box)
//                              (k_check_box_quotient, cached per handle);
//                            * the recentring (float)((double)v - x0) as the f32 subtraction it equals when x0 is an
//                              f32 value (Random.x0 = rand()/float(RAND_MAX) is one, densitymaps.cpp:188-190): double
//                              rounding through binary64 is innocuous for +,- when 53 >= 2*24+2;
//                            * both periodic wraps reduced to the one branch that can fire for in-box input, with one
//                              integer range test per particle sending anything else (r outside [0, box], -0.0, NaN)
//                              to the general code;
//                            * an fp64 projection that is accurate to ~2^-44 instead of correctly rounded (one Newton
//                              step on v_rsq_f64 / v_rcp_f64), with every result whose rounding to f32 -- or whose FOV
//                              decision -- could differ from the exact one sent to the exact code (p ~ 1e-5 per entry);
//                            * a 40-dword kernel-argument block instead of the 80-dword PassParams (no SGPR spills);
//                              the rare exact paths read PassParams straight from the kernarg segment.
//   k_project_bin_general  everything else: up to 8 planes, lateral replication (-DUSE_REPLICATION, densitymaps.cpp:
//                          377-399: up to (2n+1)^2 records per particle), any box size / centre.  Round 1's kernel.
// Both produce bit-identical records; tests/test_gpu_parity.py runs every parity case through both
// (SLICER_K1_GENERAL=1 forces the general one).
#include "slicer_binned_common.hpp"

#pragma clang fp contract(off)

namespace slicer {

namespace {

// 512 threads at 4 waves per SIMD: two workgroups (16 waves) per CU and 128 VGPRs per lane.  Round 2 ran 768 threads at
// 6 waves per SIMD (80 VGPRs); round 3 measured the same speed at the lower occupancy (108 vs 109 us: the kernel does not
// live off waves in flight) and spends the registers on two particles per lane in the projection (project_emit2).
#ifndef SLICER_K1_BLOCK
#define SLICER_K1_BLOCK 512
#endif
#ifndef SLICER_K1_WAVES_PER_SIMD
#define SLICER_K1_WAVES_PER_SIMD 4
#endif
constexpr int kK1Block = SLICER_K1_BLOCK;  // 8 waves, two workgroups per CU at 32768 particles each
#ifndef SLICER_K1_PER_THREAD
#define SLICER_K1_PER_THREAD 4
#endif
constexpr int kPerThread = SLICER_K1_PER_THREAD;  // particles per lane and round: 4 (three dwordx4 loads) or 2 (three dwordx2)
static_assert(kPerThread == 4 || kPerThread == 2, "load_round handles 2 or 4 particles per lane");
constexpr int kRound = kK1Block * kPerThread;
constexpr int kWaves = kK1Block / 64;
constexpr int kWaveQ = 64 * kPerThread + 64;  // stack capacity per wave: one round + a remainder < 64

// One lane's particles of a round as they lie in the POS block: raw[3 k + a] = coordinate a of its particle k.
// Loaded by buffer loads through a descriptor of the WORKGROUP'S batch (base = its first particle, size = its bytes):
// the hardware range check returns 0 for every dword beyond the batch, so the ragged end of a chunk needs no second
// load path -- and with a single path there is no merge point at which the compiler reconciles register layouts by
// moving freshly loaded registers around.  It did, with a vector and a scalar path: a v_mov right behind the loads
// means waiting for the next round's positions the moment they have been requested, and the "prefetch" overlapped
// nothing (round 3; the rounds of a wave then cost a full memory latency each).
struct RawRound {
    float v[3 * kPerThread];
};
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t batch_rsrc(const float *pos, uint64_t b0, uint64_t b1)
{
    // raw buffer (stride 0): num_records in bytes; word 3 as for gfx90a / gfx94x / gfx950 raw 32-bit data
    return __builtin_amdgcn_make_buffer_rsrc((void *)(pos + 3 * b0), 0, (int)((b1 - b0) * 12), 0x00020000);
}

// particles [i, i + kPerThread) of the batch (i relative to the batch's first particle)
__device__ __forceinline__ void load_round(__amdgpu_buffer_rsrc_t rsrc, unsigned i, RawRound &R)
{
    const unsigned off = i * 12u;
    if constexpr (kPerThread == 4) {
        const v4u_t a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        const v4u_t b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
        const v4u_t c = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0);
        R.v[0] = __uint_as_float(a.x); R.v[1] = __uint_as_float(a.y); R.v[2] = __uint_as_float(a.z);
        R.v[3] = __uint_as_float(a.w); R.v[4] = __uint_as_float(b.x); R.v[5] = __uint_as_float(b.y);
        R.v[6] = __uint_as_float(b.z); R.v[7] = __uint_as_float(b.w); R.v[8] = __uint_as_float(c.x);
        R.v[9] = __uint_as_float(c.y); R.v[10] = __uint_as_float(c.z); R.v[11] = __uint_as_float(c.w);
    } else {
        const v2u_t a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
        const v2u_t b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 8, 0, 0);
        const v2u_t c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16, 0, 0);
        R.v[0] = __uint_as_float(a.x); R.v[1] = __uint_as_float(a.y); R.v[2] = __uint_as_float(b.x);
        R.v[3] = __uint_as_float(b.y); R.v[4] = __uint_as_float(c.x); R.v[5] = __uint_as_float(c.y);
    }
}

// The one kernel argument of k_project_bin_fast: the wide PassParams (read by the exact epilogue only, through the
// kernarg segment in memory, so that its 80 dwords never occupy SGPRs in the hot loop) and the lean block of the loop.
struct K1Kernarg {
    PassParams P;
    K1Args A;
};

// Pointer to that argument in the kernarg segment.  Must be called in the kernel itself (in a callee the builtin folds
// to null) and handed to out-of-line functions as an argument -- by pointer, never by reference to the by-value kernel
// parameter, which would drag the whole block into scratch memory.
__device__ __forceinline__ const K1Kernarg *kernarg_block()
{
    return (const K1Kernarg *)__builtin_amdgcn_kernarg_segment_ptr();
}

// Domain of the f32 quotient q = r / box on the fast path: +0 <= q <= 1 (positive floats order like their bit patterns;
// -0.0, negatives and NaN have the top bit set or exceed 0x3F800000).  The exhaustive sweep vouches for 2^-100 <= q <= 1:
// below that the residual FMA of the two-step division runs into the subnormal range and q may be off in its last bit --
// which cannot matter, because such a q (< 1e-30) only enters the results through fma(q, +-1, 0 | 1) - c with an f32
// centre c >= 2^-20 (k1_fast_args), whose rounding it cannot reach.
constexpr unsigned kQLo = (127u - 100u) << 23, kQHi = 0x3F800000u;
__device__ __forceinline__ bool k1_quotient_in_sweep_range(float q)
{
    return __float_as_uint(q) - kQLo <= kQHi - kQLo;
}

// (float)(ang / fov + 0.5) from s = ang * RN(1/fov) + 0.5 known to within `ds`: the f32 the exact value rounds to, if
// both ends of [s - ds, s + ds] round to it; otherwise the entry is undecided.
__device__ __forceinline__ bool round_decided(double s, double ds, float &out)
{
    const float lo = (float)(s - ds), hi = (float)(s + ds);
    out = lo;
    return lo == hi;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// k_project_bin_fast
// ---------------------------------------------------------------------------------------------
// Error budget of the fast projection (validated on the device by tests/test_gpu_parity.py::
// test_fast_projection_error_budget through slicer_debug_math ops 6-9): the refined reciprocal (square root) is good to
// < 2^-48 relative, the products and the series add a few ulp, so angles come out within 2^-47 relative (< 2^-48 rad
// for the |ang| <= 0.32 served here) and the map coordinate s = ang/fov + 0.5 within (|ang|/fov) 2^-47 <= 2^-47.  The
// windows below leave a factor of 64.
constexpr double kAngWindow = 0x1p-41;  // |(|ang| - lim)| below this: FOV decision left to the exact code
constexpr double kMapWindow = 0x1p-41;  // s +- this must round to the same f32

// FACE: Random.face - 1 (the permutation is then a renaming of registers); SERIES: 9 or 15 terms.  The mass-assignment
// scheme (A.ngp) and per-particle masses (A.mass != nullptr) only touch the emit stage and are wave-uniform run-time
// branches.  Power-of-two maps only (grid_index is one multiply + floor there).
//
// No function call and no exact arithmetic inside the loop: a particle the fast code cannot decide (raw coordinate
// outside [0, box], -0.0, NaN; a projection within the error window of an f32 rounding tie or of the FOV limit:
// ~1e-5 of the entries) is only noted -- its index goes to a small LDS list -- and the workgroup reprocesses the noted
// particles with the general, exact code after the loop (process_exact).  If the list overflows (pathological input:
// more than kExcCap undecided particles out of 32768) the workgroup discards what it emitted and runs its whole batch
// through process_exact: slow, but the result never depends on the fast path's domain.
constexpr unsigned kExcCap = 256;

// ---- two-level sort: LDS staging of the workgroup's records (A.sort2) --------------------------------------------------
// Instead of scattering every record to its (unit, workgroup) region with its tile attached, the workgroup collects
// records in LDS, and whenever a round could overflow the staging area it sorts what it holds by unit (coarse bin: a
// band of tile rows of one plane; counting sort with the per-unit rank taken when the record is staged) and writes the
// sub-batch out as whole lines, with a small table of where each unit's run starts.  The sort kernel (k_sort2) gathers
// a unit's runs from the sub-batches of a group of workgroups.  No per-tile histogram, no global prefix matrix.
struct Stage {
    float2 *rec;            // [kStageCap]
    unsigned short *rank;   // [kStageCap] position of the record among its unit's records of this sub-batch
    unsigned short *perm;   // [kStageCap] flush: staged index of the record that goes to sorted position p
    unsigned char *bin;     // [kStageCap] unit
    unsigned *cnt;          // [kMaxCoarse] records per unit since the last flush
    unsigned *start;        // [kMaxCoarse + 2] flush, more than 64 units: exclusive prefix of cnt
    unsigned *tot;          // [kMaxCoarse] records per unit written by this workgroup so far
    unsigned *n;            // staged records
};
constexpr size_t kStageBytes = (size_t)kStageCap * (8 + 2 + 2 + 1) + 4 * (kMaxCoarse + kMaxCoarse + 2 + kMaxCoarse + 2);
static_assert(kStageCap > kRound && kStageCap % 8 == 0 && kStageCap < 65536, "staging holds at least one round; 16-bit ranks");
static_assert(kK1Block >= 4 * 64, "the flush gives its small serial jobs to waves 0..3");
__device__ __forceinline__ Stage stage_of(unsigned *smem)
{
    Stage S;
    S.rec = reinterpret_cast<float2 *>(smem);
    S.rank = reinterpret_cast<unsigned short *>(S.rec + kStageCap);
    S.perm = S.rank + kStageCap;
    S.bin = reinterpret_cast<unsigned char *>(S.perm + kStageCap);
    S.cnt = reinterpret_cast<unsigned *>(S.bin + kStageCap);
    S.start = S.cnt + kMaxCoarse;
    S.tot = S.start + kMaxCoarse + 2;
    S.n = S.tot + kMaxCoarse;
    return S;
}

__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = (unsigned)__shfl_up((int)v, d);
        if ((int)lane_id() >= d)
            v += y;
    }
    return v;
}

// Write the staged records out as one sub-batch if more than `threshold` of them wait.  Every thread of the workgroup
// calls it at the same points; off (records this workgroup has written) and slot (sub-batches) are the same in all.
// A wave that is alone on a piece of serial work runs it at a sixth of a SIMD's issue rate while the other workgroup of
// the CU computes (measured: ~2 us per barrier-separated serial phase and flush), so the flush has as few phases as it
// can: with at most 64 units (the usual pass) every wave scans the unit counts for itself -- no shared prefix table, no
// barrier for it -- and the small jobs (table row, running totals, zeroing) go to different waves.
__device__ __forceinline__ void stage_flush(const K1Args &A, unsigned *smem, unsigned threshold, unsigned &off,
                                            unsigned &slot)
{
    const Stage St = stage_of(smem);
    lds_barrier();  // the appends of this round are in
    const unsigned n = *St.n;
    if (n <= threshold)
        return;
    const int tid = threadIdx.x, wave = tid >> 6;
    const unsigned lane = lane_id();
    const int nu = A.n_units;
    const size_t sg = (size_t)blockIdx.x * kSubBatches + slot;
    unsigned short *row = A.sb_start + sg * (size_t)kSubRow;  // [unit] starts + the end: one or two whole lines
    if (nu <= 64) {
        const unsigned c = (int)lane < nu ? St.cnt[lane] : 0u;
        const unsigned e = wave_inclusive_scan(c) - c;  // start of unit `lane` (lanes >= nu: the total)
        for (unsigned i0 = 0; i0 < n; i0 += kK1Block) {  // (uniform trip count: the shuffle reads lanes of the whole wave;
            const unsigned i = i0 + tid;                   //  a lane that had left the loop would answer 0)
            const bool live = i < n;
            const unsigned b = live ? St.bin[i] : 0u;
            const unsigned at = (unsigned)__shfl((int)e, (int)b);
            if (live)
                St.perm[at + St.rank[i]] = (unsigned short)i;
        }
        if (wave == 1) {  // (nu may be 64: the end of the last run has no lane of its own)
            if ((int)lane < nu)
                row[lane] = (unsigned short)e;
            if (lane == 0)
                row[nu] = (unsigned short)n;
        }
        if (wave == 2 && (int)lane < nu)
            St.tot[lane] += c;  // (only these lanes ever touch tot)
    } else {
        if (tid < 64) {  // exclusive prefix of the unit counts: four units per lane of wave 0 (nu <= 256)
            unsigned v[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int u = tid * 4 + j;
                v[j] = u < nu ? St.cnt[u] : 0u;
                sum += v[j];
            }
            unsigned e = wave_inclusive_scan(sum) - sum;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int u = tid * 4 + j;
                if (u < nu)
                    St.start[u] = e;
                e += v[j];
            }
            if (tid == 0)
                St.start[nu] = n;
        }
        lds_barrier();
        if (tid <= nu)
            row[tid] = (unsigned short)St.start[tid];
        if (tid < nu)
            St.tot[tid] += St.cnt[tid];
        for (unsigned i = tid; i < n; i += kK1Block)
            St.perm[St.start[St.bin[i]] + St.rank[i]] = (unsigned short)i;
    }
    if (tid == 3 * 64)
        A.sb_off[sg] = blockIdx.x * (unsigned)A.batch + off;
    lds_barrier();  // perm complete, the unit counts are no longer read
    if (tid < nu)
        St.cnt[tid] = 0;
    if (tid == 3 * 64)
        *St.n = 0;
    float2 *out = A.c1 + (size_t)blockIdx.x * (size_t)A.batch + off;
    for (unsigned j = tid; j < n; j += kK1Block)
        out[j] = St.rec[St.perm[j]];
    off += n;
    slot++;
    lds_barrier();  // the staging area may be refilled
}

// binning + emission of one selected entry (xs, ys) of `plane`: cursor in (unit, workgroup)'s region, record, histogram
// (s_hist / s_out / s_cnt: the workgroup's histogram, record cursors [kMaxUnits] and NGP selected-entry counters)
// (gx, gy) = cell of (xs, ys), utilities.cpp:69-70
// Emission variants of the fast kernel.  Every run-time flag tested per record costs the wave a scalar branch and, with
// the kernel at its SGPR limit, `v_readlane` reloads of the spilled flag (~8 issue slots per record for the three flags
// of the generic form): the common pass (TSC, constant mass, units = whole planes, one-level sort) has them compiled out.
enum { kEmitGeneric = 0,  // NGP drop rule / band units / per-particle masses / staging, decided at run time
       kEmitLean = 1,     // TSC, constant mass, units = whole planes, one-level sort
       kEmitSort2 = 2,    // two-level sort (staging), the rest at run time
       kEmitLeanNgp = 3,  // as kEmitLean with NGP's drop rule (utilities.cpp:74)
       kEmitLeanMass = 4 };  // as kEmitLean (TSC) with per-particle masses (densitymaps.cpp:358-372)

template <int MODE, bool POW2>
__device__ __forceinline__ void emit_record(const K1Args &A, unsigned *s_hist, unsigned *s_out, unsigned *s_cnt, bool valid,
                                            int plane, float xs, float ys, int gx, int gy, unsigned idx_in_batch,
                                            uint64_t b0, float2 *out_wg, unsigned unit_stride)
{
    constexpr bool LEAN = MODE == kEmitLean || MODE == kEmitLeanNgp || MODE == kEmitLeanMass;
    const bool ngp = MODE == kEmitLeanNgp || (!LEAN && A.ngp != 0);
    const int nn = A.nn;
    bool emit = valid;
    if (ngp) {
        emit = valid && gx >= 0 && gx < nn && gy >= 0 && gy < nn;  // utilities.cpp:74 drop rule
        // The selected-entry counters come from the bin totals (sort kernel's prologue): every emitted record is a
        // selected entry.  NGP drops the off-grid ones (border ring, ~0.5 %) after selection: only those are counted here.
        if (valid && !emit)
            atomicAdd(&s_cnt[plane], 1u);
    }
    // border-ring entries of TSC (g = -1 or nn) still feed the edge pixels: binned with the clamped cell
    gx = min(max(gx, 0), nn - 1);
    gy = min(max(gy, 0), nn - 1);
    if (MODE == kEmitSort2 || (MODE == kEmitGeneric && A.sort2)) {  // two-level sort: stage the record in LDS (s_hist is the staging area here)
        const Stage St = stage_of(s_hist);
        const unsigned coarse = (unsigned)plane * (unsigned)A.units_per_plane + ((unsigned)(gy >> A.th_log2) >> A.crow_log2);
        const unsigned long long em = __ballot(emit);
        if (em != 0ull) {  // one counter add per wave instruction: the emitting lanes take consecutive slots
            const int lead = (int)__builtin_ctzll(em);
            unsigned base = 0;
            if ((int)lane_id() == lead)
                base = atomicAdd(St.n, (unsigned)__popcll(em));
            base = (unsigned)__builtin_amdgcn_readlane((int)base, lead);
            if (emit) {
                const unsigned idx = base + (unsigned)__popcll(em & ((1ull << lane_id()) - 1ull));
                const unsigned r = atomicAdd(&St.cnt[coarse], 1u);
                St.rec[idx] = make_float2(xs, ys);
                St.bin[idx] = (unsigned char)coarse;
                St.rank[idx] = (unsigned short)r;
            }
        }
        return;
    }
    if (emit) {
        const unsigned ty = (unsigned)(gy >> A.th_log2), tx = (unsigned)(gx >> A.tw_log2);
        unsigned band = 0, trow = ty;
        if (!LEAN && A.units_per_plane > 1) {  // large maps: a unit is a band of tile rows
            band = ty / (unsigned)A.rows_per_unit;
            trow = ty - band * (unsigned)A.rows_per_unit;
        }
        const unsigned unit = LEAN ? (unsigned)plane : (unsigned)plane * (unsigned)A.units_per_plane + band;
        // (lean emission on a power-of-two map: tile counts are powers of two, the launcher checks it -- shifts
        // instead of two quarter-rate integer multiplies)
        const bool shifts = LEAN && POW2;
        const unsigned tile_in_unit = shifts ? (trow << A.ntx_log2) | tx : trow * (unsigned)A.ntx + tx;
        const unsigned bin = shifts ? (unit << A.tpu_log2) | tile_in_unit : unit * (unsigned)A.tiles_per_unit + tile_in_unit;
        // one returning LDS add per lane reserves the output slot in (unit, workgroup)'s region
        const unsigned o = atomicAdd(&s_out[unit], 1u);
        const unsigned idx = unit * unit_stride + o;
        out_wg[idx] = make_float2(xs, ys);
        A.cbin[(size_t)blockIdx.x * (size_t)A.batch + idx] = (unsigned short)tile_in_unit;
        if (MODE == kEmitLeanMass || (!LEAN && A.mass != nullptr))
            A.cm[(size_t)blockIdx.x * (size_t)A.batch + idx] = A.mass[b0 + idx_in_batch];
        atomicAdd(&s_hist[bin >> 1], 1u << ((bin & 1u) * 16u));
    }
}

// One particle through the general, exact code (transform of gadget2io.cpp:204-270 operation by operation, correctly
// rounded projection): the epilogue of the fast kernel for the particles it noted.  Returns the negativity flag.
__device__ SLICER_SLOWPATH bool process_exact(const K1Kernarg *Kk, unsigned *s_hist, unsigned *s_out, unsigned *s_cnt,
                                              unsigned idx_in_batch, uint64_t b0, float2 *out_wg, unsigned unit_stride)
{
    const PassParams P = Kk->P;
    const K1Args A = Kk->A;
    const float *r = A.pos + 3 * (b0 + idx_in_batch);
    float x, y, z;
    transform(r[0], r[1], r[2], P, x, y, z);
    const bool neg = (x < 0.0f) | (y < 0.0f) | (z < 0.0f);  // densitymaps.cpp:334
    for (int p = 0; p < P.n_planes; p++) {
        if (!(z >= P.zlo[p] && z < P.zhi[p]))
            continue;
        float xs, ys;
        if (project<0>(x, y, z, 0, 0, P, xs, ys)) {
            const int gx = P.pow2 ? grid_index<true>(xs, P) : grid_index<false>(xs, P);
            const int gy = P.pow2 ? grid_index<true>(ys, P) : grid_index<false>(ys, P);
            emit_record<kEmitGeneric, false>(A, s_hist, s_out, s_cnt, true, p, xs, ys, gx, gy, idx_in_batch, b0, out_wg, unit_stride);
        }
    }
    return neg;
}

// Fast projection (A3) of one selected entry per lane + emission; entries it cannot decide are noted for the epilogue.
// After the series: accept / note for the exact epilogue / emit one entry.  sn, tn: tan(dec), tan(ra); dec, ra.
template <bool POW2, int MODE>
__device__ __forceinline__ void decide_emit(const K1Args &A, unsigned *s_hist, unsigned *s_out, unsigned *s_cnt,
                                            unsigned *s_exc, unsigned *s_nexc, bool have, float ez, double sn, double tn,
                                            double dec, double ra, unsigned tag, uint64_t b0, float2 *out_wg,
                                            unsigned unit_stride)
{
    float xs = 0.f, ys = 0.f;
    const double adec = fabs(dec), ara = fabs(ra);
    // undecided: outside the series' range (tiny z), or within the error window of the FOV limit or of an f32 rounding
    // tie of a map coordinate
    const double sx = fma(dec, A.inv_fov, 0.5), sy = fma(ra, A.inv_fov, 0.5);
    const bool dx = round_decided(sx, kMapWindow, xs), dy = round_decided(sy, kMapWindow, ys);
    bool undecided = !(fabs(sn) <= A.series_max && fabs(tn) <= A.series_max && ez > 0.0f) ||
                     fabs(adec - A.lim) <= kAngWindow || fabs(ara - A.lim) <= kAngWindow || !dx || !dy;
    // cell of the entry: on a power-of-two map the f32 product is exact; otherwise the exact f64 product, with an entry
    // exactly on a cell boundary left to the epilogue (the reference's division decides there: grid_index<false>)
    int gx, gy;
    if (POW2) {
        gx = (int)floorf(xs * A.nn_f);
        gy = (int)floorf(ys * A.nn_f);
    } else {
        const double tx = (double)xs * A.nn_d, ty = (double)ys * A.nn_d;
        const double fx = floor(tx), fy = floor(ty);
        undecided = undecided || tx == fx || ty == fy;
        gx = (int)fx;
        gy = (int)fy;
    }
    if (have && undecided)  // rare: noted for the exact epilogue
        s_exc[min(atomicAdd(s_nexc, 1u), kExcCap - 1)] = tag >> 3;
    const bool valid = have && !undecided && adec <= A.lim && ara <= A.lim;
    emit_record<MODE, POW2>(A, s_hist, s_out, s_cnt, valid, (int)(tag & 7u), xs, ys, gx, gy, tag >> 3, b0, out_wg, unit_stride);
}

template <int SERIES, bool POW2, int MODE>
__device__ __forceinline__ void project_emit(const K1Args &A, unsigned *s_hist, unsigned *s_out, unsigned *s_cnt,
                                             unsigned *s_exc, unsigned *s_nexc, bool have, float ex, float ey, float ez,
                                             unsigned tag, uint64_t b0, float2 *out_wg, unsigned unit_stride)
{
    // A3 (densitymaps.cpp:382-386, utilities.cpp:23-25) to a few ulp: both angles as arctangents through one pass over
    // the series' coefficients (atan_small_pair): ra = atan(Y / Z), dec = asin(X / d) = atan(X / sqrt(Y^2 + Z^2)); the
    // quotients through one cubic step on the hardware reciprocal (square root)
    const double X = (double)ex - 0.5, Y = (double)ey - 0.5, Z = (double)ez;
    const double R2 = fma(Y, Y, Z * Z);
    const double sn = X * rsqrt_fast(R2);  // tan(dec)
    const double tn = Y * rcp_fast(Z);     // tan(ra)
    double dec, ra;
    atan_small_pair<SERIES>(sn, tn, dec, ra);
    decide_emit<POW2, MODE>(A, s_hist, s_out, s_cnt, s_exc, s_nexc, have, ez, sn, tn, dec, ra, tag, b0, out_wg, unit_stride);
}

// Two entries per lane at once: the four arctangents share one pass over the coefficients (atan_small_quad) and give
// the wave four independent fp64 chains.
template <int SERIES, bool POW2, int MODE>
__device__ __forceinline__ void project_emit2(const K1Args &A, unsigned *s_hist, unsigned *s_out, unsigned *s_cnt,
                                              unsigned *s_exc, unsigned *s_nexc, const bool *have, const float *ex,
                                              const float *ey, const float *ez, const unsigned *tag, uint64_t b0, float2 *out_wg, unsigned unit_stride)
{
    double t[4], ang[4];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const double X = (double)ex[j] - 0.5, Y = (double)ey[j] - 0.5, Z = (double)ez[j];
        const double R2 = fma(Y, Y, Z * Z);
        t[2 * j] = X * rsqrt_fast(R2);   // tan(dec)
        t[2 * j + 1] = Y * rcp_fast(Z);  // tan(ra)
    }
    atan_small_quad<SERIES>(t, ang);
#pragma unroll
    for (int j = 0; j < 2; j++)
        decide_emit<POW2, MODE>(A, s_hist, s_out, s_cnt, s_exc, s_nexc, have[j], ez[j], t[2 * j], t[2 * j + 1], ang[2 * j],
                          ang[2 * j + 1], tag[j], b0, out_wg, unit_stride);
}

// STACK: survivors of the slab / pre-test are compacted through the wave stack so that the projection runs on full
// waves (pays when few particles survive: one plane per pass keeps ~20 %); !STACK: the projection runs in place on the
// lanes that survive (pays when most do: the four planes of a replication keep ~78 %, and the LDS traffic and the
// write -> read round trips of the stack cost more than the idle lanes).
// SORT2: the two-level sort (staging in LDS, sub-batches sorted by unit; see Stage) instead of per-record scatter + tile
// histogram.  Not combined with STACK (the wave stacks and the staging area do not both fit twice per CU).
template <int FACE, int SERIES, bool STACK, bool POW2, int MODE>
__global__ __launch_bounds__(kK1Block, SLICER_K1_WAVES_PER_SIMD) void k_project_bin_fast(K1Kernarg K)
{
    constexpr bool SORT2 = MODE == kEmitSort2;
    static_assert(!(STACK && SORT2), "the two-level sort runs without the wave stacks");
    const K1Args &A = K.A;  // K.P is read through the kernarg segment by the exact epilogue only
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    unsigned *s_hist = smem;  // per-workgroup histogram, two 16-bit counters per word (<= 65535 records per workgroup);
                              // SORT2: the staging area (stage_of)
    const int hist_words = SORT2 ? 0 : (A.nbins + 1) >> 1;
    const int tid = threadIdx.x;
    const unsigned lane = lane_id();
    const int wave = tid >> 6;
    // wave-private stack of float4 {x, y, z, plane | index-in-batch << 3}: one ds_write_b128 per push
    float4 *q4 = reinterpret_cast<float4 *>(smem + ((hist_words + 3) & ~3)) + (size_t)wave * kWaveQ;
    __shared__ unsigned s_out[kMaxUnits], s_cnt[kMaxPlanes], s_exc[kExcCap];
    __shared__ unsigned s_nexc;
    __shared__ int s_neg;

    for (int i = tid; i < hist_words; i += kK1Block)
        s_hist[i] = 0;
    if (tid < kMaxPlanes)
        s_cnt[tid] = 0;
    if (tid < kMaxUnits)
        s_out[tid] = 0;
    if (tid == 0) {
        s_neg = 0;
        s_nexc = 0;
    }
    if (SORT2) {
        const Stage St = stage_of(smem);
        if (tid < kMaxCoarse) {
            St.cnt[tid] = 0;
            St.tot[tid] = 0;
        }
        if (tid == 0)
            *St.n = 0;
    }
    __syncthreads();
    unsigned st_off = 0, st_slot = 0;  // SORT2: records / sub-batches this workgroup has written (the same in all threads)

    const uint64_t b0 = (uint64_t)blockIdx.x * A.batch;
    const uint64_t b1 = dmin<uint64_t>(A.n, b0 + A.batch);
    float2 *const out_wg = A.cxy + (size_t)blockIdx.x * (size_t)A.batch;  // this workgroup's region of unit 0
    const unsigned unit_stride = gridDim.x * (unsigned)A.batch;           // records between two units' regions
    unsigned top = 0;  // entries on this wave's stack (wave-uniform)

    const uint64_t w0 = b0 + (uint64_t)wave * (64 * kPerThread);
    RawRound cur;
    const __amdgpu_buffer_rsrc_t rsrc = batch_rsrc(A.pos, b0, b1);
    uint64_t i0 = w0 + (uint64_t)kPerThread * lane;
    int nvalid = i0 < b1 ? (int)dmin<uint64_t>(kPerThread, b1 - i0) : 0;
    load_round(rsrc, (unsigned)(i0 - b0), cur);

    // (SORT2: every wave runs the same number of rounds -- the flush after each round holds workgroup barriers -- so the
    // loop ends where the workgroup's last round ends; a wave beyond the batch's end runs that round empty)
    const uint64_t lim = SORT2 ? b1 + (w0 - b0) : b1;
    for (uint64_t r0 = w0; r0 < lim; r0 += kRound) {
        const uint64_t i1 = i0 + kRound;
        const bool more = r0 + kRound < lim;
        const int nvalid1 = (more && i1 < b1) ? (int)dmin<uint64_t>(kPerThread, b1 - i1) : 0;

        // ---- transform, slab select, conservative FOV pre-test of the round's particles; push (STACK) ----
        static_assert(kPerThread % 2 == 0, "particles are projected in pairs");
        float px[kPerThread], py[kPerThread], pz[kPerThread];
        bool psel[kPerThread];
        unsigned ptag[kPerThread];
        {
#pragma unroll
            for (int k = 0; k < kPerThread; k++) {
                // face permutation (gadget2io.cpp:222-252): output axis a reads source axis perm[a]
                constexpr int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}, {1, 0, 2}, {2, 0, 1}, {2, 1, 0}};
                const float src[3] = {cur.v[3 * k], cur.v[3 * k + 1], cur.v[3 * k + 2]};
                const float s0 = src[perms[FACE][0]], s1 = src[perms[FACE][1]], s2 = src[perms[FACE][2]];
                // q = RN32(r / box): reciprocal product, exact residual, correction (k_check_box_quotient vouches for
                // every r whose q lies in [2^-100, 1]; smaller q cannot influence the results below, see k1_fast_args);
                // (float)(sgn * ((double)r / box)) = sgn * q      gadget2io.cpp:204-206
                const float g0 = s0 * A.rb, g1 = s1 * A.rb, g2 = s2 * A.rb;
                const float q0 = fmaf(fmaf(-g0, A.boxf, s0), A.rb, g0);
                const float q1 = fmaf(fmaf(-g1, A.boxf, s1), A.rb, g1);
                const float q2 = fmaf(fmaf(-g2, A.boxf, s2), A.rb, g2);
                // domain of this path: +0 <= q <= 1 on all three axes as ONE unsigned test on the bit patterns (positive
                // floats order like their bits; -0.0, negatives and NaN have the top bit set or exceed 0x3F800000)
                const bool off = max(max(__float_as_uint(q0), __float_as_uint(q1)), __float_as_uint(q2)) > kQHi;
                // first wrap (gadget2io.cpp:209-220) for q in [0, 1]: sgn = +1 leaves q, sgn = -1 gives 1 + (-q), one f32
                // operation either way: fma(q, sgn, sgn < 0 ? 1 : 0)  (q = 0 on a mirrored axis travels as -0.0 in the
                // reference and comes out of the second wrap as RN(1 - c), which is what 1.0 - c gives here).  Recentre +
                // second wrap (gadget2io.cpp:254-269): w in [0, 1] and c in (0, 1] give d in [-1, 1), so only
                // "d < 0 -> 1 + d" can fire; (float)((double)w - c) is the f32 difference because c is an f32 value.
                const float d0 = fmaf(q0, A.ws[0], A.wo[0]) - A.c0f[0];
                const float d1 = fmaf(q1, A.ws[1], A.wo[1]) - A.c0f[1];
                const float d2 = fmaf(q2, A.ws[2], A.wo[2]) - A.c0f[2];
                const float x = d0 < 0.0f ? 1.0f + d0 : d0;
                const float y = d1 < 0.0f ? 1.0f + d1 : d1;
                const float z = (d2 < 0.0f ? 1.0f + d2 : d2) + A.rcase;  // gadget2io.cpp:270
                // slab of this particle (densitymaps.cpp:374): the planes of one box replication are consecutive slabs,
                // so plane = number of inner thresholds passed (zlo[p >= n_planes] = +inf)
                const bool in = z >= A.zlo[0] && z < A.zlast;
                const int plane = (z >= A.zlo[1] ? 1 : 0) + (z >= A.zlo[2] ? 1 : 0) + (z >= A.zlo[3] ? 1 : 0);
                // Conservative f32 pre-test of the FOV cut: true only if the entry certainly fails |ra| <= lim or
                // |dec| <= lim.  |ra| > lim <=> |Y| > Z tan(lim); given that this does not hold, sqrt(Y^2 + Z^2) <=
                // Z sec(lim)(1 + margin), so |X| > Z tan(lim) sec(lim)(1 + margin) implies |dec| > lim.  The margins
                // (3e-5 relative, ~2e-6 absolute) dwarf every f32 rounding here; z = 0 only yields "outside", which is
                // what the reference decides for it too (angles of +-pi/2 or NaN).
                const bool outside =
                    fabsf(y - 0.5f) > fmaf(z, A.k_ra, A.eps_ra) || fabsf(x - 0.5f) > fmaf(z, A.k_dec, A.eps_dec);
                const bool live = k < nvalid;
                const bool sel = live && !off && in && !outside;
                if (live && off)  // rare: noted for the exact epilogue
                    s_exc[min(atomicAdd(&s_nexc, 1u), kExcCap - 1)] = (unsigned)(i0 + k - b0);
                const unsigned tag = (unsigned)plane | ((unsigned)(i0 + k - b0) << 3);
                if (STACK) {
                    const unsigned long long mask = __ballot(sel);
                    if (sel) {
                        const unsigned slot = top + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                        q4[slot] = make_float4(x, y, z, __uint_as_float(tag));
                    }
                    top += (unsigned)__popcll(mask);
                }
                px[k] = x, py[k] = y, pz[k] = z, psel[k] = sel, ptag[k] = tag;
            }
        }
        // The raw positions are dead from here on: the next round's are requested INTO THE SAME VARIABLE (beyond the
        // batch: zeros, no memory access) and have the projection + emission below, four fifths of the round, to
        // arrive.  (Loading into a second buffer at the top of the round cost twelve register copies per round at the
        // bottom of it.)
        load_round(rsrc, (unsigned)(i1 - b0), cur);
        if (!STACK) {  // ---- fp64 projection + emission in place, two particles at a time ----
#pragma unroll
            for (int k0 = 0; k0 < kPerThread; k0 += 2)
                if (__ballot(psel[k0] || psel[k0 + 1]) != 0ull)
                    project_emit2<SERIES, POW2, MODE>(A, s_hist, s_out, s_cnt, s_exc, &s_nexc, psel + k0, px + k0, py + k0,
                                                      pz + k0, ptag + k0, b0, out_wg, unit_stride);
        }
        if (STACK) {
            lds_fence();
            // ---- fp64 projection on full waves popped from the stack ----
            while (top >= 64u || (!more && top > 0u)) {
                const unsigned take = top >= 64u ? 64u : top;
                float4 ent = make_float4(0.5f, 0.5f, 1.0f, 0.0f);
                const bool have = lane < take;
                if (have)
                    ent = q4[top - take + lane];
                top -= take;
                project_emit<SERIES, POW2, MODE>(A, s_hist, s_out, s_cnt, s_exc, &s_nexc, have, ent.x, ent.y, ent.z,
                                     __float_as_uint(ent.w), b0, out_wg, unit_stride);
            }
            lds_fence();
        }
        if (SORT2) {  // a further round must fit the staging area
            // The next round's positions (requested a round ago) are taken in BEFORE the flush puts its stores into the
            // memory queue: vmcnt counts in order, so a wait for those loads issued after the stores would also wait
            // for the stores to be acknowledged (~1-2 us per flush on the critical path).
#pragma unroll
            for (int k = 0; k < 3 * kPerThread; k++)
                asm volatile("" : "+v"(cur.v[k]));
            stage_flush(A, smem, (unsigned)(kStageCap - kRound), st_off, st_slot);
        }

        i0 = i1;
        nvalid = nvalid1;
    }

    // ---- exact epilogue: the noted particles (or, if there were too many to note, the whole batch) ----
    __syncthreads();
    const unsigned nexc = s_nexc;
    if (nexc) {
        const K1Kernarg *const Kk = kernarg_block();
        const unsigned nb = (unsigned)(b1 - b0);
        const bool redo = nexc > kExcCap;
        if (redo) {  // discard this workgroup's records and counters
            for (int i = tid; i < hist_words; i += kK1Block)
                s_hist[i] = 0;
            if (tid < kMaxPlanes)
                s_cnt[tid] = 0;
            if (tid < kMaxUnits)
                s_out[tid] = 0;
            if (SORT2) {  // ... the sub-batches it has written included: it starts over
                const Stage St = stage_of(smem);
                if (tid < kMaxCoarse) {
                    St.cnt[tid] = 0;
                    St.tot[tid] = 0;
                }
                if (tid == 0)
                    *St.n = 0;
                st_off = 0;
                st_slot = 0;
            }
            __syncthreads();
        }
        bool neg = false;
        const unsigned ne = redo ? nb : nexc;
        // (uniform trip count: with SORT2 every step ends in the workgroup-wide flush test)
        for (unsigned e0 = 0; e0 < ne; e0 += kK1Block) {
            const unsigned e = e0 + tid;
            if (e < ne)
                neg |= process_exact(Kk, s_hist, s_out, s_cnt, redo ? e : s_exc[e], b0, out_wg, unit_stride);
            if (SORT2)
                stage_flush(A, smem, (unsigned)(kStageCap - kK1Block), st_off, st_slot);
        }
        if (neg)
            s_neg = 1;
        __syncthreads();
    }
    if (SORT2) {
        stage_flush(A, smem, 0u, st_off, st_slot);  // whatever is left
        const Stage St = stage_of(smem);
        if (tid == 0) {
            A.sb_n[blockIdx.x] = st_slot;
            if (s_neg)
                atomicOr(A.neg_flag, 1);
        }
        // this workgroup's records per unit -> the total of its sort item (the sort kernel places the items by their
        // prefix) and the selected-entry counters: the records, plus (NGP) the selected entries dropped as off-grid
        if (tid < A.n_units) {
            const unsigned c = St.tot[tid];
            if (c) {
                atomicAdd(&A.item_tot[(size_t)(blockIdx.x / kSort2Blocks) * A.n_units + tid], c);
                atomicAdd(&s_cnt[tid / A.units_per_plane], c);
            }
        }
        __syncthreads();
        if (tid < A.n_planes && s_cnt[tid])
            atomicAdd(A.nsel + 6 * tid, (unsigned long long)s_cnt[tid]);
        return;
    }
    unsigned *row = A.hist16 + (size_t)blockIdx.x * hist_words;  // u16 [nbins] packed, row stride hist_words words
    for (int i = tid; i < hist_words; i += kK1Block)
        row[i] = s_hist[i];
    if (tid < A.n_units)
        A.bcount[(size_t)tid * gridDim.x + blockIdx.x] = s_out[tid];  // [unit][workgroup]
    if (tid == 0 && s_neg)
        atomicOr(A.neg_flag, 1);
    if (A.ngp && tid < A.n_planes && s_cnt[tid])
        atomicAdd(A.nsel + 6 * tid, (unsigned long long)s_cnt[tid]);
}

// ---------------------------------------------------------------------------------------------
// k_project_bin_general
// ---------------------------------------------------------------------------------------------
template <int MAS, bool POW2, bool HAS_MASS, int SERIES, bool REP>
__global__ __launch_bounds__(kK1Block, SLICER_K1_WAVES_PER_SIMD) void k_project_bin_general(
    const float *__restrict__ pos, const float *__restrict__ mass, uint64_t n, int vec, PassParams P, BinGeom G,
    float2 *__restrict__ cxy, unsigned short *__restrict__ cbin, float *__restrict__ cm, unsigned *__restrict__ hist16,
    unsigned *__restrict__ bcount, Targets T)
{
    extern __shared__ unsigned smem[];
    unsigned *s_hist = smem;
    const int hist_words = (G.nbins + 1) >> 1;
    const int tid = threadIdx.x;
    const unsigned lane = lane_id();
    const int wave = tid >> 6;
    // stack entry {x, y, z, plane | replica << 3 | index-in-batch << 10}: x, y already shifted by the replica's (ni, nj)
    float4 *q4 = reinterpret_cast<float4 *>(smem + ((hist_words + 3) & ~3)) + (size_t)wave * kWaveQ;
    __shared__ unsigned s_out[kMaxUnits], s_cnt[kMaxPlanes];
    __shared__ int s_neg;

    for (int i = tid; i < hist_words; i += kK1Block)
        s_hist[i] = 0;
    if (tid < kMaxPlanes)
        s_cnt[tid] = 0;
    if (tid < kMaxUnits)
        s_out[tid] = 0;
    if (tid == 0)
        s_neg = 0;
    __syncthreads();

    const uint64_t b0 = (uint64_t)blockIdx.x * G.batch;
    const uint64_t b1 = dmin<uint64_t>(n, b0 + G.batch);
    int nrmax = 0;  // lateral replication: entries (x + ni, y + nj), |ni|, |nj| <= nrep[plane]   densitymaps.cpp:377-381
    for (int p = 0; p < P.n_planes; p++)
        nrmax = P.nrep[p] > nrmax ? P.nrep[p] : nrmax;
    // this launch's window of the replica grid (the whole grid up to three replications per side)
    const int wi0 = P.rep_i0 > -nrmax ? P.rep_i0 : -nrmax, wi1 = P.rep_i1 < nrmax ? P.rep_i1 : nrmax;
    const int wj0 = P.rep_j0 > -nrmax ? P.rep_j0 : -nrmax, wj1 = P.rep_j1 < nrmax ? P.rep_j1 : nrmax;
    const int side = wj1 - wj0 + 1, nrep2 = (wi1 - wi0 + 1) * side;
    bool neg = false;
    unsigned top = 0;

    const uint64_t w0 = b0 + (uint64_t)wave * (64 * kPerThread);
    RawRound cur, nxt;
    const __amdgpu_buffer_rsrc_t rsrc = batch_rsrc(pos, b0, b1);
    uint64_t i0 = w0 + (uint64_t)kPerThread * lane;
    int nvalid = i0 < b1 ? (int)dmin<uint64_t>(kPerThread, b1 - i0) : 0;
    load_round(rsrc, (unsigned)(i0 - b0), cur);

    // pops 64 entries (or the rest when `flush`) and runs the exact projection on them
    auto drain = [&](bool flush) {
        while (top >= 64u || (flush && top > 0u)) {
            const unsigned take = top >= 64u ? 64u : top;
            bool emit = false, valid = false;
            float xs = 0.f, ys = 0.f, m = 0.f;
            unsigned bin = 0, unit = 0, tile_in_unit = 0;
            int plane = 0;
            if (lane < take) {
                const float4 ent = q4[top - take + lane];
                const unsigned tag = __float_as_uint(ent.w);
                plane = (int)(tag & 7u);
                // x, y carry the replica shift already: project() adds (float)0
                if (project<SERIES>(ent.x, ent.y, ent.z, 0, 0, P, xs, ys)) {
                    valid = true;
                    int gx = grid_index<POW2>(xs, P);
                    int gy = grid_index<POW2>(ys, P);
                    const int nn = P.nn;
                    if (MAS == kNGP) {
                        emit = gx >= 0 && gx < nn && gy >= 0 && gy < nn;
                    } else {
                        emit = true;
                        gx = gx < 0 ? 0 : (gx >= nn ? nn - 1 : gx);
                        gy = gy < 0 ? 0 : (gy >= nn ? nn - 1 : gy);
                    }
                    cell_to_tile(gx, gy, plane, G, unit, tile_in_unit);
                    bin = unit * (unsigned)G.tiles_per_unit + tile_in_unit;
                    if (HAS_MASS)
                        m = mass[b0 + (tag >> 10)];
                }
            }
            top -= take;
            // selected-entry counters: from the bin totals (sort kernel's prologue); NGP adds the selected entries it
            // drops as off-grid here
            if (MAS == kNGP && valid && !emit)
                atomicAdd(&s_cnt[plane], 1u);
            if (emit) {
                const unsigned o = atomicAdd(&s_out[unit], 1u);
                const uint64_t dst = ((uint64_t)unit * gridDim.x + blockIdx.x) * (uint64_t)G.region + o;
                cxy[dst] = make_float2(xs, ys);
                cbin[dst] = (unsigned short)tile_in_unit;
                if (HAS_MASS)
                    cm[dst] = m;
                atomicAdd(&s_hist[bin >> 1], 1u << ((bin & 1u) * 16u));
            }
        }
    };

    for (uint64_t r0 = w0; r0 < b1; r0 += kRound) {
        const uint64_t i1 = i0 + kRound;
        const bool more = r0 + kRound < b1;
        const int nvalid1 = (more && i1 < b1) ? (int)dmin<uint64_t>(kPerThread, b1 - i1) : 0;
        load_round(rsrc, (unsigned)(i1 - b0), nxt);  // (beyond the batch: zeros, no memory access)

        if (!REP) {
            // ---- transform, slab select, conservative FOV pre-test, push ----
#pragma unroll
            for (int k = 0; k < kPerThread; k++) {
                float x, y, z;
                transform(cur.v[3 * k], cur.v[3 * k + 1], cur.v[3 * k + 2], P, x, y, z);
                const bool live = k < nvalid;
                neg |= live & ((x < 0.0f) | (y < 0.0f) | (z < 0.0f));  // densitymaps.cpp:334
                // slabs are disjoint on this path (checked on the host); unused slots are empty intervals
                int pl = -1;
                if (P.n_planes <= 4) {
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        if (z >= P.zlo[p] && z < P.zhi[p])
                            pl = p;
                } else {
#pragma unroll
                    for (int p = 0; p < kMaxPlanes; p++)
                        if (z >= P.zlo[p] && z < P.zhi[p])
                            pl = p;
                }
                const bool sel = live && pl >= 0 && !surely_outside_fov(x, y, z, P);
                const unsigned long long mask = __ballot(sel);
                if (sel) {
                    const unsigned slot = top + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                    const unsigned tag = (unsigned)pl | (HAS_MASS ? (unsigned)(i0 + k - b0) << 10 : 0u);
                    q4[slot] = make_float4(x, y, z, __uint_as_float(tag));
                }
                top += (unsigned)__popcll(mask);
            }
            lds_fence();
            drain(!more);
            lds_fence();
        } else {
            float x[kPerThread], y[kPerThread], z[kPerThread];
            int plane[kPerThread];
#pragma unroll
            for (int k = 0; k < kPerThread; k++) {
                transform(cur.v[3 * k], cur.v[3 * k + 1], cur.v[3 * k + 2], P, x[k], y[k], z[k]);
                const bool live = k < nvalid;
                neg |= live & ((x[k] < 0.0f) | (y[k] < 0.0f) | (z[k] < 0.0f));  // densitymaps.cpp:334
                int pl = -1;
#pragma unroll
                for (int p = 0; p < kMaxPlanes; p++)
                    if (z[k] >= P.zlo[p] && z[k] < P.zhi[p])
                        pl = p;
                plane[k] = live ? pl : -1;
            }
            // one pass of the round per lateral replica (ni outer, nj inner: the reference's order, which only the
            // shot-noise path depends on): the stack never holds more than one round + a remainder
            for (int rep = 0; rep < nrep2; rep++) {
                const int ni = rep / side + wi0, nj = rep % side + wj0;
#pragma unroll
                for (int k = 0; k < kPerThread; k++) {
                    const int pl = plane[k];
                    const int nr = pl >= 0 ? P.nrep[pl] : 0;
                    const bool inrep = pl >= 0 && ni >= -nr && ni <= nr && nj >= -nr && nj <= nr;
                    const float xr = x[k] + (float)ni, yr = y[k] + (float)nj;  // densitymaps.cpp:382
                    const bool sel = inrep && !surely_outside_fov(xr, yr, z[k], P);
                    const unsigned long long mask = __ballot(sel);
                    if (sel) {
                        const unsigned slot = top + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                        const unsigned tag = (unsigned)pl | (HAS_MASS ? (unsigned)(i0 + k - b0) << 10 : 0u);
                        q4[slot] = make_float4(xr, yr, z[k], __uint_as_float(tag));
                    }
                    top += (unsigned)__popcll(mask);
                }
                lds_fence();
                drain(!more && rep == nrep2 - 1);
                lds_fence();
            }
        }

        cur = nxt;
        i0 = i1;
        nvalid = nvalid1;
    }

    if (neg)
        s_neg = 1;
    __syncthreads();
    unsigned *row = hist16 + (size_t)blockIdx.x * hist_words;
    for (int i = tid; i < hist_words; i += kK1Block)
        row[i] = s_hist[i];
    if (tid < G.n_units)
        bcount[(size_t)tid * gridDim.x + blockIdx.x] = s_out[tid];
    if (tid == 0 && s_neg)
        atomicOr(T.neg_flag, 1);
    if (MAS == kNGP && tid < P.n_planes && s_cnt[tid])
        atomicAdd(T.nsel[tid], (unsigned long long)s_cnt[tid]);
}

// ---------------------------------------------------------------------------------------------
// exhaustive check of the f32 quotient for one box size
// ---------------------------------------------------------------------------------------------
// For every non-negative binary32 r (2^31 bit patterns, infinities and NaNs included): if the fast path would accept r
// (its quotient q lies in [0, 1]), q must equal (float)((double)r / box) bit for bit.  *mismatches counts failures.
// out[0] = number of mismatches, out[1..8] = bit patterns of up to eight offending r (diagnostics)
__global__ __launch_bounds__(256) void k_check_box_quotient(double box, float boxf, float rb, unsigned *out)
{
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 31); u += stride) {
        const float r = __uint_as_float((unsigned)u);
        const float g = r * rb;
        const float q = fmaf(fmaf(-g, boxf, r), rb, g);
        if (k1_quotient_in_sweep_range(q)) {
            const float ref = (float)((double)r / box);
            if (__float_as_uint(ref) != __float_as_uint(q)) {
                const unsigned k = atomicAdd(out, 1u);
                if (k < 8)
                    out[1 + k] = (unsigned)u;
            }
        }
    }
}

hipError_t launch_check_box_quotient(double box, unsigned *d_out9, hipStream_t s)
{
    const float boxf = (float)box;
    const float rb = 1.0f / boxf;  // IEEE: correctly rounded on the host
    k_check_box_quotient<<<256 * 16, 256, 0, s>>>(box, boxf, rb, d_out9);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
size_t project_bin_lds_bytes(const BinGeom &G, bool has_mass)
{
    (void)has_mass;  // masses are fetched by index in the projection stage, not carried on the stacks
    return sizeof(unsigned) * (size_t)((((G.nbins + 1) >> 1) + 3) & ~3) + (size_t)kWaves * kWaveQ * 16;
}

size_t project_bin_sort2_lds_bytes() { return (kStageBytes + 15) & ~(size_t)15; }

template <typename Kern>
static hipError_t set_lds(Kern kern, size_t lds)
{
    if (lds > 48 * 1024)  // up to 64 KiB of histogram + the wave stacks
        return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds);
    return hipSuccess;
}

template <int FACE>
static hipError_t launch_k1_fast(bool s9, int nb, size_t lds, const PassParams &P, const K1Args &A, hipStream_t s)
{
    hipError_t e;
    K1Kernarg K;
    K.P = P;
    K.A = A;
#define K1F_(S_, ST_, P2_, S2_)                                                       \
    do {                                                                              \
        auto kern = k_project_bin_fast<FACE, S_, ST_, P2_, S2_>;                      \
        if ((e = set_lds(kern, lds)) != hipSuccess)                                   \
            return e;                                                                 \
        kern<<<nb, kK1Block, lds, s>>>(K);                                            \
    } while (0)
#define K1F(S_, ST_, S2_)                                                             \
    do {                                                                              \
        if (A.pow2)                                                                   \
            K1F_(S_, ST_, true, S2_);                                                 \
        else                                                                          \
            K1F_(S_, ST_, false, S2_);                                                \
    } while (0)
    // the lean emission where no run-time variant of it is needed (emit_record)
    const bool lean = !A.sort2 && A.units_per_plane == 1 && !(A.mass != nullptr && A.ngp) &&
                      (!A.pow2 || (A.ntx_log2 >= 0 && A.tpu_log2 >= 0));
    if (A.sort2) {  // (never with the wave stacks: the host clears `stack` when it picks the two-level sort)
        if (s9) K1F(9, false, kEmitSort2); else K1F(15, false, kEmitSort2);
    } else if (A.stack) {
        if (lean && A.ngp) {
            if (s9) K1F(9, true, kEmitLeanNgp); else K1F(15, true, kEmitLeanNgp);
        } else if (lean && A.mass != nullptr) {
            if (s9) K1F(9, true, kEmitLeanMass); else K1F(15, true, kEmitLeanMass);
        } else if (lean) {
            if (s9) K1F(9, true, kEmitLean); else K1F(15, true, kEmitLean);
        } else {
            if (s9) K1F(9, true, kEmitGeneric); else K1F(15, true, kEmitGeneric);
        }
    } else {
        if (lean && A.ngp) {
            if (s9) K1F(9, false, kEmitLeanNgp); else K1F(15, false, kEmitLeanNgp);
        } else if (lean && A.mass != nullptr) {
            if (s9) K1F(9, false, kEmitLeanMass); else K1F(15, false, kEmitLeanMass);
        } else if (lean) {
            if (s9) K1F(9, false, kEmitLean); else K1F(15, false, kEmitLean);
        } else {
            if (s9) K1F(9, false, kEmitGeneric); else K1F(15, false, kEmitGeneric);
        }
    }
#undef K1F_
#undef K1F
    return hipGetLastError();
}

template <int MAS, bool POW2, bool HAS_MASS, int SERIES>
static hipError_t launch_k1_general(bool vec, const float *pos, const float *mass, uint64_t n, const PassParams &P,
                                    const BinGeom &G, const BinWorkspace &W, const Targets &T, hipStream_t s)
{
    const int nb = (int)((n + G.batch - 1) / G.batch);
    const size_t lds = project_bin_lds_bytes(G, HAS_MASS);
    hipError_t e;
    if (G.region != G.batch) {  // lateral replication
        auto kern = k_project_bin_general<MAS, POW2, HAS_MASS, SERIES, true>;
        if ((e = set_lds(kern, lds)) != hipSuccess)
            return e;
        kern<<<nb, kK1Block, lds, s>>>(pos, mass, n, vec ? 1 : 0, P, G, W.cxy, W.cbin, W.cm, W.hist16, W.bcount, T);
    } else {
        auto kern = k_project_bin_general<MAS, POW2, HAS_MASS, SERIES, false>;
        if ((e = set_lds(kern, lds)) != hipSuccess)
            return e;
        kern<<<nb, kK1Block, lds, s>>>(pos, mass, n, vec ? 1 : 0, P, G, W.cxy, W.cbin, W.cm, W.hist16, W.bcount, T);
    }
    return hipGetLastError();
}

hipError_t launch_project_bin(const LaunchCfg &cfg, bool fast, const float *d_pos, const float *d_mass, uint64_t n,
                              const PassParams &P, const K1Args &A0, const BinGeom &G, const BinWorkspace &W,
                              const Targets &T, hipStream_t s)
{
    const bool vec = (reinterpret_cast<uintptr_t>(d_pos) & 15u) == 0 && (G.batch % 4) == 0;
    const bool pow2 = P.pow2 != 0;
    bool s9 = P.series_max < 0.2;  // every survivor of the pre-test is inside the 9-term range
    if (fast) {
        s9 = A0.series_max < 0.2;  // (the fast kernel's own bound: both of its series run on tangents)
        K1Args A = A0;
        A.pos = d_pos;
        A.mass = cfg.has_mass ? d_mass : nullptr;
        A.n = n;
        A.vec = vec ? 1 : 0;
        A.ngp = cfg.mas == kNGP;
        A.cxy = W.cxy;
        A.cbin = W.cbin;
        A.cm = W.cm;
        A.hist16 = W.hist16;
        A.bcount = W.bcount;
        A.neg_flag = T.neg_flag;
        A.nsel = T.nsel[0];
        A.tw_log2 = G.tw_log2, A.th_log2 = G.th_log2, A.ntx = G.ntx, A.tiles_per_unit = G.tiles_per_unit;
        auto log2_exact = [](int v) {
            int l = 0;
            while ((1 << l) < v)
                l++;
            return (1 << l) == v ? l : -1;
        };
        A.ntx_log2 = log2_exact(G.ntx), A.tpu_log2 = log2_exact(G.tiles_per_unit);
        A.units_per_plane = G.units_per_plane, A.rows_per_unit = G.rows_per_unit, A.n_units = G.n_units;
        A.nbins = G.nbins, A.batch = G.batch;
        A.c1 = W.c1;
        A.sb_off = W.sb_off;
        A.sb_start = W.sb_start;
        A.sb_n = W.sb_n;
        A.item_tot = W.item_tot;
        if (A.sort2)
            A.stack = 0;
        const int nb = (int)((n + G.batch - 1) / G.batch);
        const size_t lds = A.sort2 ? project_bin_sort2_lds_bytes() : project_bin_lds_bytes(G, cfg.has_mass);
        switch (A0.face) {
        case 0: return launch_k1_fast<0>(s9, nb, lds, P, A, s);
        case 1: return launch_k1_fast<1>(s9, nb, lds, P, A, s);
        case 2: return launch_k1_fast<2>(s9, nb, lds, P, A, s);
        case 3: return launch_k1_fast<3>(s9, nb, lds, P, A, s);
        case 4: return launch_k1_fast<4>(s9, nb, lds, P, A, s);
        default: return launch_k1_fast<5>(s9, nb, lds, P, A, s);
        }
    }
#define K1(MAS_, P2_, HM_)                                                                  \
    (s9 ? launch_k1_general<MAS_, P2_, HM_, 9>(vec, d_pos, d_mass, n, P, G, W, T, s)       \
        : launch_k1_general<MAS_, P2_, HM_, 15>(vec, d_pos, d_mass, n, P, G, W, T, s))
    if (cfg.mas == kNGP) {
        if (pow2)
            return cfg.has_mass ? K1(kNGP, true, true) : K1(kNGP, true, false);
        return cfg.has_mass ? K1(kNGP, false, true) : K1(kNGP, false, false);
    }
    if (pow2)
        return cfg.has_mass ? K1(kTSC, true, true) : K1(kTSC, true, false);
    return cfg.has_mass ? K1(kTSC, false, true) : K1(kTSC, false, false);
#undef K1
}

}  // namespace slicer
