// slicer_capi.cpp -- implementation of the C ABI declared in include/slicer_amd.h.
// Host-side orchestration only: buffers, streams, staging, launch selection, error mapping.
// Reference behaviour mirrored per entry point is cited in the header.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/slicer_amd.h"
#include "slicer_kernels.hpp"

using namespace slicer;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct PlaneBufs {
    DevBuf tot;
    DevBuf toti[6];
    DevBuf acc[6];  // F64/FIXED accumulators, or NGP per-file scratch
    DevBuf acc_shared;
};

struct ProfEntry {
    int name;
    hipEvent_t e0, e1;
};

const char *kKernelNames[] = {"direct_deposit", "finalize_tsc", "fold_ngp",  "synth",         "project_bin",
                              "bin_scan",       "bin_scatter",  "tile_deposit", "debug_project", "bin_sort"};
enum { KN_DIRECT = 0, KN_FINALIZE, KN_FOLD, KN_SYNTH, KN_PROJECT, KN_SCAN, KN_SCATTER, KN_TILE, KN_DEBUG, KN_SORT2, KN_COUNT };

}  // namespace

// Tuning and test knobs of one handle.  Read from the environment ONCE, in slicer_create (so that the tools/ scripts
// keep working), and changed per handle through slicer_set_option -- never getenv on a launch path: the per-GPU host
// threads of SLICER_amd --devices run concurrently.
struct Options {
    int k4_int = 1;       // integer tile cells: 0 never, 1 when a launch has >= 2048 particles per bin, 2 always
    int tile_log2 = 0;    // log2 tile width (0 = automatic); tile_h_log2 likewise for the height (0 = tile_log2)
    int tile_h_log2 = 0;
    int bin_batch = 0;    // particles per project+bin workgroup (0 = automatic)
    int unit_rows = 0;    // tile rows per unit (0 = automatic): band units on small maps, for tests
    int k3_per_cu = 2;    // persistent sort workgroups per CU
    int k1_general = 0;   // 1: always the general project+bin kernel
    int k1_stack = -1;    // fast project+bin kernel: compact survivors through the wave stack (0 / 1; -1 = by slab depth)
    int ngp_general = 0;  // 1: no in-tile NGP fold (count map + k_fold_ngp)
    int dl_quot = 1;      // maps that are not a power of two wide: allow the swept reciprocal-product quotient
    int zero_batch = 1;   // 1: the maps of a pass are cleared by one launch (0: one hipMemsetAsync each)
    int pending = 0;      // chunks per tile launch (0 = automatic: 8 ... 32 by the records a chunk brings per tile)
    int thin_host = 0;    // 1: shot-noise deviates drawn by libc rand() on the host (0: the stream continues on the device)
    int sort2 = 0;        // 1: two-level sort (project+bin sorts by coarse bin in LDS, k_sort2 by tile) where a pass
                          // qualifies.  Off by default: it moves fewer bytes but costs more instructions
};
struct OptionName {
    const char *key, *env;
    int Options::*field;
};
const OptionName kOptionNames[] = {
    {"k4_int", "SLICER_K4_INT", &Options::k4_int},
    {"tile_log2", "SLICER_TILE_LOG2", &Options::tile_log2},
    {"tile_h_log2", "SLICER_TILE_H_LOG2", &Options::tile_h_log2},
    {"bin_batch", "SLICER_BIN_BATCH", &Options::bin_batch},
    {"unit_rows", "SLICER_UNIT_ROWS", &Options::unit_rows},
    {"k3_per_cu", "SLICER_K3_PER_CU", &Options::k3_per_cu},
    {"k1_general", "SLICER_K1_GENERAL", &Options::k1_general},
    {"k1_stack", "SLICER_K1_STACK", &Options::k1_stack},
    {"ngp_general", "SLICER_NGP_GENERAL", &Options::ngp_general},
    {"dl_quot", "SLICER_DL_QUOT", &Options::dl_quot},
    {"sort2", "SLICER_SORT2", &Options::sort2},
    {"thin_host", "SLICER_THIN_HOST", &Options::thin_host},
    {"pending", "SLICER_PENDING", &Options::pending},
    {"zero_batch", "SLICER_ZERO_BATCH", &Options::zero_batch},
};

constexpr size_t kPassScalarsBytes = sizeof(unsigned long long) * SLICER_MAX_PLANES * 6 + sizeof(int) + 7 * sizeof(unsigned);

struct slicer_handle_s {
    int device = 0;
    Options opt;
    int num_cus = 256;
    unsigned items_epoch = 0;  // launches of the tile kernel on the current w_items workspace
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipStream_t own = nullptr;
    uint64_t max_chunk = 0;
    std::string err;

    bool in_plane = false, in_file = false, finalized = false;
    slicer_plane_desc desc{};
    slicer_file_desc file{};
    uint64_t npix2 = 0;
    PlaneBufs planes[SLICER_MAX_PLANES];
    unsigned long long *d_counts = nullptr;  // [SLICER_MAX_PLANES][6]
    int *d_neg = nullptr;
    // [7] bits of the largest selected per-particle mass of this pass: per species, and slot 6 for the shared accumulator
    // (want_type_maps == 0), whose pending list mixes species -- the tile kernel's quantum must cover all of them
    unsigned *d_maxmass = nullptr;
    bool type_seen[6] = {};       // in this plane pass
    bool shared_seen = false;
    int algo_mask = 0;            // bit (1 << SLICER_ALGO_*) of every algorithm that ran in this pass; bit 3 = thinning
    bool neg_remote = false;      // another rank reported the negativity guard (slicer_reduce_meta_set)
    int file_mode[6] = {};        // NGP fold mode of the current file
    bool file_partial_flush[6] = {};  // NGP: some of this file's records of the species went to the global count map
    unsigned file_serial = 0;         // counts slicer_file_begin calls (PendingList.file_id)
    float file_mconst[6] = {};
    int fixed_exp[6] = {};
    int fixed_exp_shared = 0;
    bool fixed_exp_set[6] = {};
    bool fixed_shared_set = false;

    // host->device staging (double buffered)
    float *h_stage[2] = {nullptr, nullptr};
    float *d_stage[2] = {nullptr, nullptr};
    float *h_mstage[2] = {nullptr, nullptr};
    float *d_mstage[2] = {nullptr, nullptr};
    hipEvent_t stage_free[2] = {nullptr, nullptr};
    uint64_t stage_cap = 0;  // particles

    // SLICER_ALGO_BINNED workspace (sized for max_chunk particles)
    DevBuf w_cxy, w_cbin, w_cm, w_hist, w_hist16, w_total, w_bcount, w_items;
    DevBuf w_c1, w_sboff, w_sbstart, w_sbn;  // two-level sort: project+bin output of the current chunk
    // box sizes whose f32 quotient r/box passed (true) or failed (false) the exhaustive device sweep
    // (launch_check_box_quotient): k_project_bin_fast is only used for the former
    std::vector<std::pair<double, bool>> box_verdicts;
    std::vector<std::pair<int, bool>> dl_verdicts;  // map sizes (not powers of two) whose quot_dl3 passed / failed its sweep
    bool dl_quot_ok = false;                        // ... the verdict for the current pass's npix
    unsigned *d_sweep = nullptr;
    DevBuf w_tcounts, w_tbase, w_urand;  // shot-noise thinning (snopt > 0)
    std::vector<float> h_urand;
    // libc's rand() stream on the device (slicer_rand.hip): jump tables, the 31-word state, wave start states
    DevBuf w_randtab, w_randstate, w_randwaves;
    bool rand_tab_ready = false;
    bool rand_on_device = false;  // between thin_rng_begin and thin_rng_end the device holds the stream
    // a stream of this handle's own instead of the process-global one (slicer_rand_stream_set): the reference's MPI
    // ranks each own an identically seeded copy of libc's stream; rank threads of one process get theirs this way
    ZeroList zero_list{};      // zero-fills collected between zero_begin / zero_end
    bool zero_collect = false;
    bool rand_private = false;
    uint32_t rand_state[31] = {};
    // Process-global mode (no slicer_rand_stream_set): the process's stream is read when a pass with snopt > 0 BEGINS --
    // before that call touches the HIP runtime, whose threads draw from libc's stream themselves now and then -- the pass
    // thins from this copy, and the advanced state goes back to libc when the pass ends (flush / finalize / read, the next
    // plane_begin, destroy): whatever the runtime drew in between is overwritten.
    bool rand_pass = false;
    // snopt > 0 with several planes in one pass: the reference draws its deviates plane by plane (outer loop of
    // createDensityMaps' caller), so the chunks are kept on the device and deposited plane-major when the pass ends
    struct ThinChunk {
        int file, type;
        DevBuf pos, mass;
        uint64_t n;
    };
    struct ThinFile {
        slicer_file_desc file;
        int mode[6];
        float mconst[6];
    };
    std::vector<ThinChunk> thin_chunks;
    std::vector<ThinFile> thin_files;
    // chunks binned but not yet deposited (flushed by one k_tile_deposit launch).  One list per plane group: a pass whose
    // planes go through the binned kernels in several groups (binned_chunk) keeps every group's chunks pending separately.
    struct Pending {
        PendingList L{};
        int key = -1;        // type * 2 + has_mass (or 12 + has_mass for the shared accumulator)
        int p0 = 0, np = 0;  // planes [p0, p0 + np) of the pass are behind the pending chunks
        LaunchCfg cfg{};
        PassParams P{};
        BinGeom G{};
        Targets T{};
        uint64_t particles = 0;  // particles behind the pending chunks (bounds their record count)
        DevBuf w_sxy[kMaxPending], w_base[kMaxPending];  // one sorted slot per pending chunk
        // two-level sort: the chunk's item table (w_base then holds the items' allocation cursor), the group's bin totals
        DevBuf w_ptab[kMaxPending], w_tot;
        bool sort2 = false;
        int limit = 8;  // chunks per tile launch of this list (set when its first chunk arrives)
    };
    Pending pg[SLICER_MAX_PLANES];

    bool profiling = false;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
    uint64_t prof_event_failures = 0;
    double prof_ms[KN_COUNT] = {};
    uint64_t prof_n[KN_COUNT] = {};
};

namespace {

thread_local std::string g_null_err;

int fail(slicer_handle h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h)
        h->err = buf;
    else
        g_null_err = buf;
    return code;
}

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(h, e_ == hipErrorOutOfMemory ? SLICER_ERR_NOMEM : SLICER_ERR_HIP,            \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Grow-only device buffer.  *fresh (optional) is set when the buffer was (re)allocated: its contents are undefined
// (the new allocation may even reuse the old address, so callers must not compare pointers).
int ensure(slicer_handle h, DevBuf &b, size_t bytes, bool *fresh = nullptr)
{
    if (fresh)
        *fresh = false;
    if (b.cap >= bytes)
        return SLICER_OK;
    if (b.p)
        HIPCHK(h, hipFree(b.p));  // implicit device synchronisation: happens only while a workspace still grows
    b.p = nullptr;
    b.cap = 0;
    HIPCHK(h, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    if (fresh)
        *fresh = true;
    return SLICER_OK;
}

void release(DevBuf &b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

hipEvent_t get_event(slicer_handle h)
{
    if (!h->ev_pool.empty()) {
        hipEvent_t e = h->ev_pool.back();
        h->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess)
        return nullptr;  // the scope below then skips timing for this launch
    return e;
}

struct ProfScope {
    slicer_handle h;
    int name;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(slicer_handle h_, int name_) : h(h_), name(name_)
    {
        if (h->profiling) {
            e0 = get_event(h);
            e1 = get_event(h);
            if (!e0 || !e1) {  // hipEventCreate failed: leave this launch untimed rather than record on a null event
                if (e0)
                    h->ev_pool.push_back(e0);
                if (e1)
                    h->ev_pool.push_back(e1);
                e0 = e1 = nullptr;
                h->prof_event_failures++;
            } else {
                (void)hipEventRecord(e0, h->stream);
            }
        }
    }
    ~ProfScope()
    {
        if (e0 && e1) {
            (void)hipEventRecord(e1, h->stream);
            h->prof.push_back({name, e0, e1});
        }
    }
};

void prof_collect(slicer_handle h)
{
    for (auto &p : h->prof) {
        (void)hipEventSynchronize(p.e1);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            h->prof_ms[p.name] += ms;
            h->prof_n[p.name] += 1;
        }
        h->ev_pool.push_back(p.e0);
        h->ev_pool.push_back(p.e1);
    }
    h->prof.clear();
}

float ceil_to_f32(double v)
{
    // smallest float >= v
    float f = (float)v;
    if ((double)f < v)
        f = std::nextafterf(f, INFINITY);
    return f;
}

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

int env_int(const char *name, int dflt = 0)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

int acc_kind(const slicer_plane_desc &d, bool has_mass)
{
    if (d.mas == SLICER_MAS_NGP)
        return has_mass ? kF32 : kCountU32;
    switch (d.accum) {
    case SLICER_ACC_F64: return kF64;
    case SLICER_ACC_FIXED64: return kFixed64;
    default: return kF32;
    }
}

size_t acc_elem_size(int kind) { return (kind == kF64 || kind == kFixed64) ? 8 : 4; }

// Build the uniform parameter block for (current file, type).
void make_params(slicer_handle h, int type, bool has_mass, PassParams &P)
{
    const slicer_plane_desc &d = h->desc;
    const slicer_file_desc &f = h->file;
    memset(&P, 0, sizeof P);
    P.box = f.boxsize;
    P.inv_box = 1.0 / f.boxsize;
    for (int a = 0; a < 3; a++) {
        P.c0[a] = f.center[a];
        P.sgn[a] = (float)f.sgn[a];
    }
    // gadget2io.cpp:222-252: face -> (x,y,z) = wrapped[perm]
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}, {1, 0, 2}, {2, 0, 1}, {2, 1, 0}};
    int fi = (f.face >= 1 && f.face <= 6) ? f.face - 1 : 0;  // any other value leaves case 1 (switch falls through)
    for (int a = 0; a < 3; a++) {
        P.perm[a] = perms[fi][a];
        for (int c = 0; c < 3; c++)
            P.pm[a][c] = perms[fi][a] == c ? 0xFFFFFFFFu : 0u;
    }
    P.rcase = f.rcase;
    P.n_planes = d.n_planes;
    for (int p = 0; p < d.n_planes; p++) {
        double minDist = d.ld[p] / f.boxsize * 1.e+3 / 1.0;   // densitymaps.cpp:346 (POS_U = 1.0)
        double maxDist = d.ld2[p] / f.boxsize * 1.e+3 / 1.0;  // densitymaps.cpp:347
        P.zlo[p] = ceil_to_f32(minDist);
        P.zhi[p] = ceil_to_f32(maxDist);
        P.nrep[p] = d.nrepperp[p];
    }
    for (int p = d.n_planes; p < kMaxPlanes; p++) {  // unused slots select nothing (kernels may unroll over all 8)
        P.zlo[p] = INFINITY;
        P.zhi[p] = -INFINITY;
    }
    P.rep_i0 = P.rep_j0 = -64;  // every lateral replica (the binned path narrows this per launch)
    P.rep_i1 = P.rep_j1 = 64;
    P.fov = d.fov_rad;
    P.inv_fov = 1.0 / d.fov_rad;
    P.lim = d.fov_rad * (1. + 2. / d.npix) * 0.5;  // densitymaps.cpp:383
    if (P.lim < 1.5) {
        P.tan_lim_hi = (float)(std::tan(P.lim) * (1.0 + 1e-5));
        P.sin2_lim_hi = (float)(std::sin(P.lim) * std::sin(P.lim) * (1.0 + 1e-5));
    }
    P.force_libm = d.debug_flags & 1;
    // every entry that reaches the series passed the f32 pre-test (|tan ra|, |sin dec| within 1e-5 of the limit's)
    // or, on the direct path, may lie anywhere: there, and with debug bit 1, keep the wide 15-term range
    P.series_max = (!(d.debug_flags & 2) && P.lim < 1.5 && std::tan(P.lim) * 1.001 < kSeriesMax9) ? kSeriesMax9
                                                                                                   : kSeriesMax15;
    P.nn = d.npix;
    P.pow2 = is_pow2(d.npix) ? 1 : 0;
    P.dl = 1. / double(d.npix);  // utilities.cpp:50
    P.nn_d = (double)d.npix;
    P.half_dl = 0.5 * P.dl;
    P.onehalf_dl = 0.5 * 3.0 * P.dl;
    P.dl_f = (float)P.dl;
    P.nn_f = (float)P.nn_d;
    P.half_dl_f = (float)P.half_dl;
    P.onehalf_dl_f = (float)P.onehalf_dl;
    auto round_down = [](double x) {
        float f = (float)x;
        return (double)f > x ? std::nextafterf(f, 0.0f) : f;
    };
    P.inv_dl = 1.0 / P.dl;
    P.dl_quot_ok = (!P.pow2 && h->dl_quot_ok) ? 1 : 0;
    P.half_dl_lo = round_down(P.half_dl);
    P.onehalf_dl_lo = round_down(P.onehalf_dl);
    P.mconst = (float)f.massarr[type];  // densitymaps.cpp:372
    P.sm_const = sqrtf(P.mconst);       // glibc sqrtf is correctly rounded, as std::sqrt(float)
    int e = d.want_type_maps ? h->fixed_exp[type] : h->fixed_exp_shared;
    P.fixed_scale = std::ldexp(1.0, e);
    {
        int le = 10;  // MAX_M = 1e3 < 2^10
        if (P.mconst > 0 && std::isfinite(P.mconst))
            le = std::ilogb(P.mconst) + 1;
        P.tile_scale = std::ldexp(1.0, 49 - le);
        P.tile_inv_scale = std::ldexp(1.0, le - 49);
        P.tile_cmin = std::ldexp(1.0f, le - 25);
    }
    (void)has_mass;
}

int pick_fixed_exp(const slicer_plane_desc &d, double m, bool has_mass)
{
    int frac = d.fixed_frac_bits > 0 ? d.fixed_frac_bits : 40;
    int le = 10;  // MAX_M = 1e3 < 2^10
    if (!has_mass && m > 0 && std::isfinite(m))
        le = std::ilogb(m) + 1;
    return frac - le;
}

// Zero-fills on the handle's stream.  Between zero_begin and zero_end they are collected and go out as ONE launch
// (launch_zero_many): a pass clears four to fourteen maps, and every dispatch costs a few microseconds of idle GPU.
int zero_flush(slicer_handle h)
{
    ZeroList &Z = h->zero_list;
    if (Z.n == 1) {
        HIPCHK(h, hipMemsetAsync(Z.p[0], 0, Z.words[0] * 4, h->stream));
    } else if (Z.n > 1) {
        HIPCHK(h, launch_zero_many(Z, h->stream));
    }
    Z.n = 0;
    Z.quad0[0] = 0;
    return SLICER_OK;
}

int zero_async(slicer_handle h, void *p, size_t bytes)
{
    if (!h->zero_collect || (bytes & 3) || bytes == 0) {
        HIPCHK(h, hipMemsetAsync(p, 0, bytes, h->stream));
        return SLICER_OK;
    }
    ZeroList &Z = h->zero_list;
    if (Z.n == kZeroMax) {
        int rc = zero_flush(h);
        if (rc)
            return rc;
    }
    Z.p[Z.n] = p;
    Z.words[Z.n] = bytes / 4;
    Z.quad0[Z.n + 1] = Z.quad0[Z.n] + (bytes / 4 + 3) / 4;
    Z.n++;
    return SLICER_OK;
}

void zero_begin(slicer_handle h)
{
    h->zero_collect = h->opt.zero_batch != 0;
    h->zero_list.n = 0;
    h->zero_list.quad0[0] = 0;
}

int zero_end(slicer_handle h)
{
    h->zero_collect = false;
    return zero_flush(h);
}

// Make sure the destination buffers of `type` exist and are zeroed for this plane pass.
static int prepare_type_maps(slicer_handle h, int type, bool has_mass)
{
    const slicer_plane_desc &d = h->desc;
    const size_t n4 = h->npix2 * 4;
    const int kind = acc_kind(d, has_mass);
    const bool ngp = d.mas == SLICER_MAS_NGP;
    const bool shared = !ngp && !d.want_type_maps;
    if (shared) {
        if (!h->shared_seen) {
            if (!h->fixed_shared_set) {
                // From the mass table alone, which every sub-file of a snapshot carries identically -- not from which
                // types this particular file holds -- so that ranks owning different sub-files pick the same scale
                // (their FIXED64 accumulators are summed as integers: slicer_plane_accumulators).
                double mm = 0;
                for (int t = 0; t < 6; t++)
                    mm = std::max(mm, h->file.massarr[t]);
                h->fixed_exp_shared = pick_fixed_exp(d, mm, d.hydro != 0);
                h->fixed_shared_set = true;
            }
            for (int p = 0; p < d.n_planes; p++) {
                int rc = ensure(h, h->planes[p].acc_shared, h->npix2 * acc_elem_size(kind));
                if (rc)
                    return rc;
                rc = zero_async(h, h->planes[p].acc_shared.p, h->npix2 * acc_elem_size(kind));
                if (rc)
                    return rc;
            }
            h->shared_seen = true;
        }
        return SLICER_OK;
    }
    if (!h->type_seen[type]) {
        if (!h->fixed_exp_set[type]) {
            h->fixed_exp[type] = pick_fixed_exp(d, h->file.massarr[type], has_mass);
            h->fixed_exp_set[type] = true;
        }
        for (int p = 0; p < d.n_planes; p++) {
            int rc = SLICER_OK;
            if (!ngp || d.want_type_maps) {  // NGP without per-type outputs only needs the count scratch
                rc = ensure(h, h->planes[p].toti[type], n4);
                if (rc)
                    return rc;
                rc = zero_async(h, h->planes[p].toti[type].p, n4);
                if (rc)
                    return rc;
            }
            if (ngp || kind != kF32) {
                size_t b = h->npix2 * (ngp ? 4 : acc_elem_size(kind));
                rc = ensure(h, h->planes[p].acc[type], b);
                if (rc)
                    return rc;
                rc = zero_async(h, h->planes[p].acc[type].p, b);
                if (rc)
                    return rc;
            }
        }
        h->type_seen[type] = true;
    }
    return SLICER_OK;
}

int prepare_type(slicer_handle h, int type, bool has_mass)
{
    zero_begin(h);  // the maps of all planes are cleared by one launch
    const int rc = prepare_type_maps(h, type, has_mass);
    const int rcz = zero_end(h);
    return rc ? rc : rcz;
}

void fill_targets(slicer_handle h, int type, bool has_mass, Targets &T)
{
    const slicer_plane_desc &d = h->desc;
    const int kind = acc_kind(d, has_mass);
    const bool ngp = d.mas == SLICER_MAS_NGP;
    const bool shared = !ngp && !d.want_type_maps;
    memset(&T, 0, sizeof T);
    for (int p = 0; p < d.n_planes; p++) {
        if (shared)
            T.acc[p] = h->planes[p].acc_shared.p;
        else if (ngp || kind != kF32)
            T.acc[p] = h->planes[p].acc[type].p;
        else
            T.acc[p] = h->planes[p].toti[type].p;
        T.nsel[p] = h->d_counts + (size_t)p * 6 + type;
    }
    T.neg_flag = h->d_neg;
    T.max_mass = h->d_maxmass + (shared ? 6 : type);
}

// Lateral replication (densitymaps.cpp:377-381): a pass with n replications per side has (2n+1)^2 replicas per particle.
// One launch of the binned project kernel takes a window of at most 7 x 7 of them; the side (2n+1) is cut into equal parts.
static int rep_windows(int nrmax) { return (2 * nrmax + 1 + 6) / 7; }
static int rep_window_side(int nrmax) { return (2 * nrmax + 1 + rep_windows(nrmax) - 1) / rep_windows(nrmax); }

constexpr int kBinBatch = 32768;  // particles per K1 workgroup (sweep: tools/sweep.sh)
constexpr int kUnitBins = 8192;   // up to this many bins the units are whole planes

// Tile geometry of the binned path.  Tiles are powers of two so that pixel -> tile is a shift.  4-byte
// LDS cells (NGP counts): up to 128 x 128 (+halo = 67.6 KiB of LDS, two workgroups per CU); 8-byte: 64 x 128.
// Small maps get smaller tiles so that the grid still has >= ~1024 workgroups.
bool choose_geom(const slicer_plane_desc &d, int acc, const Options &opt, BinGeom &G)
{
    int nrmax = 0;
    for (int p = 0; p < d.n_planes; p++)
        nrmax = std::max(nrmax, d.nrepperp[p]);
    // (2n+1)^2 records per particle must fit the 16-bit per-workgroup counters at a 1024-particle batch: beyond three
    // replications per side the replica grid is walked in windows of at most 7 x 7, one run of K1-K3 per window
    const int ws = rep_window_side(nrmax);
    const int reps = ws * ws;
    for (int p = 0; p < d.n_planes; p++)  // slabs must be disjoint: a particle enters at most one bin
        for (int q = p + 1; q < d.n_planes; q++)
            if (d.ld[p] < d.ld2[q] && d.ld[q] < d.ld2[p])
                return false;
    // tuning / test knobs of the handle (slicer_set_option)
    const int env_s = opt.tile_log2, env_h = opt.tile_h_log2;
    const int env_b = opt.bin_batch;
    const bool wide = acc != kCountU32;  // every mode but the NGP counts keeps 8-byte cells in LDS
    int s = 7;  // log2 tile side
    auto tiles = [&](int sl) {
        int tw = 1 << sl, th = 1 << (wide ? sl - 1 : sl);
        return (long)((d.npix + tw - 1) / tw) * (long)((d.npix + th - 1) / th);
    };
    while (s > 4 && tiles(s) * d.n_planes < 1024)
        s--;
    G.tw_log2 = s;
    G.th_log2 = wide ? s - 1 : s;
    if (env_s) {
        G.tw_log2 = env_s;
        G.th_log2 = env_h ? env_h : env_s;
    }
    G.ntx = (d.npix + (1 << G.tw_log2) - 1) >> G.tw_log2;
    G.nty = (d.npix + (1 << G.th_log2) - 1) >> G.th_log2;
    // units: whole planes while everything fits kUnitBins tiles, otherwise bands of tile rows (large maps)
    const int env_rows = opt.unit_rows;  // tests
    const long tiles_plane = (long)G.ntx * G.nty;
    if (tiles_plane * d.n_planes <= kUnitBins && !env_rows) {
        G.units_per_plane = 1;
        G.rows_per_unit = G.nty;
    } else {
        G.rows_per_unit = env_rows ? std::min(env_rows, G.nty) : std::max(1, 2048 / G.ntx);
        // at most kMaxUnits units per pass (a test override may ask for thinner bands than that allows)
        const int max_upp = std::max(1, kMaxUnits / d.n_planes);
        G.rows_per_unit = std::max(G.rows_per_unit, (G.nty + max_upp - 1) / max_upp);
        G.units_per_plane = (G.nty + G.rows_per_unit - 1) / G.rows_per_unit;
    }
    G.tiles_per_unit = G.rows_per_unit * G.ntx;
    G.n_units = d.n_planes * G.units_per_plane;
    const long nb = (long)G.n_units * G.tiles_per_unit;
    if (G.n_units > kMaxUnits || G.tiles_per_unit > 8192 || nb > kMaxBins)
        return false;
    G.nbins = (int)nb;
    // tuning overrides (tile_log2 / tile_h_log2 / bin_batch): the batch must keep every
    // workgroup's first particle 16-byte aligned (multiple of 4; kept at a multiple of 1024) and fit the 16-bit
    // per-workgroup counters
    G.batch = env_b ? std::min(std::max((env_b / 1024) * 1024, 1024), 64512) : kBinBatch;
    G.batch = std::min(G.batch, std::max(1024, 65535 / reps / 1024 * 1024));  // lateral replicas multiply the records
    G.region = G.batch * reps;
    if (G.tw_log2 < 3 || G.tw_log2 > 8 || G.th_log2 < 3 || G.th_log2 > 8)
        return false;
    return true;
}

// persistent K3 workgroups: two per CU (their LDS and registers allow it), so that one workgroup's load / LDS /
// store phases overlap the other's; option k3_per_cu overrides (tuning knob)
static int scatter_workgroups(slicer_handle h)
{
    return h->num_cus * std::max(1, h->opt.k3_per_cu);
}

int run_box_sweep(slicer_handle h, double box, unsigned out[9])
{
    if (!h->d_sweep)
        HIPCHK(h, hipMalloc((void **)&h->d_sweep, 9 * sizeof(unsigned)));
    HIPCHK(h, hipMemsetAsync(h->d_sweep, 0, 9 * sizeof(unsigned), h->stream));
    HIPCHK(h, launch_check_box_quotient(box, h->d_sweep, h->stream));
    out[0] = 1;
    HIPCHK(h, hipMemcpyAsync(out, h->d_sweep, 9 * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SLICER_OK;
}

// Has the f32 form of r / box (k_project_bin_fast) been proven for this box size?  One exhaustive device sweep over
// all 2^31 non-negative floats per distinct box size and handle (a few milliseconds), then cached.
int box_quotient_ok(slicer_handle h, double box, bool &ok)
{
    for (auto &v : h->box_verdicts)
        if (v.first == box) {
            ok = v.second;
            return SLICER_OK;
        }
    unsigned out[9];
    int rc = run_box_sweep(h, box, out);
    if (rc)
        return rc;
    ok = out[0] == 0;
    h->box_verdicts.emplace_back(box, ok);
    return SLICER_OK;
}

// Maps that are not a power of two wide: may the grid arithmetic use quot_dl3 instead of f64 divisions by dl = 1/npix?
// One exhaustive device sweep (2^30 operands, ~1 ms) per distinct npix and handle, cached.  Option dl_quot = 0 says no.
int dl_quotient_ok(slicer_handle h, int npix, bool &ok, unsigned *examples9 = nullptr)
{
    if (!examples9)
        for (auto &v : h->dl_verdicts)
            if (v.first == npix) {
                ok = v.second;
                return SLICER_OK;
            }
    if (!h->d_sweep)
        HIPCHK(h, hipMalloc((void **)&h->d_sweep, 9 * sizeof(unsigned)));
    HIPCHK(h, hipMemsetAsync(h->d_sweep, 0, 9 * sizeof(unsigned), h->stream));
    HIPCHK(h, launch_check_dl_quotient(1. / double(npix), h->d_sweep, h->stream));
    unsigned out[9] = {1};
    HIPCHK(h, hipMemcpyAsync(out, h->d_sweep, sizeof out, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    ok = out[0] == 0;
    if (examples9)
        memcpy(examples9, out, sizeof out);
    else
        h->dl_verdicts.emplace_back(npix, ok);
    return SLICER_OK;
}

// Kernel arguments of k_project_bin_fast and whether this (file, pass) qualifies for it; see the conditions in
// slicer_project_bin.hip.  Option k1_general = 1 forces the general kernel (tests run both).
int k1_fast_args(slicer_handle h, const PassParams &P, const BinGeom &G, int nblocks, K1Args &A, bool &fast)
{
    memset(&A, 0, sizeof A);
    fast = false;
    if (h->opt.k1_general || P.n_planes > 4 || !(P.lim < 1.5) || G.region != G.batch ||
        (uint64_t)G.n_units * (uint64_t)nblocks * (uint64_t)G.region >= (1ull << 31))
        return SLICER_OK;
    for (int p = 0; p < P.n_planes; p++)
        if (P.nrep[p] != 0)
            return SLICER_OK;
    for (int p = 0; p + 1 < P.n_planes; p++)  // consecutive slabs (the planes of one box replication)
        if (!(P.zhi[p] == P.zlo[p + 1] && P.zlo[p] <= P.zhi[p]))
            return SLICER_OK;
    if (!(P.rcase >= 0.0f) || !std::isfinite(P.rcase) || !std::isfinite((float)P.box) || (float)P.box <= 0.0f)
        return SLICER_OK;
    for (int a = 0; a < 3; a++) {
        const double c = P.c0[a];
        // the recentring runs in f32: exact iff the centre is an f32 value (rand()/float(RAND_MAX) is one,
        // densitymaps.cpp:188-190)
        // (centres below 2^-20 -- e.g. the exact 0 of -DUSE_FIXED_PLC_VERTEX -- are where the reference's -0.0 and the
        // last bit of a quotient below 2^-100 could reach the result: left to the general kernel)
        if (!((double)(float)c == c) || !(c >= 0x1p-20 && c <= 1.0))
            return SLICER_OK;
    }
    bool ok = false;
    int rc = box_quotient_ok(h, P.box, ok);
    if (rc)
        return rc;
    if (!ok)
        return SLICER_OK;
    A.boxf = (float)P.box;
    A.rb = 1.0f / A.boxf;
    for (int a = 0; a < 3; a++) {
        const float sg = P.sgn[P.perm[a]];
        A.ws[a] = sg;
        A.wo[a] = sg < 0.0f ? 1.0f : 0.0f;
        A.c0f[a] = (float)P.c0[a];
    }
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}, {1, 0, 2}, {2, 0, 1}, {2, 1, 0}};
    A.face = 0;
    for (int f = 0; f < 6; f++)
        if (perms[f][0] == P.perm[0] && perms[f][1] == P.perm[1] && perms[f][2] == P.perm[2])
            A.face = f;
    A.rcase = P.rcase;
    A.n_planes = P.n_planes;
    {
        // expected fraction of particles that reach the projection: the slabs' share of the unit box depth
        // (positions are uniform in z to first order); option k1_stack = 0 / 1 overrides
        double depth = 0;
        for (int p = 0; p < P.n_planes; p++)
            depth += std::max(0.0, std::min<double>(P.zhi[p], P.rcase + 1.0) - std::max<double>(P.zlo[p], P.rcase));
        A.stack = h->opt.k1_stack >= 0 ? h->opt.k1_stack : (depth < 0.6 ? 1 : 0);
    }
    for (int p = 0; p < 4; p++)
        A.zlo[p] = P.zlo[p];  // +inf beyond n_planes (make_params)
    A.zlast = P.zhi[P.n_planes - 1];
    const double tl = std::tan(P.lim);
    A.k_ra = ceil_to_f32(tl * (1.0 + 3e-5));
    A.eps_ra = 2e-6f;
    A.k_dec = ceil_to_f32(tl * std::sqrt(1.0 + (double)A.k_ra * (double)A.k_ra) * (1.0 + 3e-5));
    A.eps_dec = ceil_to_f32(tl * 2.2e-6 + 1e-6);
    // both series of the fast kernel run on tangents: |tan ra| <= k_ra and |tan dec| = |X| / sqrt(Y^2 + Z^2) <= |X| / Z <=
    // k_dec (plus the pre-test's absolute slack) for every entry that passes the pre-test; entries beyond the chosen
    // range (15 terms: 0.3125) are left to the exact epilogue by the kernel
    A.series_max = (!(h->desc.debug_flags & 2) && (double)A.k_dec * 1.001 + 1e-4 < kSeriesMax9) ? kSeriesMax9 : kSeriesMax15;
    A.lim = P.lim;
    A.inv_fov = P.inv_fov;
    A.nn_f = P.nn_f;
    A.nn_d = P.nn_d;
    A.nn = P.nn;
    A.pow2 = P.pow2;
    fast = true;
    return SLICER_OK;
}

int ensure_bin_workspace(slicer_handle h, bool has_mass, int group, int slot, uint64_t n, const BinGeom &G,
                         BinWorkspace &W)
{
    auto &Q = h->pg[group];
    const uint64_t nb = (n + G.batch - 1) / G.batch;
    const uint64_t region = (uint64_t)G.n_units * nb * G.region;  // compact records: [unit][workgroup][region]
    const uint64_t nrec = n * (uint64_t)(G.region / G.batch);     // most records this chunk can emit
    int rc;
    if ((rc = ensure(h, h->w_cxy, region * 8)) || (rc = ensure(h, h->w_cbin, region * 2)) ||
        (rc = ensure(h, h->w_hist, nb * (uint64_t)G.nbins * 4)) ||
        (rc = ensure(h, h->w_hist16, nb * (uint64_t)(G.nbins + 2) * 2)) ||
        (rc = ensure(h, h->w_total, (kMaxBins + kMaxBins / 32 + 1) * 4)) ||
        (rc = ensure(h, h->w_bcount, nb * kMaxUnits * 4)) || (rc = ensure(h, Q.w_sxy[slot], nrec * (has_mass ? 12 : 8))) ||  // float2, or Rec3 with per-particle masses
        (rc = ensure(h, Q.w_base[slot], (kMaxBins + 1) * 4)))
        return rc;
    if (has_mass && (rc = ensure(h, h->w_cm, region * 4)))
        return rc;
    W.cxy = (float2 *)h->w_cxy.p;
    W.cbin = (unsigned short *)h->w_cbin.p;
    W.cm = (float *)h->w_cm.p;
    W.sxy = (float2 *)Q.w_sxy[slot].p;
    W.sm = has_mass ? (float *)Q.w_sxy[slot].p : nullptr;  // (the masses travel inside the 12-byte sorted records)
    W.hist = (unsigned *)h->w_hist.p;
    W.hist16 = (unsigned *)h->w_hist16.p;
    W.total = (unsigned *)h->w_total.p;
    W.base = (unsigned *)Q.w_base[slot].p;
    W.bcount = (unsigned *)h->w_bcount.p;
    return SLICER_OK;
}

// Two-level sort: the units of the pass become coarse bins -- bands of 2^crow_log2 tile rows of one plane -- chosen so
// that a pass has about 64 of them (runs of ~0.5 KB in the project+bin kernel's sub-batches as well as in the sort
// kernel's items) within the 8-bit ids of both kernels.  False if the pass does not fit (the one-level sort serves it).
bool sort2_geom(const BinGeom &G, int n_planes, BinGeom &G2, int &crow_log2)
{
    G2 = G;
    crow_log2 = 0;
    auto units = [&](int cl) { return n_planes * ((G.nty + (1 << cl) - 1) >> cl); };
    while (units(crow_log2) > 64 && (2 << crow_log2) * G.ntx <= kMaxCoarseTiles)
        crow_log2++;
    G2.rows_per_unit = 1 << crow_log2;
    G2.units_per_plane = (G.nty + G2.rows_per_unit - 1) / G2.rows_per_unit;
    G2.tiles_per_unit = G2.rows_per_unit * G.ntx;
    G2.n_units = n_planes * G2.units_per_plane;
    G2.nbins = G2.n_units * G2.tiles_per_unit;
    return G2.n_units <= kMaxCoarse && G2.tiles_per_unit <= kMaxCoarseTiles && G.region == G.batch && G.batch <= 32768;
}

constexpr int kSort2Slots = kSort2Blocks * kSubBatches;  // sub-batch slots per item of the sort kernel

int ensure_sort2_workspace(slicer_handle h, int group, int slot, uint64_t n, const BinGeom &G, int ngroups, BinWorkspace &W)
{
    auto &Q = h->pg[group];
    const uint64_t nb = (n + G.batch - 1) / G.batch, nslots = nb * kSubBatches;
    int rc;
    bool fresh_tot = false;
    if ((rc = ensure(h, h->w_c1, nb * (uint64_t)G.batch * 8)) || (rc = ensure(h, h->w_sboff, nslots * 4)) ||
        (rc = ensure(h, h->w_sbstart, (uint64_t)kSubRow * nslots * 2)) || (rc = ensure(h, h->w_sbn, nb * 4)) ||
        (rc = ensure(h, Q.w_sxy[slot], n * 8)) || (rc = ensure(h, Q.w_base[slot], (kMaxBins + 1) * 4)) ||
        (rc = ensure(h, Q.w_ptab[slot], (uint64_t)G.n_units * ngroups * (G.tiles_per_unit + 1) * 4)) ||
        (rc = ensure(h, Q.w_tot, (uint64_t)kMaxCoarse * kMaxCoarseTiles * 4, &fresh_tot)))
        return rc;
    memset(&W, 0, sizeof W);
    W.c1 = (float2 *)h->w_c1.p;
    W.sb_off = (unsigned *)h->w_sboff.p;
    W.sb_start = (unsigned short *)h->w_sbstart.p;
    W.sb_n = (unsigned *)h->w_sbn.p;
    W.sxy = (float2 *)Q.w_sxy[slot].p;
    W.ptab = (unsigned *)Q.w_ptab[slot].p;
    W.item_tot = (unsigned *)Q.w_base[slot].p;
    W.tot = (unsigned *)Q.w_tot.p;
    W.base = (unsigned *)Q.w_base[slot].p;
    if (slot == 0 || fresh_tot)  // a new pending list starts from zero totals (the tile launch's item builder re-zeroes them)
        HIPCHK(h, hipMemsetAsync(W.tot, 0, (size_t)G.nbins * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(W.item_tot, 0, (size_t)ngroups * G.n_units * 4, h->stream));
    return SLICER_OK;
}

bool ngp_foldable(slicer_handle h, int type)
{
    int species = 0;
    for (int t = 0; t < 6; t++)
        species += h->file.npart[t] > 0;
    return species == 1 && h->file.npart[type] > 0 && !h->file_partial_flush[type] && !h->opt.ngp_general;
}

// NGP: some of the open file's records of this species are (about to be) in the global count map, so none of them may
// be folded inside the tile kernel: the per-file sum needs the file's complete count per pixel (k_fold_ngp does it)
void ngp_spoil_file(slicer_handle h, int type)
{
    h->file_partial_flush[type] = true;
    for (auto &Q : h->pg)
        if (Q.L.n && Q.key < 12 && Q.key / 2 == type)
            for (int c = 0; c < Q.L.n; c++)
                if (!Q.L.done[c])
                    Q.L.fold[c] = 0;
}

// Deposit the pending (binned) chunks of one plane group with one tile-kernel launch.
int flush_group(slicer_handle h, int group)
{
    auto &Q = h->pg[group];
    if (Q.L.n == 0)
        return SLICER_OK;
    NgpFold F;
    memset(&F, 0, sizeof F);
    if (Q.cfg.mas == kNGP && Q.cfg.acc == kCountU32 && Q.key < 12) {
        const int ptype = Q.key / 2;
        for (int c = 0; c < Q.L.n; c++)
            if (!Q.L.done[c]) {  // a flush in mid-file: the open file's counts are partial
                ngp_spoil_file(h, ptype);
                break;
            }
        for (int c = 0; c < Q.L.n; c++)
            F.on |= Q.L.fold[c];
        for (int p = 0; p < Q.np; p++) {
            F.tot[p] = (float *)h->planes[Q.p0 + p].tot.p;
            F.toti[p] = h->desc.want_type_maps ? (float *)h->planes[Q.p0 + p].toti[ptype].p : nullptr;
        }
    }
    bool fresh = false;
    int rc = ensure(h, h->w_items, tile_items_bytes(Q.G, Q.particles), &fresh);
    if (rc)
        return rc;
    if (fresh) {  // fresh workspace: both work-item counters start at zero
        HIPCHK(h, hipMemsetAsync(h->w_items.p, 0, 16, h->stream));
        h->items_epoch = 0;
    }
    Q.L.run0[0] = 0;
    for (int c = 0; c < Q.L.n; c++)
        Q.L.run0[c + 1] = Q.L.run0[c] + (Q.L.ptab[c] ? Q.L.ngroups[c] : 1);
    Q.L.tot = Q.sort2 ? (unsigned *)Q.w_tot.p : nullptr;
    {
        ProfScope ps(h, KN_TILE);
        bool int_cells = false;
        HIPCHK(h, launch_tile_deposit(Q.cfg, Q.P, Q.G, Q.L, Q.T, F, h->w_items.p, h->items_epoch++, Q.particles,
                                      h->opt.k4_int, &int_cells, h->stream));
        if (int_cells)
            h->algo_mask |= 1 << 6;
    }
    Q.L.n = 0;
    Q.key = -1;
    Q.particles = 0;
    return SLICER_OK;
}

int flush_pending(slicer_handle h)
{
    for (int g = 0; g < SLICER_MAX_PLANES; g++) {
        int rc = flush_group(h, g);
        if (rc)
            return rc;
    }
    return SLICER_OK;
}

// End of a pass in process-global mode: libc gets its stream back, advanced by the pass's draws.
void pass_stream_return(slicer_handle h)
{
    if (h->rand_pass) {
        (void)libc_rand_put(h->rand_state);
        h->rand_pass = false;
    }
}

// The libc stream moves to the device for a run of thinned chunks: thin_rng_begin reads the process-global generator
// state and uploads it, thin_rng_end brings the advanced state back and installs it (one synchronisation).  False from
// begin: the stream stays on the host (option thin_host, a generator other than glibc's TYPE_3, or the layout check of
// slicer_rand.hip failed) and thin_chunk draws with rand() as the reference does.
bool thin_rng_begin(slicer_handle h, int &rc)
{
    rc = SLICER_OK;
    if (h->opt.thin_host)
        return false;
    uint32_t v[31];
    if (h->rand_private || h->rand_pass)
        memcpy(v, h->rand_state, sizeof v);
    else if (!libc_rand_grab(v))
        return false;
    if ((rc = ensure(h, h->w_randtab, rand_tables_bytes())) || (rc = ensure(h, h->w_randstate, 32 * 4)))
        return false;
    if (!h->rand_tab_ready) {
        if (rand_tables_upload(h->w_randtab.p, h->stream) != hipSuccess) {
            rc = fail(h, SLICER_ERR_HIP, "upload of the generator tables failed: %s", hipGetErrorString(hipGetLastError()));
            return false;
        }
        h->rand_tab_ready = true;
    }
    if (hipMemcpyAsync(h->w_randstate.p, v, sizeof v, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {  // v is a stack array
        rc = fail(h, SLICER_ERR_HIP, "upload of the generator state failed: %s", hipGetErrorString(hipGetLastError()));
        return false;
    }
    h->rand_on_device = true;
    return true;
}

int thin_rng_end(slicer_handle h)
{
    if (!h->rand_on_device)
        return SLICER_OK;
    h->rand_on_device = false;
    uint32_t v[31];
    HIPCHK(h, hipMemcpyAsync(v, h->w_randstate.p, sizeof v, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->rand_private || h->rand_pass)
        memcpy(h->rand_state, v, sizeof v);
    else if (!libc_rand_put(v))
        return fail(h, SLICER_ERR_STATE, "the process changed its libc generator during a thinned pass");
    return SLICER_OK;
}

// Shot-noise thinning of one chunk into plane slot 0 of (P, T).
int thin_chunk(slicer_handle h, PassParams P, const Targets &T, const LaunchCfg &cfg, const float *d_pos,
               const float *d_mass, uint64_t n)
{
    // densitymaps.cpp:387-397: one libc rand() per selected entry, in selection order.  Count on the device, then
    // either continue the process-global stream on the device (thin_rng_begin) or draw on the host from it -- the
    // same deviates either way, exactly what the reference consumes -- and deposit.
    P.series_max = kSeriesMax15;  // no pre-test on this path either
    const uint64_t nchunks = (n + 63) / 64;
    int rc;
    if ((rc = ensure(h, h->w_tcounts, nchunks * 4)) || (rc = ensure(h, h->w_tbase, (nchunks + 1) * 8)))
        return rc;
    {
        ProfScope ps(h, KN_DIRECT);
        HIPCHK(h, launch_thin_count(d_pos, n, P, (unsigned *)h->w_tcounts.p, (unsigned long long *)h->w_tbase.p,
                                    h->d_neg, h->stream));
    }
    const double pw = std::pow(2, h->desc.snopt);
    const unsigned long long *d_nsel = (unsigned long long *)h->w_tbase.p + nchunks;
    const uint64_t reps = (uint64_t)(2 * P.nrep[0] + 1) * (uint64_t)(2 * P.nrep[0] + 1);
    if (h->rand_on_device) {
        // at most one draw per (particle, replica); with lateral replicas the buffer is sized by the real count
        unsigned long long max_draws = n * reps;
        if (reps > 1) {
            HIPCHK(h, hipMemcpyAsync(&max_draws, d_nsel, sizeof max_draws, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        if ((rc = ensure(h, h->w_urand, std::max<size_t>(max_draws, 1) * 4)) ||
            (rc = ensure(h, h->w_randwaves, rand_wave_states_bytes(max_draws))))
            return rc;
        ProfScope ps(h, KN_DIRECT);
        HIPCHK(h, launch_rand_deviates(d_nsel, (uint32_t *)h->w_randstate.p, (uint32_t *)h->w_randwaves.p,
                                       h->w_randtab.p, (float *)h->w_urand.p, max_draws, h->stream));
        HIPCHK(h, launch_thin_deposit(cfg, d_pos, d_mass, n, P, T, (const unsigned long long *)h->w_tbase.p,
                                      (const float *)h->w_urand.p, 1. / pw, pw, h->stream));
        h->algo_mask |= (1 << 3) | (1 << 8);
        return SLICER_OK;
    }
    unsigned long long nsel = 0;
    HIPCHK(h, hipMemcpyAsync(&nsel, d_nsel, sizeof nsel, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->h_urand.resize(nsel);
    if (h->rand_private || h->rand_pass)
        libc_rand_model_fill(h->rand_state, h->h_urand.data(), nsel);
    else
        for (unsigned long long k = 0; k < nsel; k++)
            h->h_urand[k] = rand() / float(RAND_MAX);
    if ((rc = ensure(h, h->w_urand, std::max<size_t>(nsel, 1) * 4)))
        return rc;
    if (nsel)
        HIPCHK(h, hipMemcpyAsync(h->w_urand.p, h->h_urand.data(), nsel * 4, hipMemcpyHostToDevice, h->stream));
    {
        ProfScope ps(h, KN_DIRECT);
        HIPCHK(h, launch_thin_deposit(cfg, d_pos, d_mass, n, P, T, (const unsigned long long *)h->w_tbase.p,
                                      (const float *)h->w_urand.p, 1. / pw, pw, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));  // h_urand is reused by the next chunk
    h->algo_mask |= 1 << 3;
    return SLICER_OK;
}

bool thin_deferred(slicer_handle h) { return h->desc.snopt > 0 && h->desc.n_planes > 1; }

void thin_drop(slicer_handle h)
{
    for (auto &c : h->thin_chunks) {
        release(c.pos);
        release(c.mass);
    }
    h->thin_chunks.clear();
    h->thin_files.clear();
}

// NGP: fold this file's per-type count / mass maps of plane p into its float maps (densitymaps.cpp:405-412 adds each
// file's mapxyi into the running maps)
int fold_file_plane(slicer_handle h, int p)
{
    bool any = false;
    for (int t = 0; t < 6; t++)
        any |= h->file_mode[t] != 0;
    if (!any)
        return SLICER_OK;
    FoldArgs A;
    memset(&A, 0, sizeof A);
    for (int t = 0; t < 6; t++) {
        A.mode[t] = h->file_mode[t];
        A.mconst[t] = h->file_mconst[t];
        A.scratch[t] = h->file_mode[t] ? h->planes[p].acc[t].p : nullptr;
        A.toti[t] = (h->file_mode[t] && h->desc.want_type_maps) ? (float *)h->planes[p].toti[t].p : nullptr;
    }
    A.tot = (float *)h->planes[p].tot.p;
    A.npix2 = h->npix2;
    ProfScope ps(h, KN_FOLD);
    HIPCHK(h, launch_fold_ngp(A, h->stream));
    return SLICER_OK;
}

// snopt > 0 with several planes: deposit the retained chunks plane-major, files and species in their original order
// inside each plane -- the order in which the reference (one createDensityMaps call per plane) consumes rand().
static int thin_replay_chunks(slicer_handle h);

// Called wherever a pass ends (flush / finalize / read): deposits the retained chunks of a multi-plane thinned pass and
// hands libc its stream back.
int thin_replay(slicer_handle h)
{
    const int rc = thin_replay_chunks(h);
    pass_stream_return(h);
    return rc;
}

static int thin_replay_chunks(slicer_handle h)
{
    if (!thin_deferred(h) || (h->thin_chunks.empty() && h->thin_files.empty()))
        return SLICER_OK;
    const slicer_plane_desc &d = h->desc;
    const slicer_file_desc file_saved = h->file;
    int rc = SLICER_OK;
    thin_rng_begin(h, rc);  // (false: the deviates come from the host loop)
    for (int p = 0; p < d.n_planes && !rc; p++) {
        size_t ci = 0;
        for (size_t f = 0; f < h->thin_files.size() && !rc; f++) {
            const auto &F = h->thin_files[f];
            h->file = F.file;
            for (int t = 0; t < 6; t++) {
                h->file_mode[t] = F.mode[t];
                h->file_mconst[t] = F.mconst[t];
            }
            for (; ci < h->thin_chunks.size() && h->thin_chunks[ci].file == (int)f && !rc; ci++) {
                const auto &c = h->thin_chunks[ci];
                const bool has_mass = c.mass.p != nullptr;
                PassParams P;
                make_params(h, c.type, has_mass, P);
                Targets T;
                fill_targets(h, c.type, has_mass, T);
                P.zlo[0] = P.zlo[p];
                P.zhi[0] = P.zhi[p];
                P.nrep[0] = P.nrep[p];
                T.acc[0] = T.acc[p];
                T.nsel[0] = T.nsel[p];
                LaunchCfg cfg{d.mas == SLICER_MAS_NGP ? kNGP : kTSC, acc_kind(d, has_mass), has_mass};
                rc = thin_chunk(h, P, T, cfg, (const float *)c.pos.p, (const float *)c.mass.p, c.n);
            }
            if (!rc && d.mas == SLICER_MAS_NGP)
                rc = fold_file_plane(h, p);
        }
    }
    const int rce = thin_rng_end(h);
    rc = rc ? rc : rce;
    h->file = file_saved;
    for (int t = 0; t < 6; t++)
        h->file_mode[t] = 0;
    thin_drop(h);
    return rc;
}

// One chunk through K1-K3 for the planes [p0, p0 + np) of the pass (P and T already hold them in slots 0 .. np - 1);
// the sorted records wait in the pending list for the tile kernel.
int binned_chunk(slicer_handle h, const LaunchCfg &cfg, const PassParams &P, const Targets &T, BinGeom G, int type,
                 int group, int p0, int np, const float *d_pos, const float *d_mass, uint64_t n)
{
    auto &Q = h->pg[group];
    const slicer_plane_desc &d = h->desc;
    const bool has_mass = d_mass != nullptr;
    if (!h->opt.bin_batch) {
        // K1 keeps two workgroups per CU resident: size the batch so that the workgroups of this call fill whole
        // rounds of resident slots instead of leaving a short tail round
        const uint64_t slots = 2ull * (uint64_t)h->num_cus;
        const uint64_t rounds = (n + slots * kBinBatch - 1) / (slots * kBinBatch);
        const uint64_t per = (n + slots * rounds - 1) / (slots * rounds);
        const int reps = G.region / G.batch;
        G.batch = (int)std::min<uint64_t>(G.batch, std::max<uint64_t>(std::min(8192, G.batch), (per + 1023) / 1024 * 1024));
        G.region = G.batch * reps;
    }
    const bool shared = d.mas != SLICER_MAS_NGP && !d.want_type_maps;
    const int key = (shared ? 12 : type * 2) + (has_mass ? 1 : 0);
    int rc;
    int nblocks = (int)((n + G.batch - 1) / G.batch);
    K1Args A;
    bool fast = false;
    if ((rc = k1_fast_args(h, P, G, nblocks, A, fast)))
        return rc;
    // two-level sort where the pass qualifies: the fast project+bin kernel without the wave stacks, constant mass, a unit
    // table within the 8-bit ids, at most kMaxSortGroups items per unit
    BinGeom G2;
    int crow_log2 = 0;
    const bool sort2 = fast && h->opt.sort2 && !has_mass && !A.stack && sort2_geom(G, P.n_planes, G2, crow_log2) &&
                       (nblocks * kSubBatches + kSort2Slots - 1) / kSort2Slots <= kMaxSortGroups;
    if (sort2) {
        G = G2;
        A.sort2 = 1;
        A.crow_log2 = crow_log2;
    }
    if (Q.L.n && (Q.key != key || Q.p0 != p0 || Q.np != np || Q.L.n >= Q.limit || Q.sort2 != sort2) &&
        (rc = flush_group(h, group)))
        return rc;
    const int slot = Q.L.n;
    BinWorkspace W;
    h->algo_mask |= fast ? (1 << 4) : (1 << 5);
    if (sort2) {
        const int ngroups = (nblocks * kSubBatches + kSort2Slots - 1) / kSort2Slots;
        if ((rc = ensure_sort2_workspace(h, group, slot, n, G, ngroups, W)))
            return rc;
        h->algo_mask |= 1 << 7;
        {
            ProfScope ps(h, KN_PROJECT);
            HIPCHK(h, launch_project_bin(cfg, true, d_pos, d_mass, n, P, A, G, W, T, h->stream));
        }
        {
            ProfScope ps(h, KN_SORT2);
            HIPCHK(h, launch_sort2(nblocks, kSort2Slots, ngroups, scatter_workgroups(h), P, G, W, h->stream));
        }
        Q.L.ptab[slot] = W.ptab;
        Q.L.ngroups[slot] = ngroups;
    } else {
        if ((rc = ensure_bin_workspace(h, has_mass, group, slot, n, G, W)))
            return rc;
        {
            ProfScope ps(h, KN_PROJECT);
            HIPCHK(h, launch_project_bin(cfg, fast, d_pos, d_mass, n, P, A, G, W, T, h->stream));
        }
        {
            ProfScope ps(h, KN_SCAN);
            HIPCHK(h, launch_bin_scan(cfg, nblocks, P.n_planes, G, W, T, h->stream));
        }
        {
            ProfScope ps(h, KN_SCATTER);
            HIPCHK(h, launch_bin_scatter(cfg, nblocks, P.n_planes, scatter_workgroups(h), G, W, T, h->stream));
        }
        Q.L.ptab[slot] = nullptr;
        Q.L.ngroups[slot] = 1;
    }
    if (slot == 0) {
        // chunks per tile launch: enough for ~16384 records per bin (what eight chunks of the headline case bring),
        // judged by the first chunk; option `pending` overrides
        const uint64_t per_bin = std::max<uint64_t>(1, n * (uint64_t)(G.region / G.batch) / (uint64_t)std::max(1, G.nbins));
        int limit = (int)std::min<uint64_t>(kMaxPending, std::max<uint64_t>(8, (16384 + per_bin - 1) / per_bin));
        if (h->opt.pending > 0)
            limit = std::min(h->opt.pending, kMaxPending);
        Q.limit = sort2 ? std::min(limit, kMaxPendingRuns) : limit;
        Q.key = key;
        Q.sort2 = sort2;
        Q.p0 = p0;
        Q.np = np;
        Q.cfg = cfg;
        Q.P = P;
        Q.G = G;
        Q.T = T;
    }
    Q.L.sxy[slot] = W.sxy;
    Q.L.sm[slot] = has_mass ? W.sm : nullptr;
    Q.L.base[slot] = W.base;
    Q.L.mconst[slot] = P.mconst;
    Q.L.file_id[slot] = (unsigned short)h->file_serial;
    Q.L.done[slot] = 0;
    Q.L.fold[slot] = cfg.mas == kNGP && cfg.acc == kCountU32 && ngp_foldable(h, type);
    if (Q.L.fold[slot] && G.tw_log2 + G.th_log2 > 14) {  // the tile kernel keeps 16 pixels per lane (tile size overrides)
        ngp_spoil_file(h, type);
        Q.L.fold[slot] = 0;
    }
    Q.L.sm_const[slot] = P.sm_const;
    Q.L.n = slot + 1;
    Q.particles += n * (uint64_t)(G.region / G.batch);  // bounds the records behind the pending chunks
    return SLICER_OK;
}

int deposit_device_chunk(slicer_handle h, int type, const float *d_pos, const float *d_mass, uint64_t n)
{
    const slicer_plane_desc &d = h->desc;
    const bool has_mass = d_mass != nullptr;
    PassParams P;
    make_params(h, type, has_mass, P);
    Targets T;
    fill_targets(h, type, has_mass, T);
    LaunchCfg cfg{d.mas == SLICER_MAS_NGP ? kNGP : kTSC, acc_kind(d, has_mass), has_mass};
    if (d.snopt > 0) {
        if (d.n_planes == 1) {
            if (d.mas == SLICER_MAS_NGP)
                ngp_spoil_file(h, type);  // counts into the global map: the file's fold is k_fold_ngp's, not the tile kernel's
            int rc = SLICER_OK;
            thin_rng_begin(h, rc);
            if (!rc)
                rc = thin_chunk(h, P, T, cfg, d_pos, d_mass, n);
            const int rce = thin_rng_end(h);  // the host's stream is current again before the call returns
            return rc ? rc : rce;
        }
        // several planes: keep the chunk, thin_replay deposits it once per plane in the reference's order
        slicer_handle_s::ThinChunk c{};
        c.file = (int)h->thin_files.size();
        c.type = type;
        c.n = n;
        int rc = ensure(h, c.pos, n * 12);
        if (!rc && has_mass)
            rc = ensure(h, c.mass, n * 4);
        if (rc) {
            release(c.pos);
            release(c.mass);
            return rc;
        }
        HIPCHK(h, hipMemcpyAsync(c.pos.p, d_pos, n * 12, hipMemcpyDeviceToDevice, h->stream));
        if (has_mass)
            HIPCHK(h, hipMemcpyAsync(c.mass.p, d_mass, n * 4, hipMemcpyDeviceToDevice, h->stream));
        h->thin_chunks.push_back(c);
        return SLICER_OK;
    }
    // One pass of the binned pipeline holds at most kMaxBins (plane, tile) bins and needs disjoint slabs.  A pass beyond
    // that (four 16384^2 planes; overlapping slabs) takes its planes in groups, each group a binned sub-pass over the
    // same chunk, before the fused global-atomic kernel is considered.
    int gsize = d.n_planes;
    BinGeom G;
    auto fits = [&](int p0, int np, BinGeom &Gs) {
        slicer_plane_desc sub = d;
        sub.n_planes = np;
        for (int j = 0; j < np; j++) {
            sub.ld[j] = d.ld[p0 + j];
            sub.ld2[j] = d.ld2[p0 + j];
            sub.nrepperp[j] = d.nrepperp[p0 + j];
        }
        return choose_geom(sub, cfg.acc, h->opt, Gs) && scatter_lds_bytes(Gs, has_mass) <= 160 * 1024 - 256;
    };
    bool binned = d.algo != SLICER_ALGO_DIRECT && fits(0, d.n_planes, G);
    if (!binned && d.algo != SLICER_ALGO_DIRECT)
        for (int g = d.n_planes - 1; g >= 1 && !binned; g--) {
            bool ok = true;
            for (int p0 = 0; p0 < d.n_planes && ok; p0 += g)
                ok = fits(p0, std::min(g, d.n_planes - p0), G);
            if (ok) {
                binned = true;
                gsize = g;
            }
        }
    if (!binned && d.algo == SLICER_ALGO_BINNED)
        return fail(h, SLICER_ERR_UNSUPPORTED,
                    "SLICER_ALGO_BINNED cannot serve this pass (a tile table beyond the limits even for a single "
                    "plane); SLICER_ALGO_AUTO falls back to the fused global-atomic kernel");
    if (binned && d.algo == SLICER_ALGO_AUTO && n < 65536)
        binned = false;  // several launches are not worth it for a tiny chunk
    h->algo_mask |= 1 << (binned ? SLICER_ALGO_BINNED : SLICER_ALGO_DIRECT);
    if (!binned) {
        if (d.mas == SLICER_MAS_NGP)
            ngp_spoil_file(h, type);  // counts into the global map
        ProfScope ps(h, KN_DIRECT);
        P.series_max = kSeriesMax15;  // no pre-test on this path: entries far outside the field reach project()
        HIPCHK(h, launch_direct(cfg, d_pos, d_mass, n, P, T, h->stream));
        return SLICER_OK;
    }
    for (int p0 = 0; p0 < d.n_planes; p0 += gsize) {
        const int np = std::min(gsize, d.n_planes - p0);
        PassParams Pg = P;
        Targets Tg = T;
        if (np != d.n_planes) {  // this group's planes move to the front
            fits(p0, np, G);
            Pg.n_planes = np;
            for (int j = 0; j < np; j++) {
                Pg.zlo[j] = P.zlo[p0 + j];
                Pg.zhi[j] = P.zhi[p0 + j];
                Pg.nrep[j] = P.nrep[p0 + j];
                Tg.acc[j] = T.acc[p0 + j];
                Tg.nsel[j] = T.nsel[p0 + j];
            }
            for (int j = np; j < kMaxPlanes; j++) {  // as make_params leaves the slots beyond the pass
                Pg.zlo[j] = INFINITY;
                Pg.zhi[j] = -INFINITY;
                Pg.nrep[j] = 0;
                Tg.acc[j] = nullptr;
                Tg.nsel[j] = nullptr;
            }
        }
        int nr = 0;
        for (int j = 0; j < np; j++)
            nr = std::max(nr, Pg.nrep[j]);
        const int nwin = rep_windows(nr), ws = rep_window_side(nr);
        for (int wi = 0; wi < nwin; wi++)
            for (int wj = 0; wj < nwin; wj++) {
                if (nwin > 1) {
                    Pg.rep_i0 = -nr + wi * ws;
                    Pg.rep_i1 = std::min(nr, Pg.rep_i0 + ws - 1);
                    Pg.rep_j0 = -nr + wj * ws;
                    Pg.rep_j1 = std::min(nr, Pg.rep_j0 + ws - 1);
                }
                int rc = binned_chunk(h, cfg, Pg, Tg, G, type, p0 / gsize, p0, np, d_pos, d_mass, n);
                if (rc)
                    return rc;
            }
    }
    return SLICER_OK;
}

// Device -> host copy of a map into the caller's (pageable) array.  (Pinning the destination with hipHostRegister
// for the duration of the copy was measured on MI355X and bought nothing -- 8.5 ms per createDensityMaps call either
// way: registering 64 MiB costs what the direct DMA saves -- so the plain copy stays.)
int copy_map_to_host(slicer_handle h, void *dst, const void *d_src, size_t bytes)
{
    HIPCHK(h, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return SLICER_OK;
}

int check_deposit_args(slicer_handle h, int type, const void *pos, const void *mass, uint64_t n)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!h->in_plane || !h->in_file)
        return fail(h, SLICER_ERR_STATE, "deposit outside slicer_plane_begin/slicer_file_begin");
    if (type < 0 || type > 5)
        return fail(h, SLICER_ERR_ARG, "particle type %d out of range 0..5", type);
    if (n && !pos)
        return fail(h, SLICER_ERR_ARG, "null position pointer with n = %llu", (unsigned long long)n);
    (void)mass;
    return SLICER_OK;
}

int begin_type(slicer_handle h, int type, bool has_mass)
{
    const slicer_plane_desc &d = h->desc;
    int rc = prepare_type(h, type, has_mass);
    if (rc)
        return rc;
    if (d.mas == SLICER_MAS_NGP) {
        int mode = has_mass ? 2 : 1;
        if (h->file_mode[type] && h->file_mode[type] != mode)
            return fail(h, SLICER_ERR_ARG, "type %d deposited both with and without per-particle masses in one file",
                        type);
        h->file_mode[type] = mode;
        h->file_mconst[type] = (float)h->file.massarr[type];
        if (d.snopt > 0)  // kept entries carry (float)(pow(2, snopt) * m)   densitymaps.cpp:394
            h->file_mconst[type] = (float)(std::pow(2, d.snopt) * (double)(float)h->file.massarr[type]);
    }
    return SLICER_OK;
}

int ensure_staging(slicer_handle h, bool need_mass)
{
    uint64_t cap = h->max_chunk;
    if (h->stage_cap < cap) {
        for (int i = 0; i < 2; i++) {
            if (h->h_stage[i])
                (void)hipHostFree(h->h_stage[i]);
            if (h->d_stage[i])
                (void)hipFree(h->d_stage[i]);
            h->h_stage[i] = h->d_stage[i] = nullptr;
            if (h->h_mstage[i])
                (void)hipHostFree(h->h_mstage[i]);
            if (h->d_mstage[i])
                (void)hipFree(h->d_mstage[i]);
            h->h_mstage[i] = h->d_mstage[i] = nullptr;
            HIPCHK(h, hipHostMalloc((void **)&h->h_stage[i], cap * 12, hipHostMallocDefault));
            HIPCHK(h, hipMalloc((void **)&h->d_stage[i], cap * 12));
            if (!h->stage_free[i])
                HIPCHK(h, hipEventCreateWithFlags(&h->stage_free[i], hipEventDisableTiming));
        }
        h->stage_cap = cap;
    }
    if (need_mass && !h->h_mstage[0]) {
        for (int i = 0; i < 2; i++) {
            HIPCHK(h, hipHostMalloc((void **)&h->h_mstage[i], h->stage_cap * 4, hipHostMallocDefault));
            HIPCHK(h, hipMalloc((void **)&h->d_mstage[i], h->stage_cap * 4));
        }
    }
    return SLICER_OK;
}

}  // namespace

extern "C" {

int slicer_version(void) { return SLICER_AMD_VERSION; }

const char *slicer_last_error(slicer_handle h) { return h ? h->err.c_str() : g_null_err.c_str(); }

int slicer_create(int device, uint64_t max_chunk, slicer_handle *out)
{
    if (!out)
        return fail(nullptr, SLICER_ERR_ARG, "slicer_create: out is null");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SLICER_ERR_NO_DEVICE, "no HIP device available (%s)",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return fail(nullptr, SLICER_ERR_ARG, "device %d out of range (have %d)", device, ndev);
    slicer_handle h = new (std::nothrow) slicer_handle_s;
    if (!h)
        return fail(nullptr, SLICER_ERR_NOMEM, "out of host memory");
    h->device = device;
    for (const OptionName &o : kOptionNames)  // the environment seeds the knobs once; slicer_set_option changes them
        h->opt.*(o.field) = env_int(o.env, h->opt.*(o.field));
    // record cursors are 32-bit: one kernel pass carries at most 2^30 particles
    h->max_chunk = max_chunk ? std::min<uint64_t>(max_chunk, 1ull << 30) : (1ull << 24);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->own) != hipSuccess ||
        // one block: the selected-entry counters, the guard flag, the mass maxima (one memset per pass clears them)
        hipMalloc((void **)&h->d_counts, kPassScalarsBytes) != hipSuccess) {
        int rc = fail(nullptr, SLICER_ERR_HIP, "device %d initialisation failed: %s", device,
                      hipGetErrorString(hipGetLastError()));
        delete h;
        return rc;
    }
    h->d_neg = reinterpret_cast<int *>(h->d_counts + SLICER_MAX_PLANES * 6);
    h->d_maxmass = reinterpret_cast<unsigned *>(h->d_neg + 1);
    h->stream = h->own;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        h->num_cus = cus;
    *out = h;
    return SLICER_OK;
}

int slicer_destroy(slicer_handle h)
{
    if (!h)
        return SLICER_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    prof_collect(h);
    thin_drop(h);
    pass_stream_return(h);
    for (auto e : h->ev_pool)
        (void)hipEventDestroy(e);
    for (auto &pl : h->planes) {
        release(pl.tot);
        release(pl.acc_shared);
        for (int t = 0; t < 6; t++) {
            release(pl.toti[t]);
            release(pl.acc[t]);
        }
    }
    if (h->d_sweep)
        (void)hipFree(h->d_sweep);
    for (DevBuf *b : {&h->w_cxy, &h->w_cbin, &h->w_cm, &h->w_hist, &h->w_hist16, &h->w_total, &h->w_bcount, &h->w_items,
                      &h->w_tcounts, &h->w_tbase, &h->w_urand, &h->w_randtab, &h->w_randstate, &h->w_randwaves, &h->w_c1, &h->w_sboff, &h->w_sbstart, &h->w_sbn})
        release(*b);
    for (auto &Q : h->pg) {
        release(Q.w_tot);
        for (int i = 0; i < kMaxPending; i++) {
            release(Q.w_sxy[i]);
            release(Q.w_base[i]);
            release(Q.w_ptab[i]);
        }
    }
    for (int i = 0; i < 2; i++) {
        if (h->h_stage[i]) (void)hipHostFree(h->h_stage[i]);
        if (h->d_stage[i]) (void)hipFree(h->d_stage[i]);
        if (h->h_mstage[i]) (void)hipHostFree(h->h_mstage[i]);
        if (h->d_mstage[i]) (void)hipFree(h->d_mstage[i]);
        if (h->stage_free[i]) (void)hipEventDestroy(h->stage_free[i]);
    }
    if (h->d_counts) (void)hipFree(h->d_counts);  // (d_neg and d_maxmass live in the same block)
    if (h->own) (void)hipStreamDestroy(h->own);
    delete h;
    return SLICER_OK;
}

int slicer_set_option(slicer_handle h, const char *key, int32_t value)
{
    if (!h || !key)
        return fail(h, SLICER_ERR_ARG, "null argument");
    // (a pass stays "open" until the next slicer_plane_begin so that its maps can be read; what must not see a knob
    // change is work in flight: an open file, or binned chunks still waiting for their tile launch)
    bool busy = h->in_file;
    for (auto &Q : h->pg)
        busy = busy || Q.L.n > 0;
    if (busy)
        return fail(h, SLICER_ERR_STATE, "slicer_set_option with deposits in flight (open file or pending chunks)");
    for (const OptionName &o : kOptionNames)
        if (!strcmp(o.key, key)) {
            h->opt.*(o.field) = value;
            return SLICER_OK;
        }
    return fail(h, SLICER_ERR_ARG, "unknown option '%s'", key);
}

int slicer_get_option(slicer_handle h, const char *key, int32_t *value)
{
    if (!h || !key || !value)
        return fail(h, SLICER_ERR_ARG, "null argument");
    for (const OptionName &o : kOptionNames)
        if (!strcmp(o.key, key)) {
            *value = h->opt.*(o.field);
            return SLICER_OK;
        }
    return fail(h, SLICER_ERR_ARG, "unknown option '%s'", key);
}

int slicer_rand_stream_set(slicer_handle h, const uint32_t *v31)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (h->in_file)
        return fail(h, SLICER_ERR_STATE, "slicer_rand_stream_set inside a file");
    pass_stream_return(h);
    h->rand_private = v31 != nullptr;
    if (v31)
        memcpy(h->rand_state, v31, sizeof h->rand_state);
    return SLICER_OK;
}

int slicer_rand_stream_get(slicer_handle h, uint32_t *v31)
{
    if (!h || !v31)
        return fail(h, SLICER_ERR_ARG, "null argument");
    if (!h->rand_private)
        return fail(h, SLICER_ERR_STATE, "the handle draws from the process-global stream (slicer_rand_stream_set)");
    memcpy(v31, h->rand_state, sizeof h->rand_state);
    return SLICER_OK;
}

int slicer_libc_rand_supported(void)
{
    uint32_t v[31];
    return libc_rand_grab(v) ? 1 : 0;
}

int slicer_libc_rand_state_get(uint32_t *v31)
{
    if (!v31)
        return fail(nullptr, SLICER_ERR_ARG, "null argument");
    return libc_rand_grab(v31) ? SLICER_OK : fail(nullptr, SLICER_ERR_UNSUPPORTED, "libc generator state not accessible");
}

int slicer_libc_rand_state_set(const uint32_t *v31)
{
    if (!v31)
        return fail(nullptr, SLICER_ERR_ARG, "null argument");
    return libc_rand_put(v31) ? SLICER_OK : fail(nullptr, SLICER_ERR_UNSUPPORTED, "libc generator state not accessible");
}

int slicer_set_stream(slicer_handle h, void *hip_stream)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (h->in_plane)
        return fail(h, SLICER_ERR_STATE, "cannot change stream inside a plane pass");
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own;
    return SLICER_OK;
}

int slicer_plane_begin(slicer_handle h, const slicer_plane_desc *desc)
{
    if (!h || !desc)
        return fail(h, SLICER_ERR_ARG, "null argument");
    if (desc->npix <= 0 || desc->npix > 65536)
        return fail(h, SLICER_ERR_ARG, "npix %d out of range", desc->npix);
    if (desc->n_planes < 1 || desc->n_planes > SLICER_MAX_PLANES)
        return fail(h, SLICER_ERR_ARG, "n_planes %d out of range 1..%d", desc->n_planes, SLICER_MAX_PLANES);
    if (desc->mas != SLICER_MAS_TSC && desc->mas != SLICER_MAS_NGP)
        return fail(h, SLICER_ERR_ARG, "unknown mass-assignment scheme %d", desc->mas);
    if (desc->accum < SLICER_ACC_F32 || desc->accum > SLICER_ACC_FIXED64)
        return fail(h, SLICER_ERR_ARG, "unknown accumulator %d", desc->accum);
    if (desc->snopt < 0 || desc->snopt > 30)
        return fail(h, SLICER_ERR_ARG, "snopt = %d out of range 0..30", desc->snopt);
    if (!(desc->fov_rad > 0))
        return fail(h, SLICER_ERR_ARG, "fov_rad must be > 0");
    for (int p = 0; p < desc->n_planes; p++)
        if (desc->nrepperp[p] < 0 || desc->nrepperp[p] > 8)
            return fail(h, SLICER_ERR_ARG, "nrepperp[%d] = %d out of range 0..8", p, desc->nrepperp[p]);
    // shot-noise thinning in process-global mode: libc's stream is read here, before this call touches the HIP runtime
    pass_stream_return(h);  // (a pass that was never read or flushed)
    if (desc->snopt > 0 && !h->rand_private && libc_rand_grab(h->rand_state))
        h->rand_pass = true;
    HIPCHK(h, hipSetDevice(h->device));
    h->desc = *desc;
    h->dl_quot_ok = false;
    if (!is_pow2(desc->npix) && h->opt.dl_quot) {
        int rcq = dl_quotient_ok(h, desc->npix, h->dl_quot_ok);
        if (rcq)
            return rcq;
    }
    h->npix2 = (uint64_t)desc->npix * (uint64_t)desc->npix;
    h->in_plane = true;
    h->in_file = false;
    h->finalized = false;
    h->shared_seen = false;
    h->fixed_shared_set = false;
    h->algo_mask = 0;
    thin_drop(h);
    h->neg_remote = false;
    for (auto &Q : h->pg) {
        Q.L.n = 0;
        Q.key = -1;
        Q.particles = 0;
    }
    for (int t = 0; t < 6; t++) {
        h->type_seen[t] = false;
        h->fixed_exp_set[t] = false;
        h->file_mode[t] = 0;
    }
    zero_begin(h);
    int rcp = SLICER_OK;
    for (int p = 0; p < desc->n_planes && !rcp; p++) {
        rcp = ensure(h, h->planes[p].tot, h->npix2 * 4);
        if (!rcp && desc->mas == SLICER_MAS_NGP)  // the NGP fold accumulates into tot; TSC finalize overwrites it
            rcp = zero_async(h, h->planes[p].tot.p, h->npix2 * 4);
    }
    const int rcz = zero_end(h);
    if (rcp || rcz)
        return rcp ? rcp : rcz;
    HIPCHK(h, hipMemsetAsync(h->d_counts, 0, kPassScalarsBytes, h->stream));  // counters, guard flag, mass maxima
    return SLICER_OK;
}

int slicer_file_begin(slicer_handle h, const slicer_file_desc *file)
{
    if (!h || !file)
        return fail(h, SLICER_ERR_ARG, "null argument");
    if (!h->in_plane || h->finalized)
        return fail(h, SLICER_ERR_STATE, "slicer_file_begin outside a plane pass");
    if (h->in_file)
        return fail(h, SLICER_ERR_STATE, "slicer_file_begin: previous file not ended");
    if (!(file->boxsize > 0))
        return fail(h, SLICER_ERR_ARG, "boxsize must be > 0");
    for (int a = 0; a < 3; a++)
        if (file->sgn[a] != 1 && file->sgn[a] != -1)
            return fail(h, SLICER_ERR_ARG, "sgn[%d] = %d is not +-1", a, file->sgn[a]);
    h->file = *file;
    h->in_file = true;
    h->file_serial++;
    for (int t = 0; t < 6; t++) {
        h->file_mode[t] = 0;
        h->file_partial_flush[t] = false;
    }
    return SLICER_OK;
}

int slicer_deposit_device(slicer_handle h, int type, const float *d_pos, const float *d_mass, uint64_t n)
{
    int rc = check_deposit_args(h, type, d_pos, d_mass, n);
    if (rc)
        return rc;
    if (n == 0)
        return SLICER_OK;
    HIPCHK(h, hipSetDevice(h->device));
    rc = begin_type(h, type, d_mass != nullptr);
    if (rc)
        return rc;
    // one kernel pass per max_chunk particles keeps the workspace bounded
    for (uint64_t off = 0; off < n; off += h->max_chunk) {
        uint64_t c = std::min<uint64_t>(h->max_chunk, n - off);
        rc = deposit_device_chunk(h, type, d_pos + 3 * off, d_mass ? d_mass + off : nullptr, c);
        if (rc)
            return rc;
    }
    return SLICER_OK;
}

namespace {
struct HostSpan {
    const float *pos;
    const float *mass;
};
int copy_fill(void *user, float *dst_pos, float *dst_mass, uint64_t first, uint64_t count)
{
    const HostSpan *s = static_cast<const HostSpan *>(user);
    memcpy(dst_pos, s->pos + 3 * first, count * 12);
    if (dst_mass)
        memcpy(dst_mass, s->mass + first, count * 4);
    return 0;
}
}  // namespace

int slicer_deposit_stream(slicer_handle h, int type, uint64_t n, int has_mass, slicer_fill_fn fill, void *user)
{
    int rc = check_deposit_args(h, type, fill ? (const void *)h : nullptr, nullptr, n);
    if (rc)
        return rc;
    if (n == 0)
        return SLICER_OK;
    if (!fill)
        return fail(h, SLICER_ERR_ARG, "null fill callback");
    HIPCHK(h, hipSetDevice(h->device));
    rc = begin_type(h, type, has_mass != 0);
    if (rc)
        return rc;
    rc = ensure_staging(h, has_mass != 0);
    if (rc)
        return rc;
    int slot = 0;
    for (uint64_t off = 0; off < n; off += h->stage_cap, slot ^= 1) {
        uint64_t c = std::min<uint64_t>(h->stage_cap, n - off);
        // the slot is reusable once the kernel that read it has finished; meanwhile the other slot's
        // H2D copy and kernels run, so filling (file read) overlaps with device work
        HIPCHK(h, hipEventSynchronize(h->stage_free[slot]));
        if (fill(user, h->h_stage[slot], has_mass ? h->h_mstage[slot] : nullptr, off, c) != 0)
            return fail(h, SLICER_ERR_ARG, "fill callback failed at particle %llu", (unsigned long long)off);
        HIPCHK(h, hipMemcpyAsync(h->d_stage[slot], h->h_stage[slot], c * 12, hipMemcpyHostToDevice, h->stream));
        const float *dm = nullptr;
        if (has_mass) {
            HIPCHK(h, hipMemcpyAsync(h->d_mstage[slot], h->h_mstage[slot], c * 4, hipMemcpyHostToDevice, h->stream));
            dm = h->d_mstage[slot];
        }
        rc = deposit_device_chunk(h, type, h->d_stage[slot], dm, c);
        if (rc)
            return rc;
        HIPCHK(h, hipEventRecord(h->stage_free[slot], h->stream));
    }
    return SLICER_OK;
}

int slicer_deposit_host(slicer_handle h, int type, const float *pos, const float *mass, uint64_t n)
{
    int rc = check_deposit_args(h, type, pos, mass, n);
    if (rc)
        return rc;
    HostSpan span{pos, mass};
    return slicer_deposit_stream(h, type, n, mass != nullptr, copy_fill, &span);
}

int slicer_file_end(slicer_handle h)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!h->in_file)
        return fail(h, SLICER_ERR_STATE, "slicer_file_end without slicer_file_begin");
    h->in_file = false;
    if (thin_deferred(h)) {  // deposited (and folded) plane by plane in thin_replay
        slicer_handle_s::ThinFile f;
        f.file = h->file;
        for (int t = 0; t < 6; t++) {
            f.mode[t] = h->file_mode[t];
            f.mconst[t] = h->file_mconst[t];
        }
        h->thin_files.push_back(f);
        return SLICER_OK;
    }
    if (h->desc.mas == SLICER_MAS_NGP) {
        for (auto &Q : h->pg)
            for (int c = 0; c < Q.L.n; c++)
                Q.L.done[c] = 1;  // every chunk still pending belongs to a closed file now
        // A species whose counts all wait in the pending lists marked `fold` needs nothing more here: the tile kernel
        // folds them, file by file, when the lists are flushed (NgpFold).  Anything else has counts in the global count
        // maps (or is about to: its pending chunks carry fold = 0) and is folded by the map-wide kernel, now.
        bool need_kernel = false;
        for (int t = 0; t < 6; t++) {
            if (!h->file_mode[t])
                continue;
            if (h->file_mode[t] == 1 && ngp_foldable(h, t))
                h->file_mode[t] = 0;
            else
                need_kernel = true;
        }
        if (need_kernel) {
            int rcf = flush_pending(h);
            if (rcf)
                return rcf;
            for (int p = 0; p < h->desc.n_planes; p++) {
                int rc = fold_file_plane(h, p);
                if (rc)
                    return rc;
            }
        }
    }
    return SLICER_OK;
}

int slicer_plane_finalize(slicer_handle h)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!h->in_plane)
        return fail(h, SLICER_ERR_STATE, "slicer_plane_finalize outside a plane pass");
    if (h->in_file)
        return fail(h, SLICER_ERR_STATE, "slicer_plane_finalize: file not ended");
    if (h->finalized)
        return SLICER_OK;
    const slicer_plane_desc &d = h->desc;
    {
        int rcf = thin_replay(h);
        if (!rcf)
            rcf = flush_pending(h);
        if (rcf)
            return rcf;
    }
    if (d.mas == SLICER_MAS_TSC) {
        const int kind = acc_kind(d, false);
        for (int p = 0; p < d.n_planes; p++) {
            FinalizeArgs A;
            memset(&A, 0, sizeof A);
            bool any = false;
            if (!d.want_type_maps && kind == kF32 && h->shared_seen) {
                // the shared f32 accumulator already is the all-types map: hand the buffer over instead of copying
                std::swap(h->planes[p].tot, h->planes[p].acc_shared);
                continue;
            }
            if (!d.want_type_maps) {
                A.acc_shared = h->shared_seen ? h->planes[p].acc_shared.p : nullptr;
                A.inv_scale_shared = std::ldexp(1.0, -h->fixed_exp_shared);
                any = h->shared_seen;
            } else {
                for (int t = 0; t < 6; t++) {
                    if (!h->type_seen[t])
                        continue;
                    any = true;
                    A.acc[t] = kind == kF32 ? h->planes[p].toti[t].p : h->planes[p].acc[t].p;
                    A.toti[t] = (float *)h->planes[p].toti[t].p;
                    A.inv_scale[t] = std::ldexp(1.0, -h->fixed_exp[t]);
                }
            }
            A.tot = (float *)h->planes[p].tot.p;
            A.npix2 = h->npix2;
            if (!any) {
                int rc = zero_async(h, A.tot, h->npix2 * 4);
                if (rc)
                    return rc;
                continue;
            }
            ProfScope ps(h, KN_FINALIZE);
            HIPCHK(h, launch_finalize_tsc(kind, A, h->stream));
        }
    }
    h->finalized = true;
    return SLICER_OK;
}

int slicer_plane_device_maps(slicer_handle h, int plane, float **d_tot, float **d_toti)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!h->in_plane || !h->finalized)
        return fail(h, SLICER_ERR_STATE, "maps are available after slicer_plane_finalize");
    if (plane < 0 || plane >= h->desc.n_planes)
        return fail(h, SLICER_ERR_ARG, "plane %d out of range", plane);
    if (d_tot)
        *d_tot = (float *)h->planes[plane].tot.p;
    if (d_toti)
        for (int t = 0; t < 6; t++)
            d_toti[t] = (h->type_seen[t] && h->desc.want_type_maps) ? (float *)h->planes[plane].toti[t].p : nullptr;
    return SLICER_OK;
}

int slicer_get_stream(slicer_handle h, void **hip_stream)
{
    if (!h || !hip_stream)
        return fail(h, SLICER_ERR_ARG, "null argument");
    *hip_stream = (void *)h->stream;
    return SLICER_OK;
}

int slicer_plane_info(slicer_handle h, int32_t *npix, int32_t *n_planes)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!h->in_plane)
        return fail(h, SLICER_ERR_STATE, "no plane pass is open");
    if (npix)
        *npix = h->desc.npix;
    if (n_planes)
        *n_planes = h->desc.n_planes;
    return SLICER_OK;
}

int slicer_plane_device_counts(slicer_handle h, int plane, uint64_t **d_counts)
{
    if (!h || !d_counts)
        return fail(h, SLICER_ERR_ARG, "null argument");
    if (plane < 0 || plane >= SLICER_MAX_PLANES)
        return fail(h, SLICER_ERR_ARG, "plane %d out of range", plane);
    *d_counts = (uint64_t *)(h->d_counts + (size_t)plane * 6);
    return SLICER_OK;
}

int slicer_plane_algo_mask(slicer_handle h, int32_t *mask)
{
    if (!h || !mask)
        return fail(h, SLICER_ERR_ARG, "null argument");
    *mask = h->algo_mask;
    return SLICER_OK;
}

int slicer_plane_status(slicer_handle h)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!h->in_plane)
        return fail(h, SLICER_ERR_STATE, "slicer_plane_status outside a plane pass");
    HIPCHK(h, hipSetDevice(h->device));
    int neg = 0;
    HIPCHK(h, hipMemcpyAsync(&neg, h->d_neg, sizeof neg, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (neg || h->neg_remote)
        return fail(h, SLICER_ERR_NEGATIVE_COORD,
                    "a transformed coordinate is negative (positions outside [0, 2*boxsize]?): the reference stops "
                    "here (densitymaps.cpp:334-345)%s", neg ? "" : " [reported by another rank]");
    return SLICER_OK;
}

int slicer_plane_flush(slicer_handle h)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!h->in_plane || h->in_file)
        return fail(h, SLICER_ERR_STATE, "slicer_plane_flush: needs an open plane pass and no open file");
    if (h->finalized)
        return fail(h, SLICER_ERR_STATE, "slicer_plane_flush after slicer_plane_finalize");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = thin_replay(h);
    return rc ? rc : flush_pending(h);
}

namespace {
// which accumulator slots (types 0..5, 6 = shared / all-types) take part in a cross-rank sum, and their element kind
void reduce_slots(slicer_handle h, bool live[7], int &elem)
{
    const slicer_plane_desc &d = h->desc;
    for (int s = 0; s < 7; s++)
        live[s] = false;
    if (d.mas == SLICER_MAS_NGP) {
        // the per-file fold (densitymaps.cpp:511-513) already produced f32 maps: they are what the reference sums
        elem = SLICER_ELEM_F32;
        live[6] = true;
        if (d.want_type_maps)
            for (int t = 0; t < 6; t++)
                live[t] = h->type_seen[t];
        return;
    }
    elem = d.accum == SLICER_ACC_F64 ? SLICER_ELEM_F64 : d.accum == SLICER_ACC_FIXED64 ? SLICER_ELEM_FIXED64 : SLICER_ELEM_F32;
    if (!d.want_type_maps) {
        live[6] = h->shared_seen;
        return;
    }
    for (int t = 0; t < 6; t++)
        live[t] = h->type_seen[t];
}
}  // namespace

namespace {
// the host-known part of the reduce meta: which accumulators are live and their FIXED64 scales (v[21..23] = 0)
int reduce_meta_local(slicer_handle h, slicer_reduce_meta *m, const char *who)
{
    if (!h || !m)
        return fail(h, SLICER_ERR_ARG, "null argument");
    if (!h->in_plane || h->in_file || h->finalized)
        return fail(h, SLICER_ERR_STATE, "%s: after the last slicer_file_end, before finalize", who);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = thin_replay(h);
    if (!rc)
        rc = flush_pending(h);
    if (rc)
        return rc;
    bool live[7];
    int elem;
    reduce_slots(h, live, elem);
    for (int s = 0; s < 7; s++) {
        m->v[s] = live[s] ? 1 : 0;
        const bool fx = live[s] && elem == SLICER_ELEM_FIXED64;
        const int e = s < 6 ? h->fixed_exp[s] : h->fixed_exp_shared;
        m->v[7 + s] = fx ? e : INT32_MIN;
        m->v[14 + s] = fx ? -e : INT32_MIN;
    }
    m->v[21] = m->v[22] = m->v[23] = 0;
    return SLICER_OK;
}
}  // namespace

int slicer_reduce_meta_get(slicer_handle h, slicer_reduce_meta *m)
{
    int rc = reduce_meta_local(h, m, "slicer_reduce_meta_get");
    if (rc)
        return rc;
    int neg = 0;
    HIPCHK(h, hipMemcpyAsync(&neg, h->d_neg, sizeof neg, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    m->v[21] = neg ? 1 : 0;
    return SLICER_OK;
}

int slicer_reduce_meta_get_async(slicer_handle h, slicer_reduce_meta *m)
{
    return reduce_meta_local(h, m, "slicer_reduce_meta_get_async");
}

int slicer_plane_device_guard(slicer_handle h, int32_t **d_flag)
{
    if (!h || !d_flag)
        return fail(h, SLICER_ERR_ARG, "null argument");
    *d_flag = (int32_t *)h->d_neg;
    return SLICER_OK;
}

int slicer_reduce_meta_set(slicer_handle h, const slicer_reduce_meta *m)
{
    if (!h || !m)
        return fail(h, SLICER_ERR_ARG, "null argument");
    if (!h->in_plane || h->in_file || h->finalized)
        return fail(h, SLICER_ERR_STATE, "slicer_reduce_meta_set: after the last slicer_file_end, before finalize");
    HIPCHK(h, hipSetDevice(h->device));
    const slicer_plane_desc &d = h->desc;
    bool live[7];
    int elem;
    reduce_slots(h, live, elem);
    const bool ngp = d.mas == SLICER_MAS_NGP;
    const size_t esz = elem == SLICER_ELEM_F32 ? 4 : 8;
    for (int s = 0; s < 7; s++) {
        if (!m->v[s])
            continue;
        if (elem == SLICER_ELEM_FIXED64) {
            if (m->v[7 + s] == INT32_MIN || m->v[7 + s] != -m->v[14 + s])
                return fail(h, SLICER_ERR_UNSUPPORTED,
                            "ranks scaled FIXED64 accumulator %d differently (2^%d vs 2^%d): their integer sums cannot be "
                            "added; pass the same mass table on every rank", s, m->v[7 + s], -m->v[14 + s]);
            if (live[s] && (s < 6 ? h->fixed_exp[s] : h->fixed_exp_shared) != m->v[7 + s])
                return fail(h, SLICER_ERR_UNSUPPORTED, "FIXED64 scale of accumulator %d differs from the combined one", s);
        }
        if (live[s])
            continue;
        // this rank never saw the slot: zero-filled stand-ins keep the set of collectives rank-invariant
        const bool shared_layout = !ngp && !d.want_type_maps;
        const bool valid = s == 6 ? shared_layout : (ngp ? d.want_type_maps != 0 : !shared_layout);
        if (!valid)
            return fail(h, SLICER_ERR_ARG, "combined reduce meta names accumulator %d, which this pass layout lacks", s);
        for (int p = 0; p < d.n_planes; p++) {
            int rc;
            if (s == 6) {  // shared TSC accumulator (NGP's slot 6 is tot: always live)
                if ((rc = ensure(h, h->planes[p].acc_shared, h->npix2 * esz)) ||
                    (rc = zero_async(h, h->planes[p].acc_shared.p, h->npix2 * esz)))
                    return rc;
            } else {
                if ((rc = ensure(h, h->planes[p].toti[s], h->npix2 * 4)) ||
                    (rc = zero_async(h, h->planes[p].toti[s].p, h->npix2 * 4)))
                    return rc;
                if (!ngp && elem != SLICER_ELEM_F32 &&
                    ((rc = ensure(h, h->planes[p].acc[s], h->npix2 * esz)) ||
                     (rc = zero_async(h, h->planes[p].acc[s].p, h->npix2 * esz))))
                    return rc;
            }
        }
        if (s == 6) {
            h->shared_seen = true;
            h->fixed_exp_shared = elem == SLICER_ELEM_FIXED64 ? m->v[7 + s] : h->fixed_exp_shared;
            h->fixed_shared_set = true;
        } else {
            h->type_seen[s] = true;
            if (elem == SLICER_ELEM_FIXED64)
                h->fixed_exp[s] = m->v[7 + s];
            h->fixed_exp_set[s] = true;
        }
    }
    h->neg_remote = m->v[21] != 0;
    return SLICER_OK;
}

int slicer_plane_accumulators(slicer_handle h, int plane, void **acc, int32_t *elem_kind)
{
    if (!h || !acc || !elem_kind)
        return fail(h, SLICER_ERR_ARG, "null argument");
    if (!h->in_plane || h->in_file || h->finalized)
        return fail(h, SLICER_ERR_STATE, "accumulators are available after the last slicer_file_end, before finalize");
    if (plane < 0 || plane >= h->desc.n_planes)
        return fail(h, SLICER_ERR_ARG, "plane %d out of range", plane);
    for (auto &Q : h->pg)
        if (Q.L.n)
            return fail(h, SLICER_ERR_STATE, "call slicer_plane_flush (or slicer_reduce_meta_get) first");
    bool live[7];
    int elem;
    reduce_slots(h, live, elem);
    const bool ngp = h->desc.mas == SLICER_MAS_NGP;
    for (int s = 0; s < 7; s++) {
        acc[s] = nullptr;
        if (!live[s])
            continue;
        if (s == 6)
            acc[s] = ngp ? h->planes[plane].tot.p : h->planes[plane].acc_shared.p;
        else
            acc[s] = (ngp || elem == SLICER_ELEM_F32) ? h->planes[plane].toti[s].p : h->planes[plane].acc[s].p;
    }
    *elem_kind = elem;
    return SLICER_OK;
}

int slicer_synchronize(slicer_handle h)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SLICER_OK;
}

int slicer_plane_read(slicer_handle h, int plane, float *tot, float *toti, int64_t *nsel)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!h->in_plane)
        return fail(h, SLICER_ERR_STATE, "slicer_plane_read outside a plane pass");
    if (plane < 0 || plane >= h->desc.n_planes)
        return fail(h, SLICER_ERR_ARG, "plane %d out of range", plane);
    int rc = slicer_plane_finalize(h);
    if (rc)
        return rc;
    rc = slicer_plane_status(h);  // synchronises; densitymaps.cpp:334-345
    if (rc)
        return rc;
    const size_t n4 = h->npix2 * 4;
    if (tot && (rc = copy_map_to_host(h, tot, h->planes[plane].tot.p, n4)))
        return rc;
    if (toti) {
        for (int t = 0; t < 6; t++) {
            if (h->type_seen[t] && h->desc.want_type_maps) {
                if ((rc = copy_map_to_host(h, toti + h->npix2 * t, h->planes[plane].toti[t].p, n4)))
                    return rc;
            } else {
                memset(toti + h->npix2 * t, 0, n4);
            }
        }
    }
    if (nsel) {
        unsigned long long c[6];
        HIPCHK(h, hipMemcpy(c, h->d_counts + (size_t)plane * 6, sizeof c, hipMemcpyDeviceToHost));
        for (int t = 0; t < 6; t++)
            nsel[t] = (int64_t)c[t];
    }
    return SLICER_OK;
}

int slicer_device_malloc(slicer_handle h, size_t bytes, void **d_ptr)
{
    if (!h || !d_ptr)
        return fail(h, SLICER_ERR_ARG, "null argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMalloc(d_ptr, bytes));
    return SLICER_OK;
}

int slicer_device_free(slicer_handle h, void *d_ptr)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipFree(d_ptr));
    return SLICER_OK;
}

int slicer_copy_to_device(slicer_handle h, void *d_dst, const void *src, size_t bytes)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    HIPCHK(h, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SLICER_OK;
}

int slicer_copy_to_host(slicer_handle h, void *dst, const void *d_src, size_t bytes)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return copy_map_to_host(h, dst, d_src, bytes);
}

int slicer_synth_positions(slicer_handle h, float *d_pos, uint64_t first, uint64_t count, double boxsize,
                           uint64_t seed, int clustered)
{
    if (!h || (!d_pos && count))
        return fail(h, SLICER_ERR_ARG, "null argument");
    ProfScope ps(h, KN_SYNTH);
    HIPCHK(h, launch_synth(d_pos, first, count, boxsize, seed, clustered, h->stream));
    return SLICER_OK;
}

int slicer_debug_project(slicer_handle h, int type, const float *d_pos, uint64_t n, float *d_xs, float *d_ys,
                         int32_t *d_plane, uint64_t *d_src, uint64_t capacity, uint64_t *n_out)
{
    int rc = check_deposit_args(h, type, d_pos, nullptr, n);
    if (rc)
        return rc;
    PassParams P;
    make_params(h, type, false, P);
    unsigned long long *d_count = nullptr;
    HIPCHK(h, hipMalloc((void **)&d_count, sizeof(unsigned long long)));
    HIPCHK(h, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), h->stream));
    {
        ProfScope ps(h, KN_DEBUG);
        HIPCHK(h, launch_debug_project(d_pos, n, P, d_xs, d_ys, d_plane, d_src, capacity, d_count, h->d_neg,
                                       h->stream));
    }
    unsigned long long c = 0;
    HIPCHK(h, hipMemcpyAsync(&c, d_count, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(d_count));
    if (n_out)
        *n_out = c;
    return SLICER_OK;
}

int slicer_debug_box_quotient(slicer_handle h, double box, uint32_t *n_bad, uint32_t *examples8)
{
    if (!h || !n_bad)
        return fail(h, SLICER_ERR_ARG, "null argument");
    HIPCHK(h, hipSetDevice(h->device));
    unsigned out[9];
    int rc = run_box_sweep(h, box, out);
    if (rc)
        return rc;
    *n_bad = out[0];
    if (examples8)
        memcpy(examples8, out + 1, 8 * sizeof(uint32_t));
    return SLICER_OK;
}

int slicer_debug_dl_quotient(slicer_handle h, int32_t npix, uint32_t *n_bad, uint32_t *examples8)
{
    if (!h || !n_bad || npix < 1 || npix > 65536)
        return fail(h, SLICER_ERR_ARG, "slicer_debug_dl_quotient: bad arguments");
    HIPCHK(h, hipSetDevice(h->device));
    unsigned out[9];
    bool ok;
    int rc = dl_quotient_ok(h, npix, ok, out);
    if (rc)
        return rc;
    *n_bad = out[0];
    if (examples8)
        memcpy(examples8, out + 1, 8 * sizeof(uint32_t));
    return SLICER_OK;
}

int slicer_debug_math(slicer_handle h, int op, const double *d_a, const double *d_b, double *d_out, uint64_t n)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (op < 0 || op > 11 || (n && (!d_a || !d_out || ((op == 1 || op >= 10) && !d_b))) || n > (1ull << 31))
        return fail(h, SLICER_ERR_ARG, "slicer_debug_math: bad arguments");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, launch_debug_math(op, d_a, d_b, d_out, n, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SLICER_OK;
}

int slicer_profile_enable(slicer_handle h, int on)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    if (!on && h->profiling)
        prof_collect(h);
    h->profiling = on != 0;
    return SLICER_OK;
}

int slicer_profile_reset(slicer_handle h)
{
    if (!h)
        return fail(h, SLICER_ERR_ARG, "null handle");
    prof_collect(h);
    for (int i = 0; i < KN_COUNT; i++) {
        h->prof_ms[i] = 0;
        h->prof_n[i] = 0;
    }
    return SLICER_OK;
}

int slicer_profile_get(slicer_handle h, slicer_kernel_time *out, int capacity, int *n_out)
{
    if (!h || !n_out)
        return fail(h, SLICER_ERR_ARG, "null argument");
    prof_collect(h);
    int k = 0;
    for (int i = 0; i < KN_COUNT; i++) {
        if (!h->prof_n[i])
            continue;
        if (out && k < capacity) {
            memset(&out[k], 0, sizeof out[k]);
            strncpy(out[k].name, kKernelNames[i], sizeof(out[k].name) - 1);
            out[k].launches = h->prof_n[i];
            out[k].total_ms = h->prof_ms[i];
        }
        k++;
    }
    *n_out = k;
    return SLICER_OK;
}

}  // extern "C"
