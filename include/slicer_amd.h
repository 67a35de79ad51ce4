/*
 * slicer_amd.h -- C ABI of the MI355X-native particle->grid mass-assignment path.
 *
 * This is the drop-in boundary behind SLICER's createDensityMaps()
 * (reference: SLICER/densitymaps.h:161-165, body densitymaps.cpp:419-524, called
 * from slicer-v2.cpp:204-206).  The reference has no FFI; its seam is a C++ free
 * function over std::valarray / std::vector / ifstream.  The adapter in
 * slicer_amd/csrc/densitymaps_amd.{hpp,cpp} keeps that C++ signature and forwards
 * to the entry points below, which are what a binding for this path would bind:
 * plain pointers, sizes and PODs only -- no C++ types, no torch types.
 *
 * Call sequence for one lens plane (= one createDensityMaps call), per device:
 *
 *   slicer_create()                                  once per device
 *   slicer_plane_begin(desc)                         densitymaps.cpp:426-431  (resize/zero maps)
 *   for each snapshot sub-file ff in [ffmin,ffmax):  densitymaps.cpp:432
 *     slicer_file_begin(file)                        Header + Random entry    (:438, gadget2io.cpp:204-270)
 *     for each particle type t with npart[t] > 0:
 *       slicer_deposit_host|device(t, pos, mass, n)  readPos + mapParticles + gridist_w
 *     slicer_file_end()                              densitymaps.cpp:511-513  (per-file accumulation)
 *   slicer_plane_finalize()                          device maps final (ready for the cross-rank sum,
 *                                                    slicer-v2.cpp:214-217 -> RCCL, see slicer_amd_rccl.h)
 *   slicer_plane_read(...)                           D2H into the caller's valarrays; reports the
 *                                                    negativity guard of densitymaps.cpp:334-345
 *
 * All functions return 0 on success and a SLICER_ERR_* code otherwise; they never
 * exit(), abort() or throw across the boundary (the reference returns 1 and lets
 * main() MPI_Abort: slicer-v2.cpp:204-207; gridist_w calls exit(-1): utilities.cpp:55-64).
 * A handle is not thread-safe; different handles are independent (one per GPU).
 */
#ifndef SLICER_AMD_H
#define SLICER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLICER_AMD_VERSION 200 /* 0.2.0 */
#define SLICER_MAX_PLANES 8

/* status codes */
#define SLICER_OK 0
#define SLICER_ERR_NEGATIVE_COORD 1 /* densitymaps.cpp:334-345 "I will STOP here" -> reference returns 1 */
#define SLICER_ERR_ARG 2
#define SLICER_ERR_STATE 3
#define SLICER_ERR_HIP 4
#define SLICER_ERR_NOMEM 5
#define SLICER_ERR_UNSUPPORTED 6 /* e.g. SLICER_ALGO_BINNED asked for a pass it cannot serve */
#define SLICER_ERR_NO_DEVICE 7

/* mass-assignment scheme: DO_NGP is a compile-time macro in the reference (densitymaps.h:22),
 * a run-time argument of gridist_w (utilities.h:140) and a run-time field here. */
#define SLICER_MAS_TSC 0
#define SLICER_MAS_NGP 1

/* accumulator of the TSC maps (NGP always uses exact u32 counts when the mass is constant) */
#define SLICER_ACC_F32 0     /* f32 atomics: fastest, order-dependent in the last bits            */
#define SLICER_ACC_F64 1     /* f64 atomics, rounded to f32 once at finalize                       */
#define SLICER_ACC_FIXED64 2 /* 64-bit fixed point: order-independent => bitwise reproducible sums (also across ranks).
                              * ABSOLUTE accuracy: every contribution is rounded to 2^-fixed_frac_bits (default 2^-40) of
                              * the mass scale 2^ceil(log2 m) -- or of 2^10 (MAX_M) with per-particle masses -- so a pixel
                              * that holds nothing but a vanishing TSC weight (<< 1e-6 m) loses RELATIVE precision; pixels
                              * holding >= 1e-3 m agree with the f64 accumulator to < 1 f32 ulp.  F32 / F64 keep the
                              * relative per-pixel bound.  A cell overflows at 2^(64 - fixed_frac_bits) mass scales. */

/* deposit algorithm */
#define SLICER_ALGO_AUTO 0
#define SLICER_ALGO_DIRECT 1 /* fused project + global atomics                                   */
#define SLICER_ALGO_BINNED 2 /* project -> tile bins -> LDS-privatised tiles -> shaped row flush.  A pass with more
                             * (plane, tile) bins than one run holds, or with overlapping slabs, goes in plane groups;
                             * more than three lateral replications per side go in windows of the replica grid.
                             * An explicit BINNED request that cannot be honoured (a tile table beyond the limits) returns
                             * SLICER_ERR_UNSUPPORTED; only SLICER_ALGO_AUTO falls back to DIRECT
                             * (slicer_plane_algo_mask tells which ran). */

/* element kind of an accumulator map (slicer_plane_accumulators) */
#define SLICER_ELEM_F32 0
#define SLICER_ELEM_F64 1
#define SLICER_ELEM_FIXED64 2 /* u64, value = integer * 2^-fixed_exp */

typedef struct slicer_handle_s *slicer_handle;

/* One pass = up to SLICER_MAX_PLANES lens planes cut from the same box replication (same Random
 * entry and rcase: densitymaps.cpp:227-246, slicer-v2.cpp:184-185).  n_planes = 1 reproduces one
 * createDensityMaps call.  ld/ld2 are Lens.ld/ld2 in Mpc/h (data.h:111-112). */
typedef struct {
    int32_t npix;                          /* InputParams.npix (data.h:31)                       */
    int32_t n_planes;                      /* 1..SLICER_MAX_PLANES                               */
    int32_t mas;                           /* SLICER_MAS_*                                       */
    int32_t accum;                         /* SLICER_ACC_*  (TSC)                                */
    int32_t algo;                          /* SLICER_ALGO_*                                      */
    int32_t hydro;                         /* InputParams.hydro (data.h:35)                      */
    int32_t snopt;                         /* InputParams.snopt (data.h:45); > 0 draws libc rand(), plane-major */
    int32_t want_type_maps;                /* 1: keep the six per-type maps (mapxytoti)          */
    double fov_rad;                        /* fovradiants                                        */
    double ld[SLICER_MAX_PLANES];          /* Lens.ld[isnap]                                     */
    double ld2[SLICER_MAX_PLANES];         /* Lens.ld2[isnap]                                    */
    int32_t nrepperp[SLICER_MAX_PLANES];   /* Lens.nrepperp[isnap] (data.h:114)                  */
    int32_t fixed_frac_bits;               /* SLICER_ACC_FIXED64: fractional bits, 0 = auto (40) */
    int32_t debug_flags;                   /* bit 0: always use libm (OCML) asin/atan2, no small-angle series */
} slicer_plane_desc;

/* Per sub-file state: the fields of Header (data.h:59-79) and of the plane's Random entry
 * (data.h:126-131) that the path reads, plus rcase (slicer-v2.cpp:184-185). */
typedef struct {
    int32_t npart[6];   /* Header.npart                                  */
    double massarr[6];  /* Header.massarr                                */
    double boxsize;     /* Header.boxsize (kpc/h; POS_U = 1)             */
    int32_t sgn[3];     /* Random.sgnX/Y/Z[isnap]  (+1 / -1)             */
    int32_t face;       /* Random.face[isnap]      (1..6)                */
    double center[3];   /* Random.x0/y0/z0[isnap]                        */
    float rcase;        /* box-replication offset along the line of sight */
    int32_t reserved;
} slicer_file_desc;

typedef struct {
    char name[32];
    uint64_t launches;
    double total_ms; /* summed HIP-event time of those launches */
} slicer_kernel_time;

int slicer_version(void);

/* device = HIP device ordinal.  max_chunk = largest number of particles one deposit call may
 * carry in one kernel pass (bigger inputs are looped internally); sizes the workspace. */
int slicer_create(int device, uint64_t max_chunk, slicer_handle *out);
int slicer_destroy(slicer_handle h);
const char *slicer_last_error(slicer_handle h); /* valid until the next call on h; h may be NULL */

/* Tuning and test knobs of a handle (integers).  The environment seeds them once, in slicer_create (SLICER_<KEY> in
 * upper case); afterwards only this call changes them -- no launch path reads the environment, so handles driven by
 * different host threads (SLICER_amd --devices) do not share mutable state.  Not while deposits are in flight (an
 * open file, or chunks waiting for their tile launch: SLICER_ERR_STATE).  Keys:
 *   k4_int       integer (u64) LDS tile cells: 0 never, 1 automatic (>= 2048 particles per bin), 2 always
 *   tile_log2, tile_h_log2, bin_batch, unit_rows    tile / batch / unit geometry overrides (0 = automatic)
 *   k3_per_cu    persistent sort workgroups per CU          k1_general   1: always the general project+bin kernel
 *   k1_stack     fast project+bin kernel: -1 automatic, 0 / 1 project in place / through the wave stack
 *   ngp_general  1: no in-tile NGP fold                      dl_quot      0: no reciprocal-product grid quotient
 *   sort2        1: the two-level sort wherever a pass qualifies (default 0: the one-level sort)
 *   pending      chunks binned before one tile-kernel launch deposits them (0 = automatic: 8 ... 32)
 *   zero_batch   1 (default): the maps of a pass are cleared by one launch; 0: one hipMemsetAsync per map
 *   thin_host    1: shot-noise deviates (snopt > 0) drawn by libc rand() on the host, one call per selected entry;
 *                default 0: the process-global rand() stream continues on the device (slicer_libc_rand_supported)
 * Unknown keys return SLICER_ERR_ARG. */
int slicer_set_option(slicer_handle h, const char *key, int32_t value);
int slicer_get_option(slicer_handle h, const char *key, int32_t *value);

/* Shot-noise thinning (InputParams.snopt > 0) consumes the process-global libc rand() stream, one deviate per selected
 * entry (densitymaps.cpp:393).  The library continues that stream on the device: slicer_plane_begin of a pass with
 * snopt > 0 reads the generator state (before it touches the HIP runtime), the pass jumps and generates from that copy on
 * the GPU, and the advanced state is installed in libc when the pass ends -- before slicer_plane_flush / finalize / read
 * returns (or the next slicer_plane_begin, or slicer_destroy) -- so the caller's rand() calls after the pass see exactly
 * the stream position the reference would leave.  The host must not draw from rand() itself between plane_begin and
 * that point.  This needs glibc's default TYPE_3 generator (no initstate() with another size by the process) and passes
 * a layout self-check on a private state
 * array: 1 if so, 0 if thinning falls back to rand() calls on the host (same deviates, ~25x slower).  No GPU needed.
 * Reading and installing the state switches libc to a scratch state array for a few instructions (initstate / setstate):
 * like rand() itself next to srand(), not to be raced by rand() calls of other threads of the process.
 * CAUTION: once the HIP runtime runs, its own threads call rand() now and then (a kernel's first launch loads its code
 * object, allocations, ...), which moves the process-global stream at unpredictable points.  What they draw DURING a pass
 * is overwritten when the pass ends; what they draw between the host's own last draw and slicer_plane_begin (any HIP
 * work of the process in that window: creating the handle, other passes) is not.  A host that needs the reference's exact
 * thinning reads the stream BEFORE the first HIP call of the process (slicer_libc_rand_state_get needs no GPU) and gives
 * the handle its own copy (slicer_rand_stream_set below) -- what the createDensityMaps adapter and SLICER_amd do. */
int slicer_libc_rand_supported(void);
/* The process-global generator state as 31 words, oldest first (x[n-31] ... x[n-1] of x[n] = x[n-31] + x[n-3],
 * rand() = x[n] >> 1): read, and install.  Test hooks of the above; SLICER_ERR_UNSUPPORTED if not supported. */
int slicer_libc_rand_state_get(uint32_t *v31);
int slicer_libc_rand_state_set(const uint32_t *v31);
/* A stream of the handle's own for the shot-noise deviates, in the same 31-word form, instead of the process-global
 * one: the reference's MPI ranks each own an identically seeded copy of libc's stream and consume it independently
 * (densitymaps.cpp:187-217 seeds it in every rank alike); threads of ONE process that drive one device each get the same
 * by reading the process state once after the plan is made (slicer_libc_rand_state_get) and handing every handle a copy.
 * The handle's state advances with its draws (slicer_rand_stream_get reads it back); libc's own stream is not touched.
 * v31 = NULL returns the handle to the process-global stream.  Not inside a file. */
int slicer_rand_stream_set(slicer_handle h, const uint32_t *v31);
int slicer_rand_stream_get(slicer_handle h, uint32_t *v31);

/* Use an existing hipStream_t (e.g. the caller framework's current stream); NULL = handle-owned. */
int slicer_set_stream(slicer_handle h, void *hip_stream);

int slicer_plane_begin(slicer_handle h, const slicer_plane_desc *desc);
int slicer_file_begin(slicer_handle h, const slicer_file_desc *file);
/* pos: AoS [n][3] f32 exactly as in the POS block (raw file units), host memory.  mass: per-particle
 * f32 masses (hydro types with massarr == 0: MASS / BHMA stream, densitymaps.cpp:358-372) or NULL. */
int slicer_deposit_host(slicer_handle h, int type, const float *pos, const float *mass, uint64_t n);
/* Same as slicer_deposit_host, but the library pulls the particles: `fill(user, dst_pos, dst_mass, first, count)`
 * must write particles [first, first+count) of this type straight into the pinned staging buffers it is given
 * (dst_mass is NULL when has_mass == 0) and return 0; e.g. an fread from the POS block.  Saves the pageable
 * copy of the whole block and overlaps file reads with H2D and kernels (double-buffered).  SURVEY S8f row N1. */
typedef int (*slicer_fill_fn)(void *user, float *dst_pos, float *dst_mass, uint64_t first, uint64_t count);
int slicer_deposit_stream(slicer_handle h, int type, uint64_t n, int has_mass, slicer_fill_fn fill, void *user);
/* same, operands already resident in this device's HBM */
int slicer_deposit_device(slicer_handle h, int type, const float *d_pos, const float *d_mass, uint64_t n);
int slicer_file_end(slicer_handle h);
int slicer_plane_finalize(slicer_handle h);

/* Device pointers of the finalized f32 maps of plane `plane` (npix*npix each): *d_tot, and
 * d_toti[0..5] (NULL for types that never appeared or when want_type_maps == 0). */
int slicer_plane_device_maps(slicer_handle h, int plane, float **d_tot, float **d_toti);
/* Synchronise, run the guard check, copy maps to host.  tot: npix^2 floats; toti: 6*npix^2 floats or
 * NULL; nsel: 6 int64 (true number of selected particles per type; the reference's own out-parameter
 * is always 0 because of the shadowed array at densitymaps.cpp:497) or NULL. */
int slicer_plane_read(slicer_handle h, int plane, float *tot, float *toti, int64_t *nsel);
int slicer_synchronize(slicer_handle h);
/* the hipStream_t all work of this handle is enqueued on, and the device array of selected-entry counters
 * of `plane` (6 x uint64), for callers that reduce across ranks themselves (slicer_amd_rccl.h) */
int slicer_get_stream(slicer_handle h, void **hip_stream);
int slicer_plane_info(slicer_handle h, int32_t *npix, int32_t *n_planes); /* of the current plane pass */
int slicer_plane_device_counts(slicer_handle h, int plane, uint64_t **d_counts);

/* Which deposit algorithms ran since slicer_plane_begin: bit SLICER_ALGO_DIRECT, bit SLICER_ALGO_BINNED
 * (1 << value), bit 3 = the shot-noise thinning kernels (snopt > 0); bits 4 / 5 tell which project+bin kernel the
 * binned path used (4: the f32-transform fast variant, 5: the general one; see slicer_project_bin.hip); bit 6: a tile
 * kernel launch kept its tiles as integer (u64) cells (constant-mass TSC, F32 / F64 accumulators, enough records per
 * tile; option k4_int = 0 / 2 forbids / forces them -- a tuning and test knob); bit 7: a chunk went through the two-level
 * sort (project+bin kernel sorts by coarse bin in LDS, k_sort2 by tile; option sort2 = 1 allows it); bit 8: shot-noise
 * deviates came from the device continuation of the libc stream. */
int slicer_plane_algo_mask(slicer_handle h, int32_t *mask);
/* Synchronise and report the negativity guard (densitymaps.cpp:334-345) without copying maps: SLICER_OK or
 * SLICER_ERR_NEGATIVE_COORD.  Callers that hand the device maps on (cross-rank reduce) call this first. */
int slicer_plane_status(slicer_handle h);

/* ---- cross-rank sum in the accumulator type ("reduce in the accumulator type, convert after") ----
 * Sequence per plane pass, on every rank, after the last slicer_file_end():
 *   slicer_plane_flush()              all deposits issued into the accumulators (no conversion yet)
 *   slicer_reduce_meta_get(&m)        which accumulators this rank holds + FIXED64 scales
 *   <combine m across ranks: element-wise MAX of m.v[], e.g. one small all-reduce>
 *   slicer_reduce_meta_set(&m)        allocates and zero-fills the accumulators this rank lacks, so that the SET OF
 *                                     COLLECTIVES IS THE SAME ON EVERY RANK (the reference reduces all 7 maps
 *                                     unconditionally, slicer-v2.cpp:214-217); fails if two ranks scaled the same
 *                                     FIXED64 accumulator differently
 *   slicer_plane_accumulators(p,...)  device pointers + element kind; sum them over ranks (ncclFloat / ncclDouble /
 *                                     ncclUint64: a FIXED64 N-rank sum is bitwise the 1-rank sum)
 *   slicer_plane_finalize()           on the root: accumulators -> f32 maps (once, after the sum)
 * slicer_amd_rccl.h wraps this sequence for RCCL; slicer_amd/parallel.py for torch.distributed. */
#define SLICER_REDUCE_META_INTS 24
typedef struct {
    /* v[0..6]   1 if accumulator slot s is live (s = type 0..5; s = 6: the shared all-types accumulator), else 0
     * v[7..13]  FIXED64 exponent of slot s, or INT32_MIN when not live / not FIXED64
     * v[14..20] minus that exponent, or INT32_MIN (so that an element-wise MAX exposes disagreeing ranks)
     * v[21]     negativity-guard flag of this rank (0/1)    v[22..23] reserved (0) */
    int32_t v[SLICER_REDUCE_META_INTS];
} slicer_reduce_meta;
int slicer_plane_flush(slicer_handle h);
int slicer_reduce_meta_get(slicer_handle h, slicer_reduce_meta *m);
int slicer_reduce_meta_set(slicer_handle h, const slicer_reduce_meta *m);
/* The same sequence without a host synchronisation (slicer_reduce_meta_get waits for the deposits in order to read the
 * negativity guard, which stalls a pipelined caller once per plane pass): _get_async fills in only what the host knows
 * (live accumulators, scales; v[21] = 0) and the caller combines the guard on the device instead -- one MAX all-reduce
 * of the int32 at *d_flag (slicer_plane_device_guard), issued with the map sums; slicer_plane_status /
 * slicer_plane_read then report a guard raised on any rank. */
int slicer_reduce_meta_get_async(slicer_handle h, slicer_reduce_meta *m);
int slicer_plane_device_guard(slicer_handle h, int32_t **d_flag);
/* acc[s] (s as above; NULL when not live), element kind SLICER_ELEM_* (one kind per pass), npix^2 elements each. */
int slicer_plane_accumulators(slicer_handle h, int plane, void **acc /* [7] */, int32_t *elem_kind);

/* --- utilities for benches and tests (device-side synthetic boxes) --- */
int slicer_device_malloc(slicer_handle h, size_t bytes, void **d_ptr);
int slicer_device_free(slicer_handle h, void *d_ptr);
int slicer_copy_to_device(slicer_handle h, void *d_dst, const void *src, size_t bytes);
int slicer_copy_to_host(slicer_handle h, void *dst, const void *d_src, size_t bytes);
int slicer_synth_positions(slicer_handle h, float *d_pos, uint64_t first, uint64_t count, double boxsize,
                           uint64_t seed, int clustered);
/* project only: writes xs, ys (and the plane index) of every selected entry of one chunk, in no
 * particular order, plus the source particle index; returns the count.  For parity tests of A1-A3. */
int slicer_debug_project(slicer_handle h, int type, const float *d_pos, uint64_t n, float *d_xs, float *d_ys,
                         int32_t *d_plane, uint64_t *d_src, uint64_t capacity, uint64_t *n_out);

/* debug: the device arithmetic primitives of the projection on arbitrary operands (device pointers, n doubles each).
 * op 0: out = sqrt(a)  (unscaled Newton iteration, valid for 2^-500 <= a <= 2^500)
 * op 1: out = a / b    (same operand range; a may be 0)
 * op 2: out = asin(a)  small-angle series, |a| <= 0.3125      op 3: out = atan(a), |a| <= 0.3125
 * op 4 / 5: the 9-term variants of 2 / 3, |a| <= 0.155
 * op 6 / 7: the raw hardware estimates v_rsq_f64(a), v_rcp_f64(a)      op 8 / 9: their one-step refinements
 *           (rsqrt_fast / rcp_fast of the fast project+bin kernel: not correctly rounded, < 2^-48 relative)
 * op 10: out = cell index floor((double)(float)a / dl) on a map of npix = (int)b & 0xFFFFF pixels (dl = 1 / npix, any
 *           npix: utilities.cpp:69-70)      op 11: TSC weight number ((int)b >> 20) & 3 (0, 1, 2) of that coordinate;
 *           bit 22 of b selects the reciprocal-product quotient a clean slicer_debug_dl_quotient sweep licenses
 *           (utilities.cpp:4-16, 82-88) -- the device evaluates both without the reference's f64 divisions except on
 *           exact ties
 * Lets the tests compare these with correctly rounded host results bit by bit (densitymaps.cpp:382-384 uses
 * sqrt, /, asin, atan2 of libm). */
int slicer_debug_math(slicer_handle h, int op, const double *d_a, const double *d_b, double *d_out, uint64_t n);

/* debug: the exhaustive sweep that licenses the f32 form of r / box in the fast project+bin kernel.  For every one of
 * the 2^31 non-negative binary32 values r whose fast quotient lies in the fast path's domain (0, or 2^-100..1) the
 * device compares it with (float)((double)r / box); *n_bad = number of mismatches (0 = licensed), examples8 = bit
 * patterns of up to eight offending r.  The library runs the same sweep once per handle and box size. */
int slicer_debug_box_quotient(slicer_handle h, double box, uint32_t *n_bad, uint32_t *examples8);
/* The same kind of proof for maps that are not a power of two wide: the grid arithmetic divides by dl = 1/npix in
 * f64 (utilities.cpp:69-70, :4-16); the device replaces the division by a reciprocal product + one FMA correction
 * (quot_dl3, slicer_device.hpp) for a map size only after this sweep -- every non-negative f32 operand below 2 against
 * the IEEE division, ~1 ms, run once per npix and handle -- found no mismatch.  n_bad = mismatches (0 expected). */
int slicer_debug_dl_quotient(slicer_handle h, int32_t npix, uint32_t *n_bad, uint32_t *examples8);

/* per-kernel HIP-event timing (off by default; adds two event records per launch) */
int slicer_profile_enable(slicer_handle h, int on);
int slicer_profile_reset(slicer_handle h);
int slicer_profile_get(slicer_handle h, slicer_kernel_time *out, int capacity, int *n_out);

#ifdef __cplusplus
}
#endif
#endif /* SLICER_AMD_H */
