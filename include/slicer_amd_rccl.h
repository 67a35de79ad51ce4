/*
 * slicer_amd_rccl.h -- the cross-rank sum of the plane maps over RCCL / xGMI.
 *
 * Replaces the reference's only collective: slicer-v2.cpp:214-217,
 *     MPI_Reduce(&mapxytot[0],     ..., npix*npix, MPI_FLOAT, MPI_SUM, 0, MPI_COMM_WORLD)
 *     MPI_Reduce(&mapxytoti[i][0], ...)   for i < 6
 * One rank per GPU.  The reduce runs on the slicer handle's stream, in place on the device maps that
 * slicer_plane_finalize() produced, before slicer_plane_read() copies the root's maps to the host.
 * Separate library (libslicer_amd_rccl.so) so that libslicer_amd.so does not depend on RCCL.
 */
#ifndef SLICER_AMD_RCCL_H
#define SLICER_AMD_RCCL_H

#include "slicer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slicer_rccl_comm_s *slicer_rccl_comm;

#define SLICER_RCCL_ID_BYTES 128
/* rank 0 creates the id and hands the 128 bytes to the other ranks (e.g. MPI_Bcast, a file, a socket) */
int slicer_rccl_unique_id(void *id128);
int slicer_rccl_comm_init_rank(slicer_rccl_comm *out, int nranks, int rank, const void *id128, int device);
/* single process driving ndev GPUs (one host thread per device calls the reduce) */
int slicer_rccl_comm_init_all(slicer_rccl_comm *out /* [ndev] */, int ndev, const int *devices);
int slicer_rccl_comm_destroy(slicer_rccl_comm c);
const char *slicer_rccl_last_error(void);

/* Sum the current plane pass of `h` over the ranks onto `root`.  Call after the last slicer_file_end() and
 * INSTEAD of slicer_plane_finalize(): the sum runs on the accumulators, in their own type (f32 / f64 / 64-bit
 * fixed point: SURVEY S8e "reduce in the accumulator type, convert after"), and the conversion to f32 maps
 * happens once, afterwards.  A FIXED64 N-rank result is therefore bitwise the 1-rank result.  The set of
 * collectives is the same on every rank even when a rank's sub-files hold no particle of some type (it
 * contributes a zero map, like the reference, which reduces all 7 maps unconditionally); the negativity guard
 * of any rank reaches every rank (slicer_plane_read / slicer_plane_status then return SLICER_ERR_NEGATIVE_COORD).
 * per_type is ignored (kept for source compatibility): the per-type maps are summed iff the pass keeps them
 * (want_type_maps).  Counters of selected particles are summed too (the reference forgets to). */
int slicer_rccl_plane_reduce(slicer_handle h, slicer_rccl_comm c, int root, int per_type);

/* The same sum with the algorithm chosen by the caller (an argument, not an environment variable):
 *   SLICER_RCCL_REDUCE_ROOTED  one ncclReduce per map (what slicer_rccl_plane_reduce does): RCCL's ring / tree
 *   SLICER_RCCL_REDUCE_DIRECT  SURVEY S5: in-place ncclReduceScatter (rank j ends up with the sum of slice j of every
 *                              map) + ncclSend / ncclRecv of the slices to the root, grouped per plane.  xGMI links
 *                              are point to point, so 1/N of a map per link and hop instead of the whole map through
 *                              every link of a ring.  Any map size (the n % N tail goes through a small rooted reduce).
 * Sums run in the accumulator type with both, so a FIXED64 result does not depend on the choice. */
#define SLICER_RCCL_REDUCE_ROOTED 0
#define SLICER_RCCL_REDUCE_DIRECT 1
int slicer_rccl_plane_reduce_ex(slicer_handle h, slicer_rccl_comm c, int root, int algo);

#ifdef __cplusplus
}
#endif
#endif
