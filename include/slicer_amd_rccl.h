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

/* Sum every finalized map of the current plane pass of `h` onto `root`: the all-types map of each
 * plane, the six per-type maps when per_type != 0 (the reference always sends them, even when
 * partinplanes == false and nothing reads them), and the selected-particle counters. */
int slicer_rccl_plane_reduce(slicer_handle h, slicer_rccl_comm c, int root, int per_type);

#ifdef __cplusplus
}
#endif
#endif
