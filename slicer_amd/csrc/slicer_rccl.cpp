// slicer_rccl.cpp -- slicer-v2.cpp:214-217's MPI_Reduce(SUM) as ncclReduce over xGMI (RCCL).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/slicer_amd_rccl.h"

struct slicer_rccl_comm_s {
    ncclComm_t comm;
    int device;
    int nranks;
};

namespace {
thread_local std::string g_err;
int fail(const char *what, const char *detail)
{
    g_err = std::string(what) + ": " + detail;
    return SLICER_ERR_HIP;
}
}  // namespace

#define NCCLCHK(expr)                                        \
    do {                                                     \
        ncclResult_t r_ = (expr);                            \
        if (r_ != ncclSuccess)                               \
            return fail(#expr, ncclGetErrorString(r_));      \
    } while (0)

extern "C" {

const char *slicer_rccl_last_error(void) { return g_err.c_str(); }

int slicer_rccl_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) <= SLICER_RCCL_ID_BYTES, "id buffer too small");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memset(id128, 0, SLICER_RCCL_ID_BYTES);
    memcpy(id128, &id, sizeof id);
    return SLICER_OK;
}

int slicer_rccl_comm_init_rank(slicer_rccl_comm *out, int nranks, int rank, const void *id128, int device)
{
    if (!out || !id128)
        return fail("slicer_rccl_comm_init_rank", "null argument");
    if (hipSetDevice(device) != hipSuccess)
        return fail("hipSetDevice", hipGetErrorString(hipGetLastError()));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    slicer_rccl_comm c = new slicer_rccl_comm_s{nullptr, device, nranks};
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail("ncclCommInitRank", ncclGetErrorString(r));
    }
    *out = c;
    return SLICER_OK;
}

int slicer_rccl_comm_init_all(slicer_rccl_comm *out, int ndev, const int *devices)
{
    if (!out || ndev < 1 || ndev > 64)
        return fail("slicer_rccl_comm_init_all", "bad argument");
    ncclComm_t comms[64];
    NCCLCHK(ncclCommInitAll(comms, ndev, devices));
    for (int i = 0; i < ndev; i++)
        out[i] = new slicer_rccl_comm_s{comms[i], devices ? devices[i] : i, ndev};
    return SLICER_OK;
}

int slicer_rccl_comm_destroy(slicer_rccl_comm c)
{
    if (!c)
        return SLICER_OK;
    ncclCommDestroy(c->comm);
    delete c;
    return SLICER_OK;
}

int slicer_rccl_plane_reduce(slicer_handle h, slicer_rccl_comm c, int root, int per_type)
{
    if (!h || !c)
        return fail("slicer_rccl_plane_reduce", "null argument");
    void *sp = nullptr;
    if (slicer_get_stream(h, &sp) != SLICER_OK)
        return fail("slicer_get_stream", slicer_last_error(h));
    hipStream_t stream = (hipStream_t)sp;
    if (slicer_plane_finalize(h) != SLICER_OK)
        return fail("slicer_plane_finalize", slicer_last_error(h));
    int32_t npix = 0, n_planes = 0;
    if (slicer_plane_info(h, &npix, &n_planes) != SLICER_OK)
        return fail("slicer_plane_info", slicer_last_error(h));
    const size_t n = (size_t)npix * (size_t)npix;
    for (int p = 0; p < n_planes; p++) {
        float *tot = nullptr, *toti[6];
        if (slicer_plane_device_maps(h, p, &tot, toti) != SLICER_OK)
            return fail("slicer_plane_device_maps", slicer_last_error(h));
        NCCLCHK(ncclGroupStart());
        NCCLCHK(ncclReduce(tot, tot, n, ncclFloat, ncclSum, root, c->comm, stream));
        if (per_type)
            for (int t = 0; t < 6; t++)
                if (toti[t])
                    NCCLCHK(ncclReduce(toti[t], toti[t], n, ncclFloat, ncclSum, root, c->comm, stream));
        uint64_t *cnt = nullptr;
        if (slicer_plane_device_counts(h, p, &cnt) == SLICER_OK)
            NCCLCHK(ncclReduce(cnt, cnt, 6, ncclUint64, ncclSum, root, c->comm, stream));
        NCCLCHK(ncclGroupEnd());
    }
    return SLICER_OK;
}

}  // extern "C"
