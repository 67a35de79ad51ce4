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
    int32_t *d_meta = nullptr;  // device scratch of the reduce-meta all-reduce
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
    if (c->d_meta) {
        (void)hipSetDevice(c->device);
        (void)hipFree(c->d_meta);
    }
    ncclCommDestroy(c->comm);
    delete c;
    return SLICER_OK;
}

int slicer_rccl_plane_reduce(slicer_handle h, slicer_rccl_comm c, int root, int per_type)
{
    (void)per_type;  // kept for source compatibility: the live accumulators decide what is summed
    if (!h || !c)
        return fail("slicer_rccl_plane_reduce", "null argument");
    void *sp = nullptr;
    if (slicer_get_stream(h, &sp) != SLICER_OK)
        return fail("slicer_get_stream", slicer_last_error(h));
    hipStream_t stream = (hipStream_t)sp;
    if (hipSetDevice(c->device) != hipSuccess)
        return fail("hipSetDevice", hipGetErrorString(hipGetLastError()));
    int32_t npix = 0, n_planes = 0;
    if (slicer_plane_info(h, &npix, &n_planes) != SLICER_OK)
        return fail("slicer_plane_info", slicer_last_error(h));

    // 1. make the set of collectives rank-invariant: element-wise MAX of the reduce meta (which accumulators are
    //    live anywhere, their FIXED64 scales, the negativity guard), then zero-filled stand-ins where this rank has none
    slicer_reduce_meta m;
    if (slicer_reduce_meta_get(h, &m) != SLICER_OK)
        return fail("slicer_reduce_meta_get", slicer_last_error(h));
    if (!c->d_meta && hipMalloc((void **)&c->d_meta, sizeof m) != hipSuccess)
        return fail("hipMalloc", hipGetErrorString(hipGetLastError()));
    if (hipMemcpyAsync(c->d_meta, &m, sizeof m, hipMemcpyHostToDevice, stream) != hipSuccess)
        return fail("hipMemcpyAsync", hipGetErrorString(hipGetLastError()));
    NCCLCHK(ncclAllReduce(c->d_meta, c->d_meta, SLICER_REDUCE_META_INTS, ncclInt32, ncclMax, c->comm, stream));
    if (hipMemcpyAsync(&m, c->d_meta, sizeof m, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return fail("meta all-reduce", hipGetErrorString(hipGetLastError()));
    if (slicer_reduce_meta_set(h, &m) != SLICER_OK)
        return fail("slicer_reduce_meta_set", slicer_last_error(h));

    // 2. one rooted sum per live accumulator and plane, in the accumulator's own type (slicer-v2.cpp:214-217 sums
    //    f32 maps; f64 / 64-bit fixed-point accumulators are summed before their single rounding to f32)
    const size_t n = (size_t)npix * (size_t)npix;
    for (int p = 0; p < n_planes; p++) {
        void *acc[7];
        int32_t elem = 0;
        if (slicer_plane_accumulators(h, p, acc, &elem) != SLICER_OK)
            return fail("slicer_plane_accumulators", slicer_last_error(h));
        const ncclDataType_t dt = elem == SLICER_ELEM_F64 ? ncclDouble : elem == SLICER_ELEM_FIXED64 ? ncclUint64 : ncclFloat;
        NCCLCHK(ncclGroupStart());
        for (int s = 0; s < 7; s++)
            if (acc[s])
                NCCLCHK(ncclReduce(acc[s], acc[s], n, dt, ncclSum, root, c->comm, stream));
        uint64_t *cnt = nullptr;
        if (slicer_plane_device_counts(h, p, &cnt) == SLICER_OK)
            NCCLCHK(ncclReduce(cnt, cnt, 6, ncclUint64, ncclSum, root, c->comm, stream));
        NCCLCHK(ncclGroupEnd());
    }
    // 3. accumulators -> f32 maps (on the root these are the sums; elsewhere the rank's own partial maps)
    if (slicer_plane_finalize(h) != SLICER_OK)
        return fail("slicer_plane_finalize", slicer_last_error(h));
    return SLICER_OK;
}

}  // extern "C"
