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

namespace {
// One accumulator of n elements summed onto `root`.  Must be called inside ncclGroupStart / ncclGroupEnd.
//   SLICER_RCCL_REDUCE_ROOTED   ncclReduce: the library's choice (a ring or tree over xGMI)
//   SLICER_RCCL_REDUCE_DIRECT   SURVEY S5 / S8e: in-place reduce-scatter (rank j ends up with the sum of slice j:
//                               1/N of the map per link) + the slices sent to the root; the n % N tail through a
//                               small rooted reduce
ncclResult_t sum_to_root(void *acc, size_t n, ncclDataType_t dt, size_t esz, int root, int algo, int rank, int nranks,
                         ncclComm_t comm, hipStream_t stream, bool second_phase)
{
    if (algo == SLICER_RCCL_REDUCE_ROOTED || nranks == 1)
        return second_phase ? ncclSuccess : ncclReduce(acc, acc, n, dt, ncclSum, root, comm, stream);
    const size_t q = n / (size_t)nranks;
    char *base = (char *)acc;
    ncclResult_t r = ncclSuccess;
    if (!second_phase) {
        if (q)
            r = ncclReduceScatter(base, base + (size_t)rank * q * esz, q, dt, ncclSum, comm, stream);
        if (r == ncclSuccess && n > q * (size_t)nranks)
            r = ncclReduce(base + q * nranks * esz, base + q * nranks * esz, n - q * nranks, dt, ncclSum, root, comm, stream);
        return r;
    }
    if (!q)
        return ncclSuccess;
    if (rank == root) {
        for (int j = 0; j < nranks && r == ncclSuccess; j++)
            if (j != root)
                r = ncclRecv(base + (size_t)j * q * esz, q, dt, j, comm, stream);
    } else {
        r = ncclSend(base + (size_t)rank * q * esz, q, dt, root, comm, stream);
    }
    return r;
}
}  // namespace

int slicer_rccl_plane_reduce_ex(slicer_handle h, slicer_rccl_comm c, int root, int algo)
{
    if (!h || !c)
        return fail("slicer_rccl_plane_reduce", "null argument");
    if (algo != SLICER_RCCL_REDUCE_ROOTED && algo != SLICER_RCCL_REDUCE_DIRECT)
        return fail("slicer_rccl_plane_reduce", "unknown algorithm");
    void *sp = nullptr;
    if (slicer_get_stream(h, &sp) != SLICER_OK)
        return fail("slicer_get_stream", slicer_last_error(h));
    hipStream_t stream = (hipStream_t)sp;
    if (hipSetDevice(c->device) != hipSuccess)
        return fail("hipSetDevice", hipGetErrorString(hipGetLastError()));
    int32_t npix = 0, n_planes = 0;
    if (slicer_plane_info(h, &npix, &n_planes) != SLICER_OK)
        return fail("slicer_plane_info", slicer_last_error(h));
    int rank = 0;
    NCCLCHK(ncclCommUserRank(c->comm, &rank));

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

    // 2. one sum per live accumulator and plane, in the accumulator's own type (slicer-v2.cpp:214-217 sums f32 maps;
    //    f64 / 64-bit fixed-point accumulators are summed before their single rounding to f32).  An error inside a
    //    group still closes the group before it is reported (an open group would swallow every later call).
    const size_t n = (size_t)npix * (size_t)npix;
    for (int p = 0; p < n_planes; p++) {
        void *acc[7];
        int32_t elem = 0;
        if (slicer_plane_accumulators(h, p, acc, &elem) != SLICER_OK)
            return fail("slicer_plane_accumulators", slicer_last_error(h));
        const ncclDataType_t dt = elem == SLICER_ELEM_F64 ? ncclDouble : elem == SLICER_ELEM_FIXED64 ? ncclUint64 : ncclFloat;
        const size_t esz = elem == SLICER_ELEM_F32 ? 4 : 8;
        uint64_t *cnt = nullptr;
        const bool have_cnt = slicer_plane_device_counts(h, p, &cnt) == SLICER_OK;
        for (int phase = 0; phase < (algo == SLICER_RCCL_REDUCE_DIRECT && c->nranks > 1 ? 2 : 1); phase++) {
            NCCLCHK(ncclGroupStart());
            ncclResult_t r = ncclSuccess;
            for (int s = 0; s < 7 && r == ncclSuccess; s++)
                if (acc[s])
                    r = sum_to_root(acc[s], n, dt, esz, root, algo, rank, c->nranks, c->comm, stream, phase == 1);
            if (r == ncclSuccess && phase == 0 && have_cnt)
                r = ncclReduce(cnt, cnt, 6, ncclUint64, ncclSum, root, c->comm, stream);
            const ncclResult_t re = ncclGroupEnd();
            if (r != ncclSuccess)
                return fail("plane sum (inside the group)", ncclGetErrorString(r));
            if (re != ncclSuccess)
                return fail("ncclGroupEnd", ncclGetErrorString(re));
        }
    }
    // 3. accumulators -> f32 maps (on the root these are the sums; elsewhere the rank's own partial maps)
    if (slicer_plane_finalize(h) != SLICER_OK)
        return fail("slicer_plane_finalize", slicer_last_error(h));
    return SLICER_OK;
}

int slicer_rccl_plane_reduce(slicer_handle h, slicer_rccl_comm c, int root, int per_type)
{
    (void)per_type;  // kept for source compatibility: the live accumulators decide what is summed
    return slicer_rccl_plane_reduce_ex(h, c, root, SLICER_RCCL_REDUCE_ROOTED);
}

}  // extern "C"
