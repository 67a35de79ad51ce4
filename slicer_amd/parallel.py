"""Multi-GPU layout of the path: one process per GPU, the reference's own work split, RCCL for the sum.

Reference: slicer-v2.cpp:162-175 (contiguous sub-file range per rank, last rank takes the remainder) and
slicer-v2.cpp:214-217 (7 x MPI_Reduce(MPI_FLOAT, MPI_SUM, root 0) per plane).  Here the reduce runs over
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests) directly on the
device maps owned by the slicer handle -- no staging copy.
"""
import numpy as np


def file_range(numfiles, numprocs, myid):
    """ffmin, ffmax of rank `myid` (slicer-v2.cpp:162-175)."""
    intdiv, remaindiv = numfiles // numprocs, numfiles % numprocs
    ffmin, ffmax = myid * intdiv, (myid + 1) * intdiv
    if myid == numprocs - 1:
        ffmax += remaindiv
    return ffmin, ffmax


def device_tensor(torch, ptr, n, dtype="<f4"):
    """Zero-copy torch view of `n` elements at device pointer `ptr` (memory owned by the slicer handle)."""
    class _Iface:
        pass
    o = _Iface()
    o.__cuda_array_interface__ = {"shape": (int(n),), "typestr": dtype, "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(o, device="cuda")


def reduce_planes(S, dist, torch, root=0, per_type=False):
    """Sum the finalized device maps of every plane of the current pass onto `root` (in place on root).
    Replaces the MPI_Reduce calls of slicer-v2.cpp:214-217; per_type=False skips the six per-type maps
    the reference reduces even when it never writes them (partinplanes == false)."""
    n = S.npix * S.npix
    for p in range(S.n_planes):
        d_tot, d_toti = S.plane_device_maps(p)
        dist.reduce(device_tensor(torch, d_tot, n), dst=root, op=dist.ReduceOp.SUM)
        if per_type:
            for t in range(6):
                if d_toti[t]:
                    dist.reduce(device_tensor(torch, d_toti[t], n), dst=root, op=dist.ReduceOp.SUM)


def reduce_host_maps(dist, torch, maps, root=0):
    """Same sum for host arrays (gloo): maps is a list of float32 numpy arrays, reduced in place on root."""
    for m in maps:
        t = torch.from_numpy(np.ascontiguousarray(m))
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
        if dist.get_rank() == root:
            m[...] = t.numpy().reshape(m.shape)
    return maps
