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


_ELEM_TYPESTR = {0: "<f4", 1: "<f8", 2: "<i8"}  # FIXED64 sums are integer sums mod 2^64: int64 adds the same bits


def _as_tensor(torch, ref, n, elem):
    """ref: a device pointer (int, memory owned by the slicer handle) or a host numpy array (CPU rehearsal)."""
    if isinstance(ref, np.ndarray):
        return torch.from_numpy(ref.reshape(-1).view(_ELEM_TYPESTR[elem]))
    return device_tensor(torch, ref, n, _ELEM_TYPESTR[elem])


def combine_meta(dist, torch, meta, on_device):
    """Element-wise MAX of the 24-int reduce meta over all ranks (which accumulators are live anywhere, their
    FIXED64 scales, the negativity guard): one small all-reduce that makes the later set of collectives rank-invariant."""
    t = torch.tensor(meta, dtype=torch.int32, device="cuda" if on_device else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(x) for x in t.tolist()]


def reduce_planes(S, dist, torch, root=0, per_type=None, async_op=False, finalize=True):
    """Sum the current plane pass over the ranks onto `root`, in the accumulator's own type, then convert to f32 maps.

    Replaces the MPI_Reduce calls of slicer-v2.cpp:214-217.  Call after the last file_end() and INSTEAD of
    plane_finalize().  Every rank issues the same collectives even if its sub-files lack a particle type (zero-filled
    stand-ins, as the reference reduces all seven maps unconditionally); f64 / fixed-point accumulators are summed
    before their single rounding, so a FIXED64 N-rank result is bitwise the 1-rank result.
    S: slicer_amd.Slicer, or any object with reduce_meta_get/set, plane_accumulators, plane_device_counts,
    plane_finalize, npix, n_planes (the gloo rehearsal in tests/ uses a host stand-in).
    async_op=True returns the pending works without finalizing: wait on them, then call S.plane_finalize()."""
    del per_type  # the live accumulators decide (kept for source compatibility)
    on_device = dist.get_backend() != "gloo"
    S.reduce_meta_set(combine_meta(dist, torch, S.reduce_meta_get(), on_device))
    n = S.npix * S.npix
    works = []
    for p in range(S.n_planes):
        acc, elem = S.plane_accumulators(p)
        for ref in acc:
            if ref is not None:
                works.append(dist.reduce(_as_tensor(torch, ref, n, elem), dst=root, op=dist.ReduceOp.SUM,
                                         async_op=async_op))
        cnt = S.plane_device_counts(p)
        if cnt is not None:
            works.append(dist.reduce(_as_tensor(torch, cnt, 6, 2), dst=root, op=dist.ReduceOp.SUM, async_op=async_op))
    if async_op:
        return works
    if finalize:
        S.plane_finalize()
    return []


def reduce_host_maps(dist, torch, maps, root=0):
    """Same sum for host arrays (gloo): maps is a list of float32 numpy arrays, reduced in place on root."""
    for m in maps:
        t = torch.from_numpy(np.ascontiguousarray(m))
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
        if dist.get_rank() == root:
            m[...] = t.numpy().reshape(m.shape)
    return maps


class StepGather:
    """Strong scaling by whole steps: step i (one snapshot -> its plane maps) belongs to rank i % world, which builds
    it alone -- no partial sums exist, so nothing is reduced -- and every finished map then travels to `root` point to
    point (torch.distributed isend / irecv = RCCL send / recv over the direct xGMI link of that pair), overlapped with
    the steps that follow.  The root ends up with every map, as rank 0 does in slicer-v2.cpp:214-222, but each link only
    carries the maps its own rank built (a rooted reduce would push every map through every rank).

    Root side: `expect(i)` posts the receives of step i (owner != root) into a ring of `depth` slots per peer.
    Owner side: `send(i, tensors)` posts the sends of step i and returns the works to wait on before the buffers are
    reused.  `finish()` waits for everything still in flight.  Tensors: CPU (gloo rehearsal) or device."""

    def __init__(self, dist, torch, world, rank, n_maps, numel, dtype, device, root=0, depth=2):
        self.dist, self.world, self.rank, self.root, self.depth = dist, world, rank, root, depth
        self.n_maps = n_maps
        self.slots, self.inflight = {}, {}
        if rank == root:
            for peer in range(world):
                if peer != root:
                    self.slots[peer] = [[torch.empty(numel, dtype=dtype, device=device) for _ in range(n_maps)]
                                        for _ in range(depth)]
                    self.inflight[peer] = [None] * depth
        self.sent = []

    def owner(self, i):
        return i % self.world

    def expect(self, i):
        """Root: post the receives of step i into its ring slot; `complete(i)` makes the slot's tensors valid."""
        peer = self.owner(i)
        assert self.rank == self.root and peer != self.root
        k = (i // self.world) % self.depth
        self._wait_slot(peer, k)  # the slot's previous occupant must have arrived (and been consumed by the caller)
        # (tags keep gloo's matching unambiguous with several messages of one pair in flight; RCCL matches in order)
        works = [self.dist.irecv(t, src=peer, tag=i * self.n_maps + p) for p, t in enumerate(self.slots[peer][k])]
        self.inflight[peer][k] = works
        return self.slots[peer][k]

    def complete(self, i):
        """Root: wait until the maps of step i have arrived; returns them (valid until the slot is reused by
        expect(i + depth * world))."""
        peer = self.owner(i)
        k = (i // self.world) % self.depth
        self._wait_slot(peer, k)
        return self.slots[peer][k]

    def _wait_slot(self, peer, k):
        works = self.inflight[peer][k]
        if works is not None:  # each work is waited on exactly once (a second wait on a gloo receive blocks)
            for w in works:
                w.wait()
            self.inflight[peer][k] = None

    def send(self, i, tensors):
        """Owner (not the root): post the sends of step i."""
        assert self.owner(i) == self.rank and self.rank != self.root and len(tensors) == self.n_maps
        works = [self.dist.isend(t, dst=self.root, tag=i * self.n_maps + p) for p, t in enumerate(tensors)]
        self.sent.append(works)
        return works

    def finish(self):
        for works in self.sent:
            for w in works:
                w.wait()
        self.sent = []
        for peer, slots in self.inflight.items():
            for k in range(len(slots)):
                self._wait_slot(peer, k)
