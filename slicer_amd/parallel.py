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


REDUCE_ALGOS = ("rooted", "rs_gather", "p2p")


def _slices(n, world):
    """Slice j of an n-element accumulator: [j * q, (j + 1) * q), the last one also takes the n % world tail."""
    q = n // world
    return [(j * q, (j + 1) * q if j < world - 1 else n) for j in range(world)]


class _Works:
    """Pending communication of one reduce_planes call: `wait()` completes it (stages run in order)."""

    def __init__(self):
        self.stages = []

    def add(self, works, then=None):
        self.stages.append(([w for w in works if w is not None], then))

    def wait(self):
        for works, then in self.stages:
            for w in works:
                w.wait()
            if then is not None:
                then()
        self.stages = []


def _sum_rooted(dist, t, root, async_op):
    """slicer-v2.cpp:214: one rooted sum, left to the library (RCCL: a ring or tree over xGMI)."""
    return [dist.reduce(t, dst=root, op=dist.ReduceOp.SUM, async_op=async_op)], None


def _sum_rs_gather(dist, t, root, async_op, world, rank):
    """reduce-scatter (every rank ends up with the sum of its own 1/N slice) + the slices gathered on
    the root -- with point-to-point xGMI links each slice crosses one link per hop instead of the whole map crossing
    every link of a ring.  RCCL's reduce_scatter; the tail n % N goes through a small rooted reduce."""
    n = t.numel()
    q = n // world
    works = []
    if q:
        body = t[:q * world]
        works.append(dist.reduce_scatter_tensor(body[rank * q:(rank + 1) * q], body, op=dist.ReduceOp.SUM,
                                                async_op=async_op))
        works.append(dist.gather(body[rank * q:(rank + 1) * q],
                                 [body[j * q:(j + 1) * q] for j in range(world)] if rank == root else None, dst=root,
                                 async_op=async_op))
    if n - q * world:
        works.append(dist.reduce(t[q * world:], dst=root, op=dist.ReduceOp.SUM, async_op=async_op))
    return works, None


def _sum_p2p(dist, torch, t, root, world, rank, tag0):
    """The direct form written out with sends and receives (works on every backend; the gloo rehearsal of the tests
    runs it): rank r sends slice j to rank j for every j != r -- one message per dedicated link --, adds the N - 1
    slices it receives to its own, and the reduced slices then travel to the root.  Every exchange is one
    batch_isend_irecv (a grouped RCCL send/recv: sends and receives of a pair progress together, no ordering deadlock).
    Returns (works, then): `then` runs after the works (local sums) and returns the second hop's works."""
    n = t.numel()
    sl = _slices(n, world)
    lo, hi = sl[rank]
    tmp = {}
    ops = []
    for j in range(world):
        if j == rank:
            continue
        if sl[j][1] > sl[j][0]:
            ops.append(dist.P2POp(dist.isend, t[sl[j][0]:sl[j][1]], j, tag=tag0 + rank))
        if hi > lo:
            tmp[j] = torch.empty(hi - lo, dtype=t.dtype, device=t.device)
            ops.append(dist.P2POp(dist.irecv, tmp[j], j, tag=tag0 + j))
    works = dist.batch_isend_irecv(ops) if ops else []

    def second_hop():
        mine = t[lo:hi]
        for j in sorted(tmp):  # fixed order: rank-invariant arithmetic for the f32 / f64 accumulators
            mine += tmp[j]
        hop = []
        if rank == root:
            for j in range(world):
                if j != root and sl[j][1] > sl[j][0]:
                    hop.append(dist.P2POp(dist.irecv, t[sl[j][0]:sl[j][1]], j, tag=tag0 + world + j))
        elif hi > lo:
            hop.append(dist.P2POp(dist.isend, mine, root, tag=tag0 + world + rank))
        return dist.batch_isend_irecv(hop) if hop else []

    return works, second_hop


def reduce_planes(S, dist, torch, root=0, per_type=None, async_op=False, finalize=True, algo="rooted", sync_free=None):
    """Sum the current plane pass over the ranks onto `root`, in the accumulator's own type, then convert to f32 maps.

    Replaces the MPI_Reduce calls of slicer-v2.cpp:214-217.  Call after the last file_end() and INSTEAD of
    plane_finalize().  Every rank issues the same collectives even if its sub-files lack a particle type (zero-filled
    stand-ins, as the reference reduces all seven maps unconditionally); f64 / fixed-point accumulators are summed
    before their single rounding, so a FIXED64 N-rank result is bitwise the 1-rank result (with any `algo`: integer
    sums do not depend on the order).
    S: slicer_amd.Slicer, or any object with reduce_meta_get/set, plane_accumulators, plane_device_counts,
    plane_finalize, npix, n_planes (the gloo rehearsal in tests/ uses a host stand-in).
    algo: "rooted" (one library reduce per map), "rs_gather" (reduce-scatter + gather of the slices, RCCL only) or
    "p2p" (the same direct pattern written with sends / receives and local sums; any backend).
    STREAMS (device backends): the handle must work on torch's CURRENT stream (S.set_stream(torch.cuda.current_stream()
    .cuda_stream) and call this under that stream): the collectives are ordered behind the deposits -- and
    plane_finalize behind the collectives -- through that stream; a handle on a stream of its own would race both.
    This is checked.  async_op=True returns an object whose wait() completes the communication (call it under the same
    stream), after which S.plane_finalize() may run.
    sync_free (default: whenever S offers reduce_meta_get_async and the backend is a device one): no host
    synchronisation with the deposit stream -- the reduce meta is combined from host-known entries on a side stream and
    the negativity guard by a MAX all-reduce of the device flag next to the map sums."""
    del per_type  # the live accumulators decide (kept for source compatibility)
    assert algo in REDUCE_ALGOS, algo
    on_device = dist.get_backend() != "gloo"
    world, rank = dist.get_world_size(), dist.get_rank()
    if on_device and hasattr(S, "get_stream"):
        cur = int(torch.cuda.current_stream().cuda_stream)
        if int(S.get_stream()) != cur:
            raise RuntimeError("reduce_planes: the slicer handle does not work on torch's current stream "
                               "(S.set_stream(torch.cuda.current_stream().cuda_stream)); its deposits and finalize "
                               "would not be ordered with the collectives")
    if sync_free is None:
        sync_free = on_device and hasattr(S, "reduce_meta_get_async")
    if sync_free:
        meta = S.reduce_meta_get_async()
        side = getattr(reduce_planes, "_side", None)
        if side is None:
            side = reduce_planes._side = torch.cuda.Stream()
        with torch.cuda.stream(side):  # nothing of the deposit stream is waited for: host-known integers only
            meta = combine_meta(dist, torch, meta, True)
        S.reduce_meta_set(meta)
    else:
        S.reduce_meta_set(combine_meta(dist, torch, S.reduce_meta_get(), on_device))
    n = S.npix * S.npix
    out = _Works()
    first, hops = [], []
    tag = 1 << 20
    for p in range(S.n_planes):
        acc, elem = S.plane_accumulators(p)
        for ref in acc:
            if ref is None:
                continue
            t = _as_tensor(torch, ref, n, elem)
            if algo == "rooted" or world == 1:
                w, then = _sum_rooted(dist, t, root, async_op)
            elif algo == "rs_gather":
                w, then = _sum_rs_gather(dist, t, root, async_op, world, rank)
            else:
                w, then = _sum_p2p(dist, torch, t, root, world, rank, tag)
                tag += 2 * world
            first += w
            if then is not None:
                hops.append(then)
        cnt = S.plane_device_counts(p)
        if cnt is not None:
            first.append(dist.reduce(_as_tensor(torch, cnt, 6, 2), dst=root, op=dist.ReduceOp.SUM, async_op=async_op))
    if sync_free:
        flag = device_tensor(torch, S.plane_device_guard(), 1, "<i4")
        first.append(dist.all_reduce(flag, op=dist.ReduceOp.MAX, async_op=async_op))
    out.add(first)
    if hops:
        second = []

        def run_hops():
            for h in hops:
                second.extend(h())
        out.stages[-1] = (out.stages[-1][0], run_hops)
        out.stages.append((second, None))  # filled in by run_hops before it is waited on
    if async_op:
        return out
    out.wait()
    if finalize:
        S.plane_finalize()
    return out


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
