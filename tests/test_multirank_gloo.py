"""World-size-2 rehearsal of the N>1 path on CPU (gloo): the reference's sub-file partition
(slicer-v2.cpp:162-175) + rank sum (slicer-v2.cpp:214-217) give the single-rank maps.  The per-rank
compute is the oracle here (no GPU in this container); on the GPU box the same partition and the same
reduce call run over RCCL on the device maps (bench.py --shard files, slicer_amd.parallel.reduce_planes)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import oracle
    from slicer_amd import parallel, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    BOX = 1000.0
    files, first = [], 0
    for ff in range(5):
        n = 3000 + 17 * ff
        files.append(dict(npart=[0, n, 0, 0, 0, 0], massarr=[0, 0.25, 0, 0, 0, 0], boxsize=BOX,
                          pos=synth.positions(first, n, BOX)))
        first += n
    args = (32, False, True, 3.0, 4.0, 0, 0.25, (-1, 1, -1), 3, (0.3, 0.6, 0.1), 3.0)
    ffmin, ffmax = parallel.file_range(len(files), world, rank)
    rc, tot, toti, nsel = oracle.create_density_maps(files, ffmin, ffmax, *args)
    maps = [tot] + [toti[t] for t in range(6)]           # the 7 reduces of slicer-v2.cpp:214-217
    parallel.reduce_host_maps(dist, torch, maps, root=0)
    cnt = torch.from_numpy(nsel.copy())
    dist.reduce(cnt, dst=0, op=dist.ReduceOp.SUM)
    dist.barrier()
    if rank == 0:
        rc, ref_tot, ref_toti, ref_nsel = oracle.create_density_maps(files, 0, len(files), *args)
        q.put((bool(np.array_equal(maps[0], ref_tot)), bool(np.array_equal(maps[2], ref_toti[1])),
               bool(np.array_equal(cnt.numpy(), ref_nsel)), (ffmin, ffmax)))
    dist.destroy_process_group()


def _files_with_missing_species(BOX=1000.0):
    """4 sub-files; particle type 4 (stars) lives only in sub-file 0, so with 2 ranks (files 0-1 | 2-3) rank 1 never
    sees it -- the case in which per-rank `if (seen[t])` collectives deadlock."""
    from slicer_amd import synth
    files, first = [], 0
    for ff in range(4):
        n1, n4 = 2500 + 13 * ff, (700 if ff == 0 else 0)
        pos = synth.positions(first, n1 + n4, BOX)
        first += n1 + n4
        files.append(dict(npart=[0, n1, 0, 0, n4, 0], massarr=[0, 0.25, 0, 0, 0.0625, 0], boxsize=BOX, pos=pos))
    return files


def _worker_acc(rank, world, port, q, algo="rooted"):
    """reduce_planes (the code path bench.py and the C RCCL wrapper follow) on host stand-ins of the handles."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from host_rank import HostRankPass
    from slicer_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rnd = dict(sgn=(-1, 1, -1), face=3, center=(0.3, 0.6, 0.1), rcase=3.0)
    files = _files_with_missing_species()
    lo, hi = parallel.file_range(len(files), world, rank)
    out = {}
    for mode, wtm in (("fixed64", True), ("fixed64", False), ("f64", True), ("f32", True), ("f32", False), ("ngp", True),
                      ("ngp", False)):
        mine = HostRankPass(files[lo:hi], 32, 0.25, 3.0, 4.0, rnd, mode=mode, want_type_maps=wtm)
        had_type4 = mine.acc[4] is not None
        parallel.reduce_planes(mine, dist, torch, root=0, algo=algo)
        assert mine.finalized
        if rank == 0:
            one = HostRankPass(files, 32, 0.25, 3.0, 4.0, rnd, mode=mode, want_type_maps=wtm)
            tot, toti = mine.maps()
            rtot, rtoti = one.maps()
            if mode == "fixed64":  # integer sums: bitwise the single-rank result, accumulators and maps alike
                ok = all((a is None) == (b is None) and (a is None or np.array_equal(a, b))
                         for a, b in zip(mine.acc, one.acc))
                ok = ok and np.array_equal(tot.view(np.uint32), rtot.view(np.uint32))
                ok = ok and np.array_equal(toti.view(np.uint32), rtoti.view(np.uint32))
            elif mode == "ngp":    # masses are powers of two: every f32 sum is exact, any order
                ok = np.array_equal(tot, rtot) and np.array_equal(toti, rtoti)
            else:
                ok = np.allclose(tot, rtot, rtol=2e-6, atol=0) and np.allclose(toti, rtoti, rtol=2e-6, atol=0)
            ok = ok and np.array_equal(mine.counts, one.counts)
            if wtm:
                ok = ok and float(rtoti[4].sum()) > 0 and np.allclose(toti[4], rtoti[4], rtol=2e-6, atol=0)
            out[f"{mode}/{int(wtm)}"] = bool(ok)
        else:
            # rank 1 really lacked the species (with 2 or 3 ranks type 4 lives in rank 0's range only)
            out[f"{mode}/{int(wtm)}"] = (not had_type4) if wtm else True
    # the negativity guard of any rank reaches the root
    bad = dict(npart=[0, 4, 0, 0, 0, 0], massarr=[0, 0.25, 0, 0, 0, 0], boxsize=1000.0,
               pos=np.full((4, 3), 2600.0 if rank == 1 else 500.0, np.float32))
    g = HostRankPass([bad], 32, 0.25, 3.0, 4.0, dict(sgn=(-1, -1, -1), face=1, center=(0., 0., 0.), rcase=0.0),
                     mode="f32")
    parallel.reduce_planes(g, dist, torch, root=0, algo=algo)
    out["neg"] = bool(g.neg_remote) and (g.neg == (1 if rank == 1 else 0))
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,algo", [(2, "rooted"), (2, "p2p"), (3, "p2p")])
def test_two_rank_accumulator_typed_reduce_with_a_species_missing_on_one_rank(world, algo):
    """Rank-invariant collective set (no deadlock when rank 1's sub-files lack type 4), sums in the accumulator type
    (FIXED64 N-rank == 1-rank bitwise), selected-particle counters summed, negativity guard propagated -- with the
    rooted library reduce and with the direct reduce-scatter + gather-to-root written as sends / receives (SURVEY S5;
    three ranks: the 1024-pixel maps do not divide evenly, the last slice takes the tail)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_acc, args=(r, world, port, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert all(res[r].values()), (r, res[r])


def _worker_steps(rank, world, port, q):
    """StepGather (bench.py --gather on, timed in every --gpus N run as reduce_layout.steps_gather) on CPU tensors: owners build whole steps, the root collects."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from slicer_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_maps, numel, steps = 3, 1000, 11
    G = parallel.StepGather(dist, torch, world, rank, n_maps, numel, torch.float32, "cpu", root=0, depth=2)

    def build(i):  # what the owner of step i produces
        return [torch.full((numel,), float(100 * i + p)) for p in range(n_maps)]
    got = {}
    pending = []
    for i in range(steps):
        if G.owner(i) == rank:
            maps = build(i)
            if rank == 0:
                got[i] = [m.clone() for m in maps]
            else:
                G.send(i, maps)
        elif rank == 0:
            peer_steps = [j for j in pending if G.owner(j) == G.owner(i)]
            if len(peer_steps) >= 2:  # ring depth 2 per peer: consume the older step before its slot is reused
                j = peer_steps[0]
                got[j] = [t.clone() for t in G.complete(j)]
                pending.remove(j)
            G.expect(i)
            pending.append(i)
    for j in pending:
        got[j] = [t.clone() for t in G.complete(j)]
    G.finish()
    dist.barrier()
    ok = True
    if rank == 0:
        ok = sorted(got) == list(range(steps)) and all(
            bool((got[i][p] == float(100 * i + p)).all()) for i in range(steps) for p in range(n_maps))
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_step_gather_delivers_every_map_to_the_root(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_steps, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res.values()), res


def test_file_partition_matches_reference_rule():
    from slicer_amd import parallel
    import oracle
    for nf in (1, 4, 5, 8, 13):
        for np_ in (1, 2, 3, 4, 8):
            got = [parallel.file_range(nf, np_, r) for r in range(np_)]
            assert got == [oracle.file_range(nf, np_, r) for r in range(np_)]
            covered = [f for a, b in got for f in range(a, b)]
            assert covered == list(range(nf))


@pytest.mark.timeout(300)
def test_two_rank_partition_and_reduce_equals_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[:3] == (True, True, True), res
