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
