#!/usr/bin/env python3
"""bench.py -- headline benchmark of the mass-assignment hot path on MI355X.

Metric (BASELINE.json): particles/s deposited (TSC, 4096^2 map); max |dpixel| vs ref.
Workload (BASELINE config[2], the configuration the metric is quoted on; it fits one GPU):
  4 synthetic GADGET-2 snapshots of 512^3 particles (8 sub-files of 2^24 each, raw POS blocks resident
  in HBM before the timed region), 4096^2 TSC maps, the 4 lens planes of one box replication built in
  one pass (S8d geometry: rcase=3, ld=3+{0,.25,.5,.75}, fov=0.25 rad, face 3, signs (-,+,-),
  centre (.3,.6,.1)).  One step = one snapshot -> its 4 plane maps: plane_begin (zero) + 8 sub-file
  deposits + finalize.  Steps cycle through the resident snapshots.
N > 1 (one process per GPU; `python bench.py --gpus N` starts the N ranks itself when no launcher did):
  --shard steps (default, "scaling": "strong"): the SAME job and the same steps; step i (one snapshot -> its 4 plane
      maps) is built by rank i % N alone, and every finished map then travels to rank 0 point to point over RCCL/xGMI
      (rank 0 ends up with every map, as in slicer-v2.cpp:214-222), overlapped with the steps that follow.  Snapshots
      and lens planes shard without partial sums, so nothing needs adding up -- and at 5e10 particles/s a per-snapshot
      sum of 4 x 64 MiB over xGMI costs more than the snapshot.
  --shard files ("strong"): split the way the reference splits it -- every snapshot's sub-files in contiguous ranges
      over the ranks (slicer-v2.cpp:162-175) -- followed by the per-plane SUM to rank 0 (slicer-v2.cpp:214-217) over
      RCCL/xGMI, in the accumulator's own type; the sum of step i overlaps the deposits of step i+1.
  --shard snapshots ("weak"): every rank owns its own boxes, no data-path collective (SURVEY S8e level 2).
value = deposits of all ranks / max-over-ranks time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BOX = 1000.0
# Random.x0/y0/z0 are rand()/float(RAND_MAX) in the reference (densitymaps.cpp:188-190): binary32 values stored in
# doubles.  The S8d centre (0.3, 0.6, 0.1) is therefore used as the reference could produce it: rounded to binary32.
_F32 = lambda v: float(__import__("numpy").float32(v))  # noqa: E731
RND = dict(sgn=(-1, 1, -1), face=3, center=(_F32(0.3), _F32(0.6), _F32(0.1)), rcase=3.0)
LDS = [3.0, 3.25, 3.5, 3.75]
LD2S = [3.25, 3.5, 3.75, 4.0]
FOV = 0.25
MASS = 0.0123
SEED = 0x51CE2
HBM_PEAK = 8.0e12    # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec); 6.29e12 measured-achievable
PCIE_PEAK = 63.0e9   # B/s, PCIe Gen5 x16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--side", type=int, default=512, help="particles per snapshot = side^3")
    ap.add_argument("--npix", type=int, default=4096)
    ap.add_argument("--snapshots", type=int, default=4, help="resident snapshots per rank")
    ap.add_argument("--files", type=int, default=8, help="sub-files per snapshot")
    ap.add_argument("--planes", type=int, default=4, choices=[1, 2, 4])
    ap.add_argument("--mas", default="tsc", choices=["tsc", "ngp"])
    ap.add_argument("--accum", default="f32", choices=["f32", "f64", "fixed64"])
    ap.add_argument("--algo", default="auto", choices=["auto", "direct", "binned"])
    ap.add_argument("--clustered", action="store_true")
    ap.add_argument("--hydro", action="store_true",
                    help="per-particle masses (type 0, massarr = 0: densitymaps.cpp:358-372) instead of one mass per type")
    ap.add_argument("--shard", default="auto", choices=["auto", "snapshots", "files", "steps"],
                    help="auto: steps (strong scaling: step i is built by rank i % N, whole) when N > 1")
    ap.add_argument("--streams", type=int, default=1,
                    help="snapshots in flight per GPU: consecutive steps alternate between this many handles, each on "
                         "its own HIP stream.  With more than one the job is timed on ONE stream first (`single_stream`; "
                         "the per-kernel durations of `roofline` / `kernels` belong to that run) and `value` is the "
                         "multi-stream run.  Measured gain of 2 over 1: about 2 %, inside the run-to-run scatter "
                         "(DESIGN.md S5), hence the default of 1")
    ap.add_argument("--gather", default="off", choices=["on", "off"],
                    help="--shard steps: also send every finished map to rank 0 (point to point, overlapped).  Off: the maps "
                         "stay on the rank that built them, as for N = 1 (snapshots and lens planes are independent: each "
                         "rank would write its own planes); the gathering variant is timed in the same run as "
                         "config.reduce_layout.steps_gather")
    ap.add_argument("--secondary-timeout", type=float, default=120.0,
                    help="N > 1: seconds the secondary layouts (config.reduce_layout) may take before the line is printed "
                         "without them")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: wait for each step's rank sum before the next step")
    ap.add_argument("--reduce-algo", default="rooted", choices=["rooted", "rs_gather", "p2p"],
                    help="--shard files: the per-plane rank sum as one library reduce per map (rooted), as reduce-scatter "
                         "+ gather of the slices (rs_gather), or as the same direct pattern written with sends / receives "
                         "and local sums (p2p); slicer_amd/parallel.py")
    ap.add_argument("--reduce-layout", default="auto", choices=["auto", "on", "off"],
                    help="after the main layout, also time the reference's own layout in the same run -- sub-files split "
                         "over the ranks + the per-plane RCCL sum to rank 0 (slicer-v2.cpp:162-175, 214-217) -- and report "
                         "it as config.reduce_layout (auto: when N > 1 and the main layout is another one)")
    ap.add_argument("--cpu-baseline", dest="cpu", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--cpu-particles", type=int, default=1 << 24, help="particles per CPU-baseline worker file")
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--parity", default="auto", choices=["auto", "on", "off"],
                    help="max |dpixel| vs the oracle on one sub-file of the workload (outside the timed region)")
    ap.add_argument("--e2e", default="auto", choices=["auto", "on", "off"],
                    help="file -> C++ createDensityMaps -> host maps timing (outside the timed region)")
    ap.add_argument("--profile-steps", type=int, default=2)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a scalar port of the reference path) on the host cores, parallelised the
# way the reference is (one process per contiguous sub-file range, slicer-v2.cpp:162-175).  Each
# worker builds the planes of its own sub-file with one createDensityMaps-equivalent call per plane,
# exactly what slicer-v2.cpp's plane loop does (the reference re-reads the snapshot for every plane).
#   compute-only:        positions already in RAM when the clock starts (the oracle still zeroes six per-type
#                        maps per file and runs the map-wide sums of densitymaps.cpp:496-513, as the reference does)
#   reference-faithful:  + the reference's read pattern: three 4-byte stream reads per particle from the
#                        (page-cached) POS block, once per plane (gadget2io.cpp:200-202)
# Worker 0 owns sub-file 0 of snapshot 0 of the GPU workload and hands its maps back: they are the
# oracle side of "max |dpixel| vs ref".
# ------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    first, n, npix, ngp, planes, clustered, faithful, want_maps = args
    import oracle
    from slicer_amd import synth
    pos = synth.positions(first, n, BOX, seed=SEED, clustered=clustered)
    f = dict(npart=[0, n, 0, 0, 0, 0], massarr=[0, MASS, 0, 0, 0, 0], boxsize=BOX, pos=pos)
    oracle.lib()
    path = None
    if faithful:
        fd, path = tempfile.mkstemp(prefix="slicer_cpu_", suffix=".pos", dir="/tmp")
        with os.fdopen(fd, "wb") as fh:
            fh.write(pos.tobytes())
    t0 = time.perf_counter()
    dep, maps = 0, []
    for p in range(planes):
        if faithful:
            _faithful_read(path, n)  # buffered stream, 3 reads of 4 bytes per particle
        rc, tot, toti, nsel = oracle.create_density_maps([f], 0, 1, npix, False, ngp, LDS[p], LD2S[p], 0, FOV,
                                                         RND["sgn"], RND["face"], RND["center"], RND["rcase"])
        dep += int(nsel[1])
        if want_maps:
            maps.append(tot.copy())
    dt = time.perf_counter() - t0
    if path:
        os.unlink(path)
    return dt, dep, maps


def _faithful_read(path, n):
    """3 x fin.read(4 bytes) per particle (gadget2io.cpp:200-202) through a buffered stream; the loop is native
    (oracle/slicer_oracle.c: orc_stream_read_pos)."""
    import ctypes

    import oracle
    L = oracle.lib()
    L.orc_stream_read_pos.restype = ctypes.c_long
    L.orc_stream_read_pos.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.POINTER(ctypes.c_float)]
    sink = ctypes.c_float()
    got = L.orc_stream_read_pos(path.encode(), n, ctypes.byref(sink))
    assert got == n, (got, n)


def cpu_baseline(a, want_maps):
    import multiprocessing as mp
    avail = len(os.sched_getaffinity(0))
    cores = a.cpu_cores or max(1, min(16, avail))
    n = a.cpu_particles
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    out = {"unit": "particles/s", "kind": "port"}
    ref_maps = None

    def run(ncores, faithful, maps):
        jobs = [(i * n, n, a.npix, a.mas == "ngp", a.planes, a.clustered, faithful, maps and i == 0)
                for i in range(ncores)]
        with ctx.Pool(ncores) as pool:
            res = pool.map(_cpu_worker, jobs)
        wall = max(r[0] for r in res)
        return sum(r[1] for r in res) / wall, ncores * n * a.planes / wall, wall, res[0][2]

    v, vin, wall, ref_maps = run(cores, False, want_maps)
    out.update(value=v, cores=cores, n_in_per_s=vin,
               sample=f"compute-only: {cores} sub-files x {n} particles, {a.planes} planes each ({a.planes} "
                      f"createDensityMaps-equivalent oracle calls per sub-file, one process per sub-file), {a.npix}^2 "
                      f"{a.mas.upper()}; {wall:.1f} s wall, inputs in RAM")
    if a.cpu != "off":
        v1, vin1, w1, _ = run(1, False, False)
        vf, vinf, wf, _ = run(cores, True, False)
        out["variants"] = {
            "compute_only_R1": {"value": v1, "cores": 1, "n_in_per_s": vin1, "wall_s": w1},
            f"reference_faithful_R{cores}": {"value": vf, "cores": cores, "n_in_per_s": vinf, "wall_s": wf,
                                             "what": "compute-only + 3 buffered stream reads of 4 B per particle from a "
                                                     "page-cached POS file, per plane (gadget2io.cpp:200-202)"},
        }
    out["total_s"] = time.perf_counter() - t0
    return out, ref_maps


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the rank processes ourselves, before anything touches a GPU
# ------------------------------------------------------------------------------------------------
def launch_ranks(a):
    import torch  # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} needs {a.gpus} devices, this node shows {have} "
                         f"(it will not silently run on fewer)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    line = [ln for ln in out0.decode().splitlines() if ln.startswith("{")]
    if any(codes) or not line:
        raise SystemExit(f"rank exit codes {codes}; no result line" if not line else f"rank exit codes {codes}")
    print(line[-1], flush=True)


# ------------------------------------------------------------------------------------------------
def main():
    a = parse()
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE is 1: refusing to report a {a.gpus}-GPU number from one rank")
        a.gpus = world
    shard = a.shard if a.shard != "auto" else ("steps" if world > 1 else "snapshots")

    cpu, ref_maps = None, None
    want_parity = a.parity == "on" or (a.parity == "auto" and a.mas == "tsc" and a.side ** 3 // a.files >= a.cpu_particles
                                       and not a.hydro)
    if rank == 0 and world == 1 and a.cpu != "off":
        cpu, ref_maps = cpu_baseline(a, want_parity)  # before anything touches the GPU in this process
    elif rank == 0 and want_parity and a.parity == "on":
        import torch  # noqa: F401  (the worker imports slicer_amd in THIS process: torch's HIP runtime has to be first)
        _, _, ref_maps = _cpu_worker((0, a.cpu_particles, a.npix, a.mas == "ngp", a.planes, a.clustered, False, True))

    import torch  # first, so that libslicer_amd.so binds to the HIP runtime torch already loaded
    import torch.distributed as dist

    import slicer_amd
    from slicer_amd import parallel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: needs device {local_rank}, node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("SLICER_BENCH_FORCE_DIST") == "1"  # the latter: rehearse on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints its version banner on STDOUT when the first communicator comes up; the driver reads one JSON
        # line from stdout, so stdout is pointed at stderr until the communicator exists.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    n_snap = a.side ** 3
    files = a.files
    per_file = n_snap // files
    assert per_file * files == n_snap
    mas = slicer_amd.MAS_NGP if a.mas == "ngp" else slicer_amd.MAS_TSC
    accum = dict(f32=slicer_amd.ACC_F32, f64=slicer_amd.ACC_F64, fixed64=slicer_amd.ACC_FIXED64)[a.accum]
    algo = dict(auto=slicer_amd.ALGO_AUTO, direct=slicer_amd.ALGO_DIRECT, binned=slicer_amd.ALGO_BINNED)[a.algo]
    lds, ld2s = LDS[:a.planes], LD2S[:a.planes]
    if a.planes == 1:
        lds, ld2s = [3.0], [3.25]
    ptype = 0 if a.hydro else 1
    npix2 = a.npix * a.npix
    chunk = per_file
    if os.environ.get("SLICER_BENCH_CHUNK_LOG2"):  # experiment: several kernel passes per sub-file
        chunk = min(per_file, 1 << int(os.environ["SLICER_BENCH_CHUNK_LOG2"]))

    # resident raw POS blocks: [snapshot][file] -> torch buffer (HBM).  Every rank keeps every sub-file of its boxes:
    # the same boxes on all ranks (layouts "steps" and "files", which pick their sub-files out of them) or boxes of its
    # own ("snapshots").
    seed0 = SEED + (1000 * rank if shard == "snapshots" else 0)
    gen = slicer_amd.Slicer(local_rank, max_chunk=chunk)
    gen.set_stream(torch.cuda.current_stream().cuda_stream)
    pos = []
    for s in range(a.snapshots):
        row = []
        for ff in range(files):
            buf = torch.empty(per_file * 3, dtype=torch.float32, device="cuda")
            gen.synth_positions(buf.data_ptr(), ff * per_file, per_file, BOX, seed=seed0 + s, clustered=a.clustered)
            row.append(buf)
        pos.append(row)
    masses = None
    if a.hydro:  # one block of per-particle masses, shared by every sub-file (values in (0.5, 1.5) * MASS)
        masses = (torch.rand(per_file, dtype=torch.float32, device="cuda") + 0.5) * MASS
    torch.cuda.synchronize()
    gen.close()

    class Layout:
        """One way of spreading the job over the ranks: its handles, its step function, its timed run."""

        def __init__(self, shard, reduce_algo="rooted", streams=None, gather=None):
            self.shard, self.reduce_algo = shard, reduce_algo
            streams = a.streams if streams is None else streams
            self.reduce_steps = shard == "files" and use_dist
            self.gather_steps = shard == "steps" and use_dist  # whole steps per rank (step i: rank i % N)
            self.gather = self.gather_steps and (a.gather == "on" if gather is None else gather)  # ... maps to rank 0
            self.overlap = (self.reduce_steps or self.gather) and not a.no_overlap
            # Two handles when the rank sum of step i overlaps the deposits of step i+1: each owns its maps and
            # workspace and works on its own stream; RCCL runs on torch.distributed's communication stream.
            self.n_handles = max(2 if self.overlap else 1, streams)
            self.handles = [slicer_amd.Slicer(local_rank, max_chunk=chunk) for _ in range(self.n_handles)]
            self.streams = ([torch.cuda.Stream() for _ in range(self.n_handles)] if self.n_handles > 1
                            else [torch.cuda.current_stream()])
            for S, st in zip(self.handles, self.streams):
                S.set_stream(st.cuda_stream)
            if shard == "files":  # sub-files split as slicer-v2.cpp:162-175
                lo, hi = parallel.file_range(files, world, rank)
                self.my_files = list(range(lo, hi))
            else:                  # whole snapshots: all sub-files
                self.my_files = list(range(files))
            self.pending = [None] * self.n_handles  # async rank-sum / send works of the handle's previous step
            self.G = None
            if self.gather:
                self.G = parallel.StepGather(dist, torch, world, rank, len(lds), npix2, torch.float32, "cuda", root=0,
                                             depth=2)
            self.own_count = 0

        def close(self):
            for S in self.handles:
                S.close()
            self.handles = []

        def settle(self, k):
            """Step k's rank sum has to be complete before its accumulators become f32 maps (and are reused); sends of
            the handle's previous step have to be complete before its maps are zeroed again."""
            if self.pending[k] is not None:
                with torch.cuda.stream(self.streams[k]):
                    if hasattr(self.pending[k], "wait"):
                        self.pending[k].wait()
                    else:
                        for w in self.pending[k]:
                            w.wait()
                    if self.reduce_steps:
                        self.handles[k].plane_finalize()
                self.pending[k] = None

        def deposit(self, S, s):
            S.plane_begin(a.npix, FOV, lds, ld2s, mas=mas, accum=accum, algo=algo, want_type_maps=False, hydro=a.hydro)
            for ff in self.my_files:
                if a.hydro:
                    S.file_begin([per_file, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], BOX, RND["sgn"], RND["face"],
                                 RND["center"], RND["rcase"])
                    S.deposit_device(0, pos[s][ff].data_ptr(), per_file, masses.data_ptr())
                else:
                    S.file_begin([0, per_file, 0, 0, 0, 0], [0, MASS, 0, 0, 0, 0], BOX, RND["sgn"], RND["face"],
                                 RND["center"], RND["rcase"])
                    S.deposit_device(1, pos[s][ff].data_ptr(), per_file)
                S.file_end()

        def local_step(self, i, k=0):
            """One snapshot -> its finished plane maps on this rank, no communication."""
            with torch.cuda.stream(self.streams[k]):
                self.deposit(self.handles[k], i % a.snapshots)
                self.handles[k].plane_finalize()

        def step(self, i):
            s = i % a.snapshots
            G = self.G
            if self.gather_steps:
                if i % world == rank:  # this rank builds the whole step (with --gather on the root gets the maps)
                    k = self.own_count % self.n_handles
                    self.own_count += 1
                    self.settle(k)
                    with torch.cuda.stream(self.streams[k]):
                        self.deposit(self.handles[k], s)
                        self.handles[k].plane_finalize()
                        if self.gather and rank != 0:
                            maps = [parallel.device_tensor(torch, self.handles[k].plane_device_maps(p)[0], npix2)
                                    for p in range(len(lds))]
                            self.pending[k] = G.send(i, maps)
                            if not self.overlap:
                                self.settle(k)
                elif self.gather and rank == 0:
                    G.expect(i)  # (waits for the ring slot's previous occupant first)
                    if not self.overlap:
                        G.complete(i)
                return
            k = i % self.n_handles
            S = self.handles[k]
            self.settle(k)
            with torch.cuda.stream(self.streams[k]):
                self.deposit(S, s)
                if self.reduce_steps:
                    # slicer-v2.cpp:214: MPI_Reduce(mapxytot, SUM, root 0) per plane -> RCCL over xGMI, on the
                    # accumulators (f32 / f64 / fixed point), converted to f32 maps once, after the sum
                    if self.overlap:
                        self.pending[k] = parallel.reduce_planes(S, dist, torch, root=0, async_op=True,
                                                                 algo=self.reduce_algo)
                    else:
                        parallel.reduce_planes(S, dist, torch, root=0, algo=self.reduce_algo)
                else:
                    S.plane_finalize()

        def drain(self):
            for k in range(self.n_handles):
                self.settle(k)
            if self.G is not None:
                self.G.finish()

        def run(self):
            """Counts, warm-up, the timed K steps (barrier + synchronize on both sides, max over ranks)."""
            # deposits per step (identical for a given snapshot every time it is processed); with --shard files the
            # counters on rank 0 are the rank sums (reduced with the maps)
            dep_per_snap = []
            for s in range(a.snapshots):
                if self.gather_steps:
                    self.local_step(s)  # every rank learns the counts of every snapshot (outside the timed region)
                else:
                    self.step(s)
                    self.drain()
                dep_per_snap.append(sum(int(_counts(self.handles[0 if self.gather_steps else s % self.n_handles], p)[ptype])
                                        for p in range(len(lds))))
            self.algo_mask = self.handles[0].algo_mask()
            for i in range(a.warmup):
                self.step(i)
            self.drain()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                self.step(i)
            self.drain()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0

            my_dep = sum(dep_per_snap[i % a.snapshots] for i in range(a.steps))
            my_in = a.steps * per_file * len(self.my_files)
            if use_dist:
                t = torch.tensor([dt], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            if self.gather_steps:  # the K steps are the job, whichever rank built them: my_dep / my_in are the totals
                tot_dep, tot_in = float(my_dep), float(my_in)
            elif use_dist:
                # --shard files: rank 0's counters already hold the sums over ranks; the others' are partial
                c = torch.tensor([my_dep if (not self.reduce_steps or rank == 0) else 0, my_in], dtype=torch.float64,
                                 device="cuda")
                dist.all_reduce(c, op=dist.ReduceOp.SUM)
                tot_dep, tot_in = float(c[0].item()), float(c[1].item())
            else:
                tot_dep, tot_in = float(my_dep), float(my_in)
            self.my_dep = my_dep
            return dt, tot_dep, tot_in

    # With several snapshots in flight the kernels of different steps share the GPU and their individual durations say
    # nothing about the kernels themselves.  So the job is timed twice: first on ONE stream (`single_stream`, the number
    # the per-kernel HIP-event durations add up to; the profile and parity phases below run on that layout too), then
    # with `--streams` snapshots in flight -- that second run is `value`.
    single_stream, Lmain = None, None
    if a.streams > 1 and not use_dist:
        L = Layout(shard, a.reduce_algo, streams=1)
        dt1, dep1, in1 = L.run()
        single_stream = {"value": dep1 / dt1, "ms_per_step": 1e3 * dt1 / a.steps, "n_in_per_s": in1 / dt1, "streams": 1}
    else:
        L = Layout(shard, a.reduce_algo)
        dt, tot_dep, tot_in = L.run()
        Lmain = L
    reduce_steps, gather_steps, overlap, gathered = L.reduce_steps, L.gather_steps, L.overlap, L.gather
    algo_mask = L.algo_mask
    n_handles, handles, my_files, my_dep = L.n_handles, L.handles, L.my_files, L.my_dep
    S0 = handles[0]
    step, local_step, drain = L.step, L.local_step, L.drain

    # per-kernel durations, live, with HIP events on the launch stream (a few extra untimed steps)
    S0.profile_reset()
    S0.profile_enable(True)
    for i in range(max(1, a.profile_steps) * n_handles):
        if gather_steps:
            if i % n_handles == 0:
                local_step(i)
        else:
            step(i)
    drain()
    torch.cuda.synchronize()
    S0.profile_enable(False)
    prof = S0.profile_get()
    dom_name, (dom_n, dom_ms) = max(prof.items(), key=lambda kv: kv[1][1])
    kernels = {k: {"launches": v[0], "avg_us": 1e3 * v[1] / max(v[0], 1)} for k, v in prof.items()}
    # algorithmic bytes per launch of the dominant kernel (DESIGN.md "Algorithmic bytes")
    # (the profiled handle ran max(1, profile_steps) steps; every project_bin launch streams one sub-file)
    dep_per_step_rank = (my_dep / max(a.steps, 1)) * (len(my_files) / files if reduce_steps else 1.0)
    if gather_steps:
        dep_per_step_rank = my_dep / max(a.steps, 1)
    if dom_name in ("direct_deposit", "project_bin"):
        alg_bytes = 12.0 * per_file
    elif dom_name in ("tile_deposit", "bin_scatter", "bin_sort"):
        alg_bytes = 8.0 * dep_per_step_rank * max(1, a.profile_steps) / max(dom_n, 1)
    else:
        alg_bytes = 4.0 * a.npix * a.npix
    achieved = alg_bytes / (1e-3 * dom_ms / max(dom_n, 1))
    traffic, traffic_step = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(dom_name)
            if ent and ent.get("workload") == f"{a.side}^3/{a.npix}/{a.mas}/{a.algo}/{a.accum}":
                traffic = ent.get("hbm_bytes_per_launch")
            ws = tj.get("whole_step")
            if ws and ws.get("workload") == f"{a.side}^3/{a.npix}/{a.mas}/{a.algo}/{a.accum}":
                traffic_step = ws
        except Exception:
            traffic = None

    # ---- max |dpixel| vs ref: sub-file 0 of snapshot 0 through the same configuration, against the oracle's maps
    parity = None
    if rank == 0 and ref_maps:
        import numpy as np
        n_par = a.cpu_particles
        S0.plane_begin(a.npix, FOV, lds, ld2s, mas=mas, accum=accum, algo=algo, want_type_maps=False)
        S0.file_begin([0, n_par, 0, 0, 0, 0], [0, MASS, 0, 0, 0, 0], BOX, RND["sgn"], RND["face"], RND["center"], RND["rcase"])
        buf = torch.empty(n_par * 3, dtype=torch.float32, device="cuda")
        S0.synth_positions(buf.data_ptr(), 0, n_par, BOX, seed=SEED, clustered=a.clustered)
        S0.deposit_device(1, buf.data_ptr(), n_par)
        S0.file_end()
        max_abs, max_rel, max_ref = 0.0, 0.0, 0.0
        for p in range(len(lds)):
            got, _, _ = S0.plane_read(p, want_types=False)
            ref = ref_maps[p].reshape(got.shape)
            d = np.abs(got.astype(np.float64) - ref)
            nz = ref > 0
            max_abs = max(max_abs, float(d.max()))
            max_rel = max(max_rel, float((d[nz] / ref[nz]).max()))
            max_ref = max(max_ref, float(ref.max()))
        parity = {"max_abs_dpixel": max_abs, "max_rel_dpixel": max_rel, "max_pixel": max_ref,
                  "sample": f"sub-file 0 of snapshot 0 ({n_par} particles), {len(lds)} planes, {a.npix}^2 "
                            f"{a.mas.upper()}, accum {a.accum}, vs oracle (parity unpinned by a reference build: DESIGN.md S2)",
                  "algo_mask": S0.algo_mask()}
        del buf

    if Lmain is None:  # the timed run proper: `--streams` snapshots in flight
        L.close()
        L = Lmain = Layout(shard, a.reduce_algo)
        dt, tot_dep, tot_in = L.run()
        algo_mask = L.algo_mask
        n_handles = L.n_handles

    # ---- end to end: page-cached format-2 file -> C++ createDensityMaps adapter -> maps in host memory
    e2e = None
    if rank == 0 and world == 1 and (a.e2e == "on" or (a.e2e == "auto" and a.cpu != "off")):
        e2e = end_to_end(a)

    out = None
    reduce_layout = None
    if rank == 0:
        out = {
            "metric": "particles/s deposited (TSC, 4096^2 map)" if (a.mas == "tsc" and a.npix == 4096)
            else f"particles/s deposited ({a.mas.upper()}, {a.npix}^2 map)",
            "value": tot_dep / dt,
            "unit": "particles/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True,
            "scaling": "weak" if shard == "snapshots" else "strong",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f64": "f64", "fixed64": "int64"}[a.accum] if a.mas == "tsc" else "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{a.side}^3-particle GADGET-2 boxes ({files} sub-files{', per-particle masses' if a.hydro else ''}), {a.snapshots} snapshots "
                            f"resident in HBM, {a.npix}^2 {a.mas.upper()}, {len(lds)} lens planes per pass, "
                            f"{'clustered' if a.clustered else 'uniform'}",
                "shard": shard, "algo": a.algo, "accum": a.accum, "algo_mask": algo_mask,
                "streams": max(1, a.streams) if not use_dist else None,
                "reduce_algo": a.reduce_algo if reduce_steps else None,
                "reduce_layout": None,  # (filled in below, after the main number is safe)
                "collective": (("per-plane sum to rank 0 in the accumulator type over RCCL, " if reduce_steps else
                                "finished plane maps sent to rank 0 point to point over RCCL, ")
                               + ("overlapped with the following steps" if overlap else "not overlapped"))
                if (reduce_steps or gathered) else None,
                "maps": ("finalized on the device of the rank that built the step (snapshots and lens planes are "
                         "independent: no data-path collective; each rank would write its own planes)")
                if (gather_steps and not gathered) else None,
                "particles_in_per_step": per_file * (files if shard in ("files", "steps") else len(my_files)),
                "particles_deposited_per_step": tot_dep / max(a.steps, 1) / (1 if shard in ("files", "steps") else world),
            },
            "n_in_per_s": tot_in / dt,
            "hbm_read_roofline_frac_whole_step": 12.0 * tot_in / dt / HBM_PEAK / world,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "alg_bytes_per_launch": alg_bytes, "avg_launch_us": 1e3 * dom_ms / max(dom_n, 1),
                         "whole_step_traffic": traffic_step},
            "kernels": kernels,
            "kernels_measured": "HIP events on one stream, in extra untimed steps right after the single-stream timed run (with "
                                "several snapshots in flight the kernels of different steps share the GPU); they add up to "
                                "single_stream.ms_per_step, or to ms_per_step under --streams 1",
            "single_stream": single_stream,
            "max_rel_dpixel": parity["max_rel_dpixel"] if parity else None,
            "parity": parity,
            "e2e": e2e,
            "cpu_baseline": cpu,
        }

    # ---- the reference's own multi-rank layout in the same run: sub-files of every snapshot split over
    # the ranks (slicer-v2.cpp:162-175) + the per-plane sum to rank 0 over RCCL (slicer-v2.cpp:214-217), timed with the
    # library's rooted reduce and with the direct reduce-scatter + gather-to-root, on the same boxes
    want_reduce_layout = a.reduce_layout == "on" or (a.reduce_layout == "auto" and world > 1)
    if use_dist and want_reduce_layout and shard != "files":
        L.close()
        reduce_layout = {}
        # The secondary layouts are collectives that have never met real multi-GPU hardware before the scaling run: if one
        # of them hangs, the main number -- measured and complete at this point -- must still reach stdout.  Every rank
        # arms the same timer; when it fires rank 0 prints the line with what finished and all ranks leave.
        import threading
        finished = threading.Event()

        def bail():
            if finished.is_set():
                return
            if rank == 0:
                rl = dict(reduce_layout)
                rl["error"] = f"the secondary layouts did not finish within {a.secondary_timeout} s and were cut off"
                out["config"]["reduce_layout"] = rl
                print(json.dumps(out), flush=True)
            os._exit(0)

        watchdog = threading.Timer(a.secondary_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        for ralgo in ("rooted", "p2p"):
            try:
                L2 = Layout("files", ralgo)
                dt2, dep2, in2 = L2.run()
                reduce_layout[ralgo] = {"ms_per_step": 1e3 * dt2 / a.steps, "value": dep2 / dt2, "n_in_per_s": in2 / dt2,
                                        "overlap": L2.overlap}
                L2.close()
            except Exception as e:  # the main number must survive a failing secondary layout
                reduce_layout[ralgo] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if shard == "steps" and not gathered:  # the step layout WITH the maps travelling to rank 0
            try:
                L2 = Layout("steps", a.reduce_algo, gather=True)
                dt2, dep2, in2 = L2.run()
                reduce_layout["steps_gather"] = {"ms_per_step": 1e3 * dt2 / a.steps, "value": dep2 / dt2,
                                                 "n_in_per_s": in2 / dt2, "overlap": L2.overlap}
                L2.close()
            except Exception as e:
                reduce_layout["steps_gather"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        reduce_layout["what"] = ("--shard files: every snapshot's sub-files in contiguous ranges over the ranks "
                                 "(slicer-v2.cpp:162-175) + per-plane sum to rank 0 in the accumulator type over RCCL "
                                 "(slicer-v2.cpp:214-217); rooted = one library reduce per map, p2p = direct "
                                 "reduce-scatter + gather-to-root as grouped sends / receives (slicer_amd/parallel.py); "
                                 "steps_gather = the step layout of `value` with every finished map also sent to rank 0 "
                                 "point to point (parallel.StepGather); same boxes, steps and timing protocol as `value`")
        finished.set()
        watchdog.cancel()


    if rank == 0:
        out["config"]["reduce_layout"] = reduce_layout
        print(json.dumps(out), flush=True)
    L.close()
    if use_dist:
        dist.destroy_process_group()


def end_to_end(a):
    """One sub-file (2^24 particles) written as a format-2 file, page cached; tests/cpp/adapter_driver calls the C++
    createDensityMaps adapter on it five times (file read -> pinned staging -> H2D -> kernels -> D2H of the maps)."""
    import numpy as np  # noqa: F401

    from slicer_amd import gadget, synth
    drv = os.path.join(ROOT, "tests", "cpp", "adapter_driver")
    if not os.path.exists(drv):
        return {"error": "tests/cpp/adapter_driver not built"}
    n = 1 << 24
    d = tempfile.mkdtemp(prefix="slicer_e2e_", dir="/tmp")
    base = os.path.join(d, "snap_000")
    try:
        gadget.write_snapshot(base + ".0", synth.positions(0, n, BOX, seed=SEED), [0, n, 0, 0, 0, 0],
                              [0, MASS, 0, 0, 0, 0], BOX)
        env = dict(os.environ, ADAPTER_REPEAT="5", SLICER_AMD_READ_THREADS="8")
        r = subprocess.run([drv, base, "0", "1", str(a.npix), str(FOV), "3.0", "3.25", "3.0", "0", "0",
                            os.path.join(d, "m.bin")], capture_output=True, env=env, text=True, timeout=300)
        ms = [float(ln.split(": ")[1].split()[0]) for ln in r.stderr.splitlines() if ln.startswith("createDensityMaps call")]
        if r.returncode or len(ms) < 2:
            return {"error": f"adapter_driver rc={r.returncode}"}
        best = min(ms[1:])
        # the same call when the caller does not keep per-type planes (partinplanes = 0) and opts out of the six
        # per-type maps it would discard (slicer_amd_adapter_skip_type_maps): half the device-to-host traffic
        skip = None
        envs = dict(env, SLICER_AMD_SKIP_TYPE_MAPS="1", ADAPTER_PARTINPLANES="0")
        rs = subprocess.run([drv, base, "0", "1", str(a.npix), str(FOV), "3.0", "3.25", "3.0", "0", "0",
                             os.path.join(d, "ms.bin")], capture_output=True, env=envs, text=True, timeout=300)
        mss = [float(ln.split(": ")[1].split()[0]) for ln in rs.stderr.splitlines() if ln.startswith("createDensityMaps call")]
        if rs.returncode == 0 and len(mss) >= 2:
            bs = min(mss[1:])
            skip = {"ms_per_call": bs, "value": n / (bs * 1e-3), "pcie_frac": 12.0 * n / (bs * 1e-3) / PCIE_PEAK,
                    "what": "partinplanes = 0 with slicer_amd_adapter_skip_type_maps(1): all-types map only"}
        # the same sub-file for the four planes of one box replication, called plane by plane as slicer-v2.cpp does: the
        # adapter deposits all four during the first call and serves the other three from the device
        four = None
        env4 = dict(os.environ, ADAPTER_REPEAT="3", SLICER_AMD_READ_THREADS="8")
        r4 = subprocess.run([drv, base, "0", "1", str(a.npix), str(FOV), "3.0,3.25,3.5,3.75", "3.25,3.5,3.75,4.0", "3.0", "0",
                             "0", os.path.join(d, "m4.bin")], capture_output=True, env=env4, text=True, timeout=300)
        ms4 = [float(ln.split(": ")[1].split()[0]) for ln in r4.stderr.splitlines() if ln.startswith("createDensityMaps call")]
        if r4.returncode == 0 and len(ms4) == 12:
            loops = [ms4[4 * k:4 * k + 4] for k in range(1, 3)]
            bl = min(loops, key=sum)
            four = {"calls_ms": bl, "ms_per_plane": sum(bl) / 4.0, "input_particles_per_s_per_plane": n / (sum(bl) / 4.0 * 1e-3),
                    "what": "four createDensityMaps calls (planes of one replication) on the same sub-file; best of 2 warm loops"}
        return {"value": n / (best * 1e-3), "unit": "input particles/s", "ms_per_call": best, "calls_ms": ms,
                "four_planes": four, "without_type_maps": skip,
                "pcie_frac": 12.0 * n / (best * 1e-3) / PCIE_PEAK,
                "what": f"page-cached format-2 file ({n} particles) -> C++ createDensityMaps (one plane, {a.npix}^2 TSC) -> "
                        "all-types map + populated per-type map in host memory; 8 read threads; best of 4 warm calls"}
    finally:
        subprocess.call(["rm", "-rf", d])


def _counts(S, plane):
    import numpy as np

    from slicer_amd import api
    nsel = np.zeros(6, np.int64)
    rc = api._L.slicer_plane_read(S._h, plane, None, None, nsel.ctypes.data)
    if rc:
        raise RuntimeError(f"slicer_plane_read failed: {rc}")
    return nsel


if __name__ == "__main__":
    main()
