#!/usr/bin/env python3
"""bench.py -- headline benchmark of the mass-assignment hot path on MI355X.

Metric (BASELINE.json): particles/s deposited (TSC, 4096^2 map).
Workload at N=1 (BASELINE config[2], the configuration the metric is quoted on; it fits one GPU):
  4 synthetic GADGET-2 snapshots of 512^3 particles (8 sub-files of 2^24 each, raw POS blocks resident
  in HBM before the timed region), 4096^2 TSC maps, the 4 lens planes of one box replication built in
  one pass (S8d geometry: rcase=3, ld=3+{0,.25,.5,.75}, fov=0.25 rad, face 3, signs (-,+,-),
  centre (.3,.6,.1)).  One step = one snapshot -> its 4 plane maps: plane_begin (zero) + 8 sub-file
  deposits + finalize.  Steps cycle through the resident snapshots.
N>1: one process per GPU (torchrun), each rank owns its own snapshots (planes / snapshots shard with no
  data-path exchange: SURVEY S8e level 2) => "scaling": "weak"; value = all ranks' deposits / max time.
  --shard files switches to the reference's own partition (slicer-v2.cpp:162-175: sub-files of every
  snapshot split over ranks) followed by the per-plane sum to rank 0 (slicer-v2.cpp:214-217) over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BOX = 1000.0
RND = dict(sgn=(-1, 1, -1), face=3, center=(0.3, 0.6, 0.1), rcase=3.0)
LDS = [3.0, 3.25, 3.5, 3.75]
LD2S = [3.25, 3.5, 3.75, 4.0]
FOV = 0.25
MASS = 0.0123
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec); 6.29e12 measured-achievable


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
    ap.add_argument("--shard", default="snapshots", choices=["snapshots", "files"])
    ap.add_argument("--cpu-baseline", dest="cpu", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--cpu-particles", type=int, default=1 << 24, help="particles per CPU-baseline worker file")
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--profile-steps", type=int, default=2)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a scalar port of the reference path) on the host cores, parallelised the
# way the reference is (one process per contiguous sub-file range, slicer-v2.cpp:162-175).  Each
# worker builds the 4 planes of its own sub-file with 4 createDensityMaps-equivalent calls, exactly
# what slicer-v2.cpp's plane loop does (the reference re-reads the snapshot for every plane).
# ------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    first, n, npix, ngp, planes, clustered = args
    import numpy as np  # noqa: F401

    import oracle
    from slicer_amd import synth
    pos = synth.positions(first, n, BOX, clustered=clustered)
    f = dict(npart=[0, n, 0, 0, 0, 0], massarr=[0, MASS, 0, 0, 0, 0], boxsize=BOX, pos=pos)
    oracle.lib()
    t0 = time.perf_counter()
    dep = 0
    for p in range(planes):
        rc, tot, toti, nsel = oracle.create_density_maps([f], 0, 1, npix, False, ngp, LDS[p], LD2S[p], 0, FOV,
                                                         RND["sgn"], RND["face"], RND["center"], RND["rcase"])
        dep += int(nsel[1])
    return time.perf_counter() - t0, dep


def cpu_baseline(a):
    import multiprocessing as mp
    cores = a.cpu_cores or max(1, min(16, len(os.sched_getaffinity(0))))
    n = a.cpu_particles
    ctx = mp.get_context("spawn")
    jobs = [(i * n, n, a.npix, a.mas == "ngp", a.planes, a.clustered) for i in range(cores)]
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = max(r[0] for r in res)  # compute-only: workers generate their inputs before the clock starts
    dep = sum(r[1] for r in res)
    return {"value": dep / wall, "unit": "particles/s", "cores": cores, "kind": "port",
            "sample": f"{cores} sub-files x {n} particles, {a.planes} planes each ({a.planes} createDensityMaps-"
                      f"equivalent oracle calls per sub-file, one process per sub-file), {a.npix}^2 "
                      f"{a.mas.upper()}; {wall:.1f} s wall, inputs in RAM",
            "n_in_per_s": cores * n * a.planes / wall, "total_s": time.perf_counter() - t0}


# ------------------------------------------------------------------------------------------------
def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world

    cpu = None
    if rank == 0 and world == 1 and a.cpu != "off":
        cpu = cpu_baseline(a)  # before anything touches the GPU in this process

    import torch  # first, so that libslicer_amd.so binds to the HIP runtime torch already loaded
    import torch.distributed as dist

    import slicer_amd
    from slicer_amd import parallel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
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

    S = slicer_amd.Slicer(local_rank, max_chunk=per_file)
    stream = torch.cuda.current_stream()
    S.set_stream(stream.cuda_stream)

    # which sub-files of which snapshots this rank deposits
    if a.shard == "snapshots":
        my_snaps = list(range(a.snapshots))
        seed0 = 0x51CE2 + 1000 * rank          # every rank owns different boxes
        my_files = list(range(files))
    else:
        my_snaps = list(range(a.snapshots))
        seed0 = 0x51CE2                         # same boxes everywhere, sub-files split as slicer-v2.cpp:162-175
        lo, hi = parallel.file_range(files, world, rank)
        my_files = list(range(lo, hi))

    # resident raw POS blocks: [snapshot][file] -> torch buffer (HBM)
    pos = []
    for s in my_snaps:
        row = []
        for ff in my_files:
            buf = torch.empty(per_file * 3, dtype=torch.float32, device="cuda")
            S.synth_positions(buf.data_ptr(), ff * per_file, per_file, BOX, seed=seed0 + s, clustered=a.clustered)
            row.append(buf)
        pos.append(row)
    masses = None
    if a.hydro:  # one block of per-particle masses, shared by every sub-file (values in (0.5, 1.5) * MASS)
        masses = (torch.rand(per_file, dtype=torch.float32, device="cuda") + 0.5) * MASS
    torch.cuda.synchronize()

    def step(i):
        s = i % len(my_snaps)
        S.plane_begin(a.npix, FOV, lds, ld2s, mas=mas, accum=accum, algo=algo, want_type_maps=False, hydro=a.hydro)
        for j, ff in enumerate(my_files):
            if a.hydro:
                S.file_begin([per_file, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], BOX, RND["sgn"], RND["face"],
                             RND["center"], RND["rcase"])
                S.deposit_device(0, pos[s][j].data_ptr(), per_file, masses.data_ptr())
            else:
                S.file_begin([0, per_file, 0, 0, 0, 0], [0, MASS, 0, 0, 0, 0], BOX, RND["sgn"], RND["face"],
                             RND["center"], RND["rcase"])
                S.deposit_device(1, pos[s][j].data_ptr(), per_file)
            S.file_end()
        S.plane_finalize()
        if a.shard == "files" and use_dist:
            # slicer-v2.cpp:214: MPI_Reduce(mapxytot, SUM, root 0) per plane -> RCCL reduce over xGMI
            parallel.reduce_planes(S, dist, torch, root=0, per_type=False)

    # deposits per step (identical for a given snapshot every time it is processed)
    dep_per_snap = []
    for s in range(len(my_snaps)):
        step(s)
        d = 0
        for p in range(len(lds)):
            d += int(_counts(S, p)[0 if a.hydro else 1])
        dep_per_snap.append(d)

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    my_dep = sum(dep_per_snap[i % len(my_snaps)] for i in range(a.steps))
    my_in = a.steps * per_file * len(my_files)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([my_dep, my_in], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot_dep, tot_in = float(c[0].item()), float(c[1].item())
        if a.shard == "files":
            pass
    else:
        tot_dep, tot_in = float(my_dep), float(my_in)

    # per-kernel durations, live, with HIP events on the launch stream (a few extra untimed steps)
    S.profile_reset()
    S.profile_enable(True)
    for i in range(max(1, a.profile_steps)):
        step(i)
    torch.cuda.synchronize()
    S.profile_enable(False)
    prof = S.profile_get()
    dom_name, (dom_n, dom_ms) = max(prof.items(), key=lambda kv: kv[1][1])
    kernels = {k: {"launches": v[0], "avg_us": 1e3 * v[1] / max(v[0], 1)} for k, v in prof.items()}
    # algorithmic bytes per launch of the dominant kernel (DESIGN.md "Algorithmic bytes")
    per_launch_particles = per_file
    if dom_name in ("direct_deposit", "project_bin"):
        alg_bytes = 12.0 * per_launch_particles
    elif dom_name in ("tile_deposit", "bin_scatter"):
        alg_bytes = 8.0 * (my_dep / max(a.steps, 1)) / max(len(my_files), 1)
    else:
        alg_bytes = 4.0 * a.npix * a.npix
    achieved = alg_bytes / (1e-3 * dom_ms / max(dom_n, 1))
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(dom_name)
            if ent and ent.get("workload") == f"{a.side}^3/{a.npix}/{a.mas}/{a.algo}/{a.accum}":
                traffic = ent.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

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
            "scaling": "weak" if a.shard == "snapshots" else "strong",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f64": "f64", "fixed64": "int64"}[a.accum] if a.mas == "tsc" else "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{a.side}^3-particle GADGET-2 boxes ({files} sub-files{', per-particle masses' if a.hydro else ''}), {a.snapshots} snapshots/rank "
                            f"resident in HBM, {a.npix}^2 {a.mas.upper()}, {len(lds)} lens planes per pass, "
                            f"{'clustered' if a.clustered else 'uniform'}",
                "shard": a.shard, "algo": a.algo, "accum": a.accum,
                "particles_in_per_step": per_file * len(my_files),
                "particles_deposited_per_step": my_dep / max(a.steps, 1),
            },
            "n_in_per_s": tot_in / dt,
            "hbm_read_roofline_frac_whole_step": 12.0 * tot_in / dt / HBM_PEAK / world,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "alg_bytes_per_launch": alg_bytes, "avg_launch_us": 1e3 * dom_ms / max(dom_n, 1)},
            "kernels": kernels,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    S.close()
    if use_dist:
        dist.destroy_process_group()


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
