"""Longer run of tests/test_gpu_parity.py::test_random_configurations_binned_path (needs a GPU).
usage: python tools/fuzz_binned.py [first_seed last_seed]   (default 12 260)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import test_gpu_parity as T
import slicer_amd
S = slicer_amd.Slicer(0, max_chunk=1 << 20)
bad = 0
lo_seed, hi_seed = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 260)
for seed in range(lo_seed, hi_seed):
    for levels in (0, 1):  # one-level sort, and the two-level sort where a pass qualifies (option sort2)
        S.set_option("sort2", levels)
        try:
            T.test_random_configurations_binned_path(S, levels, seed)
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, "sort levels", levels + 1, str(e)[:200], flush=True)
        S.plane_begin(16, 0.25, [3.0], [4.0])
        S.set_option("sort2", 0)
    if seed % 200 == 0:
        print("seed", seed, "ok so far, failures:", bad, flush=True)
print("done, failures:", bad)
S.close()

# second flavour: several particle types per file, per-particle masses for the types with massarr == 0 (hydro), TSC:
# binned and direct kernels must give the same FIXED64 maps (total and per type) bit for bit
S = slicer_amd.Slicer(0, max_chunk=1 << 20)
bad = 0
for seed in range(300, 380):
    rng = np.random.default_rng(seed)
    npix = int(rng.choice([64, 100, 256, 300, 1024]))
    n_planes = int(rng.integers(1, 5))
    edges = np.repeat(np.linspace(0.0, 1.0, n_planes + 1), 2)[1:-1]
    lds = [3.0 + float(e) for e in edges[0::2]]
    ld2s = [3.0 + float(e) for e in edges[1::2]]
    fov = float(rng.uniform(0.05, 0.25))
    rnd = dict(sgn=tuple(int(v) for v in rng.choice([-1, 1], 3)), face=int(rng.integers(1, 7)),
               center=tuple(float(v) for v in rng.random(3)), rcase=3.0)
    files, first = [], 0
    for _ in range(int(rng.integers(1, 4))):
        npart = [int(rng.integers(0, 90000)) if rng.random() < 0.6 else 0 for _ in range(6)]
        if sum(npart) == 0:
            npart[1] = 50000
        massarr = [0.0 if rng.random() < 0.4 else float(rng.uniform(0.01, 3.0)) for _ in range(6)]
        n = sum(npart)
        mass = {t: (rng.random(npart[t]).astype(np.float32) * 2.0 + 0.01) for t in range(6) if npart[t] and massarr[t] == 0}
        for t in mass:  # a few above MAX_M, which the reference zeroes
            mass[t][: max(1, npart[t] // 50)] = 2000.0
        files.append(dict(npart=npart, massarr=massarr, boxsize=T.BOX, pos=T.synth.positions(first, n, T.BOX), mass=mass))
        first += n
    a = T.run_gpu(S, files, npix, fov, lds, ld2s, accum=slicer_amd.ACC_FIXED64, algo=slicer_amd.ALGO_BINNED, rnd=rnd, hydro=True)
    b = T.run_gpu(S, files, npix, fov, lds, ld2s, accum=slicer_amd.ACC_FIXED64, algo=slicer_amd.ALGO_DIRECT, rnd=rnd, hydro=True)
    for p in range(n_planes):
        ok = (np.array_equal(a[p][2], b[p][2]) and np.array_equal(a[p][0].view(np.uint32), b[p][0].view(np.uint32))
              and np.array_equal(a[p][1].view(np.uint32), b[p][1].view(np.uint32)))
        if not ok:
            bad += 1
            print("FAIL hydro seed", seed, "plane", p, flush=True)
print("hydro flavour done, failures:", bad)
S.close()

# third flavour: NGP under SLICER_ALGO_AUTO with sub-files of one or several species, small kernel-pass chunks (files in
# several chunks, tiny remainders through the fused kernel), with and without per-type maps, sometimes overlapping
# slabs: every map bit for bit against the oracle (the per-file folds: inside the tile kernel, through the count map,
# and mixtures of both in one pass)
bad = 0
for seed in range(400, 400 + (int(sys.argv[3]) if len(sys.argv) > 3 else 120)):
    rng = np.random.default_rng(seed)
    S = slicer_amd.Slicer(0, max_chunk=int(rng.choice([70000, 100000, 1 << 20])))
    npix = int(rng.choice([64, 100, 128, 256, 317, 1000, 1024]))
    n_planes = int(rng.integers(1, 5))
    edges = np.repeat(np.linspace(0.0, 1.0, n_planes + 1), 2)[1:-1]
    lds = [3.0 + float(e) for e in edges[0::2]]
    ld2s = [3.0 + float(e) for e in edges[1::2]]
    if n_planes > 1 and rng.random() < 0.25:
        ld2s[0] = lds[1] + 0.1      # overlapping slabs: plane groups
    fov = float(rng.uniform(0.1, 0.25))
    rnd = dict(sgn=tuple(int(v) for v in rng.choice([-1, 1], 3)), face=int(rng.integers(1, 7)),
               center=tuple(float(np.float32(v)) if rng.random() < 0.7 else float(v) for v in rng.random(3)), rcase=3.0)
    files, first = [], 0
    for _ in range(int(rng.integers(1, 12))):
        npart = [0] * 6
        if rng.random() < 0.7:      # one species
            npart[int(rng.integers(0, 6))] = int(rng.choice([30000, 69999, 70000, 100001, 150000, 210000]))
        else:
            for t in rng.choice(6, int(rng.integers(2, 4)), replace=False):
                npart[int(t)] = int(rng.choice([20000, 70001, 130000]))
        massarr = [float(rng.uniform(0.01, 3.0)) for _ in range(6)]
        n = sum(npart)
        files.append(dict(npart=npart, massarr=massarr, boxsize=T.BOX, pos=T.synth.positions(first, n, T.BOX,
                                                                                             clustered=rng.random() < 0.5)))
        first += n
    types = bool(rng.random() < 0.5)
    out = T.run_gpu(S, files, npix, fov, lds, ld2s, ngp=True, rnd=rnd, want_type_maps=types)
    for p in range(n_planes):
        ref_tot, ref_toti, nsel = T.run_oracle(files, npix, fov, lds[p], ld2s[p], ngp=True, rnd=rnd)
        ok = np.array_equal(out[p][2], nsel) and np.array_equal(out[p][0].view(np.uint32), ref_tot.view(np.uint32))
        if types:
            ok = ok and np.array_equal(out[p][1].view(np.uint32), ref_toti.view(np.uint32))
        if not ok:
            bad += 1
            print("FAIL ngp seed", seed, "plane", p, flush=True)
    S.close()
print("ngp flavour done, failures:", bad)
