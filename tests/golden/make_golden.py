"""Generates tests/golden/*.npz -- small input/output vectors for the mass-assignment path.

PROVENANCE: produced by THIS repository's oracle (oracle/slicer_oracle.c), not by a build of the reference: the
reference path cannot be compiled in this image (GSL / CCfits headers are absent), and it ships no
fixtures of its own.  The vectors pin the oracle against silent drift (tests/test_golden.py, CPU) and give the GPU
path a fixed target that does not depend on the oracle being rebuilt (tests/test_gpu_parity.py::test_golden_*).
Inputs are stored too, so a future run of the real reference on them can be compared directly.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from slicer_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
BOX = 1000.0


def case(name, files, npix, fov, ld, ld2, ngp, rnd, hydro=False, nrep=0):
    rc, tot, toti, nsel = oracle.create_density_maps(files, 0, len(files), npix, hydro, ngp, ld, ld2, nrep, fov,
                                                     rnd["sgn"], rnd["face"], rnd["center"], rnd["rcase"])
    assert rc == 0
    arrays = dict(npix=npix, fov=fov, ld=ld, ld2=ld2, ngp=int(ngp), hydro=int(hydro), nrep=nrep,
                  sgn=np.array(rnd["sgn"]), face=rnd["face"], center=np.array(rnd["center"]), rcase=rnd["rcase"],
                  boxsize=BOX, nfiles=len(files), tot=tot, toti=toti, nsel=nsel)
    for i, f in enumerate(files):
        arrays[f"pos{i}"] = np.asarray(f["pos"], np.float32)
        arrays[f"npart{i}"] = np.array(f["npart"], np.int32)
        arrays[f"massarr{i}"] = np.array(f["massarr"], np.float64)
        for t, m in f.get("mass", {}).items():
            arrays[f"mass{i}_{t}"] = np.asarray(m, np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print(name, "selected", nsel.tolist(), "sum", float(tot.sum()))


def main():
    rnd_a = dict(sgn=(-1, 1, -1), face=3, center=(0.3, 0.6, 0.1), rcase=3.0)
    rnd_b = dict(sgn=(1, -1, 1), face=5, center=(0.9, 0.05, 0.5), rcase=0.0)
    f1 = dict(npart=[0, 4096, 0, 0, 0, 0], massarr=[0, 0.0123, 0, 0, 0, 0], boxsize=BOX, pos=synth.positions(0, 4096, BOX))
    case("tsc_64_single", [f1], 64, 0.25, 3.0, 4.0, False, rnd_a)
    case("ngp_64_single", [f1], 64, 0.25, 3.0, 4.0, True, rnd_a)
    case("tsc_16_wide_fov", [f1], 16, 0.9, 0.0, 1.0, False, rnd_b)
    files, first = [], 0
    rng = np.random.default_rng(42)
    for ff in range(3):
        npart = [301, 2003, 0, 157, 0, 29] if ff != 1 else [0, 1501, 203, 0, 0, 0]
        n = sum(npart)
        m0 = rng.uniform(0.001, 0.05, npart[0]).astype(np.float32)
        if len(m0):
            m0[::41] = 5000.0  # above MAX_M
        files.append(dict(npart=npart, massarr=[0.0, 0.0123, 0.3, 0.07, 0, 1.5], boxsize=BOX,
                          pos=synth.positions(first, n, BOX), mass={0: m0} if len(m0) else {}))
        first += n
    case("tsc_32_hydro_3files_5types", files, 32, 0.25, 3.0, 4.0, False, rnd_a, hydro=True)
    case("ngp_24_hydro_3files_5types", files, 24, 0.25, 3.0, 4.0, True, rnd_a, hydro=True)
    case("ngp_32_nrep1", [f1], 32, 0.6, 3.0, 4.0, True, rnd_a, nrep=1)


if __name__ == "__main__":
    main()
