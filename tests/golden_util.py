"""Loader of tests/golden/*.npz (oracle-generated vectors; provenance in tests/golden/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    files = []
    for i in range(int(z["nfiles"])):
        mass = {}
        for t in range(6):
            key = f"mass{i}_{t}"
            if key in z.files:
                mass[t] = z[key]
        files.append(dict(npart=z[f"npart{i}"].tolist(), massarr=z[f"massarr{i}"].tolist(), boxsize=float(z["boxsize"]),
                          pos=z[f"pos{i}"], mass=mass))
    cfg = dict(npix=int(z["npix"]), fov=float(z["fov"]), ld=float(z["ld"]), ld2=float(z["ld2"]), ngp=bool(z["ngp"]),
               hydro=bool(z["hydro"]), nrep=int(z["nrep"]),
               rnd=dict(sgn=tuple(int(v) for v in z["sgn"]), face=int(z["face"]),
                        center=tuple(float(v) for v in z["center"]), rcase=float(z["rcase"])))
    return files, cfg, z["tot"], z["toti"], z["nsel"]
