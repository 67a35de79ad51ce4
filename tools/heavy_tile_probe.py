"""How does the tile kernel behave when a few pixels receive a large share of the particles (halo cores)?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import slicer_amd  # noqa: E402
from slicer_amd import synth  # noqa: E402

n = 1 << 24
BOX = 1000.0
rng = np.random.default_rng(0)
S = slicer_amd.Slicer(0, max_chunk=n)
for frac, nblob, sig in ((0.0, 1, 1.0), (0.5, 4096, 4.0), (0.5, 64, 0.2), (0.5, 4, 0.05), (0.9, 1, 0.02), (0.9, 1, 0.002)):
    pos = synth.positions(0, n, BOX)
    k = int(frac * n)
    if k:
        c = rng.uniform(0.3 * BOX, 0.7 * BOX, (nblob, 3)).astype(np.float32)
        idx = rng.integers(0, nblob, k)
        pos[:k] = c[idx] + rng.normal(0, sig, (k, 3)).astype(np.float32)
        pos = np.mod(pos, BOX).astype(np.float32)
    d = S.to_device(pos)
    S.profile_reset()
    S.profile_enable(True)
    for rep in range(2):
        S.plane_begin(4096, 0.25, [3.0, 3.25, 3.5, 3.75], [3.25, 3.5, 3.75, 4.0], want_type_maps=False)
        S.file_begin([0, n, 0, 0, 0, 0], [0, 0.0123, 0, 0, 0, 0], BOX, (1, 1, 1), 1, (0.0, 0.0, 0.0), 3.0)
        S.deposit_device(1, d, n)
        S.file_end()
        S.plane_finalize()
    S.synchronize()
    S.profile_enable(False)
    prof = S.profile_get()
    print(f"frac={frac} blobs={nblob} sigma={sig} kpc/h:", {k_: round(1e3 * v[1] / v[0]) for k_, v in prof.items()})
    S.free(d)
