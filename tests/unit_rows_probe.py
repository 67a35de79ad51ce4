"""Run in a subprocess with SLICER_UNIT_ROWS set: forces the large-map layout (units = bands of tile rows) on
small maps, so that it can be compared with the oracle and with the fused kernel exactly."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
import slicer_amd  # noqa: E402
from slicer_amd import synth  # noqa: E402

BOX = 1000.0
RND = dict(sgn=(-1, 1, -1), face=3, center=(0.3, 0.6, 0.1), rcase=3.0)
n = 300000
pos = synth.positions(0, n, BOX, clustered=True)
f = dict(npart=[0, n, 0, 0, 0, 0], massarr=[0, 0.0123, 0, 0, 0, 0], boxsize=BOX, pos=pos)
S = slicer_amd.Slicer(0, max_chunk=1 << 20)


def gpu(npix, lds, ld2s, **kw):
    S.plane_begin(npix, 0.25, lds, ld2s, **kw)
    S.file_begin(f["npart"], f["massarr"], BOX, RND["sgn"], RND["face"], RND["center"], RND["rcase"])
    S.deposit_host(1, pos)
    S.file_end()
    return [S.plane_read(p, want_types=False) for p in range(len(lds))]


for npix in (512, 1000):
    lds, ld2s = [3.0, 3.5], [3.5, 4.0]
    out = gpu(npix, lds, ld2s, mas=slicer_amd.MAS_NGP, algo=slicer_amd.ALGO_BINNED)
    for p in range(2):
        rc, tot, toti, nsel = oracle.create_density_maps([f], 0, 1, npix, False, True, lds[p], ld2s[p], 0, 0.25,
                                                         RND["sgn"], RND["face"], RND["center"], RND["rcase"])
        assert np.array_equal(out[p][2], nsel)
        assert np.array_equal(out[p][0].view(np.uint32), tot.view(np.uint32)), (npix, p)
    a = gpu(npix, lds, ld2s, accum=slicer_amd.ACC_FIXED64, algo=slicer_amd.ALGO_DIRECT)
    b = gpu(npix, lds, ld2s, accum=slicer_amd.ACC_FIXED64, algo=slicer_amd.ALGO_BINNED)
    for p in range(2):
        assert np.array_equal(a[p][0].view(np.uint32), b[p][0].view(np.uint32)), (npix, p)
print("unit-rows probe ok")
