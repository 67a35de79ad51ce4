"""Run by tests/test_gpu_parity.py in a FRESH process: shot-noise thinning in process-global mode, with everything the
first pass of a process does inside it (workspace allocations, code-object loading at first launches, ...)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
import slicer_amd  # noqa: E402
from slicer_amd import synth  # noqa: E402

BOX = 1000.0
libc = C.CDLL("libc.so.6")
n = 120000
f = dict(npart=[0, n, 0, 0, 0, 0], massarr=[0, 0.0123, 0, 0, 0, 0], boxsize=BOX, pos=synth.positions(0, n, BOX))
sgn, face, center, rcase = (-1, 1, -1), 3, (0.3, 0.6, 0.1), 3.0
libc.srand(2026)
rc, ref_tot, _, nsel = oracle.create_density_maps([f], 0, 1, 256, False, True, 3.0, 4.0, 0, 0.25, sgn, face, center, rcase,
                                                  snopt=2)
after_ref = libc.rand()
S = slicer_amd.Slicer(0)          # the HIP runtime starts here
libc.srand(2026)
S.plane_begin(256, 0.25, [3.0], [4.0], mas=slicer_amd.MAS_NGP, snopt=2)
S.file_begin(f["npart"], f["massarr"], BOX, sgn, face, center, rcase)
S.deposit_host(1, f["pos"])
S.file_end()
tot, _, cnt = S.plane_read(0)
ok = rc == 0 and np.array_equal(cnt, nsel) and np.array_equal(tot.view(np.uint32), ref_tot.view(np.uint32)) \
    and libc.rand() == after_ref
print("THINNING_OK" if ok else "THINNING_MISMATCH", int(nsel[1]))
S.close()
