"""End-to-end (file -> maps on the host) timing of one createDensityMaps call, for DESIGN.md S7: the rate that
includes the file read, the pinned staging, PCIe H2D and the D2H of the maps.  Not the bench's `value`."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from slicer_amd import gadget, synth  # noqa: E402

npix = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 1 << 24
d = tempfile.mkdtemp(dir="/tmp")
base = os.path.join(d, "snap_000")
pos = synth.positions(0, n, 1000.0)
gadget.write_snapshot(base + ".0", pos, [0, n, 0, 0, 0, 0], [0, 0.0123, 0, 0, 0, 0], 1000.0)
drv = os.path.join(ROOT, "tests", "cpp", "adapter_driver")
for threads in ("1", "8"):
    out = os.path.join(d, "m.bin")
    env = dict(os.environ, ADAPTER_REPEAT="5", SLICER_AMD_READ_THREADS=threads)
    r = subprocess.run([drv, base, "0", "1", str(npix), "0.25", "3.0", "3.25", "3.0", "0", "0", out], capture_output=True,
                       env=env, text=True)
    print(f"C++ createDensityMaps (file read -> pinned staging -> H2D -> kernels -> D2H of 7 maps), "
          f"{threads} read thread(s), rc={r.returncode}:")
    print("   " + " | ".join(l.split(": ")[1] for l in r.stderr.splitlines() if l.startswith("createDensityMaps call")))
import slicer_amd  # noqa: E402
S = slicer_amd.Slicer(0, max_chunk=1 << 22)
p = slicer_amd.InputParams(npix=npix)
lens = slicer_amd.Lens(nplanes=1, ld=[3.0], ld2=[3.25], nrepperp=[0])
rnd = slicer_amd.Random(x0=[0.3], y0=[0.6], z0=[0.1], face=[3], sgnX=[-1], sgnY=[1], sgnZ=[-1])
for rep in range(3):
    t0 = time.perf_counter()
    rc, tot, toti, nt = slicer_amd.createDensityMaps(p, lens, rnd, 0, 0, 1, base, 0.25, 3.0, slicer=S, true_counts=True)
    dt = time.perf_counter() - t0
    print(f"python createDensityMaps, warm handle: {dt*1e3:.1f} ms = {n/dt:.3e} input particles/s, {int(nt[1])/dt:.3e} deposited/s")
