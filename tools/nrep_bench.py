"""Lateral replication beyond three per side: the binned kernels in replica windows against the fused global-atomic
kernel (what such passes ran on before round 3).  One sub-file of 2^22 particles, 4096^2 TSC, four planes.
usage (GPU box): python tools/nrep_bench.py [nrep ...]"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import slicer_amd  # noqa: E402

N, NPIX, BOX = 1 << 22, 4096, 1000.0
LDS, LD2S = [3.0, 3.25, 3.5, 3.75], [3.25, 3.5, 3.75, 4.0]


def run(S, buf, nrep, algo, reps=3):
    fov = 1.96 * math.atan((nrep + 0.5) / 4.0)
    best, cnt = 1e9, 0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S.plane_begin(NPIX, fov, LDS, LD2S, [nrep] * 4, algo=algo, want_type_maps=False)
        S.file_begin([0, N, 0, 0, 0, 0], [0, 0.0123, 0, 0, 0, 0], BOX, (-1, 1, -1), 3, (0.3, 0.6, 0.1), 3.0)
        S.deposit_device(1, buf.data_ptr(), N)
        S.file_end()
        S.plane_finalize()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    cnt = sum(int(S.plane_read(p, want_types=False)[2].sum()) for p in range(4))
    return best, cnt


def main():
    S = slicer_amd.Slicer(0, max_chunk=N)
    buf = torch.empty(N * 3, dtype=torch.float32, device="cuda")
    S.synth_positions(buf.data_ptr(), 0, N, BOX, seed=0x51CE2, clustered=False)
    for nrep in [int(x) for x in sys.argv[1:]] or [3, 4, 6, 8]:
        tb, cb = run(S, buf, nrep, slicer_amd.ALGO_BINNED)
        td, cd = run(S, buf, nrep, slicer_amd.ALGO_DIRECT, reps=1)
        assert cb == cd
        print(f"nrepperp {nrep}: {cb} entries; binned {1e3 * tb:.2f} ms = {cb / tb:.3e} /s; "
              f"fused {1e3 * td:.2f} ms = {cd / td:.3e} /s; x{td / tb:.1f}", flush=True)
    S.close()


if __name__ == "__main__":
    main()
