"""Shot-noise thinning (snopt > 0): the libc rand() stream continued on the device against rand() calls on the host.
One sub-file of 2^24 particles resident in HBM, 4096^2 TSC, snopt = 2; one plane, then four planes in one pass (the
chunks are replayed plane by plane).  usage (GPU box): python tools/thin_bench.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import slicer_amd  # noqa: E402

N, NPIX, BOX = 1 << 24, 4096, 1000.0
libc = C.CDLL("libc.so.6")


def run(S, buf, lds, ld2s, host, reps=3):
    S.set_option("thin_host", host)
    best, after = 1e9, None
    for _ in range(reps):
        libc.srand(4242)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S.plane_begin(NPIX, 0.25, lds, ld2s, snopt=2, want_type_maps=False)
        S.file_begin([0, N, 0, 0, 0, 0], [0, 0.0123, 0, 0, 0, 0], BOX, (-1, 1, -1), 3, (0.3, 0.6, 0.1), 3.0)
        S.deposit_device(1, buf.data_ptr(), N)
        S.file_end()
        S.plane_finalize()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        after = libc.rand()
    cnt = sum(int(S.plane_read(p, want_types=False)[2].sum()) for p in range(len(lds)))
    return best, cnt, after


def main():
    S = slicer_amd.Slicer(0, max_chunk=N)
    buf = torch.empty(N * 3, dtype=torch.float32, device="cuda")
    S.synth_positions(buf.data_ptr(), 0, N, BOX, seed=0x51CE2, clustered=False)
    for lds, ld2s in (([3.0], [3.25]), ([3.0, 3.25, 3.5, 3.75], [3.25, 3.5, 3.75, 4.0])):
        td, cd, ad = run(S, buf, lds, ld2s, 0)
        th, ch, ah = run(S, buf, lds, ld2s, 1, reps=1)
        assert cd == ch and ad == ah, "the two paths must consume the same deviates"
        print(f"{len(lds)} plane(s): {cd} draws; device stream {1e3 * td:.2f} ms; host rand() {1e3 * th:.1f} ms; "
              f"x{th / td:.0f}", flush=True)
    S.close()


if __name__ == "__main__":
    main()
