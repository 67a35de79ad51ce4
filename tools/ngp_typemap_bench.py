"""NGP with per-type maps kept (want_type_maps = 1, what the createDensityMaps adapter asks for): one 512^3 snapshot in 8
resident sub-files, 4096^2, four planes per pass; ms per snapshot and the tile kernel's share.
usage (GPU box): python tools/ngp_typemap_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import slicer_amd  # noqa: E402

N, FILES, NPIX, BOX = 1 << 24, 8, 4096, 1000.0
LDS, LD2S = [3.0, 3.25, 3.5, 3.75], [3.25, 3.5, 3.75, 4.0]


def main():
    S = slicer_amd.Slicer(0, max_chunk=N)
    if os.environ.get("STREAM") == "null":
        S.set_stream(torch.cuda.current_stream().cuda_stream)
    elif os.environ.get("STREAM") == "torch":
        st = torch.cuda.Stream()
        S.set_stream(st.cuda_stream)
    bufs = []
    for f in range(FILES):
        b = torch.empty(N * 3, dtype=torch.float32, device="cuda")
        S.synth_positions(b.data_ptr(), f * N, N, BOX, seed=0x51CE2, clustered=False)
        bufs.append(b)
    for want in (True, False):
        best = 1e9
        for rep in range(4):
            if rep == 3:
                S.profile_reset()
                S.profile_enable(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            S.plane_begin(NPIX, 0.25, LDS, LD2S, mas=int(os.environ.get("MAS", "1")), want_type_maps=want)
            for b in bufs:
                S.file_begin([0, N, 0, 0, 0, 0], [0, 0.0123, 0, 0, 0, 0], BOX, (-1, 1, -1), 3, (0.3, 0.6, 0.1), 3.0)
                S.deposit_device(1, b.data_ptr(), N)
                S.file_end()
            S.plane_finalize()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        S.profile_enable(False)
        prof = S.profile_get()
        print(f"want_type_maps={int(want)}: {1e3 * best:.3f} ms per snapshot; "
              + ", ".join(f"{k} {1e3 * v[1] / max(v[0], 1):.1f} us" for k, v in prof.items()), flush=True)
    S.close()


if __name__ == "__main__":
    main()
