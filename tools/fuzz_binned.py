"""Longer run of tests/test_gpu_parity.py::test_random_configurations_binned_path: seeds 12..259 (needs a GPU)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import test_gpu_parity as T
import slicer_amd
S = slicer_amd.Slicer(0, max_chunk=1 << 20)
bad = 0
for seed in range(12, 260):
    try:
        T.test_random_configurations_binned_path(S, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:200], flush=True)
    if seed % 40 == 0:
        print("seed", seed, "ok so far, failures:", bad, flush=True)
print("done, failures:", bad)
S.close()
