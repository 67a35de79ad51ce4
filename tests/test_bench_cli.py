"""bench.py's command line on one GPU: the JSON contract, and the one-run rehearsal of BOTH multi-rank layouts
 with a one-rank process group (SLICER_BENCH_FORCE_DIST=1: the same code path as N > 1 -- RCCL
communicator, StepGather, reduce_planes with the sync-free reduce meta -- on a single device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--side", "128", "--npix", "1024", "--files", "2", "--snapshots", "2", "--steps", "4", "--warmup", "1",
         "--cpu-baseline", "off", "--parity", "off", "--e2e", "off", "--profile-steps", "1"]


def test_one_gpu_line_has_the_contract_fields():
    d = _bench(SMALL)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["config"]["reduce_layout"] is None
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    # one snapshot in flight by default; with --streams 2 the same steps on one stream are reported next to `value`
    assert d["config"]["streams"] == 1 and d["single_stream"] is None
    two = _bench(SMALL + ["--streams", "2"])
    assert two["config"]["streams"] == 2 and two["single_stream"]["value"] > 0 and two["value"] > 0


def test_one_run_times_the_step_layout_and_the_per_plane_reduce_layout():
    d = _bench(SMALL + ["--shard", "steps", "--reduce-layout", "on"], env={"SLICER_BENCH_FORCE_DIST": "1",
                                                                             "MASTER_PORT": "29533"})
    assert d["config"]["shard"] == "steps" and d["value"] > 0
    rl = d["config"]["reduce_layout"]
    assert d["config"]["collective"] is None and "no data-path collective" in d["config"]["maps"]
    for algo in ("rooted", "p2p", "steps_gather"):
        assert "error" not in rl[algo], rl[algo]
        assert rl[algo]["value"] > 0 and rl[algo]["ms_per_step"] > 0
        # one rank: both layouts deposit the same particles, so the rates must be of the same order (the reduce of a
        # one-rank group moves nothing); a sum that stalled the pipeline would show here
        # (this tiny job is 0.2 ms per step: the fixed costs of a collective round show; the headline job: 1.9 vs 2.0 ms)
        assert rl[algo]["value"] > 0.2 * d["value"], (rl[algo], d["value"])


def test_a_stuck_secondary_layout_cannot_take_the_main_number_down():
    """N > 1 runs time the reduce layouts after the main number; those collectives meet real multi-GPU hardware for the
    first time in the scaling run.  A watchdog prints the line with whatever finished and leaves if they do not finish in
    time -- here the time allowed is zero."""
    d = _bench(SMALL + ["--shard", "steps", "--reduce-layout", "on", "--secondary-timeout", "0.001"],
               env={"SLICER_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29534"})
    assert d["value"] > 0 and d["config"]["shard"] == "steps"
    assert "cut off" in d["config"]["reduce_layout"]["error"]
