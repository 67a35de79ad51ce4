# e2e createDensityMaps timing: chunk size / read threads / pinning of the destination maps
python - <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from slicer_amd import gadget, synth
n = 1 << 24
d = tempfile.mkdtemp(prefix="e2e_", dir="/tmp")
base = os.path.join(d, "snap_000")
gadget.write_snapshot(base + ".0", synth.positions(0, n, 1000.0), [0, n, 0, 0, 0, 0], [0, 0.0123, 0, 0, 0, 0], 1000.0)
for chunk in ("24", "22"):
    for pin in ("1", "0"):
        env = dict(os.environ, ADAPTER_REPEAT="6", SLICER_AMD_READ_THREADS="8", SLICER_D2H_PIN=pin, SLICER_AMD_CHUNK_LOG2=chunk)
        r = subprocess.run(["tests/cpp/adapter_driver", base, "0", "1", "4096", "0.25", "3.0", "3.25", "3.0", "0", "0", os.path.join(d, "m.bin")], capture_output=True, env=env, text=True)
        print("chunk 2^" + chunk, "pin", pin, [l.split(": ")[1] for l in r.stderr.splitlines() if l.startswith("createDensityMaps call")])
PY
