"""profiles/traffic.json from a rocprofv3 --pmc summary (tools/pmc_summary.py output).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB and, on gfx950,
FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced reads (MI355X_MICROARCH.md "HBM"), hence the
factor 2 on the read side.  Calibration in this code base: k_project_bin reads exactly 12 B x 16 777 216 =
201.3 MB per launch with dwordx4 loads and the corrected counter gives 201.6 MB."""
import json
import re
import sys

summary, workload, out = sys.argv[1], sys.argv[2], sys.argv[3]
names = {"k_project_bin_fast": "project_bin", "k_project_bin_general": "project_bin_general", "k_bin_scatter": "bin_scatter",
         "k_tile_deposit": "tile_deposit", "k_scan_blocks": "bin_scan", "k_finalize_tsc": "finalize_tsc",
         "k_direct": "direct_deposit"}
# launches of each kernel per bench step (8 sub-files per snapshot, one tile-deposit flush per snapshot)
per_step = {"project_bin": 8, "bin_scan": 8, "bin_scatter": 8, "tile_deposit": 1}
cur, vals = None, {}
for line in open(summary):
    if not line.startswith(" "):
        cur = line.strip()
        vals[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean=(\S+)", line)
        if m and cur:
            vals[cur][m.group(1)] = float(m.group(3))
res = {}
for k, short in names.items():
    v = vals.get(k)
    if not v or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    res[short] = {"workload": workload, "fetch_size_kib": v["FETCH_SIZE"], "write_size_kib": v["WRITE_SIZE"],
                  "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0,
                  "tcc_ea0_atomic": v.get("TCC_EA0_ATOMIC_sum")}
if all(k in res for k in per_step):
    total = sum(per_step[k] * res[k]["hbm_bytes_per_launch"] for k in per_step)
    n_in, npix, planes = 8 * (1 << 24), 4096, 4
    alg = 12.0 * n_in + 4.0 * npix * npix * planes
    res["whole_step"] = {"workload": workload, "hbm_bytes": total, "algorithmic_bytes": alg, "ratio": total / alg,
                         "launches_per_step": per_step,
                         "note": "sum over the step's launches of (2 * FETCH_SIZE + WRITE_SIZE); algorithmic = 12 B per "
                                 "input particle + 4 B per map pixel (SURVEY S8d); the accumulator memsets are not counted"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
