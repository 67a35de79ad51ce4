"""Summarise rocprofv3 --pmc CSV passes: per kernel (short name), mean counter value per dispatch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        m = re.search(r"(k_[a-z_0-9]+)", name)
        short = m.group(1) if m else name[:30]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
