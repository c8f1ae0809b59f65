"""Development aid: per-kernel averages of the counters in a rocprofv3 --pmc capture (counter_collection.csv).
   python tools/pmc_rows.py <dir> [kernel name substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(files[-1])):
    name = r["Kernel_Name"].split("(")[0]
    if sub and sub not in name:
        continue
    key = (name[-50:], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key, {k: round(sum(v) / len(v), 1) for k, v in cs.items()}, "launches", max(len(v) for v in cs.values()))
