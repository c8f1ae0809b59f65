#!/usr/bin/env python3
"""Prints the kernel timeline of one steady-state bench step from a rocprofv3 kernel trace CSV."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "")[-44:] for r in rows]
idx = [i for i, n in enumerate(names) if "pair_forces" in n]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 30   # (a step of bench.py's timed region: --warmup 5 --steps 50)
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
busy = 0.0
for r, n in list(zip(rows, names))[i0:i1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    print("%9.1f us  dur %7.1f us  grid %-8s %s" % (s, d, r["Grid_Size_X"], n))
print("step span %.1f us, kernel-busy %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, busy))
