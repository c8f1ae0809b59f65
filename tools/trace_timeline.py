"""Development aid: prints the last steps of a rocprofv3 kernel trace as a timeline (start offset, duration, gap to the
previous kernel's end), to see where a step's time is kernels and where it is gaps.
   python tools/trace_timeline.py <dir with *_kernel_trace.csv> [number of kernels from the end]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.2f us  dur %7.2f  gap %6.2f  grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Grid_Size", ""), name))
    prev_end = e
