#!/usr/bin/env python3
"""Condenses rocprofv3 output (gpurun_out/<dir>/{trace,pmc_fetch,pmc_write}) into the tracked
summaries under profiles/: usage  python profiles/summarize.py gpurun_out/p2 r01

Commands that produced the inputs (run on the MI355X box through gpurun):
  rocprofv3 --kernel-trace --stats --output-format csv -d <dir>/trace     -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --w2
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/pmc_fetch       -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --w2
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/pmc_write       -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --w2
Counter units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half of a wide coalesced (16 B/lane)
streaming read, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(here, "%s_kernel_stats.csv" % tag))
    trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
        per[key].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    per = {k: [d for _, d in sorted(v)] for k, v in per.items()}   # launch order
    summary = {"kernels": [], "pmc": {}}
    # the trace command runs `--warmup 5 --steps 50`: launches 6..55 of the step's first kernel are bench.py's timed
    # region (acceptance uniforms read from memory); the later ones belong to the device-RNG extra, which reads none
    warmup, steps = 5, 50
    for (name, grid, wg), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        e = dict(kernel=name, grid=grid, workgroup=wg, calls=len(v), avg_us=sum(v) / len(v),
                 min_us=min(v), max_us=max(v), total_us=sum(v))
        if "k_pair_forces_select" in name and len(v) >= warmup + steps:
            e["avg_us_timed_region"] = sum(v[warmup:warmup + steps]) / steps
        summary["kernels"].append(e)
    for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(src, which, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == counter:
                agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        for (name, grid), v in agg.items():
            e = summary["pmc"].setdefault("%s grid=%d" % (name, grid), {})
            e[counter + "_KiB_avg"] = sum(v) / len(v)
            e[counter + "_dispatches"] = len(v)
    for k, e in summary["pmc"].items():
        if "FETCH_SIZE_KiB_avg" in e and "WRITE_SIZE_KiB_avg" in e:
            e["hbm_read_bytes_corrected"] = e["FETCH_SIZE_KiB_avg"] * 1024 * 2
            e["hbm_write_bytes"] = e["WRITE_SIZE_KiB_avg"] * 1024
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
    with open(os.path.join(here, "%s_summary.json" % tag), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("wrote", os.path.join(here, "%s_summary.json" % tag))


if __name__ == "__main__":
    main()
