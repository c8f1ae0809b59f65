#!/usr/bin/env python3
"""Condenses rocprofv3 output (gpurun_out/<dir>/{trace,pmc_fetch,pmc_write,cal_fetch,pmc_instr}, written on the MI355X
box by tools/profile_capture.sh) into the tracked summaries under profiles/:
    python profiles/summarize.py gpurun_out/p2 r02

Commands that produced the inputs (tools/profile_capture.sh, run through gpurun):
  rocprofv3 --kernel-trace --stats --output-format csv -d <dir>/trace     -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/pmc_fetch       -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/pmc_write       -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/cal_fetch       -- tools/randline 16 2097152 4
Counter units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half of a wide coalesced (16 B/lane)
streaming read, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
The guide calls other access widths uncalibrated: cal_fetch calibrates the read side on a KNOWN byte count in the
coordinate-CV lookup's own pattern (random aligned 128-byte lines, four lanes per line, 32 B per lane): the
summary records counted / true bytes (0.5 = the same factor 2).
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
    newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)   # (a re-used directory may hold older captures)
    stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(here, "%s_kernel_stats.csv" % tag))
    trace = newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
        per[key].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    per = {k: [d for _, d in sorted(v)] for k, v in per.items()}   # launch order
    summary = {"kernels": [], "pmc": {}}
    # the trace command runs `--warmup 5 --steps 50`: bench.py runs 5 warm-up steps, 2 more that switch the queue's
    # profiling on, then the 50 steps of the timed region: launches 8..57 of each kernel of the reference-order step
    warmup, steps = 7, 50
    for (name, grid, wg), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        e = dict(kernel=name, grid=grid, workgroup=wg, calls=len(v), avg_us=sum(v) / len(v),
                 min_us=min(v), max_us=max(v), total_us=sum(v))
        if any(k in name for k in ("k_pair_forces_ordered", "k_ordered_records", "k_select_prep", "k_integrals_gather")) \
                and len(v) >= warmup + steps:
            e["avg_us_timed_region"] = sum(v[warmup:warmup + steps]) / steps
        summary["kernels"].append(e)
    for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(src, which, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
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
    cal = glob.glob(os.path.join(src, "cal_fetch", "*", "*_counter_collection.csv"))
    if cal:
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(max(cal, key=os.path.getmtime)))
                if r["Counter_Name"] == "FETCH_SIZE" and "k<4>" in r["Kernel_Name"]]
        true_bytes = 2097152 * 128
        summary["fetch_size_calibration"] = dict(
            command="tools/randline 16 2097152 4  (2,097,152 random aligned 128-B lines of a 16 GiB buffer, four lanes per line)",
            true_bytes_per_launch=true_bytes, FETCH_SIZE_KiB_avg=sum(vals) / len(vals),
            counted_over_true=sum(vals) / len(vals) * 1024 / true_bytes,
            note="0.5: FETCH_SIZE tallies these 128-B requests at 64 B, the same factor the guide gives for 16-B-per-lane "
                 "streaming reads -- the read side of every kernel below is doubled")
    instr = glob.glob(os.path.join(src, "pmc_instr", "*", "*_counter_collection.csv"))
    if instr:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(max(instr, key=os.path.getmtime))):
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out = {}
        for (name, grid), cs in agg.items():
            if not any(k in name for k in ("k_lookup_quad", "k_pair_forces", "k_integrals_gather", "k_hill_gather", "k_pairlist",
                                           "k_ordered", "k_select_prep")):
                continue
            out["%s grid=%d" % (name, grid)] = {c: sum(v) / len(v) for c, v in cs.items()}
        with open(os.path.join(here, "%s_pmc_instr.json" % tag), "w") as fh:
            json.dump(out, fh, indent=1)
    with open(os.path.join(here, "%s_summary.json" % tag), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("wrote", os.path.join(here, "%s_summary.json" % tag))


if __name__ == "__main__":
    main()
