# development aid: kernel timeline of one LJ-melt step (fix edm_pair gpu_list) in the reference's order (LJ_ORDER=1) or batch order
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/lj
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lj -o lj -- python3 $GRAFT_REPO_ROOT/tools/lj_steps.py > $GRAFT_REPO_ROOT/gpurun_out/lj.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/lj.log
t=$(find $GRAFT_REPO_ROOT/gpurun_out/lj -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "")[-44:] for r in rows]
idx = [i for i, n in enumerate(names) if "pairlist_forces" in n]
i0, i1 = idx[60], idx[61]
t0 = int(rows[i0]["Start_Timestamp"])
for r, n in list(zip(rows, names))[i0:i1]:
    print("%9.1f us  dur %7.1f us  grid %-8s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], n))
print("step span %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
PY
