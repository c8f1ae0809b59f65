# development aid: instruction counters of the reference-order W1 step's kernels
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/w1pmc
W1_STEPS=20 W1_ORDER=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/w1pmc -o w1pmc -- python3 $GRAFT_REPO_ROOT/tools/w1_steps.py > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/w1pmc -name "*counter_collection.csv" | head -1)
python3 - $f <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
