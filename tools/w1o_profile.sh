cd /tmp && export TMPDIR=/tmp
W1_ORDER=1 W1_LOG=1 python $GRAFT_REPO_ROOT/tools/w1_steps.py
W1_ORDER=1 W1_LOG=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/w1o -o w1o -- python3 $GRAFT_REPO_ROOT/tools/w1_steps.py > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/w1o -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-150
t=$(find $GRAFT_REPO_ROOT/gpurun_out/w1o -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/profiles/step_timeline.py $t 100 | cut -c1-110
