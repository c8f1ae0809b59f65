# development aid: kernel timeline of the W1 step with a communicator of ONE rank (RCCL), the N > 1 code path on one GPU
cd /tmp && export TMPDIR=/tmp
export EDM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rm -rf $GRAFT_REPO_ROOT/gpurun_out/d1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/d1 -o d1 -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 60 --warmup 5 --no-w2 --no-nd --no-cpu-baseline > /dev/null 2>&1
t=$(find $GRAFT_REPO_ROOT/gpurun_out/d1 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/profiles/step_timeline.py $t 40 | cut -c1-120
