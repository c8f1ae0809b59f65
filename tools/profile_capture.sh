#!/bin/bash
# rocprofv3 captures of the default bench command (kernel trace, then the two HBM counters in separate passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes) + a calibration of FETCH_SIZE on a known byte count in the
# lookup kernel's own access pattern (tools/randline: random aligned 128-byte lines, four lanes per line).
# usage (on the GPU box, through gpurun): bash tools/profile_capture.sh <outdir under gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- $GRAFT_REPO_ROOT/tools/randline 16 2097152 4 > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_instr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-w2 > $OUT/pmc_instr.log 2>&1 || echo "instruction counters not collected"
tail -1 $OUT/trace.log | cut -c1-300
ls $OUT/*/*/ | head -40
