#!/bin/bash
# extended controller / gauss fuzz sweeps (seeds 1..N) -- run on the GPU box through gpurun
N=${1:-12}
fail=0
for seed in $(seq 1 $N); do
  EDM_FUZZ_SEED=$seed EDM_FUZZ_COUNT=60 timeout -k 10 300 python -m pytest tests/test_gpu_fuzz_controller.py -m gpu -x -q > gpurun_out/fuzz_ctrl_$seed.log 2>&1 || { fail=1; echo "controller seed $seed FAILED"; tail -5 gpurun_out/fuzz_ctrl_$seed.log; }
  echo "controller seed $seed: $(tail -1 gpurun_out/fuzz_ctrl_$seed.log)"
done
for seed in $(seq 1 4); do
  EDM_FUZZ_SEED=$seed EDM_FUZZ_REPS=6 timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/fuzz_gauss_$seed.log 2>&1 || { fail=1; echo "gauss seed $seed FAILED"; tail -5 gpurun_out/fuzz_gauss_$seed.log; }
  echo "gauss seed $seed: $(tail -1 gpurun_out/fuzz_gauss_$seed.log)"
done
exit $fail
