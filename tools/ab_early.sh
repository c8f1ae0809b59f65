#!/bin/bash
# A/B on one box: the limiter's early word on / off, alternating runs
for i in 1 2 3 4; do
  for v in 1 0; do
    EDM_HIP_EARLY_WORD=$v python bench.py --steps 300 --warmup 20 --no-nd --no-w2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('early=$v', round(d['ms_per_step']*1e3,2), round(d['ms_per_step_device_rng']*1e3,2))"
  done
done
