#!/usr/bin/env bash
# round 6, call 6: the step with / without the streaming 1x1 kernel, interleaved (FPD_C1=0: conv_pp as in round 5)
mkdir -p gpurun_out
for i in 1 2 3; do
  for v in 0 1; do
    FPD_C1=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('FPD_C1=$v', d['ms_per_step'], 'ms/step', d.get('launches_per_step'))"
  done
done | tee gpurun_out/g6_ab.txt
