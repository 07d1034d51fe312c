#!/usr/bin/env bash
# round 6, call 10: the step with / without the 3x3 strip kernel, interleaved
mkdir -p gpurun_out
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step')"; }
for i in 1 2 3; do
  run c3 ""
  run noc3 FPD_C3=0
done | tee gpurun_out/g10_ab.txt
