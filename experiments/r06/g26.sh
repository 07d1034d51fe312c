#!/usr/bin/env bash
# round 6, call 26: conv_c1's smallest launch, forward and data gradients apart (kernels alone: experiments/r06/g25.sh)
mkdir -p gpurun_out
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g26_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')" || tail -5 gpurun_out/g26_err.txt; }
for i in 1 2 3; do
  run f2048_b2048 ""
  run f512_b2048 "FPD_C1_MIN_PX=512"
  run f512_b16384 "FPD_C1_MIN_PX=512 FPD_C1_MIN_PX_BWD=16384"
  run f2048_b16384 "FPD_C1_MIN_PX_BWD=16384"
  run f512_b8192 "FPD_C1_MIN_PX=512 FPD_C1_MIN_PX_BWD=8192"
done | tee gpurun_out/g26_ab.txt
