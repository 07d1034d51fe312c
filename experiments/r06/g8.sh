#!/usr/bin/env bash
# round 6, call 8: why did the step go back to 9.67 ms?  current library / without conv_c1 / with the elementwise kernels assuming no aliasing (round-5 behaviour)
mkdir -p gpurun_out
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step')"; }
for i in 1 2; do
  run base ""
  run noc1 FPD_C1=0
  run ew_noalias FPD_AMD_LIB=build_ab/ewna/libfpd_amd.so
done | tee gpurun_out/g8_ab.txt
