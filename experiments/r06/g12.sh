#!/usr/bin/env bash
# round 6, call 12: the grid caps of the teacher's persistent kernels against the new student kernels (one box, interleaved)
mkdir -p gpurun_out
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')"; }
for i in 1 2; do
  run base ""
  run bneck96 FPD_BNECK_BLOCKS=96
  run bneck160 FPD_BNECK_BLOCKS=160
  run bneck192 FPD_BNECK_BLOCKS=192
  run bneck256 FPD_BNECK_BLOCKS=256
  run head128 FPD_HEAD_BLOCKS=128
  run head224 FPD_HEAD_BLOCKS=224
  run c1_192 FPD_C1_BLOCKS=192
  run c3_192 FPD_C3_BLOCKS=192
  run wg3_ranges48 FPD_WGRAD3_RANGES=48
  run wg3_ranges24 FPD_WGRAD3_RANGES=24
  run c3min8192 FPD_C3_MIN_PX=8192
done | tee gpurun_out/g12_caps.txt
