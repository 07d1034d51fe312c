#!/usr/bin/env bash
# round 6, call 13: persistent-grid sizes of conv_c1 / conv_c3 inside the step (one box, interleaved)
mkdir -p gpurun_out
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')"; }
for i in 1 2; do
  run base ""
  run c3_128 FPD_C3_BLOCKS=128
  run c3_160 FPD_C3_BLOCKS=160
  run c3_192 FPD_C3_BLOCKS=192
  run c3_224 FPD_C3_BLOCKS=224
  run c1_160 FPD_C1_BLOCKS=160
  run c1_192 FPD_C1_BLOCKS=192
  run c1_224 FPD_C1_BLOCKS=224
  run c1_192_c3_192 "FPD_C1_BLOCKS=192 FPD_C3_BLOCKS=192"
  run c1_224_c3_160 "FPD_C1_BLOCKS=224 FPD_C3_BLOCKS=160"
  run c1_192_c3_160 "FPD_C1_BLOCKS=192 FPD_C3_BLOCKS=160"
done | tee gpurun_out/g13_caps.txt
