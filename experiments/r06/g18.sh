#!/usr/bin/env bash
# round 6, call 18: balanced persistent grids of conv_c1 / conv_c3 (every block the same number of rounds) vs the previous build; caps re-swept
mkdir -p gpurun_out
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')"; }
for i in 1 2 3; do
  run balanced ""
  run c1_256_c3_256 "FPD_C1_BLOCKS=256 FPD_C3_BLOCKS=256"
  run c1_171_c3_128 "FPD_C1_BLOCKS=171 FPD_C3_BLOCKS=128"
  run c1_128_c3_128 "FPD_C1_BLOCKS=128 FPD_C3_BLOCKS=128"
  run c1_224_c3_171 "FPD_C1_BLOCKS=224 FPD_C3_BLOCKS=171"
done | tee gpurun_out/g18_balanced.txt
timeout 600 python -m pytest tests/test_conv_c1_gpu.py tests/test_conv_c3_gpu.py -q 2>&1 | tail -3
