#!/usr/bin/env bash
# round 6, call 20: conv_c1 with 16 operand channels (score_ forward, score data gradient) -- tests, A/B against the same library without it; then the round's trace
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_c1_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/g20_tests.txt
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g20_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step', d.get('launches_per_step'))" || tail -5 gpurun_out/g20_err.txt; }
for i in 1 2 3; do
  run base "FPD_AMD_LIB=$PWD/build_ab/no16/libfpd_amd.so"
  run c16 ""
done | tee gpurun_out/g20_ab.txt
timeout 900 bash tools/profile.sh r06a2 > gpurun_out/g20_profile.log 2>&1
ls gpurun_out/r06a2prof | head -30
