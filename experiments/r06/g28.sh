#!/usr/bin/env bash
# round 6, call 28: conv_c1 takes the 128 -> 128 data gradients (output tile in the operand tile's place, weight gradient on the lane) -- tests, step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_c1_gpu.py -q -x -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 | tee gpurun_out/g28_tests.txt
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g28_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')" || tail -5 gpurun_out/g28_err.txt; }
for i in 1 2 3; do
  run base "FPD_AMD_LIB=$PWD/build_ab/base/libfpd_amd.so"
  run bwd128 ""
done | tee gpurun_out/g28_ab.txt
