#!/usr/bin/env bash
# round 6, call 21: the blocks' range cuts as 32-bit quotients (fpd_cut) -- kernel tests, then A/B against the 64-bit build (build_ab/cut64)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_conv_c1_gpu.py tests/test_conv_c3_gpu.py tests/test_kernels_gpu.py tests/test_exact_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/g21_tests.txt
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g21_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')" || tail -5 gpurun_out/g21_err.txt; }
for i in 1 2 3; do
  run cut64 "FPD_AMD_LIB=$PWD/build_ab/cut64/libfpd_amd.so"
  run cut32 ""
done | tee gpurun_out/g21_ab.txt
