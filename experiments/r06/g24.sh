#!/usr/bin/env bash
# round 6, call 24: bn_request with every load of the TRAIN path issued in one go (no vmcnt(0) between gamma / beta and the statistics) --
# kernel tests, kernels alone, step A/B against the previous library (build_ab/base)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv_c1_gpu.py tests/test_conv_c3_gpu.py tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/g24_tests.txt
B=$PWD/build_ab/base/libfpd_amd.so
for v in base new; do L=""; [ $v = base ] && L=$B; echo "== $v"; FPD_AMD_LIB=$L timeout 600 python tools/c1_bench.py --rounds 2 2>&1 | grep "us (min" | cut -c1-200; done | tee gpurun_out/g24_kernels.txt
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g24_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')" || tail -5 gpurun_out/g24_err.txt; }
for i in 1 2 3; do
  run base "FPD_AMD_LIB=$B"
  run new ""
done | tee gpurun_out/g24_ab.txt
