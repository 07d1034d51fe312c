#!/usr/bin/env bash
# round 6, call 29: fused Bottleneck with the next tile's x rows requested in front of the last weight step (build_ab/bpre2) -- tests, kernel alone, step A/B
mkdir -p gpurun_out
V=$PWD/build_ab/bpre2/libfpd_amd.so
FPD_AMD_LIB=$V timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py -q -x -k "bottleneck" -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/g29_tests.txt
for i in 1 2; do
  for v in base pre; do
    L=""; [ $v = pre ] && L=$V
    for hw in 64 32 16; do echo -n "$v "; FPD_AMD_LIB=$L ONLY=$hw timeout 120 python tools/bneck_bench.py 2>&1 | grep fused; done
  done
done | tee gpurun_out/g29_kernel.txt
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g29_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step', 'bneck frac', d['roofline']['frac'], d['roofline'].get('us'))" || tail -5 gpurun_out/g29_err.txt; }
for i in 1 2 3; do
  run base ""
  run pre "FPD_AMD_LIB=$V"
done | tee gpurun_out/g29_ab.txt
