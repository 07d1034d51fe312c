#!/usr/bin/env bash
# round 6, call 5: conv_c1 with the first tile requested unconditionally (counted waits in the prologue): tests, bench, stamps
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_c1_gpu.py -q 2>&1 | tail -4 > gpurun_out/g5_tests.txt
cat gpurun_out/g5_tests.txt
timeout 600 python tools/c1_bench.py 2>&1 | grep "us (min" | tee gpurun_out/g5_bench.txt
FPD_AMD_LIB=build_ab/c1t/libfpd_amd.so timeout 300 python tools/c1_bench.py --iters 1 --rounds 1 --only @64 2>&1 | grep "conv_c1 C" > gpurun_out/g5_stamps_all.txt
awk '{k=$2" "$3" "$5" "$7" "$9" "$11" "$13; c[k]++; if (c[k]==4) print}' gpurun_out/g5_stamps_all.txt | sort > gpurun_out/g5_stamps.txt
cat gpurun_out/g5_stamps.txt
