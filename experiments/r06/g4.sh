#!/usr/bin/env bash
# round 6, call 4: conv_c1 after the prologue order / parallel flush / pipelined weight-gradient round: tests, bench, stamps
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_c1_gpu.py -q 2>&1 | tail -4 > gpurun_out/g4_tests.txt
cat gpurun_out/g4_tests.txt
timeout 600 python tools/c1_bench.py 2>&1 | grep "us (min" | tee gpurun_out/g4_bench.txt
FPD_AMD_LIB=build_ab/c1t/libfpd_amd.so timeout 300 python tools/c1_bench.py --iters 1 --rounds 1 --only @64 2>&1 | grep "conv_c1 C" | sort | awk 'NR%4==1' > gpurun_out/g4_stamps.txt
cat gpurun_out/g4_stamps.txt
