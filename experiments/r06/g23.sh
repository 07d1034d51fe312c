#!/usr/bin/env bash
# round 6, call 23: conv_c1 stamps of the current kernel at 64x64 and on the small maps (probe build build_ab/c1t), and the kernels alone
mkdir -p gpurun_out
timeout 600 python tools/c1_bench.py --only "@8" 2>&1 | grep "us (min" | tee gpurun_out/g23_bench.txt
timeout 600 python tools/c1_bench.py --only "@16" 2>&1 | grep "us (min" | tee -a gpurun_out/g23_bench.txt
FPD_AMD_LIB=build_ab/c1t/libfpd_amd.so timeout 300 python tools/c1_bench.py --iters 1 --rounds 1 2>&1 | grep "conv_c1 C" > gpurun_out/g23_stamps_all.txt
wc -l gpurun_out/g23_stamps_all.txt
