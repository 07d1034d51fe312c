#!/usr/bin/env bash
# round 6, call 2: conv_c1 tests (full file), cycle stamps of the streaming kernel's blocks (probe build), grid sweep
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_c1_gpu.py -q 2>&1 | tail -8 > gpurun_out/g2_tests.txt
cat gpurun_out/g2_tests.txt
FPD_AMD_LIB=build_ab/c1t/libfpd_amd.so timeout 300 python tools/c1_bench.py --iters 2 --rounds 1 --only @64 2>&1 | grep -v "us (min" > gpurun_out/g2_stamps.txt
head -60 gpurun_out/g2_stamps.txt
for b in 128 192 256 384 512; do echo "FPD_C1_BLOCKS=$b"; FPD_C1_BLOCKS=$b timeout 300 python tools/c1_bench.py --only @64 --rounds 2 2>&1 | grep "us (min"; done | tee gpurun_out/g2_blocks.txt
