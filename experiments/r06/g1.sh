#!/usr/bin/env bash
# round 6, call 1: first run of the streaming 1x1 kernel (conv_c1): its tests, then the micro-benchmark against conv_pp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_c1_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/g1_tests.txt
cat gpurun_out/g1_tests.txt
timeout 600 python tools/c1_bench.py 2>&1 | tee gpurun_out/g1_bench.txt
