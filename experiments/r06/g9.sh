#!/usr/bin/env bash
# round 6, call 9: first run of the 3x3 strip kernel (conv_c3): its tests, then the micro-benchmark against conv_pp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_c3_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/g9_tests.txt
cat gpurun_out/g9_tests.txt
timeout 600 python tools/c1_bench.py --only 3x3 2>&1 | tee gpurun_out/g9_bench.txt | tail -8
