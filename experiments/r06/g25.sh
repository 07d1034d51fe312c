#!/usr/bin/env bash
# round 6, call 25: small maps, conv_c1 against what the dispatcher picks otherwise (kernels alone); then the full GPU suite on the tree
mkdir -p gpurun_out
for k in "@4" "@8" "@16"; do timeout 600 python tools/c1_bench.py --rounds 2 --only "$k" 2>&1 | grep "us (min" | grep -v 3x3 | cut -c1-200; done | tee gpurun_out/g25_small.txt
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/gate1_tests.txt
cat gpurun_out/gate1_tests.txt
