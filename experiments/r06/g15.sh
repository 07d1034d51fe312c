#!/usr/bin/env bash
# round 6, call 15: fused Bottleneck with the register epilogue (permlane32_swap) and the next tile's loads requested early: tests, stamps, bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "bottleneck or bneck" 2>&1 | tail -4 | tee gpurun_out/g15_tests.txt
timeout 600 python -m pytest tests/test_exact_gpu.py -q 2>&1 | tail -3 | tee -a gpurun_out/g15_tests.txt
for cap in 128 256; do
  FPD_BNECK_BLOCKS=$cap ONLY=64 FPD_AMD_LIB=build_ab/bnt/libfpd_amd.so timeout 300 python tools/bneck_bench.py 2>&1 | grep "bneck W" | tail -2
  FPD_BNECK_BLOCKS=$cap timeout 300 python tools/bneck_bench.py 2>&1 | grep "fused"
done | tee gpurun_out/g15_bneck.txt
