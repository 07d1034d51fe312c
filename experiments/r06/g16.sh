#!/usr/bin/env bash
# round 6, call 16: fused Bottleneck, register epilogue with / without the early request of the next tile (interleaved)
mkdir -p gpurun_out
for i in 1 2; do
for lib in fast-human-pose-estimation.pytorch_amd/csrc/libfpd_amd.so build_ab/bnnp/libfpd_amd.so; do
  for cap in 128 256; do
    echo "$lib cap $cap: $(FPD_AMD_LIB=$lib FPD_BNECK_BLOCKS=$cap ONLY=64 timeout 300 python tools/bneck_bench.py 2>&1 | grep fused)"
  done
done
done | tee gpurun_out/g16_bneck.txt
