#!/usr/bin/env bash
# round 6, call 17: fused Bottleneck: old (LDS-staged fp32 epilogue) / register epilogue / the same without its stores / without its residual loads
mkdir -p gpurun_out
for i in 1 2; do
for v in bnold bnnp bnns bnnr; do
  for cap in 128; do
    echo "$v cap $cap: $(FPD_AMD_LIB=build_ab/$v/libfpd_amd.so FPD_BNECK_BLOCKS=$cap ONLY=64 timeout 300 python tools/bneck_bench.py 2>&1 | grep fused | cut -c1-60)"
  done
done
done | tee gpurun_out/g17_bneck.txt
