#!/usr/bin/env bash
# round 6, call 14: cycle stamps of the fused frozen Bottleneck (probe build) at 64x64 and 32x32, grid caps 128 / 256
mkdir -p gpurun_out
for cap in 128 256; do
  echo "== FPD_BNECK_BLOCKS=$cap"
  FPD_BNECK_BLOCKS=$cap ONLY=64 FPD_AMD_LIB=build_ab/bnt/libfpd_amd.so timeout 300 python tools/bneck_bench.py 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -12
done | tee gpurun_out/g14_bneck_stamps.txt
FPD_BNECK_BLOCKS=128 ONLY=64 timeout 300 python tools/bneck_bench.py 2>&1 | tail -1 | tee -a gpurun_out/g14_bneck_stamps.txt
FPD_BNECK_BLOCKS=256 ONLY=64 timeout 300 python tools/bneck_bench.py 2>&1 | tail -1 | tee -a gpurun_out/g14_bneck_stamps.txt
