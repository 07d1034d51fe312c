#!/usr/bin/env bash
# round-3 probe 11: finer cycle stamps of the conv_pp prologue (stamps relative to the block's entry)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p11; mkdir -p $O
export FPD_AMD_LIB=$PWD/build_ab/pptime/libfpd_amd.so
for sh in "3x3 64>64 @64" "1x1 128>64 @64" "1x1 64>128 @64"; do
  FPD_CONV_PP=1 timeout 120 python tools/conv_bench.py --iters 1 --only "$sh" 2>&1 | tail -3
done | tee $O/stamps.txt
