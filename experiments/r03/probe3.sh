#!/usr/bin/env bash
# round-3 probe 3: cycle stamps of the ping-pong convolution (FPD_PP_TIMING build)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p3; mkdir -p $O
export FPD_AMD_LIB=$PWD/build_ab/pptime/libfpd_amd.so
for sh in "3x3 64>64 @64" "1x1 128>64 @64" "1x1 64>128 @64" "l1 3x3"; do
  timeout 120 python tools/conv_bench.py --iters 1 --only "$sh" 2>&1 | tail -9
done | tee $O/stamps.txt
