#!/usr/bin/env bash
# round-3 probe 13: per-shape device times (hipGraph replays) of a library variant against build_ab/base
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p13; mkdir -p $O
for rep in 1 2; do
echo "== new"; timeout 300 python tools/conv_bench.py --graph 2>&1 | grep -E "@64|@128"
echo "== base"; FPD_AMD_LIB=$PWD/build_ab/base/libfpd_amd.so timeout 300 python tools/conv_bench.py --graph 2>&1 | grep -E "@64|@128"
done | tee $O/shapes.txt
