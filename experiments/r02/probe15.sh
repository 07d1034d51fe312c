#!/usr/bin/env bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p15; mkdir -p $O
ONLY=64 FPD_AMD_LIB=$PWD/build_ab/timing/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep -A1 "bneck W" | tail -6 | tee $O/stamps.txt
