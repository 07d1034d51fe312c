#!/usr/bin/env bash
# per-shape trace A/B: previous library vs the packed-pair conv_pp
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g9; mkdir -p $O
timeout 600 python tools/trace_ab.py $O build_ab/libfpd_amd_prev.so - > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-160
rm -rf $O/A $O/B
