#!/usr/bin/env bash
# round 6, call 3: cycle stamps of the streaming kernel (probe build, one printf per row)
mkdir -p gpurun_out
FPD_AMD_LIB=build_ab/c1t/libfpd_amd.so timeout 300 python tools/c1_bench.py --iters 1 --rounds 1 --only @64 2>&1 | grep -v "us (min" | sort | uniq -c | sort -k2 > gpurun_out/g3_stamps.txt
cat gpurun_out/g3_stamps.txt | head -120
