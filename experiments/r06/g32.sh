#!/usr/bin/env bash
# round 6, call 32: cycle stamps of the fused head inside the step (probe build build_ab/headt, block 0, last tile)
mkdir -p gpurun_out
FPD_AMD_LIB=$PWD/build_ab/headt/libfpd_amd.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-phase-times 2>/dev/null | grep "^head tiles" | sort | uniq -c | sort -rn | head -40 | tee gpurun_out/g32_head_stamps.txt
