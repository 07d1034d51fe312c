#!/usr/bin/env bash
# round 5 call 11: wgrad3 range count vs step time (fewer ranges = fewer slabs and fewer CUs taken from the chain, slower kernel)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g11; mkdir -p $O
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  for r in 16 24 32 48 64; do FPD_WGRAD3_RANGES=$r run r${r}_$i; done
done
