#!/usr/bin/env bash
# round-2 probe 21: weight-gradient lane batch size (the tail of the lane is exposed before Adam) -- same box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p21; mkdir -p $O
for v in ${SWEEP:-24 8 12 16 24 32}; do
  FPD_WGRAD_BATCH=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/b_$v.json 2> $O/b_$v.err
  python -c "import json;d=json.load(open('$O/b_$v.json'));print('wgrad_batch=$v', d['ms_per_step'])" || tail -3 $O/b_$v.err
done
