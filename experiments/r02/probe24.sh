#!/usr/bin/env bash
# round-2 probe 24: fused-Bottleneck grid cap on top of WGRAD_BATCH=8 (same box)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p24; mkdir -p $O
for v in 160 96 112 128 144 160 128; do
  FPD_BNECK_BLOCKS=$v FPD_HEAD_BLOCKS=${HEADB:-160} timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/b_$v.json 2> $O/b_$v.err
  python -c "import json;d=json.load(open('$O/b_$v.json'));print('bneck_blocks=$v', d['ms_per_step'], d['roofline']['avg_us'])" || tail -3 $O/b_$v.err
done
