#!/usr/bin/env bash
# round-2 probe 20: upper bound of folding the foldable BN-backward applies into their consumers (FPD_SKIP_APPLY=1 drops them;
# gradients are wrong, timing only), hourglass and HRNet, same box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p20; mkdir -p $O
for v in 0 1 0 1; do
  FPD_SKIP_APPLY=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/b_$v.json 2> $O/b_$v.err
  python -c "import json;d=json.load(open('$O/b_$v.json'));print('hourglass skip_apply=$v', d['ms_per_step'])" || tail -3 $O/b_$v.err
done
for v in 0 1; do
  FPD_SKIP_APPLY=$v timeout 300 python bench.py --config hrnet --steps 10 --warmup 3 > $O/h_$v.json 2> $O/h_$v.err
  python -c "import json;d=json.load(open('$O/h_$v.json'));print('hrnet skip_apply=$v', d['ms_per_step'])" || tail -3 $O/h_$v.err
done
