#!/usr/bin/env bash
# round-2 probe 19: ReLU mask folded into the BN-backward statistics pass of HRNet block tails (FPD_FUSE_MASK=0 restores the
# two launches): tests + same-box A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p19; mkdir -p $O
( timeout 600 python -m pytest tests/test_hrnet_gpu.py tests/test_fp8_gpu.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
for v in 1 0 1 0; do
  FPD_FUSE_MASK=$v timeout 300 python bench.py --config hrnet --steps 10 --warmup 3 > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json;d=json.load(open('$O/bench_$v.json'));print('hrnet fuse_mask=$v', d['ms_per_step'], d['value'])" || tail -5 $O/bench_$v.err
done
