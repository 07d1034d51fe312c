#!/usr/bin/env bash
# round-2 probe 22: WGRAD_BATCH default 8: HRNet A/B, hourglass confirmation, lanes test
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p22; mkdir -p $O
for v in 8 24 4 8 24; do
  FPD_WGRAD_BATCH=$v timeout 300 python bench.py --config hrnet --steps 10 --warmup 3 > $O/h_$v.json 2> $O/h_$v.err
  python -c "import json;d=json.load(open('$O/h_$v.json'));print('hrnet wgrad_batch=$v', d['ms_per_step'])" || tail -3 $O/h_$v.err
done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/b.json 2> $O/b.err; python -c "import json;d=json.load(open('$O/b.json'));print('hourglass default', d['ms_per_step'])"
( timeout 600 python -m pytest tests/test_model_gpu.py tests/test_entry_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2 )
