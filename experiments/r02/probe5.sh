#!/usr/bin/env bash
# round-2 probe 5: conv_tile on row tiles of any width (HRNet 48/24/12/6): kernel + exact + model tests, both bench lines
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p5; mkdir -p $O
( timeout 700 python -m pytest tests/test_exact_gpu.py tests/test_kernels_gpu.py tests/test_hrnet_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "not wgrad and not stem and not head and not loss" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -4 $O/tests.log
timeout 300 python bench.py --config hrnet --steps 10 --warmup 3 > $O/bench_hrnet.json 2> $O/bench_hrnet.err
python -c "import json;d=json.load(open('$O/bench_hrnet.json'));print('hrnet', d['ms_per_step'], d['value'], d['roofline']['frac'])" || tail -5 $O/bench_hrnet.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench.json'));print('hourglass', d['ms_per_step'], d['roofline']['avg_us'])"
