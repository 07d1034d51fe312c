#!/usr/bin/env bash
# round 6: the default bench line once more on the final tree (r06 parity record / PMC tables present, 12 convolution classes), and the micro-benchmark table
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06final; mkdir -p $O
timeout 900 python bench.py > $O/final_bench_line.json 2> $O/final_bench_err.txt; tail -c 300 $O/final_bench_line.json; echo
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_style_line.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/driver_style_line.json').read().strip().splitlines()[-1]);print('driver-style', d['ms_per_step'], d['value'])"
timeout 900 python tools/c1_bench.py --rounds 3 2>&1 | grep "us (min" | cut -c1-210 | tee $O/c1_bench.txt
