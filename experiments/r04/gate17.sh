#!/usr/bin/env bash
# the test files gate16 did not run, on the library with the wreduce change; then the default bench line again
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g17; mkdir -p $O
timeout 700 python -m pytest tests/test_bf16_parity_gpu.py tests/test_dist_gpu.py tests/test_entry_gpu.py tests/test_exact_gpu.py tests/test_fp8_gpu.py tests/test_fullsize_gpu.py tests/test_hrnet_gpu.py tests/test_infer_gpu.py -m gpu -q --tb=line -p no:cacheprovider > $O/t.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|Error" $O/t.txt | cut -c1-250 | head
timeout 300 python bench.py > $O/bench_line.json 2> $O/bench.err; python -c "import json;d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['avg_us'])"
