#!/usr/bin/env bash
# round-3 closing evidence in one call: tools/profile.sh r03, evidence.sh, the HRNet line, the JSON-is-last-line check
cd "$(dirname "$0")/../.." || exit 1
ROOT=$PWD; O=$ROOT/gpurun_out/r03final; mkdir -p $O
bash tools/profile.sh r03 > $O/profile.log 2>&1; tail -3 $O/profile.log
bash experiments/r03/evidence.sh > $O/evidence.log 2>&1; tail -12 $O/evidence.log
timeout 600 python bench.py --config hrnet > $O/r03_bench_line_hrnet.json 2> $O/hrnet.err; tail -1 $O/hrnet.err
python -c "import json;d=json.load(open('$O/r03_bench_line_hrnet.json'));print('hrnet', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout 300 python bench.py --force-dist --steps 10 --warmup 3 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-60
