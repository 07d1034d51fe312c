#!/usr/bin/env bash
# full GPU gate of the round: every -m gpu test, smoke(), the default bench line (parity + cpu_baseline legs included)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02full; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log )
tail -25 $O/tests_gpu.log
( timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log ); tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python -c "import json;d=json.load(open('$O/bench_default.json'));print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['step']['frac'], d.get('cpu_baseline',{}).get('value'))"
