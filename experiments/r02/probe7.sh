#!/usr/bin/env bash
# round-2 probe 7: is the 15.3 ms step of probe 6 the box or the tree?  Same box: old tree (bdcb059, 11.89 ms when measured)
# vs HEAD, eager vs hipGraph replay, host enqueue time of one step.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p7; mkdir -p $O
b() { # name, dir, extra flags
  local name=$1 dir=$2; shift 2
  ( cd $dir && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity "$@" ) > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', d['ms_per_step'], d['roofline'].get('avg_us'), d['roofline']['step']['note'][-22:])" || tail -3 $O/bench_$name.err
}
nproc; lscpu | grep -E "Model name|MHz" | head -3
b old build_ab/old
b head .
b old2 build_ab/old
b head_graphs . --graphs
b head2 .
rocm-smi --showclocks 2>/dev/null | head -20
( timeout 300 python -m pytest tests/test_infer_gpu.py -m gpu -q -p no:cacheprovider > $O/tests_infer.log 2>&1; echo "rc=$?" >> $O/tests_infer.log ); tail -4 $O/tests_infer.log
