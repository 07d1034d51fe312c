#!/usr/bin/env bash
# round 5 call 30 (last): the one-rank RCCL line on the final library, then the step's HBM counter passes if time allows
cd "$(dirname "$0")/../.." || exit 1
ROOT=$PWD; O=$ROOT/gpurun_out/r05g30; mkdir -p $O
timeout 120 python bench.py --force-dist --no-cpu-baseline --no-parity --no-phase-times > $O/bench_line_force_dist.json 2> $O/fd_err.txt
python -c "import json;d=json.loads(open('$O/bench_line_force_dist.json').read().strip().splitlines()[-1]);print('force-dist', d['ms_per_step'], d.get('config',{}).get('allreduce_exposed_us'))"
OUT=$ROOT/gpurun_out/r05prof2
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-parity --no-phase-times"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$OUT/step_$c; mkdir -p $d
  FPD_LAUNCH_LOG=$d/launch.log timeout 100 rocprofv3 --kernel-trace --pmc $c -f csv -d $d -o p -- $B --steps 3 --warmup 1 > $d.log 2>&1
done
cd $ROOT; python tools/profile_summarize.py $OUT r05 2>&1 | tail -1 | cut -c1-200
wc -l $OUT/r05_hbm_traffic.csv
