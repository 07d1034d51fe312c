#!/usr/bin/env bash
# round-3 evidence that is not part of tools/profile.sh: phase times, the --force-dist bench line, a kernel trace of it
cd "$(dirname "$0")/../.." || exit 1
ROOT=$PWD; O=$ROOT/gpurun_out/r03ev; mkdir -p $O
timeout 300 python tools/probes/phase_times.py > $O/r03_phase_times.txt 2>&1; tail -9 $O/r03_phase_times.txt
timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-parity --no-cpu-baseline > $O/r03_bench_line_force_dist.json 2> $O/fd.err; tail -2 $O/fd.err
python -c "import json;d=json.loads(open('$O/r03_bench_line_force_dist.json').read().strip().splitlines()[-1]);print('force-dist', d['ms_per_step'], d['config'].get('rccl_version'), d['config'].get('allreduce_exposed_us'))"      # RCCL's banner precedes the JSON line
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/dist_trace -- python $ROOT/bench.py --force-dist --steps 6 --warmup 2 --no-parity --no-cpu-baseline > $O/trace.log 2>&1; tail -2 $O/trace.log
cd $ROOT
python tools/dist_trace_summarize.py $O/dist_trace > $O/r03_dist_trace_summary.txt 2>&1; cat $O/r03_dist_trace_summary.txt
rm -rf $O/dist_trace
