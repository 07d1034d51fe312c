#!/usr/bin/env bash
# round 6, call 11: kernel trace of the step with conv_c1 / conv_c3 in (per-shape table, mid-round)
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06aprof; mkdir -p $OUT/stats
cd /tmp && export TMPDIR=/tmp
FPD_LAUNCH_LOG=$OUT/stats/launch.log rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o b -- python $ROOT/bench.py --no-cpu-baseline --no-parity --no-phase-times --steps 20 --warmup 5 > $OUT/stats.log 2>&1
cd $ROOT
python tools/profile_summarize.py $OUT r06a 2>&1 | tail -5
ls $OUT | head; ls $OUT/stats | head
