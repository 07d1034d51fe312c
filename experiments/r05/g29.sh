#!/usr/bin/env bash
# round 5 call 29: kernel-trace pass of tools/profile.sh on the final library (kernel stats + per-shape table; the counter passes of
# the round's profile run stay: the kernels they describe did not change)
cd "$(dirname "$0")/../.." || exit 1
ROOT=$PWD; OUT=$ROOT/gpurun_out/r05prof2; mkdir -p $OUT/stats
cd /tmp && export TMPDIR=/tmp
FPD_LAUNCH_LOG=$OUT/stats/launch.log rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o b -- python $ROOT/bench.py --no-cpu-baseline --no-parity --no-phase-times --steps 20 --warmup 5 > $OUT/stats.log 2>&1
cd $ROOT; python tools/profile_summarize.py $OUT r05 2>&1 | tail -3
ls $OUT | head; rm -f $OUT/stats/*kernel_trace.csv.bak
