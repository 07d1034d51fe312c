#!/usr/bin/env bash
# round 5: the teacher's >= 64-high ops by kind inside the pipelined step (timing only)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05whatif2; mkdir -p $O
run() {
  FPD_WHATIF="$2" timeout 200 env $3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]);print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
for i in 1 2; do
run base ""
run t_bneck_big t_bneck_big
run t_head t_head
run t_plain_big t_plain_big
run t_big t_big
run head128 "" FPD_HEAD_BLOCKS=128
run head96 "" FPD_HEAD_BLOCKS=96
done
} | tee $O/summary.txt
