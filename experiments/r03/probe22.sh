#!/usr/bin/env bash
# round-3 probe 22: where the student step waits for the teacher's map: at its start (FPD_TEACHER_WAIT=start) or in front of the loss
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p22; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  loss %s' % ('$1', d['ms_per_step'], d['config']['loss_last_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run start_1 FPD_TEACHER_WAIT=start
run loss_1 ""
run start_2 FPD_TEACHER_WAIT=start
run loss_2 ""
run start_3 FPD_TEACHER_WAIT=start
run loss_3 ""
run loss_cap160 "FPD_BNECK_BLOCKS=160"
run loss_cap96 "FPD_BNECK_BLOCKS=96"
run loss_cap64 "FPD_BNECK_BLOCKS=64"
