#!/usr/bin/env bash
# round-3 probe 26: the what-if table after the folds
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p26; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run base ""
run noapply FPD_WHATIF=noapply
run noew FPD_WHATIF=noew
run nosmall FPD_WHATIF=nosmall
run nomid FPD_WHATIF=nomid
run nowgrad FPD_WHATIF=nowgrad
run nobigconv FPD_WHATIF=nobigconv
run t_big FPD_WHATIF=t_big
run student_alone FPD_WHATIF=t_all
run base2 ""
timeout 300 python bench.py --config hrnet --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hrnet fold', d['ms_per_step'], d['config']['launches_per_step']['total'])"
FPD_FOLD_APPLY=0 timeout 300 python bench.py --config hrnet --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hrnet nofold', d['ms_per_step'], d['config']['launches_per_step']['total'])"
