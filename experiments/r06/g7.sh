#!/usr/bin/env bash
# round 6, call 7: default bench line (phase times), the what-if table with conv_c1 in, and the kernel's pixel threshold
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06whatif; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --no-parity > $O/bench_line.json 2> $O/bench.err; tail -c 3000 $O/bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('phase_times')); print(d['roofline'].get('conv_classes'))"
run() {  # name, FPD_WHATIF value, env
  FPD_WHATIF="$2" timeout 200 env $3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]);print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run base ""
run minpx8192 "" FPD_C1_MIN_PX=8192
run minpx2048 "" FPD_C1_MIN_PX=2048
run minpx512 "" FPD_C1_MIN_PX=512
run nowgrad nowgrad
run student_alone t_all
run t_nobig t_big
run t_bneck_big t_bneck_big
run t_head t_head
run t_plain_big t_plain_big
run nobigconv nobigconv
run no3x3big no3x3big
run no1x1big no1x1big
run nobig nobig
run nomid nomid
run nosmall nosmall
run noapply noapply
run noew noew
run base2 ""
run minpx8192b "" FPD_C1_MIN_PX=8192
} | tee $O/summary.txt
