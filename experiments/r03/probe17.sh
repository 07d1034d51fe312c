#!/usr/bin/env bash
# round-3 probe 17: what the weight-gradient lane costs now (what-if: timing only)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p17; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run base ""
run nowgrad FPD_WHATIF=nowgrad
run nowgrad_big FPD_WHATIF=nowgrad_big
run nowgrad_small FPD_WHATIF=nowgrad_small
run base2 ""
run student_alone FPD_WHATIF=t_all
run student_alone_nowgrad FPD_WHATIF=t_all,nowgrad
run student_alone_nowgrad_big FPD_WHATIF=t_all,nowgrad_big
