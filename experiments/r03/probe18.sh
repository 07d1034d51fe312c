#!/usr/bin/env bash
# round-3 probe 18: HRNet step what-if (timing only): what the teacher and the weight-gradient lane cost
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p18; mkdir -p $O
run() {  # name, env
  timeout 300 env $2 python bench.py --config hrnet --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run base ""
run nowgrad FPD_WHATIF=nowgrad
run student_alone FPD_WHATIF=t_all
run student_alone_nowgrad FPD_WHATIF=t_all,nowgrad
run noew FPD_WHATIF=noew
run lanes2 "FPD_SIDE_STREAMS=2 FPD_WGRAD_LANES=2"
