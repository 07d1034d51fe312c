#!/usr/bin/env bash
# round-3 probe 23: after the teacher wait moved in front of the loss: model tests + the what-if table again
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p23; mkdir -p $O
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_entry_gpu.py -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -3 $O/tests.log | cut -c1-200
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run base ""
run nowgrad FPD_WHATIF=nowgrad
run nobigconv FPD_WHATIF=nobigconv
run noew FPD_WHATIF=noew
run noapply FPD_WHATIF=noapply
run nosmall FPD_WHATIF=nosmall
run t_big FPD_WHATIF=t_big
run student_alone FPD_WHATIF=t_all
run base2 ""
run pp256 FPD_CONV_PP_BLOCKS=256
run pp192 FPD_CONV_PP_BLOCKS=192
run wb4 FPD_WGRAD_BATCH=4
run wb16 FPD_WGRAD_BATCH=16
