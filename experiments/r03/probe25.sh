#!/usr/bin/env bash
# round-3 probe 25: BN-backward applies folded into the consuming data gradient (FPD_FOLD_APPLY=1, default) vs not
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p25; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "folded or conv_pp or pair or dgrad or conv_forward" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -5 $O/tests.log | cut -c1-300
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x > $O/tests_model.log 2>&1; echo "rc=$?" >> $O/tests_model.log ); tail -5 $O/tests_model.log | cut -c1-300
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  loss %s launches %s' % ('$1', d['ms_per_step'], d['config']['loss_last_step'], d['config']['launches_per_step']['total']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run nofold_1 FPD_FOLD_APPLY=0
run fold_1 ""
run nofold_2 FPD_FOLD_APPLY=0
run fold_2 ""
run nofold_3 FPD_FOLD_APPLY=0
run fold_3 ""
run fold_pp256 "FPD_CONV_PP_BLOCKS=256"
run nofold_pp256 "FPD_FOLD_APPLY=0 FPD_CONV_PP_BLOCKS=256"
