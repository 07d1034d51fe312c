#!/usr/bin/env bash
# round-3 probe 8: weight gradient of 1x1 convolutions fused into their data-gradient launch (conv_pp WG)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p8; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "conv_pp or conv_pair or dgrad" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -8 $O/tests.log
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_entry_gpu.py tests/test_fullsize_gpu.py -m gpu -q -p no:cacheprovider -x > $O/tests2.log 2>&1; echo "rc=$?" >> $O/tests2.log )
tail -8 $O/tests2.log
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  loss %s  launches %s' % ('$1', d['ms_per_step'], d['config']['loss_last_step'], d['config']['launches_per_step']['total']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run wg0 FPD_CONV_PP_WGRAD=0
run wg1 FPD_CONV_PP_WGRAD=1
run wg0b FPD_CONV_PP_WGRAD=0
run wg1b FPD_CONV_PP_WGRAD=1
run wg1_student "FPD_CONV_PP_WGRAD=1 FPD_WHATIF=t_all"
run wg0_student "FPD_CONV_PP_WGRAD=0 FPD_WHATIF=t_all"
} | tee $O/summary.txt
