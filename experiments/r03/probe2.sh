#!/usr/bin/env bash
# round-3 probe 2: first run of the persistent ping-pong convolution (conv_pp) + deterministic weight gradients
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p2; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "conv_pp or repeatable or test_stem or conv_pair or dgrad or conv_forward or conv_wgrad" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -15 $O/tests.log
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  loss %s' % ('$1', d['ms_per_step'], d['config']['loss_last_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run pp0 FPD_CONV_PP=0
run pp1 FPD_CONV_PP=1
run pp0b FPD_CONV_PP=0
run pp1b FPD_CONV_PP=1
run pp1_128 "FPD_CONV_PP=1 FPD_CONV_PP_BLOCKS=128"
run pp1_student "FPD_CONV_PP=1 FPD_WHATIF=t_all"
run pp0_student "FPD_CONV_PP=0 FPD_WHATIF=t_all"
FPD_CONV_PP=1 timeout 300 python tools/conv_bench.py --graph 2>&1 | grep -E "^s |^l1"
FPD_CONV_PP=0 timeout 300 python tools/conv_bench.py --graph 2>&1 | grep -E "^s |^l1"
