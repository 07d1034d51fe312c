#!/usr/bin/env bash
# round-3 probe 16: slab reduction per weight-gradient batch (FPD_WREDUCE_PER_BATCH=1, new default) vs per bucket; batch sizes
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p16; mkdir -p $O
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -3 $O/tests.log
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  loss %s  launches %s' % ('$1', d['ms_per_step'], d['config']['loss_last_step'], d['config']['launches_per_step']['total']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run bucket_1 FPD_WREDUCE_PER_BATCH=0
run batch_1 ""
run bucket_2 FPD_WREDUCE_PER_BATCH=0
run batch_2 ""
run bucket_3 FPD_WREDUCE_PER_BATCH=0
run batch_3 ""
run batch_wb4 "FPD_WGRAD_BATCH=4"
run batch_wb6 "FPD_WGRAD_BATCH=6"
run batch_wb12 "FPD_WGRAD_BATCH=12"
run batch_wb4b "FPD_WGRAD_BATCH=4"
