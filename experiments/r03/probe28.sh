#!/usr/bin/env bash
# round-3 probe 28: knobs that did nothing while the two lanes were coupled, again
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p28; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run base ""
run wreduce_batch FPD_WREDUCE_PER_BATCH=1
run reuse1000 FPD_REUSE_DELAY=1000
run mintiles128 FPD_CONV_PP_MIN_TILES=128
run base2 ""
run wg3_96 FPD_WGRAD_BLOCKS_3=96
run wg1_128 FPD_WGRAD_BLOCKS_1=128
run pairs0 FPD_PAIR=0
run graphs4 "GPU_MAX_HW_QUEUES=4" 
