#!/usr/bin/env bash
# round-3 probe 27: teacher grid caps again (the teacher has 2.7 ms more slack since the wait moved in front of the loss)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p27; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run cap128 ""
run cap112 "FPD_BNECK_BLOCKS=112 FPD_HEAD_BLOCKS=112"
run cap96 "FPD_BNECK_BLOCKS=96 FPD_HEAD_BLOCKS=96"
run cap80 "FPD_BNECK_BLOCKS=80 FPD_HEAD_BLOCKS=80"
run cap64 "FPD_BNECK_BLOCKS=64 FPD_HEAD_BLOCKS=64"
run cap128_b "" 
run cap96_pp192 "FPD_BNECK_BLOCKS=96 FPD_HEAD_BLOCKS=96 FPD_CONV_PP_BLOCKS=192"
run cap144 "FPD_BNECK_BLOCKS=144"
run wb4 "FPD_WGRAD_BATCH=4"
run wb12 "FPD_WGRAD_BATCH=12"
run wg3_64 "FPD_WGRAD_BLOCKS_3=64"
run wg3_96 "FPD_WGRAD_BLOCKS_3=96"
