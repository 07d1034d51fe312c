#!/usr/bin/env bash
# round-3 probe 24: conv_pp grid sweep again now that the teacher no longer holds the student's forward back
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p24; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run b128 ""
run b256 FPD_CONV_PP_BLOCKS=256
run b384 FPD_CONV_PP_BLOCKS=384
run b512 FPD_CONV_PP_BLOCKS=512
run b128_2 ""
run b256_2 FPD_CONV_PP_BLOCKS=256
run b256_o1 "FPD_CONV_PP_BLOCKS=256 FPD_CONV_PP_OCC=1"
run b256_t128 "FPD_CONV_PP_BLOCKS=256 FPD_CONV_PP_MIN_TILES=128"
run b256_cap112 "FPD_CONV_PP_BLOCKS=256 FPD_BNECK_BLOCKS=112"
run b256_cap144 "FPD_CONV_PP_BLOCKS=256 FPD_BNECK_BLOCKS=144"
run b256_wg256 "FPD_CONV_PP_BLOCKS=256 FPD_WGRAD_BLOCKS_3=256"
run b256_wg64 "FPD_CONV_PP_BLOCKS=256 FPD_WGRAD_BLOCKS_3=64"
