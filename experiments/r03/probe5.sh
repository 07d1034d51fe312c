#!/usr/bin/env bash
# round-3 probe 5: conv_pp grid / occupancy / threshold sweep inside the pipelined step, interleaved with the conv_tile baseline
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p5; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
for rep in 1 2; do
run base_$rep FPD_CONV_PP=0
run b256_o2_$rep "FPD_CONV_PP=1"
run b128_o2_$rep "FPD_CONV_PP=1 FPD_CONV_PP_BLOCKS=128"
run b96_o2_$rep "FPD_CONV_PP=1 FPD_CONV_PP_BLOCKS=96"
run b64_o2_$rep "FPD_CONV_PP=1 FPD_CONV_PP_BLOCKS=64"
run b256_o1_$rep "FPD_CONV_PP=1 FPD_CONV_PP_OCC=1"
run b192_o1_$rep "FPD_CONV_PP=1 FPD_CONV_PP_OCC=1 FPD_CONV_PP_BLOCKS=192"
run b128_o1_$rep "FPD_CONV_PP=1 FPD_CONV_PP_OCC=1 FPD_CONV_PP_BLOCKS=128"
run b128_o2_t256_$rep "FPD_CONV_PP=1 FPD_CONV_PP_BLOCKS=128 FPD_CONV_PP_MIN_TILES=256"
run b128_o2_t128_$rep "FPD_CONV_PP=1 FPD_CONV_PP_BLOCKS=128 FPD_CONV_PP_MIN_TILES=128"
done | tee $O/summary.txt
