#!/usr/bin/env bash
# round-3 probe 7: teacher grid cap x conv_pp grid, pipelined step
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p7; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run base ""
for cap in 96 112 144 160 192; do
run cap$cap "FPD_BNECK_BLOCKS=$cap"
done
run cap160_pp96 "FPD_BNECK_BLOCKS=160 FPD_CONV_PP_BLOCKS=96"
run cap144_pp112 "FPD_BNECK_BLOCKS=144 FPD_CONV_PP_BLOCKS=112"
run cap112_pp144 "FPD_BNECK_BLOCKS=112 FPD_CONV_PP_BLOCKS=144"
run head128 "FPD_HEAD_BLOCKS=128"
run wgb4 "FPD_WGRAD_BATCH=4"
run wgb16 "FPD_WGRAD_BATCH=16"
run base2 ""
} | tee $O/summary.txt
