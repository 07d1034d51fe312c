#!/usr/bin/env bash
# round-3 probe 6: weight-gradient grid knobs, wgrad per-shape times, HRNet with / without conv_pp
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p6; mkdir -p $O
run() {  # name, env, extra args
  timeout 300 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity $3 > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run base ""
run wg1_512 "FPD_WGRAD_BLOCKS_1=512"
run wg1_128 "FPD_WGRAD_BLOCKS_1=128"
run wg3_256 "FPD_WGRAD_BLOCKS_3=256"
run wg3_64 "FPD_WGRAD_BLOCKS_3=64"
run wg1_512_3_256 "FPD_WGRAD_BLOCKS_1=512 FPD_WGRAD_BLOCKS_3=256"
run base2 ""
run hrnet_pp0 "FPD_CONV_PP=0" "--config hrnet --steps 10"
run hrnet_pp1 "FPD_CONV_PP=1" "--config hrnet --steps 10"
run hrnet_pp0b "FPD_CONV_PP=0" "--config hrnet --steps 10"
run hrnet_pp1b "FPD_CONV_PP=1" "--config hrnet --steps 10"
echo "== conv_bench --wgrad --graph"
timeout 300 python tools/conv_bench.py --wgrad --graph 2>&1 | grep -E "^s |^l1"
} | tee $O/summary.txt
