#!/usr/bin/env bash
# round-2 probe 4: in-step sweeps of the weight-gradient lane's grid caps / batch size and the head kernel cap
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p4; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/$name.json 2> $O/$name.err
  python -c "import json;d=json.load(open('$O/$name.json'));print('$name', d['ms_per_step'], d['roofline']['avg_us'])" || tail -2 $O/$name.err
}
run base A=1
run wg_128_128 FPD_WGRAD_BLOCKS_1=128 FPD_WGRAD_BLOCKS_3=128
run wg_128_64 FPD_WGRAD_BLOCKS_1=128 FPD_WGRAD_BLOCKS_3=64
run wg_64_64 FPD_WGRAD_BLOCKS_1=64 FPD_WGRAD_BLOCKS_3=64
run wg_192_96 FPD_WGRAD_BLOCKS_1=192 FPD_WGRAD_BLOCKS_3=96
run batch12 FPD_WGRAD_BATCH=12
run batch48 FPD_WGRAD_BATCH=48
run head128 FPD_HEAD_BLOCKS=128
run head224 FPD_HEAD_BLOCKS=224
run nolanes FPD_LANES=0
