#!/usr/bin/env bash
# round-2 probe 23: schedule knobs re-swept on top of WGRAD_BATCH=8 (same box, base interleaved)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p23; mkdir -p $O
b() { local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/b_$name.json 2> $O/b_$name.err
  python -c "import json;d=json.load(open('$O/b_$name.json'));print('$name', d['ms_per_step'])" || tail -3 $O/b_$name.err
}
b base X=1
b wb1_128 FPD_WGRAD_BLOCKS_1=128
b wb1_192 FPD_WGRAD_BLOCKS_1=192
b wb3_64 FPD_WGRAD_BLOCKS_3=64
b wb3_192 FPD_WGRAD_BLOCKS_3=192
b base2 X=1
b bneck128 FPD_BNECK_BLOCKS=128
b bneck192 FPD_BNECK_BLOCKS=192
b head128 FPD_HEAD_BLOCKS=128
b reuse100 FPD_REUSE_DELAY=100
b reuse1000 FPD_REUSE_DELAY=1000
b base3 X=1
