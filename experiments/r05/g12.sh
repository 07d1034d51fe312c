#!/usr/bin/env bash
# round 5 call 12: grid sizes of the other lane / teacher kernels now that the 3x3 weight gradients take 128 blocks (same-box sweeps, interleaved)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g12; mkdir -p $O
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  run base_$i
  FPD_WGRAD_BLOCKS_1=128 run w1_128_$i
  FPD_WGRAD_BLOCKS_1=64 run w1_64_$i
  FPD_BNECK_BLOCKS=112 run bn112_$i
  FPD_BNECK_BLOCKS=144 run bn144_$i
  FPD_WGRAD_BATCH=12 run wb12_$i
  FPD_CONV_PP_BLOCKS_BWD=224 run ppb224_$i
done
