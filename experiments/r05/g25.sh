#!/usr/bin/env bash
# round 5 call 25: do the scheduling defaults of rounds 2-4 still hold with the round-5 kernels?  (env knobs only, two interleaved sweeps)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g25; mkdir -p $O
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times"
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2; do
  echo "rep $rep default: $($B 2>/dev/null | ms)" | tee -a $O/knobs.txt
  for v in 4 6 12 16; do echo "rep $rep FPD_WGRAD_BATCH=$v: $(FPD_WGRAD_BATCH=$v $B 2>/dev/null | ms)" | tee -a $O/knobs.txt; done
  for v in 192 320; do echo "rep $rep FPD_CONV_PP_BLOCKS=$v: $(FPD_CONV_PP_BLOCKS=$v $B 2>/dev/null | ms)" | tee -a $O/knobs.txt; done
  for v in 192 320; do echo "rep $rep FPD_CONV_PP_BLOCKS_BWD=$v: $(FPD_CONV_PP_BLOCKS_BWD=$v $B 2>/dev/null | ms)" | tee -a $O/knobs.txt; done
  for v in 128 512; do echo "rep $rep FPD_CONV_PP_MIN_TILES=$v: $(FPD_CONV_PP_MIN_TILES=$v $B 2>/dev/null | ms)" | tee -a $O/knobs.txt; done
  for v in 192 384; do echo "rep $rep FPD_WGRAD_BLOCKS_1=$v: $(FPD_WGRAD_BLOCKS_1=$v $B 2>/dev/null | ms)" | tee -a $O/knobs.txt; done
  for v in bucket batch; do echo "rep $rep FPD_WREDUCE_MODE=$v: $(FPD_WREDUCE_MODE=$v $B 2>/dev/null | ms)" | tee -a $O/knobs.txt; done
  echo "rep $rep FPD_TEACHER_WAIT=start: $(FPD_TEACHER_WAIT=start $B 2>/dev/null | ms)" | tee -a $O/knobs.txt
  echo "rep $rep FPD_STATS... default again: $($B 2>/dev/null | ms)" | tee -a $O/knobs.txt
done
