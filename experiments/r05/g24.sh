#!/usr/bin/env bash
# round 5 call 24: grid caps of the teacher's fused head / Bottleneck kernels, two interleaved sweeps (env knobs only)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g24; mkdir -p $O
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times"
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2; do
  for cap in 160 64 80 96 112 128 144; do echo "rep $rep head cap $cap: $(FPD_HEAD_BLOCKS=$cap $B 2>/dev/null | ms)" | tee -a $O/caps.txt; done
done
for rep in 1 2; do
  for bc in 120 128 136 144; do for hc in 96 128; do echo "rep $rep bneck $bc head $hc: $(FPD_BNECK_BLOCKS=$bc FPD_HEAD_BLOCKS=$hc $B 2>/dev/null | ms)" | tee -a $O/caps.txt; done; done
done
