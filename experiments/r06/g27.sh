#!/usr/bin/env bash
# round 6, call 27: scheduling knobs of earlier rounds against the round-6 kernels (one box, interleaved)
mkdir -p gpurun_out
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g27_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')" || tail -5 gpurun_out/g27_err.txt; }
for i in 1 2; do
  run base ""
  run wgbatch4 "FPD_WGRAD_BATCH=4"
  run wgbatch12 "FPD_WGRAD_BATCH=12"
  run wgbatch16 "FPD_WGRAD_BATCH=16"
  run wgbatch24 "FPD_WGRAD_BATCH=24"
  run wreduce_bucket "FPD_WREDUCE_MODE=bucket"
  run wreduce_batch "FPD_WREDUCE_MODE=batch"
  run ewstats384 "FPD_EW_STATS_BLOCKS=384"
  run ewstats768 "FPD_EW_STATS_BLOCKS=768"
  run reuse200 "FPD_REUSE_DELAY=200"
  run reuse800 "FPD_REUSE_DELAY=800"
  run pp_blocks192 "FPD_CONV_PP_BLOCKS=192"
done | tee gpurun_out/g27_knobs.txt
