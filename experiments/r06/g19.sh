#!/usr/bin/env bash
# round 6, call 19: phase-aligned co-scheduling of teacher and student (FPD_ZIPPER=1: the teacher's big runs wait for the student's small windows)
mkdir -p gpurun_out
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g19_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')" || tail -5 gpurun_out/g19_err.txt; }
for i in 1 2; do
  run base ""
  run zipper "FPD_ZIPPER=1"
  run zipper_b192 "FPD_ZIPPER=1 FPD_BNECK_BLOCKS=192"
  run zipper_b256 "FPD_ZIPPER=1 FPD_BNECK_BLOCKS=256"
  run zipper_b256_h256 "FPD_ZIPPER=1 FPD_BNECK_BLOCKS=256 FPD_HEAD_BLOCKS=256"
done | tee gpurun_out/g19_zipper.txt
