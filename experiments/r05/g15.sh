#!/usr/bin/env bash
# round 5 call 15: the frozen network's big plain conv_tile launches on a capped persistent grid (FPD_EVAL_CONV_BLOCKS)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g15; mkdir -p $O
FPD_EVAL_CONV_BLOCKS=512 timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k "fused_vs or code_path or teacher" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -3 | cut -c1-300
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  run base_$i
  FPD_EVAL_CONV_BLOCKS=512 run cap512_$i
  FPD_EVAL_CONV_BLOCKS=768 run cap768_$i
  FPD_EVAL_CONV_BLOCKS=256 run cap256_$i
done
