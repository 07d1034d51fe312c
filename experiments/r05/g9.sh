#!/usr/bin/env bash
# round 5 call 9: full GPU suite with wgrad3 v7 + the 2^-20 statistics limbs; step A/B old 3x3 weight gradient vs wgrad3
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g9; mkdir -p $O
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config'].get('launches_per_step',{}).get('total'))"; }
for i in 1 2 3; do
  FPD_WGRAD3=0 run old$i
  run new$i
done
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -12 | cut -c1-300
