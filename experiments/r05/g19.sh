#!/usr/bin/env bash
# round 5 call 19: stem weight gradient in the space-to-depth form (stem_s2d_wgrad) vs the im2col GEMM
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g19; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "stem" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -6 | cut -c1-300
echo "== im2col"; FPD_STEM_S2D=0 timeout 200 python tools/stem_bench.py 2>&1 | tail -3
echo "== s2d"; timeout 200 python tools/stem_bench.py 2>&1 | tail -3
echo "== s2d wgrad 128 / 512 blocks"; FPD_STEM_WGRAD_BLOCKS=128 timeout 200 python tools/stem_bench.py 2>&1 | grep "weight"; FPD_STEM_WGRAD_BLOCKS=512 timeout 200 python tools/stem_bench.py 2>&1 | grep "weight"
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_exact_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider > $O/pytest2.txt 2>&1; echo "pytest2 rc=$?" >> $O/pytest2.txt; tail -3 $O/pytest2.txt | cut -c1-200
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  FPD_STEM_S2D=0 run old_$i
  run s2d_$i
done
