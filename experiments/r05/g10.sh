#!/usr/bin/env bash
# round 5 call 10: wgrad3 on the 128x128 layer (C = K = 32, one piece), conv_tile with 32-bit halo offsets (no spill), bench line with phase times
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g10; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -5 | cut -c1-300
echo "== wgrad kernels"; timeout 300 python tools/conv_bench.py --wgrad --partials --only "3x3" --iters 30 2>&1 | grep "s 3x3\|l1 3x3"
echo "== wgrad kernels, old"; FPD_WGRAD3=0 timeout 300 python tools/conv_bench.py --wgrad --partials --only "l1 3x3" --iters 30 2>&1 | grep "l1 3x3"
echo "== ranges 32"; FPD_WGRAD3_RANGES=32 timeout 300 python tools/conv_bench.py --wgrad --partials --only "3x3" --iters 30 2>&1 | grep "s 3x3\|l1 3x3"
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config'].get('launches_per_step',{}).get('total'))"; }
for i in 1 2 3; do
  FPD_AMD_LIB=$PWD/build_ab/prev/libfpd_amd.so run prev$i
  run new$i
  FPD_WGRAD3_RANGES=32 run r32_$i
done
timeout 300 python bench.py --no-cpu-baseline --no-parity > $O/bench_line.json 2> $O/err_line.txt; python -c "
import json; d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d.get('phase_times')))
for c in d['roofline']['conv_classes']['top']: print(c['kind'], c['conv'], c['us'], c['frac'])"
