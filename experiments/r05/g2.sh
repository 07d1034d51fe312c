#!/usr/bin/env bash
# round 5 call 2: wgrad3 v2 (512 threads, one barrier per tile, register prefetch depth 1/2/3)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "wgrad" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -4 | cut -c1-300
for v in d1 "" d3; do
  echo "== depth variant '$v'"; L=""; [ -n "$v" ] && L=$PWD/build_ab/$v/libfpd_amd.so
  FPD_AMD_LIB=$L timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3" --iters 30 2>&1 | tail -5
  FPD_AMD_LIB=$L FPD_WGRAD3_RANGES=32 timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3 64>64 @64" --iters 30 2>&1 | tail -1
  FPD_AMD_LIB=$L FPD_WGRAD3_RANGES=16 timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3 64>64 @64" --iters 30 2>&1 | tail -1
done
echo "== min tiles 2"; FPD_WGRAD3_MIN_TILES=2 timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3" --iters 30 2>&1 | tail -5
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config'].get('launches_per_step',{}).get('total'))"; }
for i in 1 2 3; do
  FPD_WGRAD3=0 run old$i
  run new$i
done
