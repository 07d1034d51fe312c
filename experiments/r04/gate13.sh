#!/usr/bin/env bash
# conv_tile small-launch trims, knobs separated: prev | new (private staging, block-level stats) | new with aliased staging
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g13; mkdir -p $O
timeout 700 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -4 | cut -c1-300
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  FPD_AMD_LIB=$PWD/build_ab/libfpd_amd_prev.so run prev$i
  run new$i
done
timeout 600 python tools/trace_ab.py $O build_ab/libfpd_amd_prev.so - conv_tile > $O/ab.txt 2>&1; head -40 $O/ab.txt | cut -c1-160; tail -1 $O/ab.txt
rm -rf $O/A $O/B
