#!/usr/bin/env bash
# conv_pp: packed forward path + peeled first tile, scalar backward epilogue -- bitwise tests, interleaved A/B, per-shape trace A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g10; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -5 | cut -c1-300
for i in 1 2 3 4; do
  for v in prev new; do
    if [ $v = prev ]; then export FPD_AMD_LIB=$PWD/build_ab/libfpd_amd_prev.so; else unset FPD_AMD_LIB; fi
    timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$v$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v$i', d['ms_per_step'])"
  done
done
unset FPD_AMD_LIB
timeout 600 python tools/trace_ab.py $O build_ab/libfpd_amd_prev.so - conv_pp > $O/ab.txt 2>&1; head -45 $O/ab.txt | cut -c1-160; tail -1 $O/ab.txt
rm -rf $O/A $O/B
