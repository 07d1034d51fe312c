#!/usr/bin/env bash
# conv_tile small-launch trims (all nine taps at once, branch-free halo requests, barrier-free wave epilogue): tests, stamps, A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g11; mkdir -p $O
timeout 700 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -12 | cut -c1-300
for i in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export FPD_AMD_LIB=$PWD/build_ab/libfpd_amd_prev.so; else unset FPD_AMD_LIB; fi
    timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$v$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v$i', d['ms_per_step'])"
  done
done
unset FPD_AMD_LIB
FPD_AMD_LIB=$PWD/build_ab/tiletime/libfpd_amd.so timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity 2>/dev/null | grep "^conv_tile" > $O/step_stamps_all.txt
timeout 600 python tools/trace_ab.py $O build_ab/libfpd_amd_prev.so - conv_tile > $O/ab.txt 2>&1; head -50 $O/ab.txt | cut -c1-160; tail -1 $O/ab.txt
rm -rf $O/A $O/B
