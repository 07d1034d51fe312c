#!/usr/bin/env bash
# round-4 gate 8: packed-pair arithmetic + peeled first tile in conv_pp -- bitwise tests, then an interleaved A/B against the previous library
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g8; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py tests/test_entry_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -15 | cut -c1-300
for i in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export FPD_AMD_LIB=$PWD/build_ab/libfpd_amd_prev.so; else unset FPD_AMD_LIB; fi
    timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$v$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v$i', d['ms_per_step'], d.get('phases'))" | cut -c1-400
  done
done
unset FPD_AMD_LIB
timeout 300 python tools/conv_bench.py > $O/conv_bench_new.txt 2>&1; FPD_AMD_LIB=$PWD/build_ab/libfpd_amd_prev.so timeout 300 python tools/conv_bench.py > $O/conv_bench_prev.txt 2>&1
paste $O/conv_bench_prev.txt $O/conv_bench_new.txt | head -40 | cut -c1-260
