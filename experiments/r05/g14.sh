#!/usr/bin/env bash
# round 5 call 14: wgrad3s (specialised waves: 4 multiply, 4 stage) vs the uniform wgrad3 -- tests, kernel times, stamps, step A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g14; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_entry_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "wgrad or launch_log" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -4 | cut -c1-300
for sp in 0 1; do for r in 32 64; do echo "== spec $sp ranges $r"; FPD_WGRAD3_SPEC=$sp FPD_WGRAD3_RANGES=$r timeout 300 python tools/conv_bench.py --wgrad --partials --only "3x3" --iters 30 2>&1 | grep "s 3x3 64>64 @64\|s 3x3 64>64 @32\|s 3x3 64>64 @16\|l1 3x3"; done; done
echo "== stamps"
FPD_AMD_LIB=$PWD/build_ab/w3t/libfpd_amd.so FPD_WGRAD3_RANGES=64 timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3 64>64 @64" --iters 3 2>&1 | grep stamps | tail -2 | tee $O/stamps64.txt
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  FPD_WGRAD3_SPEC=0 run uni_$i
  run spec_$i
  FPD_WGRAD3_RANGES=24 run spec24_$i
done
