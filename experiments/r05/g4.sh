#!/usr/bin/env bash
# round 5 call 4: wgrad3 v4 (256-pixel tiles, software-pipelined tap rows, staging between the MFMA groups, bias on the matrix pipe)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g4; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "wgrad" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -4 | cut -c1-300
timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3" --iters 30 2>&1 | tail -5
FPD_WGRAD3_RANGES=32 timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3 64>64 @64" --iters 30 2>&1 | tail -1
FPD_WGRAD3_RANGES=16 timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3 64>64 @64" --iters 30 2>&1 | tail -1
echo "== stamps"
FPD_AMD_LIB=$PWD/build_ab/w3t/libfpd_amd.so timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3 64>64 @64" --iters 3 2>&1 | grep stamps | tail -3 | tee $O/stamps64.txt
FPD_AMD_LIB=$PWD/build_ab/w3t/libfpd_amd.so timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3 64>64 @16" --iters 3 2>&1 | grep stamps | tail -2 | tee $O/stamps16.txt
