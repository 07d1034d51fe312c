#!/usr/bin/env bash
# wreduce with 16 slab requests in flight (same order of additions): bit-sensitive tests, weight-gradient tests, interleaved A/B vs the final-gate library
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g16; mkdir -p $O
T="tests/test_bf16_parity_gpu.py::test_trained_pair_bf16_vs_fp64_absolute_and_vs_reference_at_bf16 tests/test_fullsize_gpu.py::test_full_architecture_trained_pair_bf16_vs_fp64"
timeout 600 python -m pytest $T tests/test_kernels_gpu.py tests/test_model_gpu.py -q --tb=line -p no:cacheprovider > $O/t.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|Error" $O/t.txt | cut -c1-250 | head
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  FPD_AMD_LIB=$PWD/build_ab/libfpd_amd_fg.so run prev$i
  run new$i
done
timeout 300 python tools/trace_ab.py $O build_ab/libfpd_amd_fg.so - wreduce > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-160
rm -rf $O/A $O/B
