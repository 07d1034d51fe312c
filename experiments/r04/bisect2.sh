#!/usr/bin/env bash
# after reverting the epilogue row regrouping: the four bit-sensitive tests on the current library, then a short A/B against the c491dd8 library
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04bis2; mkdir -p $O
T="tests/test_bf16_parity_gpu.py::test_trained_pair_bf16_vs_fp64_absolute_and_vs_reference_at_bf16 tests/test_hrnet_gpu.py::test_hrnet_w32_w48_bf16_step_runs_at_coco_shape tests/test_fullsize_gpu.py::test_full_architecture_trained_pair_bf16_vs_fp64 tests/test_bf16_parity_gpu.py::test_bf16_training_converges_like_fp32"
timeout 900 python -m pytest $T tests/test_kernels_gpu.py tests/test_exact_gpu.py -q --tb=line -p no:cacheprovider > $O/t.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|Error" $O/t.txt | cut -c1-250 | head
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  FPD_AMD_LIB=$PWD/build_ab/libfpd_amd_prev.so run prev$i
  run new$i
done
