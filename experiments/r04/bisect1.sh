#!/usr/bin/env bash
# which change moved the bf16 pin / the fp32 training curve: V0 = library of commit c491dd8, V1 = current with the conv_tile / shared epilogue of c491dd8, V2 = current
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04bis; mkdir -p $O
T1="tests/test_bf16_parity_gpu.py::test_trained_pair_bf16_vs_fp64_absolute_and_vs_reference_at_bf16"
T4="tests/test_hrnet_gpu.py::test_hrnet_w32_w48_bf16_step_runs_at_coco_shape"
T3="tests/test_fullsize_gpu.py::test_full_architecture_trained_pair_bf16_vs_fp64"
T2="tests/test_bf16_parity_gpu.py::test_bf16_training_converges_like_fp32"
run() { n=$1; shift; timeout 600 python -m pytest "$@" -q --tb=line -p no:cacheprovider --durations=5 > $O/$n.txt 2>&1; echo "== $n rc=$?"; grep -E "passed|failed|Error|slowest|s call" $O/$n.txt | cut -c1-250 | head -12; }
FPD_AMD_LIB=$PWD/build_ab/libfpd_amd_prev.so run v0 $T1 $T4
FPD_AMD_LIB=$PWD/build_ab/v1_oldtile.so run v1 $T1 $T4 $T3 $T2
run v2 $T1 $T4
