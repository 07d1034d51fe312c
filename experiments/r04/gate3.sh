#!/usr/bin/env bash
# round-4 gate 3: the tests gate 2 did not reach (after the bn_coef_eval work-around), the new full-size trained pair (writes its
# golden curve), and the shape-keyed profile of the step
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g3; mkdir -p $O
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_exact_gpu.py::test_bottleneck_fused_exact tests/test_model_gpu.py tests/test_entry_gpu.py -q --tb=short -p no:cacheprovider > $O/a.txt 2>&1; echo "a rc=$?"; grep -v "^  File\|^Thread" $O/a.txt | tail -25 | cut -c1-300
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "bottleneck_fused or head or loss or pair or adam" > $O/b.txt 2>&1; echo "b rc=$?"; grep -v "^  File\|^Thread" $O/b.txt | tail -12 | cut -c1-300
FPD_WRITE_TRAINED_FULL=$PWD/$O/trained_full_curve.npz timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -s --tb=short -p no:cacheprovider > $O/c.txt 2>&1; echo "c rc=$?"; grep -v "^  File\|^Thread" $O/c.txt | tail -40 | cut -c1-300
timeout 900 bash tools/profile.sh r04a > $O/profile.txt 2>&1; tail -15 $O/profile.txt | cut -c1-300
cp gpurun_out/r04aprof/r04a_* $O/ 2>/dev/null
ls gpurun_out/r04aprof | head -30
