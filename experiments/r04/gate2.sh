#!/usr/bin/env bash
# round-4 gate 2: debug the two failures of gate 1 (dist tests, fused-Bottleneck exact test abort), then the rest of the suite
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g2; mkdir -p $O
timeout 600 python -m pytest tests/test_dist_gpu.py -q --tb=short -p no:cacheprovider > $O/dist.txt 2>&1; tail -60 $O/dist.txt
AMD_LOG_LEVEL=1 timeout 300 python -m pytest "tests/test_exact_gpu.py::test_bottleneck_fused_exact" -x -q -p no:cacheprovider > $O/bneck.txt 2>&1; grep -v "^  File\|^Thread\|^$" $O/bneck.txt | tail -30
for t in "case0-True" "case1-False" "case2-False" "case3-False"; do
  timeout 120 python -m pytest "tests/test_exact_gpu.py::test_bottleneck_fused_exact[$t]" -x -q -p no:cacheprovider > $O/bneck_$t.txt 2>&1; echo "$t rc=$?"
done
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --deselect tests/test_dist_gpu.py --deselect tests/test_exact_gpu.py::test_bottleneck_fused_exact > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -40
