#!/usr/bin/env bash
# round 5 call 22: loss kernel with several tiles per block (512 instead of 2 048 same-address fp64 atomics)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g22; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_hrnet_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "loss or fused_step or hrnet" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -3 | cut -c1-300
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$O/tr -o b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-phase-times > $R/$O/tr.log 2>&1; cd $R
grep -h "wprep\|adam_kernel\|loss_vec" $O/tr/*kernel_stats.csv | cut -c1-200
