#!/usr/bin/env bash
# round 5 call 20: wprep through LDS tile transposition (both sides coalesced)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g20; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py tests/test_hrnet_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "dgrad or pair or fused_step or module_api or hrnet or exact" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -4 | cut -c1-300
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$O/tr -o b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-phase-times > $R/$O/tr.log 2>&1; cd $R
grep -h "wprep\|adam_kernel\|loss_vec" $O/tr/*kernel_stats.csv | cut -c1-200
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  FPD_AMD_LIB=$PWD/build_ab/prev/libfpd_amd.so run prev_$i
  run new_$i
done
