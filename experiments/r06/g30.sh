#!/usr/bin/env bash
# round 6, call 30: statistics-producing elementwise ops with four pixel-vectors fetched ahead per thread -- tests, step A/B against the
# previous library (build_ab/base), per-kernel times from two short traces
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py -q -x -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 | tee gpurun_out/g30_tests.txt
B=$PWD/build_ab/base/libfpd_amd.so
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g30_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')" || tail -5 gpurun_out/g30_err.txt; }
for i in 1 2 3; do
  run base "FPD_AMD_LIB=$B"
  run fetch4 ""
done | tee gpurun_out/g30_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in base new; do
  L=""; [ $v = base ] && L=$B
  FPD_AMD_LIB=$L rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/g30_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-phase-times > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/g30_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep "ew_kernel" $f | cut -d, -f1-4 | cut -c1-120 | head -12
done | tee $GRAFT_REPO_ROOT/gpurun_out/g30_trace.txt
