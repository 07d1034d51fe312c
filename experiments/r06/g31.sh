#!/usr/bin/env bash
# round 6, call 31: the fused head's weight chunks by LDS-DMA (build_ab/hdma) -- tests, step A/B, the kernel's time inside the step from two traces
mkdir -p gpurun_out
V=$PWD/build_ab/hdma/libfpd_amd.so
FPD_AMD_LIB=$V timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py -q -x -k "head or fused_step or module_api" -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 | tee gpurun_out/g31_tests.txt
run() { env $2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2>gpurun_out/g31_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-26s' % '$1', d['ms_per_step'], 'ms/step')" || tail -5 gpurun_out/g31_err.txt; }
for i in 1 2 3; do
  run base ""
  run head_dma "FPD_AMD_LIB=$V"
done | tee gpurun_out/g31_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in base new; do
  L=""; [ $v = new ] && L=$V
  FPD_AMD_LIB=$L rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/g31_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-phase-times > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/g31_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep "head_eval" $f | cut -c1-160
done | tee $GRAFT_REPO_ROOT/gpurun_out/g31_trace.txt
