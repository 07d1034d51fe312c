#!/usr/bin/env bash
# round 5 call 26: grid cap of the elementwise ops that end in statistics atomics (512 since round 1: 16 dependent iterations with one
# vector in flight per thread at 128x128) -- build_ab/ewb + FPD_EW_STATS_BLOCKS, kernel times from traces + interleaved step A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g26; mkdir -p $O
L=$PWD/build_ab/ewb/libfpd_amd.so
R=$PWD
for v in 512 1024 2048; do
  (cd /tmp && export TMPDIR=/tmp && FPD_AMD_LIB=$L FPD_EW_STATS_BLOCKS=$v rocprofv3 --kernel-trace --stats -f csv -d $R/$O/tr$v -o b -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-phase-times > $R/$O/tr$v.log 2>&1)
  echo "== cap $v"; grep -h "ew_kernel<unsigned short, [0135]>\|ew_pair" $O/tr$v/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done 2>&1 | tee $O/kernels.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times"
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in 512 1024 2048; do echo "rep $rep cap $v: $(FPD_AMD_LIB=$L FPD_EW_STATS_BLOCKS=$v $B 2>/dev/null | ms)" | tee -a $O/ab.txt; done
done
FPD_AMD_LIB=$L FPD_EW_STATS_BLOCKS=2048 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k "ew or elementwise or stat" 2>&1 | tail -2
