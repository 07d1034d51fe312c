#!/usr/bin/env bash
# round 5 call 27: ... and fewer blocks than 512?  (call 26: 1024 / 2048 cost 0.07 / 0.15 ms per step although bnrelu_bwd_r itself gets faster)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g27; mkdir -p $O
L=$PWD/build_ab/ewb/libfpd_amd.so
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times"
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in 512 384 256 192 128; do echo "rep $rep cap $v: $(FPD_AMD_LIB=$L FPD_EW_STATS_BLOCKS=$v $B 2>/dev/null | ms)" | tee -a $O/ab.txt; done
done
R=$PWD
for v in 256; do
  (cd /tmp && export TMPDIR=/tmp && FPD_AMD_LIB=$L FPD_EW_STATS_BLOCKS=$v rocprofv3 --kernel-trace --stats -f csv -d $R/$O/tr$v -o b -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-phase-times > $R/$O/tr$v.log 2>&1)
  echo "== cap $v"; grep -h "ew_kernel<unsigned short, [0135]>\|ew_pair" $O/tr$v/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done 2>&1 | tee $O/kernels.txt
