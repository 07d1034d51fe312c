#!/usr/bin/env bash
# round 5 call 28: bnrelu_bwd_r with the mask source chosen once per pixel (the per-element form compiled to 56 basic blocks / 248
# instructions per vector; now 10 / 120) -- build_ab/ewm vs build_ab/ewb (= in-tree + the grid-cap knob), tests, trace, step A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g28; mkdir -p $O
A=$PWD/build_ab/ewb/libfpd_amd.so; L=$PWD/build_ab/ewm/libfpd_amd.so; R=$PWD
FPD_AMD_LIB=$L timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k "ew or elementwise or bnrelu or fused_step or bit_repeat or oracle" 2>&1 | tail -2
for n in ewb ewm; do
  (cd /tmp && export TMPDIR=/tmp && FPD_AMD_LIB=$R/build_ab/$n/libfpd_amd.so rocprofv3 --kernel-trace --stats -f csv -d $R/$O/tr_$n -o b -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-phase-times > $R/$O/tr_$n.log 2>&1)
  echo "== $n"; grep -h "ew_kernel<unsigned short, 1>" $O/tr_$n/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done 2>&1 | tee $O/kernels.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times"
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2 3 4; do echo "rep $rep base $(FPD_AMD_LIB=$A $B 2>/dev/null | ms)  mask-once $(FPD_AMD_LIB=$L $B 2>/dev/null | ms)" | tee -a $O/ab.txt; done
