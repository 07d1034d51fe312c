#!/usr/bin/env bash
# round-2 probe 18: conv_smallc (C = 3 / 17 direct kernel): kernel tests on all back ends, HRNet model tests, HRNet bench A/B
# against the previous build (build_ab/slices has no conv_smallc)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p18; mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py tests/test_exact_gpu.py -m gpu -q -p no:cacheprovider -k "conv_forward or stride2 or hrnet or exact" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -6 $O/tests.log
for v in new old new old; do
  L=""; [ $v = old ] && L="FPD_AMD_LIB=$PWD/build_ab/slices/libfpd_amd.so"
  env $L timeout 300 python bench.py --config hrnet --steps 10 --warmup 3 > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json;d=json.load(open('$O/bench_$v.json'));print('hrnet $v', d['ms_per_step'], d['value'])" || tail -5 $O/bench_$v.err
done
