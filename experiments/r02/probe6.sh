#!/usr/bin/env bash
# round-2 probe 6: new validate / data-pipeline / fp8 tests, the FPD_CONV_OCC (4 blocks per CU) experiment, grid-cap re-sweep,
# fp8 A/B at configs[4] shapes.  One box, everything A/B'd on it.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p6; mkdir -p $O
( timeout 600 python -m pytest tests/test_infer_gpu.py tests/test_fp8_gpu.py -m gpu -q -p no:cacheprovider -x -s > $O/tests_new.log 2>&1; echo "rc=$?" >> $O/tests_new.log )
tail -5 $O/tests_new.log
( FPD_CONV_OCC=1 timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py -m gpu -q -p no:cacheprovider -k "conv_forward or dgrad or exact" > $O/tests_occ.log 2>&1; echo "rc=$?" >> $O/tests_occ.log )
tail -3 $O/tests_occ.log
b() { # name, env...
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', d['ms_per_step'], d['roofline'].get('avg_us'))" || tail -3 $O/bench_$name.err
}
b base X=1
b occ512 FPD_CONV_OCC=512
b occ1024 FPD_CONV_OCC=1024
b base2 X=1
b bneck192 FPD_BNECK_BLOCKS=192
b bneck128 FPD_BNECK_BLOCKS=128
b wb48 FPD_WGRAD_BATCH=48
for v in "" "--no-fp8"; do
  timeout 300 python bench.py --config hrnet_fp8 --steps 8 --warmup 3 $v > $O/bench_f8$v.json 2> $O/bench_f8$v.err
  python -c "import json;d=json.load(open('$O/bench_f8$v.json'));print('hrnet_fp8 $v', d['ms_per_step'], d['value'])" || tail -5 $O/bench_f8$v.err
done
