#!/usr/bin/env bash
# round-3 probe 15: wgrad_tile on HRNet's 24-/12-wide maps (H4 variant): kernel tests, HRNet tests, HRNet step A/B vs build_ab/base
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p15; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "wgrad" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
( timeout 900 python -m pytest tests/test_hrnet_gpu.py -m gpu -q -p no:cacheprovider -x > $O/tests_hr.log 2>&1; echo "rc=$?" >> $O/tests_hr.log ); tail -4 $O/tests_hr.log
run() {  # name, env
  timeout 300 env $2 python bench.py --config hrnet --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  loss %s' % ('$1', d['ms_per_step'], d['config']['loss_last_step'])); print('   ', [(e['kind'], e['conv'], e['launches_per_step'], e['us']) for e in d['roofline']['conv_classes']['top'][:4]])" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
B=FPD_AMD_LIB=$PWD/build_ab/base/libfpd_amd.so
run base_1 $B
run new_1 ""
run base_2 $B
run new_2 ""
