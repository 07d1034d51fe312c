#!/usr/bin/env bash
# round-3 probe 29: wgrad_tile (H4 / flat 1x1) also on the 8- and 4-wide maps of the deepest hourglass levels vs build_ab/base
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p29; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "wgrad" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -3 $O/tests.log | cut -c1-200
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  loss %s' % ('$1', d['ms_per_step'], d['config']['loss_last_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
B=FPD_AMD_LIB=$PWD/build_ab/base/libfpd_amd.so
run base_1 $B
run new_1 ""
run base_2 $B
run new_2 ""
run base_3 $B
run new_3 ""
