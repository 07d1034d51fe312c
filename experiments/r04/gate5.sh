#!/usr/bin/env bash
# round-4 gate 5: the library side of the backward-tail changes (512-block SMALL weight gradient, two stem-wgrad blocks per CU,
# conv_tile 32-channel chunks on 128-wide rows) + grid of the conv_pp data-gradient kernels: tests, then same-box A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g5; mkdir -p $O
timeout 900 python -m pytest tests/test_dist_gpu.py -q --tb=short -p no:cacheprovider > $O/a.txt 2>&1; echo "a rc=$?"; grep -v "^  File\|^Thread" $O/a.txt | tail -15 | cut -c1-300
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py -q --tb=short -p no:cacheprovider -k "conv_forward or wgrad or stem" > $O/b.txt 2>&1; echo "b rc=$?"; grep -v "^  File\|^Thread" $O/b.txt | tail -8 | cut -c1-300
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-34s %7.3f ms/step  launches %d' % ('$1', d['ms_per_step'], d['config']['launches_per_step']['total']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run r3like "FPD_WREDUCE_MODE=bucket FPD_WGRAD_BLOCKS_3=128"
run default ""
run bwd128 "FPD_CONV_PP_BLOCKS_BWD=128"
run bwd192 "FPD_CONV_PP_BLOCKS_BWD=192"
run bwd384 "FPD_CONV_PP_BLOCKS_BWD=384"
run default2 ""
run wg3_256 "FPD_WGRAD_BLOCKS_3=256"
run r3like2 "FPD_WREDUCE_MODE=bucket FPD_WGRAD_BLOCKS_3=128"
timeout 300 python tools/probes/phase_times.py > $O/phase_times.txt 2>&1; tail -9 $O/phase_times.txt
