#!/usr/bin/env bash
# round-4 gate 4: tail of the backward (slab reduction in front of the batches, 512-block SMALL weight gradient), teacher 3x3 at 128^2 on
# conv_tile with 32-channel chunks, grid of the conv_pp data-gradient kernels: tests, then same-box A/B of the knobs
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g4; mkdir -p $O
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_model_gpu.py -q --tb=short -p no:cacheprovider > $O/a.txt 2>&1; echo "a rc=$?"; grep -v "^  File\|^Thread" $O/a.txt | tail -15 | cut -c1-300
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "conv_forward or wgrad or loss" > $O/b.txt 2>&1; echo "b rc=$?"; grep -v "^  File\|^Thread" $O/b.txt | tail -8 | cut -c1-300
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-34s %7.3f ms/step  launches %d' % ('$1', d['ms_per_step'], d['config']['launches_per_step']['total']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run r3like "FPD_WREDUCE_MODE=bucket FPD_WGRAD_BLOCKS_3=128"
run bucket_small512 "FPD_WREDUCE_MODE=bucket"
run lag_default ""
run batch "FPD_WREDUCE_MODE=batch"
run lag_bwd128 "FPD_CONV_PP_BLOCKS_BWD=128"
run lag_bwd192 "FPD_CONV_PP_BLOCKS_BWD=192"
run lag_default2 ""
run r3like2 "FPD_WREDUCE_MODE=bucket FPD_WGRAD_BLOCKS_3=128"
