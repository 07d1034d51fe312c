#!/usr/bin/env bash
# round-4 gate 6: conv_pp without row arithmetic for 1x1 (bitwise tests), two-shard tests, A/B: which data gradients carry their weight gradient
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g6; mkdir -p $O
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_model_gpu.py -q --tb=short -p no:cacheprovider > $O/a.txt 2>&1; echo "a rc=$?"; grep -v "^  File\|^Thread" $O/a.txt | tail -12 | cut -c1-300
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py -q --tb=short -p no:cacheprovider -k "pp or conv_forward or dgrad or fold or pair or bitwise" > $O/b.txt 2>&1; echo "b rc=$?"; grep -v "^  File\|^Thread" $O/b.txt | tail -8 | cut -c1-300
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-34s %7.3f ms/step  launches %d' % ('$1', d['ms_per_step'], d['config']['launches_per_step']['total']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run default ""
run ck8192 "FPD_CONV_PP_WGRAD_CKMAX=8192"
run kmax64 "FPD_CONV_PP_WGRAD_KMAX=64"
run default2 ""
run ck8192b "FPD_CONV_PP_WGRAD_CKMAX=8192"
run batch4 "FPD_WGRAD_BATCH=4"
run batch12 "FPD_WGRAD_BATCH=12"
