#!/usr/bin/env bash
# round-3 probe 9: fused Bottleneck with the conv1-row ring (one phase-A pass per tile)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p9; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_exact_gpu.py -m gpu -q -p no:cacheprovider -x -k "bottleneck or bneck or head" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -6 $O/tests.log
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -p no:cacheprovider -x > $O/tests2.log 2>&1; echo "rc=$?" >> $O/tests2.log )
tail -4 $O/tests2.log
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  bneck64 %.1f us (%.3f of peak)' % ('$1', d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run ring ""
run ring_b ""
run ring_cap160 "FPD_BNECK_BLOCKS=160"
run ring_cap256 "FPD_BNECK_BLOCKS=256"
run ring_teacher_only "FPD_WHATIF=nobig,nomid,nosmall,nowgrad"
timeout 120 python tools/bneck_bench.py 2>&1 | tail -12
} | tee $O/summary.txt
