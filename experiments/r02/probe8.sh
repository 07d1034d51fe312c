#!/usr/bin/env bash
# round-2 probe 8: fp8 tests; FPD_CONV_OCC A/B (interleaved, same box)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p8; mkdir -p $O
( timeout 600 python -m pytest tests/test_fp8_gpu.py -m gpu -q -p no:cacheprovider -s > $O/tests_fp8.log 2>&1; echo "rc=$?" >> $O/tests_fp8.log )
grep -E "passed|failed|rel-L2|cosine|Error|assert " $O/tests_fp8.log | head -30
b() { local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', d['ms_per_step'])" || tail -3 $O/bench_$name.err
}
b base X=1
b occ512 FPD_CONV_OCC=512
b base2 X=1
b occ256 FPD_CONV_OCC=256
b occ1024 FPD_CONV_OCC=1024
