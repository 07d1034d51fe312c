#!/usr/bin/env bash
# round-3 probe 10: teacher grid cap sweep with the ring Bottleneck
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p10; mkdir -p $O
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  bneck64 %.1f us' % ('$1', d['ms_per_step'], d['roofline']['avg_us']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run cap128 ""
for cap in 64 80 96 112; do run cap$cap "FPD_BNECK_BLOCKS=$cap"; done
run cap96_head96 "FPD_BNECK_BLOCKS=96 FPD_HEAD_BLOCKS=96"
run cap112_head112 "FPD_BNECK_BLOCKS=112 FPD_HEAD_BLOCKS=112"
run cap128_head128 "FPD_HEAD_BLOCKS=128"
run cap128b ""
} | tee $O/summary.txt
