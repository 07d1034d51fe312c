#!/usr/bin/env bash
# round-2 probe 17: s_setprio around the MFMA phases B/C of the fused Bottleneck (variant libraries)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p17; mkdir -p $O
{
for v in base prio1 prio3; do
  L=""; [ $v != base ] && L="FPD_AMD_LIB=$PWD/build_ab/$v/libfpd_amd.so"
  echo "== $v: cap 160 / uncapped"
  env $L ONLY=64 python tools/bneck_bench.py 2>&1 | grep fused
  env $L ONLY=64 FPD_BNECK_BLOCKS=256 python tools/bneck_bench.py 2>&1 | grep fused
done
} | tee $O/bneck.txt
b() { local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', d['ms_per_step'], d['roofline'].get('avg_us'))" || tail -3 $O/bench_$name.err
}
b base X=1
b prio1 FPD_AMD_LIB=$PWD/build_ab/prio1/libfpd_amd.so
b prio3 FPD_AMD_LIB=$PWD/build_ab/prio3/libfpd_amd.so
b base2 X=1
