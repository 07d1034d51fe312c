#!/usr/bin/env bash
# round-2 probe 10: statistics replicas R = 4 / 8 / 16 (variant libraries, same box): step time and the 64x64 student convs
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p10; mkdir -p $O
b() { local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', d['ms_per_step'])" || tail -3 $O/bench_$name.err
}
b r4 X=1
b r8 FPD_STATS_REPLICAS=8 FPD_AMD_LIB=$PWD/build_ab/r8/libfpd_amd.so
b r16 FPD_STATS_REPLICAS=16 FPD_AMD_LIB=$PWD/build_ab/r16/libfpd_amd.so
b r4b X=1
b r8b FPD_STATS_REPLICAS=8 FPD_AMD_LIB=$PWD/build_ab/r8/libfpd_amd.so
for r in 4 8 16; do
  echo "== R=$r"
  if [ $r = 4 ]; then python tools/conv_bench.py --only "s " --graph --iters 20 2>&1 | grep "@64\|@32"
  else FPD_STATS_REPLICAS=$r FPD_AMD_LIB=$PWD/build_ab/r$r/libfpd_amd.so python tools/conv_bench.py --only "s " --graph --iters 20 2>&1 | grep "@64\|@32"; fi
done | tee $O/conv_R.txt
