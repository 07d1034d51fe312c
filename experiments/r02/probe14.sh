#!/usr/bin/env bash
# round-2 probe 14: phase A of the fused Bottleneck with the MFMAs issued BEFORE the staging of the next chunk (variant
# libraries): stamps, micro-benchmark at all levels, exactness tests, step time -- same box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p14; mkdir -p $O
{
echo "== stamps: base"; ONLY=64 FPD_AMD_LIB=$PWD/build_ab/timing/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep "bneck W" | tail -2
echo "== stamps: mma-first"; ONLY=64 FPD_AMD_LIB=$PWD/build_ab/mmafirst_t/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep "bneck W" | tail -2
echo "== bench base (cap 160, then uncapped)"; python tools/bneck_bench.py 2>&1 | grep fused; FPD_BNECK_BLOCKS=256 ONLY=64 python tools/bneck_bench.py 2>&1 | grep fused
echo "== bench mma-first"; FPD_AMD_LIB=$PWD/build_ab/mmafirst/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep fused; FPD_BNECK_BLOCKS=256 ONLY=64 FPD_AMD_LIB=$PWD/build_ab/mmafirst/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep fused
} | tee $O/bneck.txt
( FPD_AMD_LIB=$PWD/build_ab/mmafirst/libfpd_amd.so timeout 300 python -m pytest tests/test_exact_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "bottleneck" 2>&1 | tail -2 ) | tee $O/tests.txt
b() { local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', d['ms_per_step'], d['roofline'].get('avg_us'))" || tail -3 $O/bench_$name.err
}
b base X=1
b mmafirst FPD_AMD_LIB=$PWD/build_ab/mmafirst/libfpd_amd.so
b base2 X=1
b mmafirst2 FPD_AMD_LIB=$PWD/build_ab/mmafirst/libfpd_amd.so
