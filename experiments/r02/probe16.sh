#!/usr/bin/env bash
# round-2 probe 16: phase A staging sliced between the MFMAs (variant library `slices`) vs the in-tree build, same box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p16; mkdir -p $O
V=${VARIANT:-slices}
{
echo "== stamps: $V"; ONLY=64 FPD_AMD_LIB=$PWD/build_ab/${V}_t/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep -A1 "bneck W" | tail -2
echo "== bench base"; python tools/bneck_bench.py 2>&1 | grep fused; FPD_BNECK_BLOCKS=256 ONLY=64 python tools/bneck_bench.py 2>&1 | grep fused
echo "== bench $V"; FPD_AMD_LIB=$PWD/build_ab/$V/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep fused; FPD_BNECK_BLOCKS=256 ONLY=64 FPD_AMD_LIB=$PWD/build_ab/$V/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep fused
echo "== P=64 $V"; P=64 FPD_AMD_LIB=$PWD/build_ab/$V/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep fused | head -3
} | tee $O/bneck.txt
( FPD_AMD_LIB=$PWD/build_ab/$V/libfpd_amd.so timeout 300 python -m pytest tests/test_exact_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "bottleneck" 2>&1 | tail -2 ) | tee $O/tests.txt
b() { local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', d['ms_per_step'], d['roofline'].get('avg_us'))" || tail -3 $O/bench_$name.err
}
b base X=1
b $V FPD_AMD_LIB=$PWD/build_ab/$V/libfpd_amd.so
b base2 X=1
b ${V}2 FPD_AMD_LIB=$PWD/build_ab/$V/libfpd_amd.so
