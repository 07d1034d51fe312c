#!/usr/bin/env bash
# Phase stamps of conv_tile on the low-resolution layer shapes (where a launch is all latency): builds the library with
# -DFPD_TILE_TIMING into build_ab/tiletime (CPU, ~3 min) unless it is there, then runs tools/conv_bench.py on the small shapes.
#   bash tools/probes/tile_timing.sh [out file]
cd "$(dirname "$0")/../.." || exit 1
D=build_ab/tiletime; C=fast-human-pose-estimation.pytorch_amd/csrc
if [ ! -f $D/libfpd_amd.so ]; then
  mkdir -p $D
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -munsafe-fp-atomics -Wno-unused-result -DFPD_TILE_TIMING -c $C/conv_tile.hip -o $D/conv_tile.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libfpd_amd.so $D/conv_tile.o $(ls $C/*.o | grep -v conv_tile.o)
fi
OUT=${1:-/dev/stdout}
for s in "@16" "@8" "@4"; do
  FPD_AMD_LIB=$PWD/$D/libfpd_amd.so python tools/conv_bench.py --iters 2 --only "$s" 2>&1 | grep -v amdgpu.ids
done > $OUT
