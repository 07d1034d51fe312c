#!/usr/bin/env bash
# Cross-compiles the gfx950 kernels + C ABI into libfpd_amd.so next to this script (no GPU needed).
set -euo pipefail
cd "$(dirname "$0")"
ROOT="$(cd ../.. && pwd)"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I${ROOT}/include -munsafe-fp-atomics -Wno-unused-result"
pids=()
for f in conv_tile conv_pp conv_c1 conv_c3 conv_tile_f8 conv_smallc bneck_fused head_fused wgrad_tile wgrad3 conv_mfma conv_naive stem stem_mfma stem_s2d elementwise loss_adam pck infer data api; do
  ( hipcc $FLAGS -c $f.hip -o $f.o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libfpd_amd.so conv_tile.o conv_pp.o conv_c1.o conv_c3.o conv_tile_f8.o conv_smallc.o bneck_fused.o head_fused.o wgrad_tile.o wgrad3.o conv_mfma.o conv_naive.o stem.o stem_mfma.o stem_s2d.o elementwise.o loss_adam.o pck.o infer.o data.o api.o
echo "built $(pwd)/libfpd_amd.so"
