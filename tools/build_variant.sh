#!/usr/bin/env bash
# Builds a variant of libfpd_amd.so into build_ab/<name>/ with extra compiler flags (same-box A/B through FPD_AMD_LIB):
#   tools/build_variant.sh r8 -DFPD_STATS_REPLICAS=8
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
OUT="$ROOT/build_ab/$NAME"; mkdir -p "$OUT"
SRC="$ROOT/fast-human-pose-estimation.pytorch_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I${ROOT}/include -I${SRC} -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed $*"
FILES="conv_tile conv_pp conv_tile_f8 conv_smallc bneck_fused head_fused wgrad_tile wgrad3 conv_mfma conv_naive stem stem_mfma stem_s2d elementwise loss_adam pck infer data api"
pids=()
for f in $FILES; do ( hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/$f.o" ) & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
OBJS=""; for f in $FILES; do OBJS="$OBJS $OUT/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfpd_amd.so" $OBJS
rm -f "$OUT"/*.o
echo "built $OUT/libfpd_amd.so"
