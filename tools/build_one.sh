#!/usr/bin/env bash
# Variant of libfpd_amd.so that differs from the in-tree build in ONE translation unit (seconds instead of minutes):
#   tools/build_one.sh <name> <unit> [flags]     e.g.  tools/build_one.sh d3 wgrad3 -DW3_DEPTH=3
# -> build_ab/<name>/libfpd_amd.so (select with FPD_AMD_LIB); the other objects are the ones csrc/build.sh left in csrc/.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; UNIT=$2; shift 2
OUT="$ROOT/build_ab/$NAME"; mkdir -p "$OUT"
SRC="$ROOT/fast-human-pose-estimation.pytorch_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$SRC" -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed "$@" -c "$SRC/$UNIT.hip" -o "$OUT/$UNIT.o"
OBJS=""
for o in "$SRC"/*.o; do b=$(basename "$o"); if [ "$b" = "$UNIT.o" ]; then OBJS="$OBJS $OUT/$UNIT.o"; else OBJS="$OBJS $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfpd_amd.so" $OBJS
echo "built $OUT/libfpd_amd.so"
