#!/usr/bin/env bash
# Recompile the given translation units of csrc/ (default: none) and relink csrc/libfpd_amd.so from the objects csrc/build.sh left.
#   tools/relink.sh conv_c1 [api ...]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/fast-human-pose-estimation.pytorch_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed"
pids=()
for u in "$@"; do ( hipcc $FLAGS -c "$SRC/$u.hip" -o "$SRC/$u.o" ) & pids+=($!); done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
UNITS=$(grep -o 'for f in [^;]*' "$SRC/build.sh" | sed 's/for f in //')
OBJS=""; for u in $UNITS; do OBJS="$OBJS $SRC/$u.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$SRC/libfpd_amd.so" $OBJS
echo "relinked $SRC/libfpd_amd.so"
