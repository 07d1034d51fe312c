#!/usr/bin/env bash
# Evidence of a round (argument: file prefix, default r04) (run on the GPU box from the repo root; results under gpurun_out/<prefix>prof, to be copied into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench workload  -> kernel stats + per-(kernel, grid, shape) table
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (own runs, nothing but --kernel-trace) over the fused Bottleneck
#      micro-benchmark and over the bench step -> HBM bytes per launch of the dominant kernel and of the HBM-bound kernels
#   3. one SQ-counter pass over the Bottleneck / conv micro-benchmarks (tools/pmc_kernels.sh)
#   4. HRNET=1: the same trace + counter passes over `bench.py --config hrnet` (BASELINE configs[3])
# Every pass writes the library's launch log (FPD_LAUNCH_LOG) next to its trace: tools/profile_summarize.py keys the rows on the
# tensor shape with it.
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PFX="${1:-r04}"
OUT="$ROOT/gpurun_out/${PFX}prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-parity --no-phase-times"
pass() {   # directory, rocprofv3 options..., --, command
  local d="$OUT/$1"; shift
  mkdir -p "$d"
  FPD_LAUNCH_LOG="$d/launch.log" rocprofv3 "$@" > "$d.log" 2>&1 || true
}
pass stats --kernel-trace --stats -f csv -d "$OUT/stats" -o b -- $B --steps 20 --warmup 5
for c in FETCH_SIZE WRITE_SIZE; do
  pass step_$c --kernel-trace --pmc $c -f csv -d "$OUT/step_$c" -o p -- $B --steps 3 --warmup 1
  ONLY=64 pass bneck_$c --kernel-trace --pmc $c -f csv -d "$OUT/bneck_$c" -o p -- python "$ROOT/tools/bneck_bench.py"
done
bash "$ROOT/tools/pmc_kernels.sh" > "$OUT/sq.log" 2>&1 || true
cp "$ROOT/gpurun_out/pmc_kernels/summary.txt" "$OUT/${PFX}_pmc_sq_counters.txt" 2>/dev/null || true
python "$ROOT/tools/profile_summarize.py" "$OUT" "$PFX"
if [ "${HRNET:-0}" = "1" ]; then
  pass hrnet_stats --kernel-trace --stats -f csv -d "$OUT/hrnet_stats" -o b -- $B --config hrnet --steps 10 --warmup 3
  for c in FETCH_SIZE WRITE_SIZE; do
    pass hrnet_step_$c --kernel-trace --pmc $c -f csv -d "$OUT/hrnet_step_$c" -o p -- $B --config hrnet --steps 2 --warmup 1
  done
  python "$ROOT/tools/profile_summarize.py" "$OUT" "$PFX" hrnet_
fi
