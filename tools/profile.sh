#!/usr/bin/env bash
# Evidence of a round (argument: file prefix, default r03) (run on the GPU box from the repo root; results under gpurun_out/<prefix>prof, to be copied into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench workload  -> kernel stats + per-(kernel, grid) table
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (own runs, nothing but --kernel-trace) over the fused Bottleneck
#      micro-benchmark and over the bench step -> HBM bytes per launch of the dominant kernel and of the HBM-bound kernels
#   3. one SQ-counter pass over the Bottleneck / conv micro-benchmarks (tools/pmc_kernels.sh)
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PFX="${1:-r03}"
OUT="$ROOT/gpurun_out/${PFX}prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-parity"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o b -- $B --steps 20 --warmup 5 > "$OUT/stats.log" 2>&1 || true
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -f csv -d "$OUT/step_$c" -o p -- $B --steps 3 --warmup 1 > "$OUT/step_$c.log" 2>&1 || true
  ONLY=64 rocprofv3 --kernel-trace --pmc $c -f csv -d "$OUT/bneck_$c" -o p -- python "$ROOT/tools/bneck_bench.py" > "$OUT/bneck_$c.log" 2>&1 || true
done
bash "$ROOT/tools/pmc_kernels.sh" > "$OUT/sq.log" 2>&1 || true
cp "$ROOT/gpurun_out/pmc_kernels/summary.txt" "$OUT/${PFX}_pmc_sq_counters.txt" 2>/dev/null || true
python "$ROOT/tools/profile_summarize.py" "$OUT" "$PFX"
