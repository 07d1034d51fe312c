#!/usr/bin/env bash
# SQ counters (one pass, 8 slots; --pmc with --kernel-trace only) of the round-6 student kernels at the micro-benchmark's 64x64 / 32x32
# shapes (tools/c1_bench.py: conv_c1, conv_c3 and, interleaved, what the dispatcher picks without them).
#   tools/pmc_c1.sh   ->   gpurun_out/pmc_c1/summary.txt
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_c1"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
for k in "@64" "@32"; do
  rocprofv3 --kernel-trace --pmc $C -f csv -d "$OUT/run$k" -o p -- python "$ROOT/tools/c1_bench.py" --iters 4 --rounds 1 --only "$k" > "$OUT/run$k.log" 2>&1 || true
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('%s/**/*counter_collection.csv' % out, recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if n.startswith('c1_kernel') or n.startswith('c3_kernel') or n.startswith('conv_pp_kernel'):
            agg[(n, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
lines = []
for (n, g), c in sorted(agg.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1.0)
    lines.append('%-50s grid %7s launches %3d | of wave cycles: parked (wait_any) %4.1f%%  issue stalls (wait_inst_any) %4.1f%%  of which LDS %4.1f%% | '
                 'mfma_busy %.3g  busy %.3g  lds_bank_conflict %.3g  lds_active %.3g' % (
                     n[:50], g, len(next(iter(c.values()))), 100 * m.get('SQ_WAIT_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_ANY', 0) / wc,
                     100 * m.get('SQ_WAIT_INST_LDS', 0) / wc, m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), m.get('SQ_BUSY_CYCLES', 0),
                     m.get('SQ_LDS_BANK_CONFLICT', 0), m.get('SQ_ACTIVE_INST_LDS', 0)))
open(out + '/summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
