#!/usr/bin/env bash
# SQ counters (one pass, 8 slots) of the dominant kernel families: fused Bottleneck at 64x64 (tools/bneck_bench.py) and the
# conv tile kernel at the micro-benchmark shapes (tools/conv_bench.py).  PMC only with --kernel-trace, nothing else traced.
#   tools/pmc_kernels.sh   ->   gpurun_out/pmc_kernels/summary.txt
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_kernels"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
ONLY=64 FPD_BNECK_BLOCKS=1024 rocprofv3 --kernel-trace --pmc $C -f csv -d "$OUT/bneck" -o p -- python "$ROOT/tools/bneck_bench.py" > "$OUT/bneck.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc $C -f csv -d "$OUT/conv" -o p -- python "$ROOT/tools/conv_bench.py" --iters 5 --only "@64" > "$OUT/conv.log" 2>&1 || true
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
lines = []
for sub in ('bneck', 'conv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
            if 'bneck_eval' in n or 'conv_tile' in n or 'conv_pp' in n:
                agg[(n, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for (n, g), c in sorted(agg.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        wc = m.get('SQ_WAVE_CYCLES', 1.0)
        lines.append('%-44s grid %8s  launches %3d | of wave cycles (quad-cycles): wait_any %4.1f%%  wait_inst_any %4.1f%%  wait_inst_lds %4.1f%% '
                     '| MFMA busy cycles / (busy cycles*... ) raw: mfma_busy %.3g busy %.3g  lds_bank_conflict %.3g lds_active %.3g' % (
                         n[:44], g, len(next(iter(c.values()))), 100 * m.get('SQ_WAIT_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_ANY', 0) / wc,
                         100 * m.get('SQ_WAIT_INST_LDS', 0) / wc, m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), m.get('SQ_BUSY_CYCLES', 0),
                         m.get('SQ_LDS_BANK_CONFLICT', 0), m.get('SQ_ACTIVE_INST_LDS', 0)))
open(out + '/summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
