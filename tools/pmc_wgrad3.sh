#!/usr/bin/env bash
# SQ counters of the 3x3 weight-gradient kernel at the micro-benchmark shapes (two passes of 8 slots; PMC only with --kernel-trace).
#   tools/pmc_wgrad3.sh   ->   gpurun_out/pmc_wgrad3/summary.txt
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_wgrad3"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
C2="SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY"
rocprofv3 --kernel-trace --pmc $C1 -f csv -d "$OUT/p1" -o p -- python "$ROOT/tools/conv_bench.py" --wgrad --partials --iters 5 --only "s 3x3 64>64 @64" > "$OUT/p1.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc $C2 -f csv -d "$OUT/p2" -o p -- python "$ROOT/tools/conv_bench.py" --wgrad --partials --iters 5 --only "s 3x3 64>64 @64" > "$OUT/p2.log" 2>&1 || true
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('%s/p*/**/*counter_collection.csv' % out, recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        if 'wgrad' in n:
            agg[(n, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
lines = []
for (n, g), c in sorted(agg.items()):
    lines.append('%s grid %s' % (n, g))
    for k in sorted(c):
        lines.append('    %-28s %.4g  (n=%d)' % (k, sum(c[k]) / len(c[k]), len(c[k])))
open(out + '/summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
