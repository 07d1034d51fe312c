#!/usr/bin/env bash
# HBM traffic of the dominant kernel (fused teacher Bottleneck at 64x64) from rocprofv3 PMC counters.
# Separate passes per counter (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2), no tracing domains besides --kernel-trace.
# Run on the GPU box from the repo root:  tools/pmc_bneck.sh  ->  gpurun_out/pmc_bneck/{fetch,write}/..., summary json
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_bneck"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ONLY=64 rocprofv3 --kernel-trace --pmc $c -f csv -d "$OUT/$c" -o pmc -- python "$ROOT/tools/bneck_bench.py" > "$OUT/$c.log" 2>&1 || true
done
python - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = []
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'bneck_eval_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c:
                vals.append(float(r['Counter_Value']))
    res[c] = sum(vals) / max(len(vals), 1)
    res[c + '_launches'] = len(vals)
# rocprofv3 reports KB; gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM section)
res['hbm_bytes_per_launch'] = (2.0 * res['FETCH_SIZE'] + res['WRITE_SIZE']) * 1024
res['note'] = ('fused teacher Bottleneck N=32 64x64 C=256 P=128; rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate '
               'passes, mean over launches, KB -> bytes, read side doubled per the gfx950 calibration in MI355X_MICROARCH.md')
json.dump(res, open(out + '/r01_pmc_bneck64.json', 'w'), indent=1)
print(json.dumps(res))
PY
