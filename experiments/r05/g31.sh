#!/usr/bin/env bash
# round 5 call 31: what does the one-rank RCCL path cost against the plain step on the SAME box?
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g31; mkdir -p $O
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times"
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('allreduce_exposed_us'))"; }
for rep in 1 2 3; do
  echo "rep $rep plain $($B 2>/dev/null | ms)   force-dist $($B --force-dist 2>/dev/null | ms)   force-dist q6 $(GPU_MAX_HW_QUEUES=6 $B --force-dist 2>/dev/null | ms)" | tee -a $O/ab.txt
done
