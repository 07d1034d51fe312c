#!/usr/bin/env bash
# round 5 call 32: the data-parallel default of 6 hardware queues (bench.py picks it when the process group will exist)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g32; mkdir -p $O
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times"
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('gpu_max_hw_queues'), d['config'].get('allreduce_exposed_us'))"; }
for rep in 1 2; do
  echo "rep $rep plain $($B 2>/dev/null | ms)   force-dist(default) $($B --force-dist 2>/dev/null | ms)   force-dist q8 $(GPU_MAX_HW_QUEUES=8 $B --force-dist 2>/dev/null | ms)" | tee -a $O/ab.txt
done
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times 2>/dev/null | ms | sed 's/^/torchrun N=1: /' | tee -a $O/ab.txt
timeout 100 python bench.py --force-dist --no-cpu-baseline --no-parity --no-phase-times > $O/bench_line_force_dist.json 2>/dev/null
