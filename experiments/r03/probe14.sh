#!/usr/bin/env bash
# round-3 probe 14: why is the --force-dist step slow?  hardware queues x bucketed / single / fake all-reduce
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p14; mkdir -p $O
run() {  # name, env, extra args
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity $3 2> $O/$1.err | grep '^{' > $O/$1.json
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  exposed %s' % ('$1', d['ms_per_step'], d['config'].get('allreduce_exposed_us')))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run plain ""
run fd_q8 "" --force-dist
run fd_q8_single "FPD_ALLREDUCE_BUCKETS=0" --force-dist
run fd_q8_fake "FPD_FAKE_ALLREDUCE=1" --force-dist
run fd_q4 "GPU_MAX_HW_QUEUES=4" --force-dist
run fd_q4_single "GPU_MAX_HW_QUEUES=4 FPD_ALLREDUCE_BUCKETS=0" --force-dist
run fd_q6 "GPU_MAX_HW_QUEUES=6" --force-dist
run fd_q6_single "GPU_MAX_HW_QUEUES=6 FPD_ALLREDUCE_BUCKETS=0" --force-dist
run fd_q12 "GPU_MAX_HW_QUEUES=12" --force-dist
run fake_nodist "FPD_FAKE_ALLREDUCE=1"
run plain_q4 "GPU_MAX_HW_QUEUES=4"
