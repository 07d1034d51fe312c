#!/usr/bin/env bash
# round 6 final gate: full GPU suite (writing the trained-pair parity record), smoke, default bench line, driver-style bench, HRNet line,
# one-rank RCCL line, three plain runs
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06gate; mkdir -p $O
rm -f $O/parity_trained.json
FPD_WRITE_PARITY_JSON=$PWD/$O/parity_trained.json timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.txt | tail -8 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/final_bench_line.json 2> $O/final_bench_err.txt; tail -c 400 $O/final_bench_line.json; echo
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/driver_style_line.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/driver_style_line.json').read().strip().splitlines()[-1]);print('driver-style', d['ms_per_step'], d['value'])"
timeout 600 python bench.py --config hrnet > $O/bench_line_hrnet.json 2> $O/hrnet_err.txt; python -c "import json;d=json.loads(open('$O/bench_line_hrnet.json').read().strip().splitlines()[-1]);print('hrnet', d['ms_per_step'], d['value'], d['config']['launches_per_step']['total'])"
timeout 600 python bench.py --force-dist --steps 30 --warmup 8 --no-cpu-baseline --no-parity > $O/bench_line_force_dist.json 2> $O/force_dist_err.txt; python -c "import json;d=json.loads(open('$O/bench_line_force_dist.json').read().strip().splitlines()[-1]);print('force-dist', d['ms_per_step'], d['config'].get('allreduce_exposed_us'))"
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-phase-times 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do run base_$i; done
