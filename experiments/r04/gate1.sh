#!/usr/bin/env bash
# round-4 gate 1: the whole GPU suite on the exact-statistics build + a bench line + phase times (same box)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -40 $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $O/bench.err | grep '^{' > $O/bench.json
python -c "import json;d=json.load(open('$O/bench.json'));print('bench %.3f ms/step' % d['ms_per_step'], d['config']['launches_per_step'])"
timeout 300 python tools/probes/phase_times.py > $O/phase_times.txt 2>&1; tail -12 $O/phase_times.txt
