#!/usr/bin/env bash
# round 5 call 13: evidence pass (kernel trace + stats, FETCH/WRITE_SIZE passes, SQ counters), wgrad tests after the slab-count changes, one-rank RCCL line
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g13; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "wgrad" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -3 | cut -c1-300
timeout 1500 bash tools/profile.sh r05 > $O/profile.log 2>&1; tail -5 $O/profile.log
ls gpurun_out/r05prof | head -30
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-phase-times --force-dist > $O/bench_line_force_dist.json 2> $O/err_dist.txt; tail -c 600 $O/bench_line_force_dist.json
