#!/usr/bin/env bash
# round-4 gate 7: the whole GPU suite, the default bench line (parity + CPU baseline legs), the HRNet line, the shape-keyed profiles
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g7; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=15 --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -30 | cut -c1-300
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench_line.json; echo
timeout 600 python bench.py --config hrnet > $O/bench_line_hrnet.json 2> $O/bench_hrnet.err; tail -c 400 $O/bench_line_hrnet.json; echo
HRNET=1 timeout 1200 bash tools/profile.sh r04 > $O/profile.txt 2>&1; tail -12 $O/profile.txt | cut -c1-250
cp gpurun_out/r04prof/r04_* $O/ 2>/dev/null; ls $O
