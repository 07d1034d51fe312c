#!/usr/bin/env bash
# round-4 final gate: the whole GPU suite, the default bench line (parity + CPU baseline legs), the driver's command line, the HRNet line,
# the shape-keyed profiles of the final kernels, the conv_tile phase stamps after the trims
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04g15; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -8 | cut -c1-300
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench_line.json; echo
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style', d['ms_per_step'], d['value'])"
timeout 600 python bench.py --config hrnet > $O/bench_line_hrnet.json 2> $O/bench_hrnet.err; tail -c 300 $O/bench_line_hrnet.json; echo
FPD_AMD_LIB=$PWD/build_ab/tiletime/libfpd_amd.so timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity 2>/dev/null | grep "^conv_tile" > $O/step_stamps_after.txt; wc -l $O/step_stamps_after.txt
HRNET=1 timeout 1200 bash tools/profile.sh r04 > $O/profile.txt 2>&1; tail -8 $O/profile.txt | cut -c1-250
cp gpurun_out/r04prof/r04_* $O/ 2>/dev/null; ls $O | head -40
