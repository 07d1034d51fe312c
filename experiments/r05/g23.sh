#!/usr/bin/env bash
# round 5 call 23: (a) fused Bottleneck / head epilogue with wave-level hand-over instead of three block barriers per tile
# (build_ab/ep), tests + interleaved A/B; (b) grid caps of the teacher's fused kernels under the round-5 lane (lighter weight gradients)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g23; mkdir -p $O
EP=$PWD/build_ab/ep/libfpd_amd.so
FPD_AMD_LIB=$EP timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "bneck or head or teacher or fused" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -3 | cut -c1-300
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times"
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "base $($B 2>/dev/null | ms)  ep $(FPD_AMD_LIB=$EP $B 2>/dev/null | ms)" | tee -a $O/ab.txt
done
for cap in 96 112 144 160; do echo "ep bneck cap $cap: $(FPD_AMD_LIB=$EP FPD_BNECK_BLOCKS=$cap $B 2>/dev/null | ms)" | tee -a $O/ab.txt; done
for cap in 128 192 256; do echo "ep head cap $cap: $(FPD_AMD_LIB=$EP FPD_HEAD_BLOCKS=$cap $B 2>/dev/null | ms)" | tee -a $O/ab.txt; done
echo "ep again: $(FPD_AMD_LIB=$EP $B 2>/dev/null | ms)" | tee -a $O/ab.txt
