#!/usr/bin/env bash
# round-2 probe 2: conv_tile all-taps staging / early loads / fp32-combined bf16 statistics
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p2; mkdir -p $O
( timeout 500 python -m pytest tests/test_entry_gpu.py tests/test_exact_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_bf16_parity_gpu.py -m gpu -q -s -p no:cacheprovider -k "(conv or pair or fused_step or module_api or pipelined or trained or bf16_build) and not wgrad" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -3 $O/tests.log
for allw in 0 320 4000; do
  echo "== conv ALLW=$allw" >> $O/conv.log
  FPD_CONV_ALLW=$allw timeout 120 python tools/conv_bench.py --graph --only "s 3x3 64>64" >> $O/conv.log 2>&1
done
timeout 120 python tools/conv_bench.py --graph --only "s 1x1" >> $O/conv.log 2>&1
grep -v amdgpu.ids $O/conv.log
for allw in 0 320 4000; do
  FPD_CONV_ALLW=$allw timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_allw$allw.json 2> $O/bench_allw$allw.err
  python -c "import json;d=json.load(open('$O/bench_allw$allw.json'));print('ALLW=$allw', d['ms_per_step'], d['roofline']['avg_us'])"
done
