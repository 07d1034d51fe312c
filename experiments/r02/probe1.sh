#!/usr/bin/env bash
# round-2 probe 1: rerun of the new tests, fused-Bottleneck weight-DMA A/B, conv_tile ablation, bench A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p1; mkdir -p $O
( timeout 600 python -m pytest tests/test_entry_gpu.py tests/test_bf16_parity_gpu.py tests/test_exact_gpu.py "tests/test_model_gpu.py" tests/test_kernels_gpu.py -m gpu -q -s -p no:cacheprovider -k "not full_size" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -3 $O/tests.log
for dma in 0 1; do for cap in 160 1024; do
  echo "== bneck DMA=$dma cap=$cap" >> $O/bneck.log
  FPD_BNECK_DMA=$dma FPD_BNECK_BLOCKS=$cap timeout 120 python tools/bneck_bench.py >> $O/bneck.log 2>&1
done; done
cat $O/bneck.log | grep -v amdgpu.ids
for dbg in 0 1 2 3 4 8 15; do
  echo "== conv dbg=$dbg" >> $O/conv.log
  FPD_CONV_DBG=$dbg timeout 120 python tools/conv_bench.py --graph --only "s 3x3 64>64" >> $O/conv.log 2>&1
done
timeout 120 python tools/conv_bench.py --graph --only "s 1x1" >> $O/conv.log 2>&1
grep -v amdgpu.ids $O/conv.log
for dma in 0 1; do
  FPD_BNECK_DMA=$dma timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_dma$dma.json 2> $O/bench_dma$dma.err
  python -c "import json;d=json.load(open('$O/bench_dma$dma.json'));print('DMA=$dma', d['ms_per_step'], d['roofline']['avg_us'])"
done
( time timeout 500 python bench.py > $O/bench_full.json 2> $O/bench_full.err ) 2>> $O/bench_full.err
tail -3 $O/bench_full.err
