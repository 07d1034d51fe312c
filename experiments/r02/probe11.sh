#!/usr/bin/env bash
# round-2 probe 11: epilogue decomposition of the student's 64x64 convs (FPD_EPI_DBG bits: 1 no statistics tail, 2 no atomics,
# 4 no y store)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p11; mkdir -p $O
for dbg in 0 2 1 4 5; do
  echo "== FPD_EPI_DBG=$dbg"
  FPD_EPI_DBG=$dbg python tools/conv_bench.py --only "s " --graph --iters 20 2>&1 | grep -E "@64|@32"
done | tee $O/epi_ablation.txt
echo "== no-stats pointer"; python tools/conv_bench.py --only "s " --graph --iters 20 --no-stats 2>&1 | grep -E "@64|@32" | tee -a $O/epi_ablation.txt
