#!/usr/bin/env bash
# round-2 probe 12: cycle stamps of the fused Bottleneck (DMA variant) at 64x64; capped vs uncapped vs no-DMA
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p12; mkdir -p $O
{
echo "== timing build (stamps of block 0), capped 160"
ONLY=64 FPD_AMD_LIB=$PWD/build_ab/timing/libfpd_amd.so python tools/bneck_bench.py 2>&1 | grep -E "bneck W|fused" | sort | uniq -c | sort -rn | head -8
echo "== production build: cap 160 / 256 / 1024, DMA on/off"
for cap in 160 256 1024; do for dma in 1 0; do
  echo "cap=$cap dma=$dma: $(ONLY=64 FPD_BNECK_BLOCKS=$cap FPD_BNECK_DMA=$dma python tools/bneck_bench.py 2>&1 | grep fused)"
done; done
} | tee $O/bneck.txt
