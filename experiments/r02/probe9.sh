#!/usr/bin/env bash
# round-2 probe 9: where do the student's 64x64 convolutions spend their time?  conv_bench ablations (FPD_CONV_DBG bits:
# 1 no epilogue, 2 no MFMA taps, 4 no halo store, 8 no BN tables) + phase times of the step
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p9; mkdir -p $O
for dbg in 0 1 2 3 7 15; do
  echo "== FPD_CONV_DBG=$dbg (graph replay, device time)"
  FPD_CONV_DBG=$dbg python tools/conv_bench.py --only "@64" --graph --iters 20 2>&1 | grep -E "^s |^t 3x3"
done | tee $O/conv_ablation.txt
echo "== no stats"; python tools/conv_bench.py --only "s " --graph --iters 20 --no-stats 2>&1 | grep "@64" | tee -a $O/conv_ablation.txt
echo "== eager"; python tools/conv_bench.py --only "s " --iters 50 2>&1 | tee -a $O/conv_ablation.txt
python tools/probes/phase_times.py 2>&1 | tee $O/phase_times.txt
