#!/usr/bin/env bash
# round 5 call 1: wgrad3 (3x3 weight gradient, pieces x contiguous ranges, ring halo) -- kernel tests, kernel A/B, step A/B; statistics headroom test
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05g1; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "wgrad or headroom or elementwise" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v "^  File\|^Thread" $O/pytest.txt | tail -6 | cut -c1-300
for v in 0 1; do
  echo "== FPD_WGRAD3=$v"; FPD_WGRAD3=$v timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3" --iters 30 2>&1 | tail -6
done
for r in 16 32 48; do echo "== ranges $r"; FPD_WGRAD3_RANGES=$r timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3 64>64 @64" --iters 30 2>&1 | tail -1; done
for mt in 2 8; do echo "== min tiles $mt"; FPD_WGRAD3_MIN_TILES=$mt timeout 300 python tools/conv_bench.py --wgrad --partials --only "s 3x3" --iters 30 2>&1 | tail -5; done
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity 2> $O/err_$1.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config'].get('launches_per_step',{}).get('total'))"; }
for i in 1 2 3; do
  FPD_WGRAD3=0 run old$i
  run new$i
done
