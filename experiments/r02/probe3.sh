#!/usr/bin/env bash
# round-2 probe 3: HRNet product path (kernels, golden parity, W32<-W48 bf16), hrnet bench line, Bottleneck grid cap re-sweep
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p3; mkdir -p $O
( timeout 600 python -m pytest tests/test_hrnet_gpu.py -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -4 $O/tests.log
timeout 300 python bench.py --config hrnet --steps 10 --warmup 3 > $O/bench_hrnet.json 2> $O/bench_hrnet.err
python -c "import json;d=json.load(open('$O/bench_hrnet.json'));print('hrnet', d['ms_per_step'], d['value'], d['roofline'])" || tail -5 $O/bench_hrnet.err
for cap in 160 192 224; do
  FPD_BNECK_BLOCKS=$cap timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_cap$cap.json 2> $O/bench_cap$cap.err
  python -c "import json;d=json.load(open('$O/bench_cap$cap.json'));print('cap=$cap', d['ms_per_step'], d['roofline']['avg_us'])"
done
