#!/usr/bin/env bash
# round-3 probe 4: conv_pp v2 (all waves on every phase, two blocks per CU): parity tests, step A/B, per-shape times, stamps
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p4; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "conv_pp or conv_pair or dgrad or conv_forward" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -6 $O/tests.log
run() {  # name, env
  timeout 200 env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step  loss %s' % ('$1', d['ms_per_step'], d['config']['loss_last_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
run pp0 FPD_CONV_PP=0
run pp1 FPD_CONV_PP=1
run pp0b FPD_CONV_PP=0
run pp1b FPD_CONV_PP=1
run pp1_occ1 "FPD_CONV_PP=1 FPD_CONV_PP_OCC=1"
run pp1_128 "FPD_CONV_PP=1 FPD_CONV_PP_BLOCKS=128"
run pp1_student "FPD_CONV_PP=1 FPD_WHATIF=t_all"
run pp0_student "FPD_CONV_PP=0 FPD_WHATIF=t_all"
echo "== conv_bench --graph, FPD_CONV_PP=1"
FPD_CONV_PP=1 timeout 300 python tools/conv_bench.py --graph 2>&1 | grep -E "@64|@128"
echo "== conv_bench --graph, FPD_CONV_PP=0"
FPD_CONV_PP=0 timeout 300 python tools/conv_bench.py --graph 2>&1 | grep -E "@64|@128"
export FPD_AMD_LIB=$PWD/build_ab/pptime/libfpd_amd.so
for sh in "3x3 64>64 @64" "1x1 128>64 @64" "1x1 64>128 @64"; do
  FPD_CONV_PP=1 timeout 120 python tools/conv_bench.py --iters 1 --only "$sh" 2>&1 | tail -3
done | tee $O/stamps.txt
