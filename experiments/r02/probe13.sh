#!/usr/bin/env bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02p13; mkdir -p $O
( timeout 600 python -m pytest tests/test_fp8_gpu.py tests/test_entry_gpu.py -m gpu -q -p no:cacheprovider -s -k "fp8_student or interpreter or cli_smoke" > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log )
grep -E "passed|failed|rel-L2|^E  |Error" $O/t1.log | head -20
for i in 1 2 3; do timeout 200 python -m pytest tests/test_bf16_parity_gpu.py -m gpu -q -p no:cacheprovider -s -k trained_pair 2>&1 | grep -E "trained teacher map|passed|failed"; done | tee $O/t2.log
