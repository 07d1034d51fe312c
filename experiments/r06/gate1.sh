#!/usr/bin/env bash
# round 6: full GPU suite on the current tree
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/gate1_tests.txt
cat gpurun_out/gate1_tests.txt
