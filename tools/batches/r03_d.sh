#!/bin/bash
# round 3 batch d: operator parity on a grid with config 5's planes and more than 2^24 elements per field (1024 x 1024 x 18)
set -o pipefail
O=gpurun_out/r03d; mkdir -p $O
BQ_TEST_EXTRA_GRID=1024,1024,18 timeout -k 10 1000 python -m pytest tests/test_gpu_ops.py -q -k "1024-1024" > $O/pytest_ops_big.log 2>&1; echo "ops rc=$?"; grep -E "FAILED|passed|failed|Error" $O/pytest_ops_big.log | head -40
