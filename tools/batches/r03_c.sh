#!/bin/bash
# round 3 batch c: which operator breaks parity on rows of 1024 / 1025 floats (step 2 of the 1024 x 1024 x 32 hash test differs in u)
set -o pipefail
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -k "1024" > $O/pytest_ops_1024.log 2>&1; echo "ops rc=$?"; grep -E "FAILED|passed|failed" $O/pytest_ops_1024.log | head -40
python -m pytest tests/test_gpu_solver.py -q -x -k "config5_rows or emitter" > $O/pytest_solver.log 2>&1; echo "solver rc=$?"; tail -12 $O/pytest_solver.log
