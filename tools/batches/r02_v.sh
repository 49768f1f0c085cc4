#!/bin/bash
# round 2, batch v: the reflection scheme on z-slab ranks (GPU parity, 2 and 3 ranks) + the rest of the slab tests
set -o pipefail
O=gpurun_out/r02v; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_slab_multirank.py tests/test_gpu_ops.py tests/test_gpu_solver.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; exit $rc
