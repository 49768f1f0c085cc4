#!/bin/bash
# round 2, batch y: full-size hash tests (BiMocq + the next rows)
set -o pipefail
O=gpurun_out/r02y; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -8 $O/tests.log; exit $rc
