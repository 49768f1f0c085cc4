#!/bin/bash
# round 2, batch y: slab tests + the RCCL branch on both stand-ins after moving the wall-sheet messages in front of stage 3
set -o pipefail
O=gpurun_out/r02y; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_slab_multirank.py tests/test_gpu_rccl_path.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -8 $O/tests.log; exit $rc
