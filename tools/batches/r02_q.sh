#!/bin/bash
# round 2, batch q: the RCCL branch with several ranks on one GPU (tests/fake_rccl stand-in)
set -o pipefail
O=gpurun_out/r02q; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_rccl_path.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -40 $O/tests.log; exit $rc
