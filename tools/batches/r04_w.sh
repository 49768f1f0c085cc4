#!/bin/bash
# round 4 batch w: seeded shape sweep of the fused level-0 kernels
set -o pipefail
O=gpurun_out/r04w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mgcg.py -x -q -k "random_shapes or vector_updates" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log
