#!/bin/bash
# round 4 batch s: bench.py's N > 1 watchdog (headline handed out when a later leg hangs) + the N = 2 / 4 bench tests again
set -o pipefail
O=gpurun_out/r04s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1100 python -m pytest tests/test_gpu_rccl_path.py -x -q -k "bench" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log
