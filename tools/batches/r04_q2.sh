#!/bin/bash
# round 4 batch q2: the -m gpu suite with its slowest tests listed
set -o pipefail
O=gpurun_out/r04q2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( while true; do sleep 60; echo "[progress] $(tail -c 120 $O/pytest_gpu.log 2>/dev/null | tr '\n' ' ')"; done ) &
PROG=$!
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=30 > $O/pytest_gpu.log 2>&1; rc=$?
kill $PROG 2>/dev/null
echo "pytest -m gpu rc=$rc"; tail -40 $O/pytest_gpu.log
