#!/bin/bash
# round 4 batch q: the whole -m gpu suite (what the driver runs at round end)
set -o pipefail
O=gpurun_out/r04q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( while true; do sleep 60; echo "[progress] $(tail -c 200 $O/pytest_gpu.log 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done ) &
PROG=$!
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?
kill $PROG 2>/dev/null
echo "pytest -m gpu rc=$rc"; tail -8 $O/pytest_gpu.log
