#!/bin/bash
# round 2, batch s: full GPU suite + default bench after the three-sweep kernel changes
set -o pipefail
O=gpurun_out/r02s; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; cut -c1-200 $O/bench_default.json
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver.err; cut -c1-200 $O/bench_driver_cmd.json
