#!/bin/bash
# round 4 batch k: forward-map update beside the DMC sub-steps (fl_aux_*), volatile DPP adds in the Jacobi kernels: parity, then A/B timing
set -o pipefail
O=gpurun_out/r04k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1100 python -m pytest tests/test_gpu_solver.py tests/test_gpu_projection.py tests/test_gpu_runtime.py tests/test_gpu_contexts.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "full_size_hashes" > $O/pytest_hash.log 2>&1; echo "pytest hashes rc=$?"; tail -3 $O/pytest_hash.log
B="python3 bench.py --gpus 1 --steps 40 --warmup 10 --no-extra --no-cpu-baseline --no-measure-traffic"
show() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], "Mvox/s", d["ms_per_step"], "ms", "jacobi us/launch", d["roofline"]["us_per_launch"])
except Exception as e:
    print("   unreadable:", e)
PY
}
for rep in 1 2; do
for v in "beside:" "in_turn:--bq-opt 12=0"; do
  tag=${v%%:*}; opt=${v#*:}
  timeout -k 10 300 $B $opt > $O/bench_${tag}_$rep.json 2>$O/bench_${tag}_$rep.err; echo "$tag $rep rc=$?"; show $O/bench_${tag}_$rep.json
done
done
timeout -k 10 300 python3 bench.py --gpus 1 --size 128 --steps 100 --warmup 20 --no-extra --no-cpu-baseline --no-measure-traffic > $O/bench_128.json 2>/dev/null; show $O/bench_128.json
timeout -k 10 300 python3 bench.py --gpus 1 --size 128 --steps 100 --warmup 20 --no-extra --no-cpu-baseline --no-measure-traffic --bq-opt 12=0 > $O/bench_128_in_turn.json 2>/dev/null; show $O/bench_128_in_turn.json
