#!/bin/bash
# round 4 batch zt: fl_shutdown destroys the CU-masked stream first: the reproducer three times, the exit tests, the runtime tests
set -o pipefail
O=gpurun_out/r04zt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
CMD="python3 tools/jacobi_tune.py --n 128 --reps 7 --variants 5:2:4,4:0:8,4:0:12,4:0:16,4:6:8,4:6:16,4:2:4"
for i in 1 2 3; do
  s=$(date +%s.%N); timeout -k 5 40 $CMD > $O/run$i.txt 2>&1; rc=$?; e=$(date +%s.%N)
  echo "reproducer $i rc=$rc $(python3 -c "print(round($e-$s,1))") s"
done
timeout -k 5 60 python3 tools/jacobi_tune.py --n 128 --reps 7 --variants 5:2:4,4:0:8,4:0:12,4:0:16,4:6:8,4:6:16,4:2:4 > /dev/null 2>&1; echo "again rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_runtime.py tests/test_gpu_contexts.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_rccl_path.py -x -q -k "reserved_cus or two_ranks_over" > $O/pytest2.log 2>&1; echo "pytest reserved-cus rc=$?"; tail -3 $O/pytest2.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/bench.json 2>$O/bench.err; echo "bench (with its profiled children) rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r04zt/bench.json')); print(d['value'], d['roofline']['traffic'], d['roofline']['traffic_source'][:60])"
