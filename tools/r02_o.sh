#!/bin/bash
# round 2, batch o: the lean fp64 two-sweep smoother (parity, timing against mg_smooth2_kernel, MGCG-mode step)
set -e -o pipefail
O=gpurun_out/r02o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mgcg.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/smooth_tune.py --n 256 --sweeps 32 --reps 3 > $O/smooth_256.log 2>&1; cat $O/smooth_256.log
for v in 0 3; do
  timeout -k 10 300 python bench.py --projection mgcg --steps 4 --warmup 2 --no-extra --no-cpu-baseline --fl-opt 6=$v > $O/mgcg_rows_$v.json 2> $O/mgcg_rows_$v.err || { tail -20 $O/mgcg_rows_$v.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('$O/mgcg_rows_$v.json').read().strip().splitlines()[-1]); print('rows-option=$v', d['value'], d['ms_per_step'], d['roofline'])"
done
