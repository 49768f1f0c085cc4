#!/bin/bash
# round 2, batch o: MGCG transfer operators without per-load branches, two-launch calc_sum (parity, MGCG-mode step, profile)
set -e -o pipefail
O=gpurun_out/r02o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mgcg.py tests/test_gpu_solver.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python bench.py --projection mgcg --steps 4 --warmup 2 --no-extra --no-cpu-baseline > $O/mgcg_l2.json 2> $O/mgcg_l2.err || { tail -20 $O/mgcg_l2.err; exit 1; }
python -c "import json,sys; d=json.loads(open('$O/mgcg_l2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
bash tools/batches/r02_n2.sh
