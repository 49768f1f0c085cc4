#!/bin/bash
# round 2, batch n: LDS tile smoother on the coarse multigrid levels (parity + MGCG-mode step time, on and off)
set -e -o pipefail
O=gpurun_out/r02n; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mgcg.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for v in 1 2 0 1 0; do
  timeout -k 10 300 python bench.py --projection mgcg --steps 4 --warmup 2 --no-extra --no-cpu-baseline --fl-opt 14=$v > $O/mgcg_tile_$v.json 2> $O/mgcg_tile_$v.err || { tail -20 $O/mgcg_tile_$v.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('$O/mgcg_tile_$v.json').read().strip().splitlines()[-1]); print('tile=$v', d['value'], d['ms_per_step'])"
done
