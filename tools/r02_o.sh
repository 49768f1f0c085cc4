#!/bin/bash
# round 2, batch o: the lean fp64 two-sweep smoother on odd rows / level 1 (parity, timing, MGCG-mode step)
set -e -o pipefail
O=gpurun_out/r02o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mgcg.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for n in 127 128; do
  timeout -k 10 300 python tools/smooth_tune.py --n $n --sweeps 32 --reps 3 --variants "0:0:0:0,2:3:0:0,2:0:0:1,2:0:16:1,2:0:32:1,2:0:0:2" > $O/smooth_$n.log 2>&1; cat $O/smooth_$n.log
done
timeout -k 10 300 python bench.py --projection mgcg --steps 4 --warmup 2 --no-extra --no-cpu-baseline > $O/mgcg_l1.json 2> $O/mgcg_l1.err || { tail -20 $O/mgcg_l1.err; exit 1; }
python -c "import json,sys; d=json.loads(open('$O/mgcg_l1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
