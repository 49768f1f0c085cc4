#!/bin/bash
# round 2, batch m: two-sweep kernel, rows of 2-4 waves: how the 62 lanes that need no edge column skip its loads
set -e -o pipefail
O=gpurun_out/r02m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
V="5:0:0:1,5:0:0:2"
for nz in 512 144 80; do
  timeout -k 10 300 python tools/jacobi_tune.py --n 512 --nz $nz --sweeps 100 --reps 3 --variants "$V" > $O/tune4_512x$nz.log 2>&1; cat $O/tune4_512x$nz.log
done
