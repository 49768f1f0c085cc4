#!/bin/bash
# round 3 batch s: deeper prefetch in the LDS Jacobi kernel (input ring of 5 / 6 planes) -- parity, timing, SQ counters
O=gpurun_out/r03s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "lds" > $O/pytest_lds.log 2>&1; rc=$?; echo "lds rc=$rc"; tail -3 $O/pytest_lds.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/jacobi_tune.py --n 256 --sweeps 192 --reps 3 --variants "4:4:32:18,4:4:32:118,4:4:32:218,4:4:32:18,4:4:32:118" 2>&1 | grep -v amdgpu.ids > $O/jacobi_lds_256.txt; cat $O/jacobi_lds_256.txt
for v in "4:4:32:118 lds3_p5" "4:4:32:218 lds3_p6"; do set -- $v
  bash tools/jacobi_sq.sh $1 r03s_$2 2>&1 | tail -1 >> $O/sq_jacobi.txt; done; cat $O/sq_jacobi.txt
