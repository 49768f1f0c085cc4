#!/bin/bash
# round 3 batch t: order of a step's sections in the LDS Jacobi kernel (prefetch last / first level before the LDS reads)
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "lds" > $O/pytest_lds.log 2>&1; rc=$?; echo "lds rc=$rc"; tail -3 $O/pytest_lds.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/jacobi_tune.py --n 256 --sweeps 192 --reps 3 --variants "4:4:32:18,4:4:32:318,4:4:32:418,4:4:32:518,4:4:32:18" 2>&1 | grep -v amdgpu.ids > $O/jacobi_lds_256.txt; cat $O/jacobi_lds_256.txt
for v in "4:4:32:318 lds3_o1" "4:4:32:518 lds3_o3"; do set -- $v
  bash tools/jacobi_sq.sh $1 r03t_$2 2>&1 | tail -1 >> $O/sq_jacobi.txt; done; cat $O/sq_jacobi.txt
