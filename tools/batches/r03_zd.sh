#!/bin/bash
# round 3 batch zd: the one-segment three-sweep kernel with the input level through LDS (8 / 12 output rows per block)
O=gpurun_out/r03zd; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "lds" > $O/pytest_lds.log 2>&1; rc=$?; echo "lds rc=$rc"; tail -3 $O/pytest_lds.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/jacobi_tune.py --n 256 --sweeps 192 --reps 3 --variants "4:4:32:18,4:4:32:31,4:4:0:32,4:4:32:32,4:4:64:32,4:4:32:18" 2>&1 | grep -v amdgpu.ids > $O/jacobi_256.txt; cat $O/jacobi_256.txt
for v in "4:4:32:31 l1s_w8" "4:4:0:32 l1s_w12"; do set -- $v
  bash tools/jacobi_sq.sh $1 r03zd_$2 2>&1 | tail -1 >> $O/sq_jacobi.txt; done; cat $O/sq_jacobi.txt
