#!/bin/bash
# round 3 batch u: LDS Jacobi kernel in blocks of 4 single rows (8 waves), two blocks per CU
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "lds" > $O/pytest_lds.log 2>&1; rc=$?; echo "lds rc=$rc"; tail -3 $O/pytest_lds.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/jacobi_tune.py --n 256 --sweeps 192 --reps 3 --variants "4:4:32:18,4:4:32:14,4:4:16:14,4:4:64:14" 2>&1 | grep -v amdgpu.ids > $O/jacobi_lds_256.txt; cat $O/jacobi_lds_256.txt
bash tools/jacobi_sq.sh 4:4:32:14 r03u_lds3_w4 2>&1 | tail -1 >> $O/sq_jacobi.txt; cat $O/sq_jacobi.txt
