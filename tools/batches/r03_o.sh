#!/bin/bash
# round 3 batch o: FOUR sweeps per launch (jacobi_lds_kernel<.., 4>) -- parity, then timing at 256^3
set -o pipefail
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "lds" > $O/pytest_lds.log 2>&1; rc=$?; echo "lds rc=$rc"; tail -4 $O/pytest_lds.log
[ $rc -eq 0 ] || exit 1
V="4:0:0,4:4:32:18"
for S in 24 18; do for kc in 26 32 37 43 52 64; do V="$V,4:6:$kc:$S"; done; done
timeout -k 10 600 python3 tools/jacobi_tune.py --n 256 --sweeps 196 --reps 3 --variants "$V" 2>&1 | grep -v amdgpu.ids > $O/jacobi_lds4_256.txt; cat $O/jacobi_lds4_256.txt
