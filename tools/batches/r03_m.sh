#!/bin/bash
# round 3 batch m: LDS three-sweep kernel with one row per wave (shapes 18, 19) against row pairs (24, 26) at 256^3
set -o pipefail
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "lds" > $O/pytest_lds.log 2>&1; rc=$?; echo "lds rc=$rc"; tail -3 $O/pytest_lds.log
[ $rc -eq 0 ] || exit 1
V="4:0:0,4:4:32:24,4:4:26:26"
for S in 18 19; do for kc in 16 22 24 26 32 43; do V="$V,4:4:$kc:$S"; done; done
timeout -k 10 600 python3 tools/jacobi_tune.py --n 256 --sweeps 198 --reps 3 --variants "$V" 2>&1 | grep -v amdgpu.ids > $O/jacobi_lds_256.txt; cat $O/jacobi_lds_256.txt
