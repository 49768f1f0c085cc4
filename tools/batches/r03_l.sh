#!/bin/bash
# round 3 batch l: LDS-exchanged three-sweep Jacobi kernel -- block shapes (4 / 5 / 6 output row pairs) x chunk lengths at 256^3
set -o pipefail
O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "lds" > $O/pytest_lds.log 2>&1; rc=$?; echo "lds rc=$rc"; tail -3 $O/pytest_lds.log
[ $rc -eq 0 ] || exit 1
V="4:0:0"
for W in 6 5 4; do for kc in 16 22 24 26 32 43 64; do V="$V,4:4:$kc:$W"; done; done
timeout -k 10 600 python3 tools/jacobi_tune.py --n 256 --sweeps 198 --reps 3 --variants "$V" 2>&1 | grep -v amdgpu.ids > $O/jacobi_lds_256.txt; cat $O/jacobi_lds_256.txt
