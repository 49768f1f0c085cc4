#!/bin/bash
# round 3 batch e: where the 1024-wide solver run leaves the oracle (field by field), and the 128^3 Jacobi launch geometries
set -o pipefail
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 900 python3 tools/diag_rows.py --grid 1024 1024 16 --steps 2 --iters 200 > $O/diag_1024x1024x16.txt 2>&1; echo "diag rc=$?"; cat $O/diag_1024x1024x16.txt | grep -v amdgpu.ids
timeout -k 10 300 python3 tools/jacobi_tune.py --n 128 --sweeps 199 --variants "4:0:0,4:2:4,4:2:6,4:2:8,4:2:12,4:2:16,5:2:4,5:2:8,5:2:16,5:1:4,5:1:8,5:1:16,1:0:0" > $O/jacobi_tune_128.txt 2>&1; cat $O/jacobi_tune_128.txt | grep -v amdgpu.ids
