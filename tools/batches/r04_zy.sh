#!/bin/bash
# round 4 batch zy: 128^3 Jacobi sweeps -- can the LDS multi-sweep kernels be made to apply with short chunks, and do they pay?
set -o pipefail
O=gpurun_out/r04zy; mkdir -p $O
timeout -k 10 200 python3 tools/jacobi_tune.py --n 128 --reps 7 --variants 5:2:4,4:0:8,4:0:12,4:0:16,4:6:8,4:6:16,4:2:4 > $O/n128.txt 2>&1; echo "rc=$?"; tail -10 $O/n128.txt
