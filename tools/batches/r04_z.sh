#!/bin/bash
# round 4 batch z: would the three-sweep LDS smoother pay at the V-cycle's second level (127^3, odd rows -> today the two-sweep
# register kernel)?  Proxy: the same 36 sweeps on 130^3 (the narrowest rows mg_lds3_kernel takes) through both kernels
set -o pipefail
O=gpurun_out/r04z; mkdir -p $O
timeout -k 10 200 python3 tools/smooth_tune.py --n 130 --sweeps 36 --reps 6 --variants 1:0:9:1,1:0:12:1,1:0:18:1,1:5:0:1 > $O/n130.txt 2>&1; echo "130 rc=$?"; cat $O/n130.txt | tail -8
timeout -k 10 200 python3 tools/smooth_tune.py --n 127 --sweeps 36 --reps 6 --variants 1:0:0:1,1:5:0:1 > $O/n127.txt 2>&1; echo "127 rc=$?"; cat $O/n127.txt | tail -5
