#!/bin/bash
# round 2, batch r: three-sweep kernel with loads one / two planes ahead (parity + timing)
set -e -o pipefail
O=gpurun_out/r02r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py tests/test_gpu_full_size.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/jacobi_tune.py --n 256 --sweeps 300 --reps 5 --variants "4:0:0:1,4:0:0:2,4:0:16:1,4:0:16:2,4:0:64:2,5:0:0:1" > $O/tune3_256.log 2>&1; cat $O/tune3_256.log
timeout -k 10 300 python tools/jacobi_tune.py --n 256 --nz 272 --sweeps 300 --reps 3 --variants "4:0:0:1,4:0:0:2" > $O/tune3_256x272.log 2>&1; cat $O/tune3_256x272.log
