#!/bin/bash
# round 2, batch r: alpha*div formed once per value in the fused Jacobi kernels (parity + timing)
set -e -o pipefail
O=gpurun_out/r02r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py tests/test_gpu_full_size.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/jacobi_tune.py --n 256 --sweeps 300 --reps 5 --variants "4:0:0,5:0:0:1" > $O/tune_256.log 2>&1; cat $O/tune_256.log
timeout -k 10 300 python tools/jacobi_tune.py --n 512 --nz 80 --sweeps 100 --reps 3 --variants "5:0:0:1" > $O/tune_512x80.log 2>&1; cat $O/tune_512x80.log
timeout -k 10 300 python tools/jacobi_tune.py --n 512 --sweeps 60 --reps 3 --variants "5:0:0:2" > $O/tune_512.log 2>&1; cat $O/tune_512.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd.json 2> $O/bench_driver.err; cut -c1-200 $O/bench_driver_cmd.json
