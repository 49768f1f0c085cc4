#!/bin/bash
set -o pipefail
O=gpurun_out/r02f; mkdir -p $O
python -m pytest tests/test_gpu_projection.py tests/test_gpu_solver.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
# A/B: lean (rows=2) vs the previous two-row kernel (rows=3) vs one-row (rows=1), 256^3 and 128^3
python tools/jacobi_tune.py --n 256 --sweeps 200 --reps 5 --variants 4:2:32,4:3:32,4:1:32 2>&1 | tail -5
python tools/jacobi_tune.py --n 128 --sweeps 200 --reps 5 --variants 4:2:16,4:3:16,4:1:32,4:2:32 2>&1 | tail -5
timeout -k 10 300 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_256.json 2> $O/bench_256.err; echo "bench rc=$?"; cut -c1-200 $O/bench_256.json; grep -o '"roofline.*' $O/bench_256.json | cut -c1-400
