#!/bin/bash
# round 2, batch t: three-sweep launches inside gpu_projection_jacobi (parity) and the reflection scheme's step
set -e -o pipefail
O=gpurun_out/r02t; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_projection.py tests/test_gpu_solver.py tests/test_gpu_ops.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python bench.py --scheme reflection --steps 40 --warmup 10 --no-cpu-baseline > $O/reflection_jacobi.json 2> $O/reflection_jacobi.err || { tail -20 $O/reflection_jacobi.err; exit 1; }
cut -c1-330 $O/reflection_jacobi.json
timeout -k 10 400 python bench.py --scheme reflection --projection mgcg --steps 6 --warmup 2 --no-cpu-baseline > $O/reflection_mgcg.json 2> $O/reflection_mgcg.err || { tail -20 $O/reflection_mgcg.err; exit 1; }
cut -c1-330 $O/reflection_mgcg.json
