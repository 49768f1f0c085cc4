#!/bin/bash
# round 3 batch n: the LDS three-sweep kernel as the default at 256^3 -- projection + full-size hash tests, bench lines, counter traffic
set -o pipefail
O=gpurun_out/r03n; mkdir -p $O
python -m pytest tests/test_gpu_projection.py tests/test_gpu_full_size.py tests/test_gpu_solver.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/dev/null; cut -c1-300 $O/bench_driver_cmd.json
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2>/dev/null; cut -c1-300 $O/bench_default.json
bash tools/jacobi_pmc.sh 4:0:0 r03n_256_lds3 > $O/pmc_256_lds3.txt 2>&1; echo "pmc rc=$?"; tail -3 $O/pmc_256_lds3.txt
