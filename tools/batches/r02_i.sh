#!/bin/bash
set -o pipefail
O=gpurun_out/r02i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_projection.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python tools/jacobi_tune.py --n 256 --sweeps 198 --reps 5 --variants 4:0:0,5:0:0,5:3:0 2>&1 | tail -4
python tools/jacobi_tune.py --n 512 --nz 80 --sweeps 100 --reps 3 --variants 5:0:0,5:3:0 2>&1 | tail -3
python tools/jacobi_tune.py --n 512 --sweeps 40 --reps 3 --variants 5:0:0 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_256.json 2> $O/bench_256.err; echo "bench rc=$?"; cut -c1-200 $O/bench_256.json
bash tools/jacobi_pmc.sh 4:0:0 r02i_256_lean3r > $O/pmc_3r.log 2>&1; tail -3 $O/pmc_3r.log
bash tools/jacobi_pmc.sh 5:0:0 r02i_256_lean2r > $O/pmc_2r.log 2>&1; tail -3 $O/pmc_2r.log
