#!/bin/bash
set -o pipefail
O=gpurun_out/r02k; mkdir -p $O
python -m pytest tests/test_gpu_projection.py tests/test_gpu_full_size.py tests/test_slab_multirank.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/jacobi_tune.py --n 256 --sweeps 198 --reps 5 --variants 5:0:0:2,5:0:0:1,5:0:16:2,5:0:16:1,4:0:0 2>&1 | tail -6
python tools/jacobi_tune.py --n 512 --nz 80 --sweeps 100 --reps 3 --variants 5:0:0:2,5:0:0:1 2>&1 | tail -3
python tools/jacobi_tune.py --n 512 --sweeps 40 --reps 3 --variants 5:0:0:2,5:0:0:1,5:0:43:2 2>&1 | tail -4
