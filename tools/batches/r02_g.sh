#!/bin/bash
set -o pipefail
O=gpurun_out/r02g; mkdir -p $O
python -m pytest tests/test_gpu_projection.py tests/test_gpu_full_size.py tests/test_slab_multirank.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python tools/jacobi_tune.py --n 512 --sweeps 40 --reps 3 --variants 4:2:86,4:3:86,4:2:64,4:2:43 2>&1 | tail -5
python tools/jacobi_tune.py --n 512 --nz 80 --sweeps 100 --reps 3 --variants 4:2:0,4:3:0,4:2:40,4:2:20 2>&1 | tail -5
python tools/jacobi_tune.py --n 256 --sweeps 200 --reps 3 --variants 4:2:32,4:2:16,4:2:64 2>&1 | tail -4
timeout -k 10 300 python bench.py --size 512 --steps 20 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_512.json 2> $O/bench_512.err; cut -c1-200 $O/bench_512.json; grep -o '"roofline.*' $O/bench_512.json | cut -c1-300
timeout -k 10 300 python bench.py --size 512 --emulate-slab 8 --steps 60 --warmup 20 --no-extra > $O/emul_512_r8_g8.json 2> $O/emul.err; cut -c1-330 $O/emul_512_r8_g8.json
