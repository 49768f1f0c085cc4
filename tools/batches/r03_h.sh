#!/bin/bash
# round 3 batch h: small-grid Jacobi policy (two-row kernel with one-round chunks), multi-rank bench diagnostics, policy 1 on slabs (GPU)
set -o pipefail
O=gpurun_out/r03h; mkdir -p $O
for shape in "128 128" "192 192" "256 64" "64 64" "96 96" "160 160"; do set -- $shape
  timeout -k 10 200 python3 tools/jacobi_tune.py --n $1 --nz $2 --sweeps 199 --variants "4:0:0,5:1:0,5:2:0,1:0:0" 2>&1 | grep -v amdgpu.ids >> $O/jacobi_small.txt; done; cat $O/jacobi_small.txt
python -m pytest tests/test_gpu_projection.py tests/test_gpu_bench_cli.py -x -q > $O/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -4 $O/pytest_a.log
python -m pytest tests/test_gpu_rccl_path.py tests/test_gpu_ops.py tests/test_gpu_example.py tests/test_gpu_full_size.py -x -q -k "reserved or bench_gpus_2 or nonfinite or cpp_rank or full_size_hashes" > $O/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -4 $O/pytest_b.log
SLAB_TEST_BLEND=1.0 MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29873 tests/slab_worker.py --backend gpu --dims 64 64 64 --L 1.0 --ghost 8 --steps 44 --iters 40 --dt-cells 2.0 --policy 1 > $O/policy1_slabs_64.log 2>&1; echo "policy1 rc=$?"; grep "^\[rank" $O/policy1_slabs_64.log | tail -6
timeout -k 10 300 python3 bench.py --size 128 --steps 60 --warmup 20 --no-extra > $O/bench_128.json 2>/dev/null; cut -c1-330 $O/bench_128.json
