#!/bin/bash
# round 3 batch y: the MGCG hashes at full size and the MGCG trajectories / slab runs on the three-sweep fp64 smoother
O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_solver.py tests/test_gpu_rccl_path.py tests/test_gpu_mgcg.py -x -q -k "mgcg or reflection or next_row" > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $O/pytest.log
