#!/bin/bash
# round 3 batch ze: the bottom of the V-cycle in one launch -- parity (operator tests, full-size MGCG hashes, slab MGCG), bench line
O=gpurun_out/r03ze; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mgcg.py -x -q > $O/pytest_mgcg.log 2>&1; rc=$?; echo "mgcg rc=$rc"; tail -4 $O/pytest_mgcg.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 1000 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_solver.py tests/test_gpu_rccl_path.py -x -q -k "mgcg or reflection or next_row" > $O/pytest_full.log 2>&1; rc=$?; echo "full rc=$rc"; tail -3 $O/pytest_full.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 bench.py --projection mgcg --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/bench_mgcg.json 2>$O/bench_mgcg.err; cut -c1-230 $O/bench_mgcg.json
timeout -k 10 400 python3 bench.py --projection mgcg --steps 3 --warmup 1 --no-extra --no-cpu-baseline --fl-opt 17=0 > $O/bench_mgcg_nobottom.json 2>/dev/null; cut -c1-230 $O/bench_mgcg_nobottom.json
