#!/bin/bash
# round 3 batch x: fp64 three-sweep LDS smoother -- parity, timing against the pair kernel, MGCG bench line
O=gpurun_out/r03x; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mgcg.py -x -q > $O/pytest_mgcg.log 2>&1; rc=$?; echo "mgcg rc=$rc"; tail -5 $O/pytest_mgcg.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/smooth_tune.py --n 256 --sweeps 32 --reps 3 --variants "1:0:0:0,1:5:0:0,1:0:0:14,1:0:64:0,1:0:16:0" 2>&1 | grep -v amdgpu.ids > $O/smooth_256.txt; cat $O/smooth_256.txt
timeout -k 10 600 python3 bench.py --projection mgcg --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/bench_mgcg.json 2>$O/bench_mgcg.err; cut -c1-400 $O/bench_mgcg.json
