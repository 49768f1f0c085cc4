#!/bin/bash
# round 3 batch z: the two-segment three-sweep kernel for rows of 260 .. 512 floats -- parity, 512^3 timing, 512^3 property test,
# config-4 rank geometry on two stand-in ranks, the 512^3 anchor leg of the default bench line
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "two_segment or lds" > $O/pytest_proj.log 2>&1; rc=$?; echo "proj rc=$rc"; tail -3 $O/pytest_proj.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/jacobi_tune.py --n 512 --sweeps 48 --reps 3 --variants "4:0:0,4:5:0,4:0:64,4:0:32" 2>&1 | grep -v amdgpu.ids > $O/jacobi_512.txt; cat $O/jacobi_512.txt
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_config5.py -x -q -k "512 or config4" > $O/pytest_512.log 2>&1; rc=$?; echo "512 rc=$rc"; tail -3 $O/pytest_512.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2>$O/bench_driver.err; python3 -c "
import json;l=json.load(open('$O/bench_driver.json'));print(l['value'],l['ms_per_step'],l['extra'].get('single_gpu_512_anchor'))"
timeout -k 10 400 python3 bench.py --size 512 --steps 10 --warmup 4 --no-extra --no-cpu-baseline > $O/bench_512.json 2>/dev/null; cut -c1-250 $O/bench_512.json
