#!/bin/bash
# round 4 batch a: (1) the exit-time teardown -- children with the default CU-masked copy stream and no fl_shutdown, plain and
# under rocprofv3 (tests/test_gpu_runtime.py), (2) the LDS corner-pair probe that decides how the field window is read,
# (3) this round's starting point: the driver's bench command with the profiled children on the product's stream set-up
set -o pipefail
O=gpurun_out/r04a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_runtime.py tests/test_gpu_contexts.py -x -q > $O/pytest_runtime.log 2>&1; rc=$?; echo "pytest runtime rc=$rc"; tail -5 $O/pytest_runtime.log
[ $rc -eq 0 ] || exit 1
./build/lds_pair_probe > $O/lds_pair_probe.txt 2>&1; echo "probe rc=$?"; cat $O/lds_pair_probe.txt
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>$O/bench_driver_cmd.err; echo "bench rc=$?"; cut -c1-400 $O/bench_driver_cmd.json; tail -5 $O/bench_driver_cmd.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_child -o run -- python3 tools/step_child.py --n 128 --steps 3 --warmup 1 --jacobi-iters 20 > $O/prof_child.log 2>&1; echo "prof child rc=$?"; tail -3 $O/prof_child.log | cut -c1-200
rm -f $O/prof_child/run_kernel_trace.csv
