#!/bin/bash
# round 3 batch j: counter traffic of the Jacobi kernel at 128^3 (new launch geometry) and on a config-5 rank's planes, the
# emulated config-5 rank with the CU-masked copy stream, BASELINE.md section 4's own CPU samples (128^3 x 20, 256^3 x 3), contexts test
set -o pipefail
O=gpurun_out/r03j; mkdir -p $O
python -m pytest tests/test_gpu_contexts.py tests/test_gpu_runtime.py -x -q > $O/pytest_ctx.log 2>&1; echo "ctx rc=$?"; tail -3 $O/pytest_ctx.log
N=128 bash tools/jacobi_pmc.sh 4:0:0 r03j_128 > $O/pmc_128.txt 2>&1; cat $O/pmc_128.txt | tail -4
timeout -k 10 300 python3 tools/jacobi_tune.py --n 1024 --nz 80 --sweeps 40 --reps 3 --variants "5:2:0" 2>&1 | grep -v amdgpu > $O/jacobi_1024x1024x80.txt; cat $O/jacobi_1024x1024x80.txt
CMD="bench.py --emulate-slab 8 --scene leapfrog --grid 1024 1024 512 --dump /tmp/dump5 --steps 10 --warmup 12 --no-extra --no-cpu-baseline --diag-steps 0"
timeout -k 10 400 python3 $CMD > $O/emul_cfg5_masked_copy_stream.json 2>/dev/null; echo "emul rc=$?"; cut -c150-330 $O/emul_cfg5_masked_copy_stream.json; rm -rf /tmp/dump5
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --cpu-steps 20 --cpu-256 --no-extra > $O/bench_cpu_samples.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_cpu_samples.json')); print(d['value'], d['cpu_baseline'])"
