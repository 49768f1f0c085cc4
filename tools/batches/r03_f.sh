#!/bin/bash
# round 3 batch f: config-5 tests on the off-axis (NaN-free) leapfrog scene, diagnostics of bench --gpus / --emulate-slab, the C++ rank
# driver, reserved-CU parity on the stand-ins, 128^3 Jacobi with chunks of 2-4 planes, the emulated config-5 rank with its legs
set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_config5.py tests/test_gpu_bench_cli.py tests/test_gpu_example.py -x -q > $O/pytest_a.log 2>&1; rc=$?; echo "pytest a rc=$rc"; tail -15 $O/pytest_a.log
python -m pytest tests/test_gpu_rccl_path.py tests/test_gpu_ops.py -x -q -k "reserved or bench_gpus_2 or nonfinite or emit" > $O/pytest_b.log 2>&1; rc=$?; echo "pytest b rc=$rc"; tail -15 $O/pytest_b.log
timeout -k 10 300 python3 tools/jacobi_tune.py --n 128 --sweeps 199 --variants "4:0:0,5:2:2,5:2:3,5:2:4,5:2:5,4:2:4,4:2:5" > $O/jacobi_tune_128.txt 2>&1; grep -v amdgpu.ids $O/jacobi_tune_128.txt
CMD="bench.py --emulate-slab 8 --scene leapfrog --grid 1024 1024 512 --dump /tmp/dump5 --steps 10 --warmup 12 --no-extra --no-cpu-baseline --diag-steps 6"
timeout -k 10 600 python3 $CMD > $O/emul_cfg5.json 2> $O/emul_cfg5.err; echo "emul cfg5 rc=$?"; cut -c1-600 $O/emul_cfg5.json; tail -3 $O/emul_cfg5.err
rm -rf /tmp/dump5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -o run -- python3 $CMD --diag-steps 0 > $O/prof_cfg5.log 2>&1; echo "prof rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03f/prof_cfg5/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{r['Name'][:110]:110s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
rm -rf /tmp/dump5; rm -f $O/prof_cfg5/*kernel_trace.csv $O/prof_cfg5/*/*kernel_trace.csv
