#!/bin/bash
# round 3 batch b: config-5 tests (hashes at 1024 x 1024 x 32, two ranks of 1024 x 1024 x 32 on the stream-ordered stand-in, the
# 2 GiB refusal), portable emitter trigonometry, what the CU mask really excludes, and why dmc_kernel took 10.9 ms on the
# emulated config-5 rank: the leapfrog scene on one GPU at 1024 x 1024 x 64 with a kernel table, the emulated rank without dump
set -o pipefail
O=gpurun_out/r03b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
build/cu_mask_probe 8 > $O/cu_mask_probe_8.txt 2>&1; build/cu_mask_probe 16 >> $O/cu_mask_probe_8.txt 2>&1; cat $O/cu_mask_probe_8.txt
python -m pytest tests/test_gpu_config5.py tests/test_gpu_ops.py tests/test_gpu_solver.py -x -q -k "config5 or full_grid or two_ranks or emit or smoke" > $O/pytest_cfg5.log 2>&1; rc=$?; echo "pytest cfg5 rc=$rc"; tail -15 $O/pytest_cfg5.log
CMD="bench.py --scene leapfrog --grid 1024 1024 64 --steps 5 --warmup 12 --no-extra --no-cpu-baseline"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_leap -o run -- python3 $CMD > $O/prof_leap.log 2>&1; echo "prof rc=$?"; tail -1 $O/prof_leap.log | cut -c1-300
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03b/prof_leap/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f"{r['Name'][:110]:110s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
rm -f $O/prof_leap/*/*kernel_trace.csv $O/prof_leap/*kernel_trace.csv
timeout -k 10 400 python3 bench.py --emulate-slab 8 --scene leapfrog --grid 1024 1024 512 --steps 10 --warmup 12 --no-extra --no-cpu-baseline > $O/emul_cfg5_nodump.json 2> $O/emul_cfg5_nodump.err; echo "emul nodump rc=$?"; cut -c150-420 $O/emul_cfg5_nodump.json
