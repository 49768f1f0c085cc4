#!/bin/bash
# round 4 batch i: the tabled structured look-up on other spacings (parity, then the reference binary's own grid again) and the
# first runs of the slab-shared multigrid solver on the RCCL stand-ins
set -o pipefail
O=gpurun_out/r04i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q > $O/pytest_ops.log 2>&1; rc=$?; echo "pytest ops rc=$rc"; tail -8 $O/pytest_ops.log
timeout -k 10 900 python -m pytest tests/test_gpu_rccl_path.py -x -q -k "not_a_power_of_two or multigrid_levels_shared" > $O/pytest_rccl.log 2>&1; rc2=$?; echo "pytest rccl rc=$rc2"; tail -30 $O/pytest_rccl.log
R="python3 bench.py --gpus 1 --reference-scene --steps 20 --warmup 12 --no-cpu-baseline --no-measure-traffic"
timeout -k 10 300 $R > $O/ref_bimocq_jacobi.json 2>$O/ref_bimocq_jacobi.err; echo "ref bimocq+jacobi rc=$?"; cut -c1-120 $O/ref_bimocq_jacobi.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ref_bj -o run -- $R > $O/prof_ref_bj.log 2>&1; echo "prof rc=$?"
rm -f $O/prof_ref_bj/run_kernel_trace.csv
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04i/prof_ref_bj/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
PY
