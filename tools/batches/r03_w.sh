#!/bin/bash
# round 3 batch w: the whole GPU suite on the round's state, bench lines (driver command, default), rocprofv3 kernel stats of
# the driver command, counter traffic of the three- and four-sweep LDS kernels
set -o pipefail
O=gpurun_out/r03w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>$O/bench_driver_cmd.err; cut -c1-300 $O/bench_driver_cmd.json
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2>/dev/null; cut -c1-300 $O/bench_default.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/prof_driver.log 2>&1; echo "prof rc=$?"; tail -1 $O/prof_driver.log | cut -c1-200
rm -f $O/prof_driver/run_kernel_trace.csv
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03w/prof_driver/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:96]:96s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
PY
bash tools/jacobi_pmc.sh 4:0:0 r03w_256_lds3 > $O/pmc_256_lds3.txt 2>&1; echo "pmc3 rc=$?"; tail -2 $O/pmc_256_lds3.txt
bash tools/jacobi_pmc.sh 4:6:32:24 r03w_256_lds4 > $O/pmc_256_lds4.txt 2>&1; echo "pmc4 rc=$?"; tail -2 $O/pmc_256_lds4.txt
