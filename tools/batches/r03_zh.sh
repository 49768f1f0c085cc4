#!/bin/bash
# round 3 batch zh (= zc on the later tree): the whole GPU suite on the round's final tree, then every bench line DESIGN.md section 6 quotes and the
# rocprofv3 kernel stats of the driver's command
set -o pipefail
O=gpurun_out/r03zh; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>$O/bench_driver_cmd.err; cut -c1-260 $O/bench_driver_cmd.json
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2>/dev/null; cut -c1-260 $O/bench_default.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/prof_driver.log 2>&1; echo "prof rc=$?"; tail -1 $O/prof_driver.log | cut -c1-120
rm -f $O/prof_driver/run_kernel_trace.csv
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03zh/prof_driver/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(f"{r['Name'][:96]:96s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
PY
timeout -k 10 300 python3 bench.py --size 128 --steps 100 --warmup 20 --no-extra --no-cpu-baseline > $O/bench_128.json 2>/dev/null; cut -c1-200 $O/bench_128.json
timeout -k 10 400 python3 bench.py --projection mgcg --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/bench_mgcg.json 2>/dev/null; cut -c1-220 $O/bench_mgcg.json
timeout -k 10 400 python3 bench.py --scheme reflection --projection mgcg --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/bench_reflection_mgcg.json 2>/dev/null; cut -c1-220 $O/bench_reflection_mgcg.json
