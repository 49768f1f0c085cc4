#!/bin/bash
# round 3 batch zf: kernel table of the MGCG-mode step
O=gpurun_out/r03zf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mgcg -o run -- python3 bench.py --projection mgcg --steps 3 --warmup 0 --no-extra --no-cpu-baseline > $O/prof_mgcg.log 2>&1; echo "prof rc=$?"
rm -f $O/prof_mgcg/run_kernel_trace.csv
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03zf/prof_mgcg/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total ms per step", tot / 3e6)
for r in rows[:26]:
    print(f"{r['Name'][:88]:88s} n={int(r['Calls'])/3:8.1f}/step avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/3e6:7.2f}")
PY
