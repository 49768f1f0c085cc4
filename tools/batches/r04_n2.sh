#!/bin/bash
# round 4 batch n2: kernel table of the MGCG-mode step with the level-0 vector updates fused (FL_OPT_MGCG_FUSE)
set -o pipefail
O=gpurun_out/r04n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --gpus 1 --projection mgcg --no-cpu-baseline --no-measure-traffic --size 256 --steps 3 --warmup 1 > $O/prof.log 2>&1; echo "prof rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04n/prof/**/run_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:30]:
        print(r["Name"][:100].ljust(100), "n=%6s avg_us=%8.1f pct=%5.1f" % (r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    print("total ms", tot / 1e6)
PY
