#!/bin/bash
# round 4 batch p2: wave-per-row fused kernels with blocks of 14 rows (16 waves) against 8 (10 waves): kernel averages
set -o pipefail
O=gpurun_out/r04p2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "w8:" "w8nt:--fl-opt 6=15" "w8b:"; do
  tag=${v%%:*}; opt=${v#*:}
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o run -- python3 bench.py --gpus 1 --projection mgcg --no-cpu-baseline --no-measure-traffic --size 256 --steps 3 --warmup 1 $opt > $O/prof_$tag.log 2>&1; echo "prof $tag rc=$?"
  python3 - $tag <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/r04p2/prof_%s/**/run_kernel_stats.csv" % sys.argv[1], recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:40]:
        if "fused" in r["Name"] or "lds3_kernel<8, 3, false, false" in r["Name"]:
            print(r["Name"][:90].ljust(90), "n=%6s avg_us=%8.1f" % (r["Calls"], float(r["AverageNs"]) / 1e3))
    print("total ms", tot / 1e6)
PY
done
