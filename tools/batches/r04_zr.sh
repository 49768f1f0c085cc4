#!/bin/bash
# round 4 batch zr: what the ends-first pressure schedule costs an emulated config-4 rank on the compute side (phases, kernel table)
set -o pipefail
O=gpurun_out/r04zr; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --size 512 --emulate-slab 8 --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --no-extra --diag-steps 10"
for v in "on:" "off:--no-ends-first"; do
  tag=${v%%:*}; opt=${v#*:}
  timeout -k 10 300 $B $opt > $O/ends_$tag.json 2>$O/ends_$tag.err; echo "ends-first $tag rc=$?"
  python3 - $tag <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04zr/ends_%s.json" % sys.argv[1]).read())
print("   ", d["value"], d["ms_per_step"], (d.get("diagnostics") or {}).get("per_rank", [{}])[0].get("phase_ms_per_step"))
PY
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o run -- $B --diag-steps 0 $opt > $O/prof_$tag.log 2>&1
  python3 - $tag <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/r04zr/prof_%s/**/run_kernel_stats.csv" % sys.argv[1], recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:40]:
        if "jacobi" in r["Name"]:
            print("      ", r["Name"][:70].ljust(70), "n=%6s avg_us=%8.1f total_ms=%8.2f" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
