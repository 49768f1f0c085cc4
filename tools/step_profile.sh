#!/bin/bash
# tools/step_profile.sh <tag> -- kernel-level profile of 40 bench steps (rocprofv3 --kernel-trace --stats) on the GPU box;
# output under gpurun_out/prof_<tag>/.  Run through gpurun:  gpurun -- 'bash tools/step_profile.sh mytag'
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o run -- python3 bench.py --steps 40 --warmup 60 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print(f"{r['Name'][:96]:96s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
PY
tail -1 gpurun_out/prof_$tag.log | cut -c1-300
