#!/bin/bash
# round 4 batch l: the z-march WITHOUT the field window on the exact arithmetic (map ring + carried plane stages + direct gathers)
set -o pipefail
O=gpurun_out/r04l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BQ_TEST_MARCH=1 timeout -k 10 900 python -m pytest tests/test_gpu_field_window.py -x -q -k "march_without" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
B="python3 bench.py --gpus 1 --steps 30 --warmup 8 --no-extra --no-cpu-baseline --no-measure-traffic"
show() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], "Mvox/s", d["ms_per_step"], "ms")
except Exception as e:
    print("   unreadable:", e)
PY
}
for v in "exact_oneplane:" "exact_march16:--fl-opt 20=16" "exact_march32:--fl-opt 20=32" "exact_march_auto:--fl-opt 20=1" "exact_oneplane2:"; do
  tag=${v%%:*}; opt=${v#*:}
  timeout -k 10 300 $B $opt > $O/bench_$tag.json 2>$O/bench_$tag.err; echo "$tag rc=$?"; show $O/bench_$tag.json
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- $B --fl-opt 20=16 > $O/prof.log 2>&1; echo "prof rc=$?"
rm -f $O/prof/run_kernel_trace.csv
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04l/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
PY
