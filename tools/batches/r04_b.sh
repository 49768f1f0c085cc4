#!/bin/bash
# round 4 batch b: the z-marching gather kernels with the field window (FL_OPT_FIELD_WINDOW) -- parity, then A/B timing of
# the 256^3 step in both arithmetic variants, then the kernel table of the fast variant with the window
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_field_window.py -x -q > $O/pytest_window.log 2>&1; rc=$?; echo "pytest window rc=$rc"; tail -15 $O/pytest_window.log
timeout -k 10 900 python -m pytest tests/test_gpu_rccl_path.py -x -q -k "two_ranks or single_communicator or three_ranks" > $O/pytest_rccl.log 2>&1; echo "pytest rccl rc=$?"; tail -6 $O/pytest_rccl.log
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-measure-traffic"
for v in "exact:" "exact_win:--fl-opt 18=1" "fast:--fl-opt 11=1" "fast_win:--fl-opt 11=1 --fl-opt 18=1" "fast_win8:--fl-opt 11=1 --fl-opt 18=8" "fast_win32:--fl-opt 11=1 --fl-opt 18=32"; do
  tag=${v%%:*}; opt=${v#*:}
  timeout -k 10 300 $B $opt > $O/bench_$tag.json 2>$O/bench_$tag.err; echo "$tag rc=$?"
  python3 - $O/bench_$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], "Mvox/s", d["ms_per_step"], "ms", d.get("phase_ms_per_step"))
except Exception as e:
    print("   unreadable:", e)
PY
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fast_win -o run -- $B --fl-opt 11=1 --fl-opt 18=1 > $O/prof_fast_win.log 2>&1; echo "prof rc=$?"
rm -f $O/prof_fast_win/run_kernel_trace.csv
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04b/prof_fast_win/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(f"{r['Name'][:110]:110s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
PY
