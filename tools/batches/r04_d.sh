#!/bin/bash
# round 4 batch d: the march kernels with batched, branch-free taps -- parity, timing, counters
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_field_window.py -x -q > $O/pytest_window.log 2>&1; rc=$?; echo "pytest window rc=$rc"; tail -5 $O/pytest_window.log
[ $rc -eq 0 ] || exit 1
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-measure-traffic"
for v in "fast:--fl-opt 11=1" "fast_win:--fl-opt 11=1 --fl-opt 18=1" "fast_win32:--fl-opt 11=1 --fl-opt 18=32" "exact_win:--fl-opt 18=1"; do
  tag=${v%%:*}; opt=${v#*:}
  timeout -k 10 300 $B $opt > $O/bench_$tag.json 2>$O/bench_$tag.err; echo "$tag rc=$?"
  python3 - $O/bench_$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], "Mvox/s", d["ms_per_step"], "ms")
except Exception as e:
    print("   unreadable:", e)
PY
done
bash tools/pmc_gather.sh fastwin2 --fl-opt 11=1 --fl-opt 18=1 > $O/pmc_fastwin.txt 2>&1; grep march $O/pmc_fastwin.txt | cut -c1-300
