#!/bin/bash
# round 4 batch f: march kernels as shipped (z-carried map stages in the two-field kernels only; window on by default with the one-fma
# lerps) -- parity incl. the solver-level fast-variant tests, A/B timing, counters, kernel table
set -o pipefail
O=gpurun_out/r04f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_field_window.py tests/test_gpu_solver.py tests/test_gpu_runtime.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-measure-traffic"
for v in "exact:" "fast_oneplane:--fl-opt 11=1 --fl-opt 18=0" "fast_default:--fl-opt 11=1" "fast_win16:--fl-opt 11=1 --fl-opt 18=16" "fast_win32:--fl-opt 11=1 --fl-opt 18=32" "fast_win64:--fl-opt 11=1 --fl-opt 18=64"; do
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
bash tools/pmc_gather.sh fastwin4 --fl-opt 11=1 > $O/pmc_fast_default.txt 2>&1; grep march $O/pmc_fast_default.txt | cut -c1-300
