#!/bin/bash
# round 4 batch g: march kernels, tap-by-tap form + z-carried map stages in the two-field kernels; then the reference binary's own
# grid and scene (100 x 200 x 200, h = 0.002: not a power of two) with kernel tables
set -o pipefail
O=gpurun_out/r04g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_field_window.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-measure-traffic"
show() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], "Mvox/s", d["ms_per_step"], "ms", d["config"]["workload"][:90])
except Exception as e:
    print("   unreadable:", e)
PY
}
for v in "fast_oneplane:--fl-opt 11=1 --fl-opt 18=0" "fast_default:--fl-opt 11=1" "fast_win32:--fl-opt 11=1 --fl-opt 18=32"; do
  tag=${v%%:*}; opt=${v#*:}
  timeout -k 10 300 $B $opt > $O/bench_$tag.json 2>$O/bench_$tag.err; echo "$tag rc=$?"; show $O/bench_$tag.json
done
bash tools/pmc_gather.sh fastwin5 --fl-opt 11=1 > $O/pmc_fast_default.txt 2>&1; grep march $O/pmc_fast_default.txt | cut -c1-120
R="python3 bench.py --gpus 1 --reference-scene --steps 20 --warmup 12 --no-cpu-baseline --no-measure-traffic"
timeout -k 10 300 $R > $O/ref_bimocq_jacobi.json 2>$O/ref_bimocq_jacobi.err; echo "ref bimocq+jacobi rc=$?"; show $O/ref_bimocq_jacobi.json; tail -2 $O/ref_bimocq_jacobi.err
timeout -k 10 300 $R --scheme reflection --projection mgcg --steps 6 --warmup 12 > $O/ref_reflection_mgcg.json 2>$O/ref_reflection_mgcg.err; echo "ref reflection+mgcg rc=$?"; show $O/ref_reflection_mgcg.json; tail -2 $O/ref_reflection_mgcg.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ref_bj -o run -- $R > $O/prof_ref_bj.log 2>&1; echo "prof rc=$?"
rm -f $O/prof_ref_bj/run_kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ref_rm -o run -- $R --scheme reflection --projection mgcg --steps 6 --warmup 12 > $O/prof_ref_rm.log 2>&1; echo "prof rc=$?"
rm -f $O/prof_ref_rm/run_kernel_trace.csv
python3 - <<'PY'
import csv, glob
for d in ("prof_ref_bj", "prof_ref_rm"):
    f = glob.glob(f"gpurun_out/r04g/{d}/**/*kernel_stats.csv", recursive=True)[0]
    print("==", d)
    for r in list(csv.DictReader(open(f)))[:16]:
        print(f"{r['Name'][:100]:100s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
PY
