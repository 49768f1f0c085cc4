#!/bin/bash
# round 4 batch p: the wave-per-row form of the fused level-0 kernels (rows of 256): parity, then the MGCG step against the
# two-rows-per-thread form (FL_OPT_MGCG_FUSE = 3) and with z-chunks of 16 planes; kernel table
set -o pipefail
O=gpurun_out/r04p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_mgcg.py -x -q -k "vector_updates" > $O/pytest_mgcg.log 2>&1; rc=$?; echo "pytest mgcg rc=$rc"; tail -6 $O/pytest_mgcg.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -x -q -k "mgcg" > $O/pytest_hash.log 2>&1; rc=$?; echo "pytest mgcg hashes rc=$rc"; tail -4 $O/pytest_hash.log
[ $rc -eq 0 ] || exit 1
show() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], "Mvox/s", d["ms_per_step"], "ms", d.get("phase_ms_per_step"))
except Exception as e:
    print("   unreadable:", e)
PY
}
M="python3 bench.py --gpus 1 --projection mgcg --no-cpu-baseline --no-measure-traffic"
timeout -k 10 600 $M --size 256 --steps 5 --warmup 2 > $O/mg_256_rows.json 2>$O/mg_256_rows.err; echo "256 wave-per-row rc=$?"; show $O/mg_256_rows.json
timeout -k 10 600 $M --size 256 --steps 5 --warmup 2 --fl-opt 20=3 > $O/mg_256_tworows.json 2>$O/mg_256_tworows.err; echo "256 two-rows-per-thread rc=$?"; show $O/mg_256_tworows.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --gpus 1 --projection mgcg --no-cpu-baseline --no-measure-traffic --size 256 --steps 3 --warmup 1 > $O/prof.log 2>&1; echo "prof rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04p/prof/**/run_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:16]:
        print(r["Name"][:100].ljust(100), "n=%6s avg_us=%8.1f pct=%5.1f" % (r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    print("total ms", tot / 1e6)
PY
