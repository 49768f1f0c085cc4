#!/bin/bash
# tools/jacobi_pmc.sh <variant> <tag> -- HBM traffic of a Jacobi kernel variant (tools/jacobi_tune.py --variants syntax) from two
# separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -e
v=${1:-4:2:32}; tag=${2:-x}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -o run -- python3 tools/jacobi_tune.py --n ${N:-256} --variants $v --sweeps 20 --reps 1 > gpurun_out/pmc_${tag}_$c.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "jacobi" in r["Kernel_Name"] and r["Counter_Name"] == c]
    by = {}
    for r in rows:
        by.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in by.items():
        print(c, k, "launches", len(v), "mean_KB", sum(v) / len(v))
PY
