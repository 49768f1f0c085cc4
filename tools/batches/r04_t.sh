#!/bin/bash
# round 4 batch t: where an emulated config-4 rank (512 x 512 x 64 + 16 planes, rank 4 of 8) stands on the final tree
set -o pipefail
O=gpurun_out/r04t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 600 python3 bench.py --size 512 --emulate-slab 8 --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic > $O/emul8.json 2>$O/emul8.err; echo "emul rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04t/emul8.json").read())
print(d["value"], d["ms_per_step"], d.get("phase_ms_per_step"))
print(json.dumps(d.get("diagnostics"), indent=0)[:1500])
print({k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("extra", {}).items()})
PY
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --size 512 --emulate-slab 8 --steps 10 --warmup 3 --no-cpu-baseline --no-measure-traffic --no-extra --diag-steps 0 > $O/prof.log 2>&1; echo "prof rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04t/prof/**/run_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:30]:
        print(r["Name"][:100].ljust(100), "n=%6s avg_us=%8.1f pct=%5.1f" % (r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    print("total ms", tot / 1e6)
PY
