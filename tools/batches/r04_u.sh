#!/bin/bash
# round 4 batch u: ghost depth of an emulated config-4 rank (512 x 512 x 64 owned planes, rank 4 of 8): 6, 8 (default), 9, 12
set -o pipefail
O=gpurun_out/r04u; mkdir -p $O
for g in 8 6 9 12; do
  timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 8 --ghost $g --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --no-extra --diag-steps 5 > $O/emul_g$g.json 2>$O/emul_g$g.err; echo "ghost $g rc=$?"
  python3 - $g <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04u/emul_g%s.json" % sys.argv[1]).read())
    ph = (d.get("diagnostics") or {}).get("per_rank", [{}])[0].get("phase_ms_per_step")
    print("   ", d["value"], d["ms_per_step"], ph)
except Exception as e:
    print("   unreadable", e)
PY
done
