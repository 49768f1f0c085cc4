#!/bin/bash
# round 4 batch x: soak -- 256^3 rising smoke for 1000 steps (the plume reaches the top wall), MGCG mode for 60, reflection for 200:
# no non-finite velocity, no latched error, step time stable
set -o pipefail
O=gpurun_out/r04x; mkdir -p $O
B="python3 bench.py --gpus 1 --no-cpu-baseline --no-measure-traffic --no-extra"
run() { tag=$1; shift; timeout -k 10 900 $B "$@" > $O/$tag.json 2>$O/$tag.err; echo "$tag rc=$?"
  python3 - $O/$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], d["ms_per_step"], "steps", d["steps"], "nonfinite", d["config"]["nonfinite_velocity_seen"])
except Exception as e:
    print("   unreadable", e)
PY
}
run soak_1000 --steps 1000 --warmup 5
run soak_mgcg_60 --projection mgcg --steps 60 --warmup 2
run soak_reflection_200 --scheme reflection --steps 200 --warmup 5
run soak_128_2000 --size 128 --steps 2000 --warmup 5
