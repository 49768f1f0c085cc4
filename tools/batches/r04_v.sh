#!/bin/bash
# round 4 batch v: the fused level-0 kernels on rows of 512 cells at full size (512^3 MGCG step), fusion on / off
set -o pipefail
O=gpurun_out/r04v; mkdir -p $O
M="python3 bench.py --gpus 1 --projection mgcg --no-cpu-baseline --no-measure-traffic --no-extra --size 512 --steps 2 --warmup 1"
for v in "on:" "off:--fl-opt 20=0"; do
  tag=${v%%:*}; opt=${v#*:}
  timeout -k 10 500 $M $opt > $O/mg512_$tag.json 2>$O/mg512_$tag.err; echo "512 fuse $tag rc=$?"
  python3 - $tag <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04v/mg512_%s.json" % sys.argv[1]).read())
    print("   ", d["value"], d["ms_per_step"], d["config"].get("nonfinite_velocity_seen"))
except Exception as e:
    print("   unreadable", e)
PY
done
