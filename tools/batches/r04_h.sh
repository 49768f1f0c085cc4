#!/bin/bash
# round 4 batch h: checkpoint -- the whole GPU suite, then the default bench line (with the new extras)
set -o pipefail
O=gpurun_out/r04h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>$O/bench_driver_cmd.err; echo "bench rc=$?"; cut -c1-300 $O/bench_driver_cmd.json
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04h/bench_driver_cmd.json").read())
e = d.get("extra", {})
print("value", d["value"], d["ms_per_step"])
for k in ("survey_metric", "arithmetic_variants", "single_gpu_512_anchor", "fast_lerp_variant", "dead_state_elision"):
    print(k, json.dumps(e.get(k))[:700])
PY
