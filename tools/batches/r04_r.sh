#!/bin/bash
# round 4 batch r: the evidence of the round's final state -- the driver's command and the default run (bench lines), the kernel
# table of the driver's command, smoke(), the MGCG and reflection + MGCG lines, 128^3
set -o pipefail
O=gpurun_out/r04r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
show() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], d["unit"], d["ms_per_step"], "ms", "frac", d.get("roofline", {}).get("frac"), d.get("phase_ms_per_step"))
except Exception as e:
    print("   unreadable:", e)
PY
}
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_command.json 2>$O/driver_command.err; echo "driver command rc=$?"; show $O/driver_command.json
timeout -k 10 900 python3 bench.py > $O/default.json 2>$O/default.err; echo "default rc=$?"; show $O/default.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-measure-traffic > $O/prof.log 2>&1; echo "prof rc=$?"
M="python3 bench.py --gpus 1 --no-cpu-baseline --no-measure-traffic --no-extra"
timeout -k 10 600 $M --projection mgcg --steps 5 --warmup 2 > $O/mgcg.json 2>$O/mgcg.err; echo "mgcg rc=$?"; show $O/mgcg.json
timeout -k 10 600 $M --projection mgcg --scheme reflection --steps 4 --warmup 2 > $O/reflection_mgcg.json 2>$O/reflection_mgcg.err; echo "reflection+mgcg rc=$?"; show $O/reflection_mgcg.json
timeout -k 10 300 $M --size 128 --steps 100 --warmup 20 > $O/bench_128.json 2>$O/bench_128.err; echo "128 rc=$?"; show $O/bench_128.json
timeout -k 10 300 $M --reference-scene --steps 20 --warmup 12 > $O/ref_scene.json 2>$O/ref_scene.err; echo "reference scene rc=$?"; show $O/ref_scene.json
