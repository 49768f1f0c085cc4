#!/bin/bash
# round 3 batch a: GPU suite on the round-2 state, then BASELINE config 5's rank geometry for the first time:
# emulated rank 4 of 8 of 1024 x 1024 x 512 (leapfrog scene, dump every frame) + per-kernel table of the same command
set -o pipefail
O=gpurun_out/r03a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
CMD="bench.py --emulate-slab 8 --scene leapfrog --grid 1024 1024 512 --dump /tmp/dump5 --steps 10 --warmup 12 --no-extra --no-cpu-baseline"
timeout -k 10 400 python3 $CMD > $O/emul_cfg5.json 2> $O/emul_cfg5.err; echo "emul cfg5 rc=$?"; cut -c1-400 $O/emul_cfg5.json; tail -5 $O/emul_cfg5.err
ls -la /tmp/dump5 | head -8
rm -rf /tmp/dump5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -o run -- python3 $CMD > $O/prof_cfg5.log 2>&1; echo "prof rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03a/prof_cfg5/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:40]:
    print(f"{r['Name'][:110]:110s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
rm -rf /tmp/dump5; rm -f $O/prof_cfg5/*/*kernel_trace.csv $O/prof_cfg5/*/*agent_info.csv 2>/dev/null
