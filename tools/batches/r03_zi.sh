#!/bin/bash
# round 3 batch zi: kernel stats of the 512^3 bench (the two-segment Jacobi kernel's average against bench's own timer), the MGCG-mode
# kernel table on the final tree, the emulated config-5 rank on the final tree
set -o pipefail
O=gpurun_out/r03zi; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_512 -o run -- python3 bench.py --size 512 --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/prof_512.log 2>&1; echo "prof512 rc=$?"; tail -1 $O/prof_512.log | cut -c1-200
rm -f $O/prof_512/run_kernel_trace.csv
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mgcg -o run -- python3 bench.py --projection mgcg --steps 3 --warmup 0 --no-extra --no-cpu-baseline > $O/prof_mgcg.log 2>&1; echo "profmg rc=$?"
rm -f $O/prof_mgcg/run_kernel_trace.csv
python3 - <<'PY'
import csv, glob
for tag, steps in (("prof_512", 10), ("prof_mgcg", 3)):
    f = glob.glob(f"gpurun_out/r03zi/{tag}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    print(tag, "total ms/step", sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6)
    for r in rows[:8]:
        print(f"  {r['Name'][:84]:84s} n={int(r['Calls'])/steps:8.1f}/step avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
CMD="bench.py --emulate-slab 8 --scene leapfrog --grid 1024 1024 512 --dump /tmp/dump5 --steps 10 --warmup 12 --no-extra --no-cpu-baseline --diag-steps 6"
timeout -k 10 600 python3 $CMD > $O/emul_cfg5.json 2> $O/emul_cfg5.err; echo "emul cfg5 rc=$?"; cut -c1-400 $O/emul_cfg5.json; rm -rf /tmp/dump5
