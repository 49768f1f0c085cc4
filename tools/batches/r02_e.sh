#!/bin/bash
set -o pipefail
O=gpurun_out/r02e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout -k 10 300 python bench.py --steps 60 --warmup 20 > $O/bench_256.json 2> $O/bench_256.err; echo "bench rc=$?"; cut -c1-330 $O/bench_256.json
timeout -k 10 300 python bench.py --size 512 --steps 20 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_512.json 2> $O/bench_512.err; cut -c1-330 $O/bench_512.json
timeout -k 10 300 python bench.py --size 512 --emulate-slab 8 --steps 60 --warmup 20 --no-extra > $O/emul_512_r8_g8.json 2> $O/emul.err; cut -c1-330 $O/emul_512_r8_g8.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profemul -o run -- python3 bench.py --size 512 --emulate-slab 8 --steps 20 --warmup 30 --no-extra > $O/profemul.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02e/profemul/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("emulated rank: total kernel ms per step", tot / 50 / 1e6)
for r in rows:
    n=r['Name']
    if any(k in n for k in ('box_copy','wall_fixup','clamp_box','fillBuffer','copyBuffer','maps_quarter')):
        print(f"{n[:90]:90s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/50/1e6:6.3f}")
PY
