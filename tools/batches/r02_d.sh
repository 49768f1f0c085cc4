#!/bin/bash
# round 2, batch d: full GPU suite, bench + kernel stats after the fp32 quarter-weight map lerps, issue-rate table,
# fast-lerp deviation at 128^3, emulated config-4 rank profile, 512^3 Jacobi traffic counters
set -o pipefail
O=gpurun_out/r02d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout -k 10 300 python bench.py --steps 60 --warmup 20 > $O/bench_256.json 2> $O/bench_256.err; echo "bench rc=$?"; cut -c1-330 $O/bench_256.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof256 -o run -- python3 bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-extra > $O/prof256.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02d/prof256/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:30]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
PY
timeout -k 10 120 ./build/valu_rate > $O/valu_rate.txt 2>&1; tail -50 $O/valu_rate.txt
timeout -k 10 300 python tools/fast_lerp_deviation.py --size 128 --steps 200 --out $O/fast_lerp_deviation_128.json > $O/fastlerp.log 2>&1; tail -4 $O/fastlerp.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profemul -o run -- python3 bench.py --size 512 --emulate-slab 8 --steps 20 --warmup 30 --no-extra > $O/profemul.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02d/profemul/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("emulated rank: total kernel ms per step", tot / 50 / 1e6)
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/50/1e6:6.3f}")
PY
tail -1 $O/profemul.log | cut -c1-250
N=512 bash tools/jacobi_pmc.sh 4:2:86 r02d_512_2r > $O/pmc512.log 2>&1; tail -6 $O/pmc512.log
