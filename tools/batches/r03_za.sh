#!/bin/bash
# round 3 batch za: the emulated config-4 rank (512 x 512 x 64 + 16 planes) with its kernel table, counter traffic of the
# two-segment kernel at 512^3 and of the fp64 three-sweep smoother at 256^3
set -o pipefail
O=gpurun_out/r03za; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export BQ_COPY_STREAM_CUS=0
CMD="bench.py --size 512 --emulate-slab 8 --steps 20 --warmup 10 --no-extra --no-cpu-baseline --diag-steps 6"
timeout -k 10 400 python3 $CMD > $O/emul_cfg4.json 2> $O/emul_cfg4.err; echo "emul cfg4 rc=$?"; cut -c1-400 $O/emul_cfg4.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -o run -- python3 $CMD --diag-steps 0 > $O/prof_cfg4.log 2>&1; echo "prof rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03za/prof_cfg4/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
PY
rm -f $O/prof_cfg4/*kernel_trace.csv $O/prof_cfg4/*/*kernel_trace.csv
N=512 bash tools/jacobi_pmc.sh 4:0:0 r03za_512_lds2seg > $O/pmc_512_lds2seg.txt 2>&1; echo "pmc rc=$?"; tail -4 $O/pmc_512_lds2seg.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r03za_mg_$c -o run -- python3 tools/smooth_tune.py --n 256 --sweeps 32 --reps 1 --variants 1:0:0:0 > gpurun_out/pmc_r03za_mg_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_r03za_mg_{c}/**/*counter_collection.csv", recursive=True)[0]
    by = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "mg_l" in r["Kernel_Name"]:
            by.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in by.items():
        print(c, k, "launches", len(v), "mean_KB", sum(v) / len(v))
PY
