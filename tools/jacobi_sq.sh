#!/bin/bash
# tools/jacobi_sq.sh <variant> <tag> -- SQ counters of a Jacobi kernel variant (tools/jacobi_tune.py --variants syntax): where a wave's
# cycles go (parked on s_waitcnt / s_barrier, stalled at issue, issuing), instructions per wave.  One pass, SQ block only (8 counters).
set -e
v=${1:-4:0:0}; tag=${2:-x}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/sq_$tag -o run -- python3 tools/jacobi_tune.py --n ${N:-256} --variants $v --sweeps 24 --reps 1 > gpurun_out/sq_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
f = glob.glob(f"gpurun_out/sq_{tag}/**/*counter_collection.csv", recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void bq::", "")
    if "jacobi" not in n: continue
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    w = d.get("SQ_WAVES", 1); wc = max(d.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k:40s} launches={len(acc[k]['SQ_WAVES'])} us={sum(dur[k])/len(dur[k])/1e3:7.1f} waves={w:.0f} valu/wave={d.get('SQ_INSTS_VALU',0)/w:8.1f} lds/wave={d.get('SQ_INSTS_LDS',0)/w:7.1f} "
          f"wave_cycles/wave={4*wc/w:9.0f} parked(waitcnt,barrier)={d.get('SQ_WAIT_ANY',0)/wc:.2f} issue_stall={d.get('SQ_WAIT_INST_ANY',0)/wc:.2f} issuing={d.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} busy_cycles={4*d.get('SQ_BUSY_CYCLES',0):.4g}")
PY
