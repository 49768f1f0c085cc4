#!/bin/bash
# tools/ta_pmc.sh <tag> [kernel-substring ...] -- texture-addresser / L1 counters of the gather kernels, ONE or TWO counters of a
# block per rocprofv3 pass (six TA counters in one pass exceed the block's slots: rocprofv3 error 38, see tools/sq_pmc.sh).
# The program follows `--` directly (no env/bash wrapper); each pass is a short bench run.
set -e
tag=${1:-t}; shift || true
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
passes=("TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_BUFFER_LOAD_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum")
i=0
for c in "${passes[@]}"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/ta_${tag}_$i -o run -- python3 bench.py --steps 2 --warmup 30 --jacobi-iters 6 --no-cpu-baseline --no-extra > gpurun_out/ta_${tag}_$i.log 2>&1 || { echo "pass $i ($c) failed: see gpurun_out/ta_${tag}_$i.log"; tail -3 gpurun_out/ta_${tag}_$i.log; }
  i=$((i+1))
done
python3 - "$tag" "$@" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]; subs = sys.argv[2:] or ["advect_kernel", "compensate_kernel", "cumulate_kernel"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/ta_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "fast::" in n or not any(k in n for k in subs): continue
        acc[n.split("(")[0].replace("void bq::exact::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k[:60], {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()})
PY
