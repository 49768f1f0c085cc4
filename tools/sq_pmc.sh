#!/bin/bash
# tools/sq_pmc.sh <tag> [kernel-substring ...] -- one SQ counter pass over a short bench run, per-kernel means.
# SQ block only.  Round 1 also asked for six TA_* counters (_sum and plain forms) in ONE pass: rocprofv3 aborted with
# "rocprofiler_create_counter_config ... error 38: Request exceeds the capabilities of the hardware to collect" (signal 6 with a
# dispatch incomplete, gpurun_out/gpmc_a_2.log) -- too many counters for the TA block's slots, not a hang of the pool.  TA / TCP
# counters go in their own passes of at most two per block: tools/ta_pmc.sh.
set -e
tag=${1:-q}; shift || true
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/sq_$tag -o run -- python3 bench.py --steps 3 --warmup 60 --jacobi-iters 6 --no-cpu-baseline > gpurun_out/sq_$tag.log 2>&1
python3 - "$tag" "$@" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]; subs = sys.argv[2:] or ["forward", "dmc", "advect_kernel"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
f = glob.glob(f"gpurun_out/sq_{tag}/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "fast::" in n or not any(k in n for k in subs): continue
    acc[n.split("(")[0].replace("void bq::exact::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    w = d.get("SQ_WAVES", 1)
    print(f"{k:46s} n={len(acc[k]['SQ_WAVES'])} waves={w:.0f} valu/wave={d.get('SQ_INSTS_VALU',0)/w:7.1f} vmem_rd/wave={d.get('SQ_INSTS_VMEM_RD',0)/w:6.1f} "
          f"wave_cycles/wave={4*d.get('SQ_WAVE_CYCLES',0)/w:9.0f} wait_frac={d.get('SQ_WAIT_INST_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1):.2f} busy_quadcycles={d.get('SQ_BUSY_CYCLES',0):.3g}")
PY
