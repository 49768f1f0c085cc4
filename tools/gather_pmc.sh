#!/bin/bash
# tools/gather_pmc.sh <tag> -- PMC passes over a short bench run, per-kernel means for the gather family
set -e
tag=${1:-g}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
while read -r group; do
  i=$((i+1))
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d gpurun_out/gpmc_${tag}_$i -o run -- python3 bench.py --steps 2 --warmup 1 --jacobi-iters 6 --no-cpu-baseline > gpurun_out/gpmc_${tag}_$i.log 2>&1 || echo "pass $i failed"
done <<'G'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD
TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TAGRAM0_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE
G
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/gpmc_{tag}_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "exact::" not in n or not any(k in n for k in ("advect_kernel", "compensate_kernel", "cumulate_kernel")):
            continue
        key = n.split("(")[0].replace("void bq::exact::", "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:44s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
