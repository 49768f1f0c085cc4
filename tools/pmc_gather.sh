#!/bin/bash
# tools/pmc_gather.sh <tag> <step_child options...> -- where the gather kernels' cycles go: four rocprofv3 --pmc passes (one
# counter block each) over a few steps of tools/step_child.py, reduced per kernel.  e.g.
#   tools/pmc_gather.sh fastwin --fl-opt 11=1 --fl-opt 18=1
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
passes=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
        "TA_BUSY_avr" "TCP_TCC_READ_REQ_sum")
i=0
for c in "${passes[@]}"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pg_${tag}_$i -o run -- python3 tools/step_child.py --n 256 --steps 3 --warmup 2 --jacobi-iters 6 "$@" > gpurun_out/pg_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pg_${tag}_$i.log; }
  i=$((i+1))
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pg_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("advect_kernel", "compensate_kernel", "cumulate_kernel", "gather_march", "dmc_kernel", "forward_kernel", "clamp_box")): continue
        key = n.split("(")[0].replace("void bq::exact::", "").replace("void bq::fast::", "f:")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Counter_Name"] in ("SQ_WAVES", "TA_BUSY_avr"):
            dur[key].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    w = max(d.get("SQ_WAVES", 1), 1)
    wc = max(d.get("SQ_WAVE_CYCLES", 1), 1)
    us = sum(dur[k]) / max(len(dur[k]), 1)
    print(f"{k[:58]:58s} us={us:6.1f} waves={w:7.0f} valu/w={d.get('SQ_INSTS_VALU',0)/w:7.0f} lds/w={d.get('SQ_INSTS_LDS',0)/w:6.0f} salu/w={d.get('SQ_INSTS_SALU',0)/w:6.0f} vmrd/w={d.get('SQ_INSTS_VMEM_RD',0)/w:5.0f} "
          f"wait_any={d.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={d.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active={d.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} "
          f"lds_conf/idx={d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',1),1):.2f} lds_idx/busy={d.get('SQ_LDS_IDX_ACTIVE',0)/max(d.get('SQ_BUSY_CYCLES',1),1):.2f} ta_busy={d.get('TA_BUSY_avr',0):.0f} tcp_tcc_rd={d.get('TCP_TCC_READ_REQ_sum',0):.3g}")
PY
