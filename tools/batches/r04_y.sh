#!/bin/bash
# round 4 batch y: the DRIVER's own multi-GPU invocation (torch.distributed.run, default 512^3 grid, --steps 20 --warmup 5) with the
# RCCL stand-in so that 2 and 4 ranks can share the one GPU of this box: the command line, the stage lines, the JSON line
set -o pipefail
O=gpurun_out/r04y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
FAKE=$(python3 tests/build_fake_rccl.py | tail -1)
echo "stand-in: $FAKE"
for n in 2 4; do
  BQ_RCCL_LIBRARY=$FAKE timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) \
      bench.py --gpus $n --steps 20 --warmup 5 > $O/n$n.json 2>$O/n$n.err; echo "N=$n rc=$?"
  grep "^\[bench rank 0" $O/n$n.err | tail -8
  python3 - $O/n$n.json <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
    d = json.loads(lines[-1])
    print("   lines:", len(lines), "|", d["value"], d["unit"], d["ms_per_step"], "ms n_gpus", d["n_gpus"], d["scaling"], d["config"]["global_grid"], d["config"]["parallelism"][:70])
    print("   diag:", {k: d["diagnostics"][k] for k in ("steps", "comm_exposed_ms_per_step")} if d.get("diagnostics") else None, "extra:", sorted((d.get("extra") or {}).keys()))
except Exception as e:
    print("   unreadable", e)
PY
done
