#!/bin/bash
# round 4 batch j: (1) division by the constant h in two instructions: parity on the other spacings + the reference grid again;
# (2) the slab-shared multigrid solver: mgcg_128 hashes on 2 / 4 stand-in ranks, then its compute-side cost -- emulated rank 4 of
# 8 at 512^3 (and of 2 / 4 at 256^3) against the single-GPU solve of the same grid
set -o pipefail
O=gpurun_out/r04j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -k "tabled or wild or advect_velocity" > $O/pytest_ops.log 2>&1; rc=$?; echo "pytest ops rc=$rc"; tail -4 $O/pytest_ops.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_rccl_path.py -x -q -k "mgcg_128" > $O/pytest_hash.log 2>&1; echo "pytest mgcg_128 hashes rc=$?"; tail -4 $O/pytest_hash.log
R="python3 bench.py --gpus 1 --reference-scene --steps 20 --warmup 12 --no-cpu-baseline --no-measure-traffic"
timeout -k 10 300 $R > $O/ref_bimocq_jacobi.json 2>$O/ref_bimocq_jacobi.err; echo "ref bimocq+jacobi rc=$?"
show() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("   ", d["value"], "Mvox/s", d["ms_per_step"], "ms", d.get("phase_ms_per_step"), d["config"].get("mgcg_levels_shared"), d["config"]["parallelism"][:60])
except Exception as e:
    print("   unreadable:", e)
PY
}
show $O/ref_bimocq_jacobi.json
M="python3 bench.py --gpus 1 --projection mgcg --no-cpu-baseline --no-measure-traffic"
timeout -k 10 600 $M --size 256 --steps 3 --warmup 1 > $O/mg_256_single.json 2>$O/mg_256_single.err; echo "256 single rc=$?"; show $O/mg_256_single.json
timeout -k 10 600 $M --size 256 --steps 3 --warmup 1 --emulate-slab 2 > $O/mg_256_emul2.json 2>$O/mg_256_emul2.err; echo "256 rank of 2 rc=$?"; show $O/mg_256_emul2.json
timeout -k 10 600 $M --size 256 --steps 3 --warmup 1 --emulate-slab 4 > $O/mg_256_emul4.json 2>$O/mg_256_emul4.err; echo "256 rank of 4 rc=$?"; show $O/mg_256_emul4.json
timeout -k 10 600 $M --size 512 --steps 2 --warmup 1 > $O/mg_512_single.json 2>$O/mg_512_single.err; echo "512 single rc=$?"; show $O/mg_512_single.json; tail -2 $O/mg_512_single.err
timeout -k 10 600 $M --size 512 --steps 2 --warmup 1 --emulate-slab 8 > $O/mg_512_emul8.json 2>$O/mg_512_emul8.err; echo "512 rank of 8 rc=$?"; show $O/mg_512_emul8.json; tail -2 $O/mg_512_emul8.err
