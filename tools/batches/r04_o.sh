#!/bin/bash
# round 4 batch o: blend != 1 reference-faithful on z-slab ranks (whole-grid *Prev fields): operator parity under fl_set_slab,
# 2 and 3 stand-in ranks over the RCCL branch; the slab suite again (advectVelocity / advectFields2 were touched)
set -o pipefail
O=gpurun_out/r04o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "advect_double" > $O/pytest_ops.log 2>&1; rc=$?; echo "pytest ops rc=$rc"; tail -4 $O/pytest_ops.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 1000 python -m pytest tests/test_gpu_rccl_path.py -x -q -k "blend_below_one or two_ranks_over or three_ranks_over" > $O/pytest_rccl.log 2>&1; rc=$?; echo "pytest rccl rc=$rc"; tail -6 $O/pytest_rccl.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py -x -q > $O/pytest_solver.log 2>&1; rc=$?; echo "pytest solver rc=$rc"; tail -4 $O/pytest_solver.log
