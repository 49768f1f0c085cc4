#!/bin/bash
# round 4 batch zz: FL_OPT_MGCG_FUSE off by default in the library, switched on by the host solver: parity, hashes, the MGCG line
set -o pipefail
O=gpurun_out/r04zz; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mgcg.py tests/test_abi_exports.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -k mgcg > $O/pytest_hash.log 2>&1; rc=$?; echo "hash rc=$rc"; tail -3 $O/pytest_hash.log
timeout -k 10 300 python3 bench.py --gpus 1 --projection mgcg --no-cpu-baseline --no-measure-traffic --no-extra --steps 4 --warmup 2 > $O/mgcg.json 2>$O/mgcg.err; echo "mgcg rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r04zz/mgcg.json')); print(d['value'], d['ms_per_step'])"
timeout -k 10 300 python3 -c "
import gpufluidsimulation_amd as bq
lib = bq.hip_lib(); assert lib.fl_init(0) == 0
print('library default:', lib.fl_get_option(bq._lib.FL_OPT_MGCG_FUSE))
from gpufluidsimulation_amd.solver import BimocqGPUSolver
from gpufluidsimulation_amd.scenes import rising_smoke
s = BimocqGPUSolver(64, 64, 64, 1.0, 0.0, 1.0, device=0); s.setSmoke(0.0, 1.0, rising_smoke(64, 1/64)); s.setProjection(3, 0.5, 1); s.advance(0, 0.01)
print('after a host-solver MGCG step:', lib.fl_get_option(bq._lib.FL_OPT_MGCG_FUSE)); s.close()"
