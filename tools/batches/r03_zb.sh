#!/bin/bash
# round 3 batch zb: three-sweep launches on plane ranges and the triple schedule of a slab chunk -- op parity, slab parity on
# the host-staged transport and the RCCL stand-ins, config-4 / config-5 rank geometry, the emulated config-4 rank with and without
set -o pipefail
O=gpurun_out/r03zb; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q > $O/pytest_proj.log 2>&1; rc=$?; echo "proj rc=$rc"; tail -3 $O/pytest_proj.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 1100 python -m pytest tests/test_slab_multirank.py tests/test_gpu_rccl_path.py tests/test_gpu_config5.py tests/test_gpu_bench_cli.py -x -q -m gpu > $O/pytest_slab.log 2>&1; rc=$?; echo "slab rc=$rc"; tail -5 $O/pytest_slab.log
[ $rc -eq 0 ] || exit 1
CMD="bench.py --size 512 --emulate-slab 8 --steps 20 --warmup 10 --no-cpu-baseline --diag-steps 6"
timeout -k 10 400 python3 $CMD > $O/emul_cfg4.json 2> $O/emul_cfg4.err; echo "emul cfg4 rc=$?"; python3 -c "
import json;l=json.load(open('$O/emul_cfg4.json'));print(l['value'],l['ms_per_step']);print({k:(v.get('value'),v.get('ms_per_step')) for k,v in l.get('extra',{}).items()})"
