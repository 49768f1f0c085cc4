#!/bin/bash
# round 3 batch k: the LDS-exchanged three-sweep Jacobi kernel -- parity, then timing against the register-only three-sweep kernel
set -o pipefail
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_projection.py -x -q -k "lds" > $O/pytest_lds.log 2>&1; rc=$?; echo "lds rc=$rc"; tail -8 $O/pytest_lds.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/jacobi_tune.py --n 256 --sweeps 198 --variants "4:0:0,4:4:0,4:4:16,4:4:20,4:4:24,4:4:32,4:4:48,5:2:0" 2>&1 | grep -v amdgpu.ids > $O/jacobi_lds_256.txt; cat $O/jacobi_lds_256.txt
timeout -k 10 300 python3 tools/jacobi_tune.py --n 128 --sweeps 198 --variants "4:0:0,4:4:8,4:4:12,4:4:16" 2>&1 | grep -v amdgpu.ids > $O/jacobi_lds_128.txt; cat $O/jacobi_lds_128.txt
N=128 bash tools/jacobi_pmc.sh 4:0:0 r03k_128 > $O/pmc_128.txt 2>&1; echo "pmc rc=$?"; tail -4 $O/pmc_128.txt
