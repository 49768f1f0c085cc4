#!/bin/bash
# round 2, batch w: occupancy bounds of the gather kernels (waves per SIMD for single-field / two-field kernels); the
# variants are prebuilt libraries under build/occ (not committed), swapped in on the GPU box only
set -e -o pipefail
O=gpurun_out/r02w; mkdir -p $O
cp gpufluidsimulation_amd/libbimocq_hip.so /tmp/base.so
for v in base max-ilp max-memory-clause base max-ilp; do
  cp build/occ/libbimocq_hip_$v.so gpufluidsimulation_amd/libbimocq_hip.so
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-extra > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python3 -c "import json; d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done
cp /tmp/base.so gpufluidsimulation_amd/libbimocq_hip.so
