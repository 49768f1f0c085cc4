#!/bin/bash
# round 3 batch g: why bench.py --gpus 2 fails since the diagnostics leg (stderr of the ranks)
set -o pipefail
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 300 python3 bench.py --gpus 2 --transport host --size 64 --steps 2 --warmup 1 --jacobi-iters 30 --no-extra > $O/b2.json 2> $O/b2.err; echo "rc=$?"
grep -v "Gloo\|amdgpu.ids" $O/b2.err | head -60
