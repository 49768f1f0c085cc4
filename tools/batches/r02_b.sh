#!/bin/bash
# faithful-mode and kept-border slab runs against the single-GPU faithful run, 128^3, 200 steps, 200 Jacobi iterations
set -o pipefail
O=gpurun_out/r02b; mkdir -p $O
REF=/tmp/slabref
CP="1 10 20 40 60 80 100 120 140 160 180 200"
timeout -k 10 400 python tests/slab_deviation_worker.py --make-reference $REF --size 128 --steps 200 --iters 200 --checkpoints $CP > $O/dev_ref.log 2>&1 || exit 1
for K in 0 1; do
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 2951$K \
   tests/slab_deviation_worker.py --reference $REF --size 128 --steps 200 --iters 200 --checkpoints $CP --keep-dmc-border $K --rms-tol 1 --json $O/slab_dev_128_200steps_keep$K.json > $O/dev_keep$K.log 2>&1
echo "keep$K rc=$?"; grep slab-deviation $O/dev_keep$K.log | tail -14
done
