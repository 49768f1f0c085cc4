#!/bin/bash
# round 2, batch z: first runs of the stream-ordered RCCL stand-in (short timeouts: a stalled stream must not hold the box)
set -o pipefail
O=gpurun_out/r02z; mkdir -p $O
export OMP_NUM_THREADS=4 MASTER_ADDR=127.0.0.1 BQ_RCCL_LIBRARY=$PWD/tests/_build/libfake_rccl_async.so
timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 tests/slab_worker.py --backend gpu --transport rccl --steps 3 > $O/two.log 2>&1; rc=$?
grep "rank\|fake_rccl" $O/two.log | tail -6; echo "rc=$rc"
[ $rc -eq 0 ] || { tail -20 $O/two.log; exit 1; }
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=3 --master-addr 127.0.0.1 --master-port 29612 tests/slab_worker.py --backend gpu --transport rccl --dims 24 20 36 --ghost 6 --steps 3 --iters 16 --dt-cells 1.0 > $O/three.log 2>&1; rc=$?
grep "rank\|fake_rccl" $O/three.log | tail -6; echo "rc=$rc"
[ $rc -eq 0 ] || { tail -20 $O/three.log; exit 1; }
