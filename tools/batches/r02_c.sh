#!/bin/bash
# wall sheets on the GPU: slab tests (2/3 ranks sharing the GPU), 128^3 x 200 steps deviation in the default mode,
# emulated config-4 rank with the wall sheets in the loop, VALU issue-rate microbenchmark
set -o pipefail
O=gpurun_out/r02c; mkdir -p $O
python -m pytest tests/test_slab_multirank.py tests/test_gpu_projection.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
REF=/tmp/slabref
CP="1 20 40 80 120 160 200"
timeout -k 10 400 python tests/slab_deviation_worker.py --make-reference $REF --size 128 --steps 200 --iters 200 --checkpoints $CP > $O/dev_ref.log 2>&1 && \
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29517 \
   tests/slab_deviation_worker.py --reference $REF --size 128 --steps 200 --iters 200 --checkpoints $CP --json $O/slab_dev_128_200steps_default.json > $O/dev_default.log 2>&1
echo "dev rc=$?"; grep slab-deviation $O/dev_default.log | tail -9
timeout -k 10 300 python bench.py --size 512 --emulate-slab 8 --ghost 8 --steps 60 --warmup 20 --no-extra > $O/emul_512_r8_g8.json 2> $O/emul.err; cat $O/emul_512_r8_g8.json | cut -c1-400
timeout -k 10 120 ./build/valu_rate > $O/valu_rate.txt 2>&1; cat $O/valu_rate.txt
