#!/bin/bash
# round 2, first GPU call: tests, faithful-mode slab deviation at 128^3, benches, emulated config-4 rank
set -o pipefail
O=gpurun_out/r02a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -3 $O/pytest.log
REF=/tmp/slabref; 
timeout -k 10 300 python tests/slab_deviation_worker.py --make-reference $REF --size 128 --steps 40 --iters 200 > $O/dev_ref.log 2>&1 && \
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29511 \
   tests/slab_deviation_worker.py --reference $REF --size 128 --steps 40 --iters 200 --keep-dmc-border 0 --json $O/slab_dev_128_keep0.json > $O/dev_keep0.log 2>&1
echo "dev keep0 rc=$?"; grep slab-deviation $O/dev_keep0.log | tail -12
timeout -k 10 300 python bench.py --steps 60 --warmup 20 > $O/bench_256.json 2> $O/bench_256.err; echo "bench rc=$?"; cat $O/bench_256.json
timeout -k 10 300 python bench.py --size 512 --steps 20 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_512.json 2> $O/bench_512.err; cat $O/bench_512.json
for G in 8 6; do
timeout -k 10 300 python bench.py --size 512 --emulate-slab 8 --ghost $G --steps 60 --warmup 20 --no-extra > $O/emul_512_r8_g$G.json 2> $O/emul_512_r8_g$G.err; cat $O/emul_512_r8_g$G.json
done
