#!/bin/bash
# round 2 validation batch p: full GPU suite, default bench, driver command, 512^3, emulated config-4 ranks,
# FETCH/WRITE_SIZE of the two-sweep kernel at 512^3
set -o pipefail
O=gpurun_out/r02p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; cut -c1-260 $O/bench_default.json
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver.err; cut -c1-260 $O/bench_driver_cmd.json
timeout -k 10 300 python3 bench.py --size 512 --steps 30 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_512.json 2>/dev/null; cut -c1-200 $O/bench_512.json
for G in 8 6; do timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 8 --ghost $G --steps 60 --warmup 20 --no-extra --no-cpu-baseline > $O/emul_512_r8_g$G.json 2>/dev/null; cut -c140-330 $O/emul_512_r8_g$G.json; done
timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 4 --steps 30 --warmup 20 --no-extra --no-cpu-baseline > $O/emul_512_r4.json 2>/dev/null; cut -c140-330 $O/emul_512_r4.json
timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 2 --steps 20 --warmup 20 --no-extra --no-cpu-baseline > $O/emul_512_r2.json 2>/dev/null; cut -c140-330 $O/emul_512_r2.json
N=512 timeout -k 10 600 bash tools/jacobi_pmc.sh 5:0:0:0 r02p_512 > $O/pmc_512.log 2>&1; cat $O/pmc_512.log
