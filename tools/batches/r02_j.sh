#!/bin/bash
# round 2 validation batch: full GPU suite, the default bench command (plain and under rocprofv3 --kernel-trace --stats),
# the driver's command line, 128^3 / 512^3 / emulated config-4 ranks, TA counter passes
set -o pipefail
O=gpurun_out/r02j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; cut -c1-260 $O/bench_default.json
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver.err; cut -c1-260 $O/bench_driver_cmd.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o run -- python3 bench.py > $O/bench_default_profiled.json 2> $O/prof_default.err; echo "rocprof rc=$?"; cut -c1-200 $O/bench_default_profiled.json
timeout -k 10 300 python3 bench.py --size 128 --steps 180 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_128.json 2>/dev/null; cut -c1-200 $O/bench_128.json
timeout -k 10 300 python3 bench.py --size 512 --steps 30 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_512.json 2>/dev/null; cut -c1-200 $O/bench_512.json
for G in 8 6; do timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 8 --ghost $G --steps 60 --warmup 20 --no-extra > $O/emul_512_r8_g$G.json 2>/dev/null; cut -c140-330 $O/emul_512_r8_g$G.json; done
timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 8 --emulate-rank 7 --steps 60 --warmup 20 --no-extra > $O/emul_512_r8_last.json 2>/dev/null; cut -c140-330 $O/emul_512_r8_last.json
timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 2 --steps 20 --warmup 20 --no-extra > $O/emul_512_r2.json 2>/dev/null; cut -c140-330 $O/emul_512_r2.json
timeout -k 10 600 bash tools/ta_pmc.sh r02j > $O/ta_pmc.log 2>&1; tail -12 $O/ta_pmc.log
