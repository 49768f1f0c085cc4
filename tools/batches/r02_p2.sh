#!/bin/bash
# round 2, batch p2: 512^3 and the emulated config-4 ranks on the final state of the round
set -o pipefail
O=gpurun_out/r02p2; mkdir -p $O
timeout -k 10 300 python3 bench.py --size 512 --steps 30 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_512.json 2>/dev/null; cut -c1-200 $O/bench_512.json
timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 8 --steps 60 --warmup 20 --no-extra --no-cpu-baseline > $O/emul_512_r8_g8.json 2>/dev/null; cut -c140-330 $O/emul_512_r8_g8.json
timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 4 --steps 30 --warmup 20 --no-extra --no-cpu-baseline > $O/emul_512_r4.json 2>/dev/null; cut -c140-330 $O/emul_512_r4.json
timeout -k 10 300 python3 bench.py --size 512 --emulate-slab 2 --steps 20 --warmup 20 --no-extra --no-cpu-baseline > $O/emul_512_r2.json 2>/dev/null; cut -c140-330 $O/emul_512_r2.json
timeout -k 10 300 python3 bench.py --size 128 --steps 180 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_128.json 2>/dev/null; cut -c1-200 $O/bench_128.json
