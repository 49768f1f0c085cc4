#!/bin/bash
# round 3 batch zq: emulated config-4 ranks of 2- and 4-rank runs (512 x 512 x 256 / 128 owned planes + 16): the triple schedule
# against the pair schedule (extra.jacobi_triples_off) where the plane ranges are long
O=gpurun_out/r03zq; mkdir -p $O
for n in 2 4; do
  timeout -k 10 500 python3 bench.py --size 512 --emulate-slab $n --steps 10 --warmup 6 --no-cpu-baseline --diag-steps 6 > $O/emul_cfg4_of$n.json 2> $O/emul_of$n.err; echo "emul of $n rc=$?"
  python3 -c "
import json;l=json.load(open('$O/emul_cfg4_of$n.json'));d=l['diagnostics'];print('ranks',$n,'headline',l['value'],l['ms_per_step'],'diag',d['ms_per_step'],d['phase_ms_per_step_slowest_rank']['projection'],'triples_off',l['extra']['jacobi_triples_off']['ms_per_step'],l['extra']['jacobi_triples_off']['phase_ms_per_step_slowest_rank']['projection'],'ends_first_off',l['extra']['ends_first_off']['ms_per_step'])"
done
