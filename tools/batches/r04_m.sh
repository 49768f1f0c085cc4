#!/bin/bash
# round 4 batch m: what bounds the map updates (dmc_kernel, forward_kernel) and the limiter: SQ / TA / L1->L2 counters, exact build, steps 20-24
set -o pipefail
O=gpurun_out/r04m; mkdir -p $O
bash tools/pmc_gather.sh exact_maps --warmup 20 --steps 4 > $O/pmc_exact.txt 2>&1; grep -E 'dmc|forward|clamp_box' $O/pmc_exact.txt | cut -c1-330
