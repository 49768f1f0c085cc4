#!/bin/bash
# round 4 batch c: SQ / TA / L1->L2 counters of the gather family, one-plane kernels against the z-marching window kernels, fast variant
set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
bash tools/pmc_gather.sh fast --fl-opt 11=1 > $O/pmc_fast.txt 2>&1; cat $O/pmc_fast.txt
bash tools/pmc_gather.sh fastwin --fl-opt 11=1 --fl-opt 18=1 > $O/pmc_fastwin.txt 2>&1; cat $O/pmc_fastwin.txt
