#!/bin/bash
# round 3 batch p: SQ counters of the three-sweep kernels (register-only, LDS-exchanged single rows / row pairs) and the four-sweep one
O=gpurun_out/r03p; mkdir -p $O
for v in "4:5:0:0 lean3r" "4:4:32:18 lds3_r1" "4:4:32:24 lds3_r2" "4:6:32:24 lds4_r2" "5:2:0 lean2r"; do set -- $v
  bash tools/jacobi_sq.sh $1 r03p_$2 2>&1 | tail -3 >> $O/sq_jacobi.txt; done; cat $O/sq_jacobi.txt
