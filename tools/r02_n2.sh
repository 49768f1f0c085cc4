#!/bin/bash
# round 2, batch n2: kernel statistics of the MGCG-mode step with the tile smoother on / off
set -e -o pipefail
O=$PWD/gpurun_out/r02n; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tile_$v -o s -- python3 $R/bench.py --projection mgcg --steps 2 --warmup 1 --no-extra --no-cpu-baseline --fl-opt 14=$v > $O/prof_tile_$v.log 2>&1 || { tail -20 $O/prof_tile_$v.log; exit 1; }
  f=$(find $O/prof_tile_$v -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_tile_$v.csv
  head -12 $f | cut -c1-160
  find $O/prof_tile_$v -name "*kernel_trace.csv" -exec rm {} \;
done
