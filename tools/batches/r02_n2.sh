#!/bin/bash
# round 2, batch n2: kernel statistics of the MGCG-mode step
set -e -o pipefail
O=$PWD/gpurun_out/r02n; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mgcg -o s -- python3 $R/bench.py --projection mgcg --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/prof_mgcg.log 2>&1 || { tail -20 $O/prof_mgcg.log; exit 1; }
f=$(find $O/prof_mgcg -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats_mgcg.csv
head -24 $f | cut -c1-150
find $O/prof_mgcg -name "*kernel_trace.csv" -exec rm {} \;
