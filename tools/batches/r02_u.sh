#!/bin/bash
# round 2, batch u: the default bench command under rocprofv3 --kernel-trace --stats (the summary committed under profiles/)
set -e -o pipefail
O=$PWD/gpurun_out/r02u; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o run -- python3 $R/bench.py > $O/bench_default_profiled.json 2> $O/prof_default.err || { tail -20 $O/prof_default.err; exit 1; }
cut -c1-200 $O/bench_default_profiled.json
f=$(find $O/prof_default -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_default.csv; head -8 $f | cut -c1-160
find $O/prof_default -name "*kernel_trace.csv" -exec rm {} \;
