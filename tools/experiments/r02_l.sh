#!/bin/bash
set -o pipefail
O=gpurun_out/r02l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH_FIELD_TILES=1 python -m pytest tests/test_gpu_solver.py tests/test_golden.py -m gpu -x -q -k "not mgcg" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for FTV in 0 1; do
BENCH_FIELD_TILES=$FTV timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ft$FTV -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_ft$FTV.json 2> $O/prof_ft$FTV.err; cut -c1-200 $O/bench_ft$FTV.json
python3 - $FTV <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/r02l/prof_ft{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'cumulate_kernel' in r['Name'] and 'false, true' in r['Name']:
        print(f"{r['Name'][:100]:100s} n={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
done
