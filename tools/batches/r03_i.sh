#!/bin/bash
# round 3 batch i: the whole GPU suite on the round-3 state, then the 128^3 step anatomy (kernel table) and bench lines
set -o pipefail
O=gpurun_out/r03i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for shape in "64 64" "96 96" "128 128" "160 160"; do set -- $shape
  timeout -k 10 200 python3 tools/jacobi_tune.py --n $1 --nz $2 --sweeps 199 --variants "4:0:0,5:1:0" 2>&1 | grep -v amdgpu.ids >> $O/jacobi_small.txt; done; cat $O/jacobi_small.txt
timeout -k 10 300 python3 bench.py --size 128 --steps 100 --warmup 20 --no-extra > $O/bench_128.json 2>/dev/null; cut -c1-330 $O/bench_128.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_128 -o run -- python3 bench.py --size 128 --steps 40 --warmup 20 --no-extra --no-cpu-baseline > $O/prof_128.log 2>&1; echo "prof rc=$?"
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/r03i/prof_128/run_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
starts=[i for i,r in enumerate(rows) if 'max_abs3_partial' in r['Kernel_Name']]
a,b=starts[30],starts[50]; S=20
tot=collections.defaultdict(lambda:[0,0])
for r in rows[a:b]:
    n=r['Kernel_Name'].replace('void ','').replace('bq::exact::','').replace('bq::','')
    key=n.split('(')[0][:60]
    tot[key][0]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); tot[key][1]+=1
for k,(t,c) in sorted(tot.items(), key=lambda x:-x[1][0])[:30]:
    print(f"{k:62s} {c/S:6.1f}/step {t/c/1e3:8.1f} us {t/S/1e6:7.3f} ms/step")
wall=(int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/S/1e6
print('wall ms/step', wall, 'kernel busy ms/step', sum(t for t,c in tot.values())/S/1e6, 'launches/step', sum(c for t,c in tot.values())/S)
PY
rm -f $O/prof_128/run_kernel_trace.csv
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/dev/null; cut -c1-300 $O/bench_driver_cmd.json
