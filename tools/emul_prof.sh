cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emul -o run -- python3 bench.py --emulate-slab --steps 20 --warmup 60 --no-cpu-baseline > gpurun_out/prof_emul.log 2>&1
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/prof_emul/run_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
starts=[i for i,r in enumerate(rows) if 'max_abs3_partial' in r['Kernel_Name']]
a,b=starts[60],starts[80]
tot=collections.defaultdict(lambda:[0,0])
for r in rows[a:b]:
    n=r['Kernel_Name'].replace('void ','').replace('bq::exact::','').replace('bq::','')
    key=n.split('(')[0][:44]+' gz='+r['Grid_Size_Z']+' gx='+r['Grid_Size_X']
    tot[key][0]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); tot[key][1]+=1
S=20
for k,(t,c) in sorted(tot.items(), key=lambda x:-x[1][0])[:34]:
    print(f"{k:70s} {c/S:6.1f}/step {t/c/1e3:8.1f} us {t/S/1e6:6.3f} ms/step")
wall=(int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/S/1e6
print('wall', wall, 'busy', sum(t for t,c in tot.values())/S/1e6)
PY
