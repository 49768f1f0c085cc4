#!/bin/bash
# round 4 batch zx: jacobi_tune.py --n 128 did not exit after printing its table (r04_zy): which variant, and where does it sit?
set -o pipefail
O=gpurun_out/r04zx; mkdir -p $O
for v in 5:2:4 4:0:8 4:6:8 4:2:4; do
  s=$(date +%s.%N)
  timeout -k 5 45 python3 tools/jacobi_tune.py --n 128 --reps 2 --sweeps 50 --variants $v > $O/v_$v.txt 2>&1; rc=$?
  e=$(date +%s.%N)
  echo "variant $v rc=$rc $(python3 -c "print(round($e-$s,1))") s: $(tail -1 $O/v_$v.txt | cut -c1-70)"
done
# the combination of the failing run, with C stacks of the stuck process
python3 tools/jacobi_tune.py --n 128 --reps 2 --sweeps 50 --variants 5:2:4,4:0:8,4:6:8,4:2:4 > $O/all.txt 2>&1 &
PID=$!
sleep 25
if kill -0 $PID 2>/dev/null; then
  echo "still alive after 25 s: stacks"
  timeout -k 5 60 /opt/rocm/bin/rocgdb -p $PID -batch -ex "thread apply all bt 14" > $O/stacks.txt 2>&1
  grep -E "^Thread|^#" $O/stacks.txt | head -60
  kill $PID; sleep 1; kill -9 $PID 2>/dev/null
else
  echo "exited by itself"; tail -2 $O/all.txt
fi
