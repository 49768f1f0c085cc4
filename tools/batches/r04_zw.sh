#!/bin/bash
# round 4 batch zw: the exact command of r04_zy, up to six times, with the C stacks of a process that is still there after 20 s
set -o pipefail
O=gpurun_out/r04zw; mkdir -p $O
for i in 1 2 3 4 5 6; do
  python3 tools/jacobi_tune.py --n 128 --reps 7 --variants 5:2:4,4:0:8,4:0:12,4:0:16,4:6:8,4:6:16,4:2:4 > $O/run$i.txt 2>&1 &
  PID=$!
  for t in $(seq 1 20); do sleep 1; kill -0 $PID 2>/dev/null || break; done
  if kill -0 $PID 2>/dev/null; then
    echo "run $i: still alive after 20 s ($(wc -l < $O/run$i.txt) lines printed): stacks"
    timeout -k 5 90 /opt/rocm/bin/rocgdb -p $PID -batch -ex "thread apply all bt 16" > $O/stacks$i.txt 2>&1
    grep -E "^Thread|^#" $O/stacks$i.txt | cut -c1-160 | head -70
    kill $PID; sleep 2; kill -9 $PID 2>/dev/null
    break
  else
    wait $PID; echo "run $i: exited rc=$? after ${t} s"
  fi
done
