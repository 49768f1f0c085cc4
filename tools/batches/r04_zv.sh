#!/bin/bash
# round 4 batch zv: the exit hang of r04_zy/zw -- does it need the CU-masked copy stream, and where does the process sit?
set -o pipefail
O=gpurun_out/r04zv; mkdir -p $O
CMD="python3 tools/jacobi_tune.py --n 128 --reps 7 --variants 5:2:4,4:0:8,4:0:12,4:0:16,4:6:8,4:6:16,4:2:4"
watch() { tag=$1; shift
  env "$@" $CMD > $O/$tag.txt 2>&1 &
  PID=$!
  for t in $(seq 1 20); do sleep 1; kill -0 $PID 2>/dev/null || break; done
  if kill -0 $PID 2>/dev/null; then echo "$tag: HANGS ($(wc -l < $O/$tag.txt) lines printed)"; kill $PID; sleep 2; kill -9 $PID 2>/dev/null; sleep 1
  else wait $PID; echo "$tag: exited rc=$? after ${t} s"; fi
}
watch default A=1
watch no_cu_mask BQ_COPY_STREAM_CUS=0
watch default_again A=1
echo "--- under the debugger (SIGINT after 30 s) ---"
timeout -s INT -k 20 30 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex run -ex "thread apply all bt 18" --args $CMD > $O/gdb.txt 2>&1
grep -E "^Thread|^#" $O/gdb.txt | cut -c1-170 | head -90
