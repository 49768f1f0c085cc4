#!/bin/bash
# round 4 batch zu: the exit hang with the CU-masked copy stream -- which call of fl_shutdown it is, and which ordering avoids it
set -o pipefail
O=gpurun_out/r04zu; mkdir -p $O
CMD="python3 tools/jacobi_tune.py --n 128 --reps 7 --variants 5:2:4,4:0:8,4:0:12,4:0:16,4:6:8,4:6:16,4:2:4"
watch() { tag=$1; shift
  env "$@" $CMD > $O/$tag.txt 2>&1 &
  PID=$!
  for t in $(seq 1 15); do sleep 1; kill -0 $PID 2>/dev/null || break; done
  if kill -0 $PID 2>/dev/null; then echo "$tag: HANGS; last lines: $(grep fl_shutdown $O/$tag.txt | tail -1)"; kill $PID; sleep 2; kill -9 $PID 2>/dev/null; sleep 1
  else wait $PID; echo "$tag: exited rc=$? after ${t} s"; fi
}
watch mode5_trace BQ_SHUTDOWN_MODE=5
watch mode1_device_sync_first BQ_SHUTDOWN_MODE=1
watch mode2_copy_stream_first BQ_SHUTDOWN_MODE=2
watch mode3_copy_stream_kept BQ_SHUTDOWN_MODE=3
watch mode4_both BQ_SHUTDOWN_MODE=4
