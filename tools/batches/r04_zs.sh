#!/bin/bash
# round 4 batch zs: every GPU tool once with small arguments, under a time limit -- does each still run and EXIT on the final tree
set -o pipefail
O=gpurun_out/r04zs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { tag=$1; shift; s=$(date +%s.%N); timeout -k 5 150 "$@" > $O/$tag.txt 2>&1; rc=$?; e=$(date +%s.%N); echo "$tag rc=$rc $(python3 -c "print(round($e-$s,1))") s | $(tail -1 $O/$tag.txt | cut -c1-110)"; }
run jacobi_tune_256 python3 tools/jacobi_tune.py --n 256 --reps 3 --sweeps 60
run jacobi_tune_128_all python3 tools/jacobi_tune.py --n 128 --reps 5
run jacobi_rows_check python3 tools/jacobi_rows_check.py
run smooth_tune python3 tools/smooth_tune.py --n 128 --sweeps 12 --reps 2
run mgcg_time python3 tools/mgcg_time.py
run diag_rows python3 tools/diag_rows.py --grid 64 64 16 --steps 2 --iters 20
run fast_lerp_deviation python3 tools/fast_lerp_deviation.py --size 32 --steps 10 --iters 20
run vel_probe python3 tools/vel_probe.py
run parity_report python3 tools/parity_report.py
run step_child python3 tools/step_child.py --n 64 --steps 3 --warmup 1 --jacobi-iters 20
run mgcg_fuse_ab python3 tools/mgcg_fuse_ab.py --size 256 --steps 1 --rounds 1 --iters 4
