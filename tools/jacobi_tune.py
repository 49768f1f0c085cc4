#!/usr/bin/env python3
"""Sweep the Jacobi sweep kernel's tuning knobs on the GPU and print us/sweep + algorithmic GB/s.

    python tools/jacobi_tune.py [--n 256] [--sweeps 200] [--reps 5]

Timing: fl_event pairs on the library's compute stream around `sweeps` back-to-back launches,
median over reps (interleaved variants, one process -- cdna_hip_programming.md rule 24).
"""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gpufluidsimulation_amd as bq
from gpufluidsimulation_amd import DeviceBuffer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--nz", type=int, default=0)
    ap.add_argument("--sweeps", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--variants", type=str, default="")
    a = ap.parse_args()
    lib = bq.hip_lib()
    assert lib.fl_init(0) == 0
    nx = ny = a.n
    nz = a.nz or a.n
    n = nx * ny * nz
    rng = np.random.default_rng(1)
    p = DeviceBuffer.from_numpy(rng.standard_normal(n, dtype=np.float32))
    t = DeviceBuffer.from_numpy(rng.standard_normal(n, dtype=np.float32))
    d = DeviceBuffer.from_numpy(rng.standard_normal(n, dtype=np.float32))
    e0, e1 = lib.fl_event_create(), lib.fl_event_create()
    if a.variants:
        variants = [tuple(int(x) for x in v.split(":")) for v in a.variants.split(",")]
    else:
        variants = [(1, 0, 0), (2, 2, 8), (3, 4, 16)] + [(4, 0, k) for k in (8, 16, 32, 64, 128)]
    res = {v: [] for v in variants}
    for rep in range(a.reps + 1):
        for v in variants:
            # variant 4 = fused kernels (three sweeps per launch where that kernel applies, else two) with kchunk v[2];
            # variant 5 = two sweeps per launch only
            fused = v[0] in (4, 5)
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, 0 if fused else v[0])
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, (2 if v[0] == 4 else 4) if fused else 0)
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, v[1])   # fused: 1 = one-plane prefetch, else two planes ahead
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK, (v[3] if len(v) > 3 else 0) if fused else v[2])   # fused: 4th field = prefetch distance (1; default 2)
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, v[2] if fused else 0)
            lib.fl_event_record(e0)
            lib.gpu_jacobi_sweeps(p.ptr, d.ptr, t.ptr, nx, ny, nz, a.sweeps, -1.0, 1.0 / 6.0)
            lib.fl_event_record(e1)
            ms = lib.fl_event_elapsed_ms(e0, e1)
            if rep:
                res[v].append(ms * 1e3 / a.sweeps)
    bq.check()
    print(f"grid {nx}x{ny}x{nz}, {a.sweeps} sweeps/launch-loop, median of {a.reps}")
    for v in variants:
        us = statistics.median(res[v])
        print(f"variant={v[0]} rows={v[1]} kchunk={v[2]:4d}{(' pf=' + str(v[3])) if len(v) > 3 else ''}: {us:8.2f} us/sweep  {12.0 * n / us / 1e3:8.1f} GB/s  "
              f"{12.0 * n / us / 1e3 / 8000:6.3f} of 8 TB/s   (min {min(res[v]):.2f})")


if __name__ == "__main__":
    main()
