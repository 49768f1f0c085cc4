#!/usr/bin/env python3
"""Sweep the fp64 smoothing kernels' knobs: us per sweep and algorithmic TB/s (24 B/cell/sweep).
    python tools/smooth_tune.py [--n 256] [--sweeps 32]"""
import argparse, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpufluidsimulation_amd as bq

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--sweeps", type=int, default=32)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--variants", type=str, default="", help="fuse:rows:kchunk2:pf,...")
    a = ap.parse_args()
    lib = bq.hip_lib(); assert lib.fl_init(0) == 0
    n = a.n; cells = n ** 3
    bufs = [lib.fl_malloc(cells * 8) for _ in range(3)]
    host = np.random.default_rng(1).standard_normal(cells)
    lib.fl_memcpy_h2d(bufs[1], host.ctypes.data, cells * 8)
    e0, e1 = lib.fl_event_create(), lib.fl_event_create()
    # (fuse, rows option, kchunk2, prefetch distance); rows 0 = triples through mg_lds3_kernel + the lean two-row kernel, 5 = the lean
    # two-row kernel only, 3 = mg_smooth2_kernel with 4 waves
    variants = [(0, 0, 0, 0), (1, 3, 64, 0), (1, 8, 64, 0)] + [(1, 0, k, pf) for pf in (1, 2) for k in (0, 32, 43, 64, 86, 128)]
    if a.variants:
        variants = [tuple(int(x) for x in v.split(':')) for v in a.variants.split(',')]
    res = {v: [] for v in variants}
    for rep in range(a.reps + 1):
        for v in variants:
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, v[0])
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, v[1])
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, v[2])
            lib.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK, v[3])
            lib.fl_event_record(e0)
            lib.gpu_smoothing_jacobi(bufs[0], bufs[1], bufs[2], -1.0, 1.0 / 6.0, n, n, n, a.sweeps)
            lib.fl_event_record(e1)
            ms = lib.fl_event_elapsed_ms(e0, e1)
            if rep: res[v].append(ms * 1e3 / a.sweeps)
    bq.check()
    for v in variants:
        us = statistics.median(res[v])
        print(f"fuse={v[0]} rows-option={v[1]:2d} kchunk={v[2]:3d} pf={v[3]}: {us:8.2f} us/sweep  {24.0 * cells / us / 1e6:6.2f} TB/s algorithmic"
              f"  ({24.0 * cells / (us * (2 if v[0] else 1)) / 1e6:5.2f} TB/s of compulsory traffic per launch)")

if __name__ == "__main__":
    main()
