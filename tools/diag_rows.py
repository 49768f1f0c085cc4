#!/usr/bin/env python3
"""Where do the HIP path and the CPU oracle part ways on a given grid?  Runs both on the leapfrog scene (BASELINE config 5)
and prints, per step and field, the number of differing entries and their bounding box (i, j, k) -- a debugging aid for
row / plane geometries the unit-test grids do not reach (written to find what broke parity at 1024 x 1024 x 32).

    python tools/diag_rows.py --grid 1024 1024 16 --steps 2 --iters 200 [--fl-opt ID=VALUE ...] [--solver-opt ID=VALUE ...]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs=3, default=[1024, 1024, 16])
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--fl-opt", action="append", default=[])
    ap.add_argument("--solver-opt", action="append", default=[])
    ap.add_argument("--full-state", type=int, default=1)
    a = ap.parse_args()
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd.scenes import leapfrog
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    from oracle_lib import OracleSolver
    nx, ny, nz = a.grid
    h = 1.0 / nx
    lib = bq.hip_lib()
    for kv in a.fl_opt:
        k, v = kv.split("=")
        lib.fl_set_option(int(k), int(v))
    em = leapfrog(nz, h)
    o = OracleSolver(nx, ny, nz, 1.0, 0.0, 1.0); o.set_smoke(0.0, 0.0, em); o.set_projection(a.iters, 0.5)
    s = BimocqGPUSolver(nx, ny, nz, 1.0, 0.0, 1.0); s.setSmoke(0.0, 0.0, em); s.setProjection(a.iters, 0.5)
    s.setOption(3, a.full_state)
    for kv in a.solver_opt:
        k, v = kv.split("=")
        s.setOption(int(k), int(v))
    dims = {"u": (nx + 1, ny, nz), "v": (nx, ny + 1, nz), "w": (nx, ny, nz + 1), "uinit": (nx + 1, ny, nz), "vinit": (nx, ny + 1, nz),
            "winit": (nx, ny, nz + 1)}
    names = ["bx", "by", "bz", "fx", "fy", "fz", "rho", "u", "v", "w", "p", "div", "uinit", "vinit", "winit", "rhoinit"]
    for f in range(a.steps):
        o.advance(f, 2.0 * h)
        s.advance(f, 2.0 * h)
        s._check()
        print(f"step {f + 1}: cfldt gpu {s.cfldt!r} oracle {o.cfldt!r}", flush=True)
        for name in names:
            x, y = o.field(name), s.field(name)
            bi, bj, bk = dims.get(name, (nx, ny, nz))
            same = (x == y) | (np.isnan(x) & np.isnan(y))
            bad = np.nonzero(~same)[0]
            if bad.size == 0:
                print(f"  {name:8s} identical")
                continue
            ii, jj, kk = bad % bi, (bad // bi) % bj, bad // (bi * bj)
            d = np.abs(x[bad].astype(np.float64) - y[bad].astype(np.float64))
            print(f"  {name:8s} {bad.size} of {x.size} differ: i {ii.min()}..{ii.max()} j {jj.min()}..{jj.max()} k {kk.min()}..{kk.max()}  "
                  f"max|diff| {d.max():.3e}  first (i,j,k)=({ii[0]},{jj[0]},{kk[0]}) oracle {x[bad[0]]!r} gpu {y[bad[0]]!r}", flush=True)
    o.close(); s.close()


if __name__ == "__main__":
    main()
