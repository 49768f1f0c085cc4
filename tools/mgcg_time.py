#!/usr/bin/env python3
"""Time the bimocq3D step with the fp64 multigrid-CG projection on the GPU.

    python tools/mgcg_time.py [--n 256] [--iters 50] [--steps 5]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gpufluidsimulation_amd.solver import BimocqGPUSolver


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    n = a.n
    s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0)
    s.setSmoke(0.0, 1.0, [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)])
    s.setProjection(a.iters, 0.5, 1)
    dt = 2.0 / n
    for f in range(a.steps):
        s.advance(f, dt)
        h = s.mgHistory()
        print(f"step {f}: {s.lib.bq_solver_last_ms(s.s):8.2f} ms   residual peak {h[2000]:.3e} -> {h[2000 + a.iters]:.3e}", flush=True)


if __name__ == "__main__":
    main()
