"""RMS(fast-lerp run - exact run) of rho, u, v, w on the GPU at a given size (the exact HIP path is bit-identical to the
CPU oracle, so this is the fast variant's deviation from the reference arithmetic).  FL_OPT_FAST_LERP is a library
option: it is switched around each solver's advance().

    python tools/fast_lerp_deviation.py [--size 128] [--steps 200] [--out profiles/...json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    lib = bq.hip_lib()
    n = a.size
    em = [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)]
    ex = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0); ex.setSmoke(0.0, 1.0, em); ex.setProjection(a.iters, 0.5)
    fa = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0); fa.setSmoke(0.0, 1.0, em); fa.setProjection(a.iters, 0.5)
    rows = []
    for f in range(a.steps):
        ex.advance(f, 2.0 / n)
        lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 1)
        fa.advance(f, 2.0 / n)
        lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 0)
        if (f + 1) % 20 == 0 or f + 1 == a.steps:
            row = {"step": f + 1}
            for k in ("rho", "u", "v", "w"):
                x, y = ex.field(k).astype(np.float64), fa.field(k).astype(np.float64)
                row[k] = {"rms": float(np.sqrt(np.mean((x - y) ** 2))), "max": float(np.abs(x - y).max()),
                          "rms_of_field": float(np.sqrt(np.mean(x * x)))}
            rows.append(row)
            print(f"step {f + 1:3d}: " + "  ".join(f"{k} {row[k]['rms']:.2e}" for k in ("rho", "u", "v", "w")), flush=True)
    bq.check()
    out = {"grid": [n, n, n], "steps": a.steps, "jacobi_iters": a.iters,
           "what": "RMS over all entries of (FL_OPT_FAST_LERP = 1 run) - (exact run = CPU oracle bit for bit), rising smoke",
           "worst_rms": max(r[k]["rms"] for r in rows for k in ("rho", "u", "v", "w")), "checkpoints": rows}
    print(f"worst RMS {out['worst_rms']:.3e} (north star tolerance 1e-5)")
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
