#!/usr/bin/env python3
"""SURVEY 8(d) parity figure as a file: RMS(gpu - oracle) of rho, u, v, w after 200 steps of the rising-smoke scene,
absolute and relative to RMS(oracle), for the exact arithmetic variant and for FL_OPT_FAST_LERP (against the exact
oracle), at grids the CPU oracle finishes in about a minute.  (Pick N with 0.2 N and 0.5 N not both integers: a node exactly on
the source's axis makes the reference's emitter normalise a zero vector -- NaN velocities from step 0, in the oracle and
on the GPU alike; N = 40 is such a grid.)  Test infrastructure (uses the oracle): not a product path.

    python tools/parity_report.py [--n 32 48] [--steps 200] > profiles/<name>.json
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    # the OpenMP oracle must not start more threads than the job's CPU share (a GPU box hands out 16 cores of a big
    # host: the default of one thread per visible core makes every barrier crawl)
    import oracle_lib
    oracle_lib.limit_openmp_threads()
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[32, 48])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--fl-opt", action="append", default=[], metavar="K=V",
                    help="fl_set_option(K, V) before the runs, e.g. 9=8: FL_OPT_JACOBI_KCHUNK2 = 8 puts the LDS-exchanged "
                         "three-sweep Jacobi kernel to work on these small grids")
    a = ap.parse_args()
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    from oracle_lib import OracleSolver
    lib = bq.hip_lib()
    assert lib.fl_init(0) == 0
    for kv in a.fl_opt:
        k, v = kv.split("=")
        lib.fl_set_option(int(k), int(v))
    out = {"options": a.fl_opt, "scene": "rising smoke (SURVEY 8d): L=1, dt=2h, nu=0, blend=1, one source at step 0, Jacobi %d iterations, halfrdx 0.5" % a.iters,
           "steps": a.steps, "tolerance_rms_abs": 1e-5, "grids": {}}
    for n in a.n:
        em = [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)]
        o = OracleSolver(n, n, n, 1.0, 0.0, 1.0); o.set_smoke(0.0, 1.0, em); o.set_projection(a.iters, 0.5)
        ex = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0); ex.setSmoke(0.0, 1.0, em); ex.setProjection(a.iters, 0.5)
        dt = 2.0 / n
        for f in range(a.steps):
            o.advance(f, dt); ex.advance(f, dt)
            if f % 10 == 9:
                print(f"[parity_report] {n}^3 step {f + 1}/{a.steps}", file=sys.stderr, flush=True)
        ref = {k: o.field(k).astype(np.float64) for k in ("rho", "u", "v", "w")}
        exact = {k: ex.field(k).astype(np.float64) for k in ref}
        jacobi_kernel = (lib.fl_jacobi_kernel_name() or b"").decode()
        ex.close()
        lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 1)
        fa = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0); fa.setSmoke(0.0, 1.0, em); fa.setProjection(a.iters, 0.5)
        for f in range(a.steps):
            fa.advance(f, dt)
        fast = {k: fa.field(k).astype(np.float64) for k in ref}
        fa.close()
        lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 0)
        o.close()
        g = {}
        for name, fields in (("exact", exact), ("fast_lerp", fast)):
            g[name] = {}
            for k in ref:
                rms = float(np.sqrt(np.mean((fields[k] - ref[k]) ** 2))); base = float(np.sqrt(np.mean(ref[k] ** 2)))
                g[name][k] = {"rms_abs": rms, "rms_rel": rms / base if base else 0.0, "max_abs": float(np.abs(fields[k] - ref[k]).max()),
                              "bit_identical": bool(np.array_equal(fields[k], ref[k], equal_nan=True)),
                              "all_finite": bool(np.isfinite(fields[k]).all() and np.isfinite(ref[k]).all())}
        g["jacobi_kernel"] = jacobi_kernel
        out["grids"][f"{n}^3"] = g
    bq.check()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
