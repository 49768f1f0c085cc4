#!/usr/bin/env python3
"""A few steps of the rising-smoke scene on one GPU and nothing else: the program bench.py puts under `rocprofv3 --pmc` to read
hardware counters of the step's kernels in a run of its own (no torch import, no timing, no output).

    python tools/step_child.py --n 256 --steps 3 --warmup 2 --jacobi-iters 6
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--jacobi-iters", type=int, default=6)
    ap.add_argument("--fl-opt", action="append", default=[], metavar="ID=VALUE", help="fl_set_option(ID, VALUE) before the run")
    a = ap.parse_args()
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd.scenes import rising_smoke
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    n = a.n
    for kv in a.fl_opt:
        k, v = kv.split("=")
        bq.hip_lib().fl_set_option(int(k), int(v))
    s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0)
    s.setSmoke(0.0, 1.0, rising_smoke(n, 1.0 / n))
    s.setProjection(a.jacobi_iters, 0.5)
    s.setOption(3, 1)                                   # the reference's full per-step sequence, as in bench.py
    for f in range(a.warmup + a.steps):
        s.advance(f, 2.0 / n)
    bq.hip_lib().fl_sync()
    s.close()           # (no fl_shutdown: the library tears itself down at exit, which is what this child also verifies)


if __name__ == "__main__":
    main()
