#!/usr/bin/env python3
"""A/B timing of FL_OPT_MGCG_FUSE inside ONE process (the same allocations, hence the same placement in memory -- between processes
the level-0 kernels of the multigrid solver vary by up to 15 % on the same box): the MGCG-mode step at 256^3, rising smoke, with
the fusion off (0), the wave-per-row kernels (1) and the two-rows-per-thread kernels (3), interleaved, three rounds."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    import torch
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    from gpufluidsimulation_amd.scenes import rising_smoke
    lib = bq.hip_lib()
    n = a.size
    s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0, device=0, scheme=0)
    s.setSmoke(0.0, 1.0, rising_smoke(n, 1.0 / n))
    s.setProjection(a.iters, 0.5, 1)
    s.setOption(3, 1)
    dt = 0.5 / n
    frame = 0
    for _ in range(3):
        s.advance(frame, dt); frame += 1
    lib.fl_sync()
    out = {}
    for r in range(a.rounds):
        for mode in (0, 1, 3):
            lib.fl_set_option(bq._lib.FL_OPT_MGCG_FUSE, mode)
            s.advance(frame, dt); frame += 1            # (the V-cycle graph is recaptured when the mode changes)
            lib.fl_sync()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                s.advance(frame, dt); frame += 1
            lib.fl_sync()
            out.setdefault(mode, []).append(round((time.perf_counter() - t0) / a.steps * 1e3, 3))
    lib.fl_set_option(bq._lib.FL_OPT_MGCG_FUSE, -1)
    s._check()
    s.close()
    print(json.dumps({"size": n, "mg_iters": a.iters, "ms_per_step": {"fuse_off": out[0], "wave_per_row": out[1], "two_rows_per_thread": out[3]}}))


if __name__ == "__main__":
    main()
