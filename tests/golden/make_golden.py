"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/bimocq_oracle.c).

What these fixtures are: the reference (CUDA + OpenVDB + tbb + Boost) cannot be built or run in this
environment and its repository holds no golden vectors for the bimocq3D path, so the only
numbers that come from an actual reference run are the per-step statistics SURVEY.md records;
tests/test_oracle_kat.py pins the oracle on those.  The vectors written here are the ORACLE's own
outputs on three small scenes, frozen so that (a) a later change to the oracle that alters results
is caught on the CPU, and (b) the HIP path can be checked against committed data on the GPU box
independently of the gcc build there.  Inputs are fully described by SCENES below.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

# name -> (dims, L, viscosity, blend, emitters, drop, rise, jacobi iters, halfrdx, dt in cells, steps)
SCENES = {
    "rising_smoke_16": ((16, 16, 16), 1.0, 0.0, 1.0, [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)], 0.0, 1.0, 20, 0.5, 2.0, 4),
    "noncubic_blend_20x12x10": ((20, 12, 10), 0.6, 0.0, 0.7,
                                [(0.2, 0.26, 0.21, 0.09, 1.0, 2.0, 0.0, 3), (0.4, 0.27, 0.19, 0.09, 0.5, 1.5, 0.0, 3)],
                                0.1, 1.0, 12, 1.0, 3.0, 4),
    "viscous_12x16x10": ((12, 16, 10), 1.0, 2e-3, 1.0, [(0.5, 0.3, 0.4, 0.15, 1.0, 2.0, 0.0, 2)], 0.0, 1.0, 8, 0.5, 2.0, 3),
}
FIELDS = ["rho", "T", "u", "v", "w", "p"]


def run_oracle(scene):
    from oracle_lib import OracleSolver
    dims, L, visc, blend, emitters, drop, rise, iters, hr, dt_cells, steps = scene
    o = OracleSolver(*dims, L, visc, blend)
    o.set_smoke(drop, rise, emitters)
    o.set_projection(iters, hr)
    dt = dt_cells * float(np.float32(L) / np.float32(dims[0]))
    out = {}
    for f in range(steps):
        o.advance(f, dt)
        out[f"cfldt_{f}"] = np.float32(o.cfldt)
    for name in FIELDS:
        out[name] = o.field(name).copy()
    o.close()
    return out


if __name__ == "__main__":
    import oracle_lib
    oracle_lib.build()
    for name, scene in SCENES.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **run_oracle(scene))
        print("wrote", name)
    import op_vectors
    op_vectors.generate()
