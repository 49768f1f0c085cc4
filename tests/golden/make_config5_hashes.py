"""Per-step SHA-256 of the CPU oracle's rho, u, v, w for BASELINE config 5's scene at its own ROW geometry: the
leapfrogging vortex rings (gpufluidsimulation_amd/scenes.py: leapfrog) on 1024 x 1024 x 32 cells -- rows of 1024 / 1025
floats and planes of 1 M cells as in the 1024 x 1024 x 512 grid, 32 planes deep so that the oracle finishes in minutes --
200 Jacobi iterations, halfrdx 0.5, dt = 2h, two steps (both inside the emitters' ten frames, so the velocity ring of
emit_smoke_velocity_kernel with its acosf / cosf is imposed twice).

    python tests/golden/make_config5_hashes.py          # rewrites tests/golden/config5_hashes.json (~10 min of CPU, ~10 GB)

tests/test_gpu_config5.py recomputes the hashes from the HIP path on the GPU box (no oracle in the loop there)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_hashes import FIELDS, digest_hex      # noqa: E402

OUT = os.path.join(HERE, "config5_hashes.json")
GRID = (1024, 1024, 32)
STEPS = 2


def main():
    from oracle_lib import OracleSolver
    from gpufluidsimulation_amd.scenes import leapfrog
    nx, ny, nz = GRID
    h = 1.0 / nx
    em = leapfrog(nz, h)
    out = {"scene": {"name": "leapfrog", "emitters": em, "L": 1.0, "dt": "2h", "viscosity": 0.0, "blend": 1.0, "drop": 0.0,
                     "rise": 0.0, "jacobi_iters": 200, "halfrdx": 0.5}, "grid": list(GRID), "rows": []}
    s = OracleSolver(nx, ny, nz, 1.0, 0.0, 1.0)
    s.set_smoke(0.0, 0.0, em)
    s.set_projection(200, 0.5)
    t0 = time.time()
    for f in range(STEPS):
        s.advance(f, 2.0 * h)
        row = {"step": f + 1, "cfldt": float(np.float32(s.cfldt))}
        for k in FIELDS:
            a = s.field(k)
            row[k] = digest_hex(a)
            if k in ("rho", "u"):
                row[k + "_sum"] = float(a.astype(np.float64).sum())
                row[k + "_absmax"] = float(np.abs(a).max())
        out["rows"].append(row)
        print(f"step {f + 1}/{STEPS}  {time.time() - t0:.0f} s", flush=True)
    s.close()
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
