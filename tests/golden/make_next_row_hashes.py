"""Per-step SHA-256 of the CPU oracle's rho, u, v, w for the "next" rows of SURVEY 8(f) at sizes the toy tests do not
reach: the fp64 multigrid-CG projection (N1) at 128^3, the MAC_REFLECTION scheme (N3) at 128^3 with the Jacobi
projection, the reference binary's default configuration (reflection + multigrid-CG) at 64^3, and one multigrid-CG step /
two reflection steps at BASELINE's 256^3.

    python tests/golden/make_next_row_hashes.py      # rewrites tests/golden/next_row_hashes.json (minutes of CPU)

Same hashing as make_hashes.py (value equality).  tests/test_gpu_full_size.py::test_hip_reproduces_next_row_hashes
recomputes them from the HIP path on the GPU box (no oracle in the loop there)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

OUT = os.path.join(HERE, "next_row_hashes.json")
# name -> (grid, steps, scheme (0 BiMocq / 3 reflection), projection kind (0 Jacobi / 1 multigrid-CG), iterations)
CASES = {"mgcg_128": (128, 2, 0, 1, 50), "reflection_128": (128, 3, 3, 0, 200), "reflection_mgcg_64": (64, 3, 3, 1, 50),
         "mgcg_256": (256, 1, 0, 1, 50), "reflection_256": (256, 2, 3, 0, 200)}


def main():
    from make_hashes import FIELDS, SMOKE, digest_hex
    from oracle_lib import OracleSolver
    out = {"scene": {"emitter": SMOKE, "L": 1.0, "dt": "2h", "viscosity": 0.0, "blend": 1.0, "drop": 0.0, "rise": 1.0, "halfrdx": 0.5},
           "cases": {}}
    for name, (n, steps, scheme, kind, iters) in CASES.items():
        s = OracleSolver(n, n, n, 1.0, 0.0, 1.0)
        s.set_smoke(0.0, 1.0, [SMOKE])
        s.set_projection(iters, 0.5, kind)
        if scheme:
            s.set_option(3, scheme)
        rows = []
        t0 = time.time()
        for f in range(steps):
            s.advance(f, 2.0 / n)
            row = {"step": f + 1, "cfldt": float(np.float32(s.cfldt))}
            for k in FIELDS:
                a = s.field(k)
                row[k] = digest_hex(a)
                if k in ("rho", "v"):
                    row[k + "_sum"] = float(a.astype(np.float64).sum())
                    row[k + "_absmax"] = float(np.abs(a).max())
            rows.append(row)
            print(f"{name} step {f + 1}/{steps}  {time.time() - t0:.0f} s", flush=True)
        s.close()
        out["cases"][name] = {"grid": n, "scheme": scheme, "projection_kind": kind, "iterations": iters, "rows": rows}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
