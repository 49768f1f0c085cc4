"""Per-step SHA-256 of the CPU oracle at the north star's own LENGTH (round 4):

    128_200          128^3 rising smoke, 200 steps (two DMC sub-steps per step from step ~41 on), hashes every 10th step
    128_200_fast     the same run of the oracle in its one-fma mode (orc_set_fast_lerp) -- pins the HIP library's fast variant
    256_12           256^3, 12 steps, every step (at 256^3 the scene of SURVEY 8(d) needs ~80 steps to exceed one cell per step)
    256_rise8_12     256^3 with the buoyancy coefficient `rise` = 8: CFL > 1 from step ~10 on -- fields of 2^24 elements in
                     the two-sub-step regime, displacements beyond one cell, in 12 steps

    python tests/golden/make_long_hashes.py [case ...]      # hours of CPU in all; writes tests/golden/long_run_hashes.json

A case that is already in the file is kept unless named on the command line.  Same digest as make_hashes.py (value
equality: -0 == +0, every NaN alike).  tests/test_gpu_full_size.py recomputes them from the HIP path (no oracle there)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_hashes import FIELDS, SMOKE, digest_hex      # noqa: E402

OUT = os.path.join(HERE, "long_run_hashes.json")
# name -> (grid, steps, hash every, rise, fast)
CASES = {"128_200": (128, 200, 10, 1.0, 0), "128_200_fast": (128, 200, 10, 1.0, 1),
         "256_12": (256, 12, 1, 1.0, 0), "256_rise8_12": (256, 12, 1, 8.0, 0)}


def main():
    from oracle_lib import OracleSolver, lib as oracle
    out = {"cases": {}}
    if os.path.exists(OUT):
        out = json.load(open(OUT))
    todo = sys.argv[1:] or [c for c in CASES if c not in out["cases"]]
    for name in todo:
        n, steps, every, rise, fast = CASES[name]
        s = OracleSolver(n, n, n, 1.0, 0.0, 1.0)
        s.set_smoke(0.0, rise, [SMOKE])
        s.set_projection(200, 0.5)
        rows, t0 = [], time.time()
        for f in range(steps):
            oracle().orc_set_fast_lerp(fast)
            s.advance(f, 2.0 / n)
            oracle().orc_set_fast_lerp(0)
            row = {"step": f + 1, "cfldt": float(np.float32(s.cfldt))}
            if (f + 1) % every == 0 or f + 1 == steps:
                for k in FIELDS:
                    a = s.field(k)
                    row[k] = digest_hex(a)
                    if k in ("rho", "v"):
                        row[k + "_absmax"] = float(np.abs(a).max())
            rows.append(row)
            print(f"{name} step {f + 1}/{steps}  {time.time() - t0:.0f} s  substeps {2.0 / n / s.cfldt:.2f}", flush=True)
        s.close()
        out["cases"][name] = {"grid": n, "steps": steps, "rise": rise, "fast_lerp": fast, "jacobi_iters": 200, "halfrdx": 0.5,
                              "emitter": SMOKE, "rows": rows}
        with open(OUT, "w") as fh:
            json.dump(out, fh, indent=1)
        print("wrote", name, "to", OUT, flush=True)


if __name__ == "__main__":
    main()
