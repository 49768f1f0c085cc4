"""Per-step SHA-256 of the CPU oracle's rho, u, v, w at BASELINE's full sizes (configs 2 and 3: 128^3 and 256^3
rising smoke, 200 Jacobi iterations, halfrdx 0.5) -- full-size evidence without shipping full-size fixtures.

    python tests/golden/make_hashes.py            # rewrites tests/golden/rising_smoke_hashes.json (minutes of CPU)

The hash is over the canonicalised values (op_vectors.digest: -0 == +0, every NaN alike), so it pins value equality,
the parity bar of this repository.  tests/test_golden.py::test_hip_reproduces_full_size_hashes recomputes them from the
HIP path on the GPU box (no oracle in the loop there)."""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

OUT = os.path.join(HERE, "rising_smoke_hashes.json")
SMOKE = (0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)
CASES = {"128": (128, 60), "256": (256, 2)}              # grid -> steps
FIELDS = ("rho", "u", "v", "w")


def digest_hex(a):
    c = np.where(np.isnan(a), np.float32(np.nan), a + np.float32(0.0)).astype("<f4")
    return hashlib.sha256(c.tobytes()).hexdigest()


def main():
    from oracle_lib import OracleSolver
    out = {"scene": {"emitter": SMOKE, "L": 1.0, "dt": "2h", "viscosity": 0.0, "blend": 1.0, "drop": 0.0, "rise": 1.0,
                     "jacobi_iters": 200, "halfrdx": 0.5}, "cases": {}}
    for name, (n, steps) in CASES.items():
        s = OracleSolver(n, n, n, 1.0, 0.0, 1.0)
        s.set_smoke(0.0, 1.0, [SMOKE])
        s.set_projection(200, 0.5)
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
            print(f"{name}^3 step {f + 1}/{steps}  {time.time() - t0:.0f} s", flush=True)
        s.close()
        out["cases"][name] = rows
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
