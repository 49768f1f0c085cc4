import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from gpufluidsimulation_amd.solver import BimocqGPUSolver
n = 256
s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0); s.setSmoke(0.0, 1.0, [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)]); s.setProjection(200, 0.5)
dt = 2.0 / n
mx = 0
for f in range(200):
    s.advance(f, dt)
    v = (1.0 / n) / s.cfldt
    mx = max(mx, v)
    if f % 20 == 19: print(f, "max|vel|", round(v, 4), "cells/step", round(v * dt * n, 3), "ms", round(s.lib.bq_solver_last_ms(s.s), 2), flush=True)
print("peak", mx)
rho = s.field("rho"); print("rho sum", float(rho.sum()), "finite", bool(np.isfinite(rho).all()))
