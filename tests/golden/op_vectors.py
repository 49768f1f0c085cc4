"""Per-operator golden vectors (deterministic analytic inputs on a non-cubic 24x20x16 grid and on
32^3, outputs frozen).  ONE table drives both back ends: the oracle (orc_<op>, numpy arrays) and the HIP
library (gpu_<op>, device buffers) receive the same argument list, so a case reads like the reference's own
call site.  Stored per case: every array argument the call CHANGED -- the values themselves on the small grid, a SHA-256
of the canonicalised values (-0 -> +0, one NaN pattern) on 32^3 -- so the whole set stays under 2 MB.

    python tests/golden/make_golden.py        # regenerates ops_*.npz from the oracle as well
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fields as F                                                          # noqa: E402

GRIDS = {"24x20x16": (24, 20, 16, float(np.float32(1.0 / 24))), "32": (32, 32, 32, float(np.float32(1.0 / 32)))}


def cases(ni, nj, nk, h):
    """name -> (op, [array arguments in call order], [scalar arguments])"""
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    u, v, w = F.velocity(ni, nj, nk, h)
    fwd, back, backp = F.warped_maps(ni, nj, nk, h, 0.8, 0.3), F.warped_maps(ni, nj, nk, h, -0.7, 1.1), F.warped_maps(ni, nj, nk, h, 0.5, 2.0)
    ui, vi, wi = F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)
    z = lambda c: np.zeros(c, np.float32)                                   # noqa: E731
    g = [h, ni, nj, nk]
    cfldt = 0.9 * h / 0.35
    out = {
        "solve_forward": ("solve_forward", [u, v, w] + [a.copy() for a in F.identity_maps(ni, nj, nk, h)], g + [cfldt, 2.2 * h]),
        "solve_backwardDMC": ("solve_backwardDMC", [u, v, w] + [a.copy() for a in back] + [z(n), z(n), z(n)], g + [cfldt]),
        "advect_velocity": ("advect_velocity", [z(nu), z(nv), z(nw), ui, vi, wi] + back, g + [0]),
        "advect_field": ("advect_field", [z(n), F.scalar(ni, nj, nk, 0.4)] + back, g + [0]),
        "advect_field_point": ("advect_field", [z(n), F.scalar(ni, nj, nk, 0.4)] + back, g + [1]),
        "advect_vel_double": ("advect_vel_double", [ui.copy(), vi.copy(), wi.copy(), F.scalar(ni + 1, nj, nk, 1.1), F.scalar(ni, nj + 1, nk, 1.2),
                                                    F.scalar(ni, nj, nk + 1, 1.3)] + back + backp, g + [0, 0.6]),
        "accumulate_velocity": ("accumulate_velocity", [u, v, w, ui.copy(), vi.copy(), wi.copy()] + fwd, g + [0, 2.0]),
        "accumulate_field": ("accumulate_field", [F.scalar(ni, nj, nk, 0.5), F.scalar(ni, nj, nk, 1.5)] + fwd, g + [0, -0.5]),
        "compensate_velocity": ("compensate_velocity", [u.copy(), v.copy(), w.copy(), ui.copy(), vi.copy(), wi.copy(), z(nu), z(nv), z(nw)] + fwd + back, g + [0]),
        "compensate_field": ("compensate_field", [F.scalar(ni, nj, nk, 0.4), F.scalar(ni, nj, nk, 0.9), z(nu)] + fwd + back, g + [0]),
        "estimate_distortion": ("estimate_distortion", [z(n)] + back + fwd, g),
        "semilag_u": ("semilag", [z(nu), ui, u, v, w], [1, 0, 0] + g + [cfldt, -1.5 * h]),
        "semilag_scalar": ("semilag", [z(n), F.scalar(ni, nj, nk, 0.4), u, v, w], [0, 0, 0] + g + [cfldt, 1.5 * h]),
        "emit_smoke": ("emit_smoke", [u.copy(), v.copy(), w.copy(), F.scalar(ni, nj, nk, 0.4), F.scalar(ni, nj, nk, 0.9)],
                       g + [0.43 * ni * h, 0.37 * nj * h, 0.52 * nk * h, 0.21 * nj * h, 1.0, 2.0, 0.0]),
        "add_buoyancy": ("add_buoyancy", [v.copy(), F.scalar(ni, nj, nk, 0.4), F.scalar(ni, nj, nk, 0.9)], [ni, nj, nk, 0.3, 1.7, 0.05]),
        "diffuse_field_u": ("diffuse_field", [u.copy(), z(nu), z(nu)], [ni + 1, nj, nk, 7, 0.13]),
        "projection_jacobi": ("projection_jacobi", [u.copy(), v.copy(), w.copy(), z(n), z(n), z(n), None], [ni, nj, nk, 10, 0.5, -1.0, float(np.float32(1.0 / 6.0))]),
        "clamp_extrema_v": ("clamp_extrema", [vi, (vi + F.scalar(ni, nj + 1, nk, 2.3, amp=0.6)).astype(np.float32), u, v, w],
                            [ni, nj + 1, nk, 0, 1, 0, 0.0, 0.5, 0.0, h, 1.7 * h / 0.35]),
    }
    return out


def run_oracle(case):
    from oracle_lib import fp, lib
    op, arrays, scalars = case
    arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float32).copy() for a in arrays]
    getattr(lib(), "orc_" + op)(*[None if a is None else fp(a) for a in arrs], *scalars)
    return [a for a in arrs if a is not None]


def run_hip(case):
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd import DeviceBuffer
    op, arrays, scalars = case
    hip = bq.hip_lib()
    bufs = [None if a is None else DeviceBuffer.from_numpy(a) for a in arrays]
    getattr(hip, "gpu_" + op)(*[None if b is None else b.ptr for b in bufs], *scalars)
    bq.check()
    return [b.numpy() for b in bufs if b is not None]


def path(grid):
    return os.path.join(HERE, f"ops_{grid}.npz")


def digest(a):
    """SHA-256 of the values: -0 and +0, and all NaNs, hash alike (parity is value equality)"""
    import hashlib
    c = np.where(np.isnan(a), np.float32(np.nan), a + np.float32(0.0)).astype("<f4")
    return np.frombuffer(hashlib.sha256(c.tobytes()).digest(), dtype=np.uint8)


def changed_outputs(case, arrays_after):
    """(index, array) of the array arguments whose content differs from what went in"""
    ins = [a for a in case[1] if a is not None]
    return [(q, a) for q, (a, b) in enumerate(zip(arrays_after, ins))
            if not np.array_equal(a, np.asarray(b, dtype=np.float32), equal_nan=True)]


def check(grid, name, arrays_after, want):
    """compare a back end's results with the stored vectors; returns a list of mismatching argument indices"""
    case = cases(*GRIDS[grid])[name]
    bad = []
    stored = sorted(int(k.split(".")[1]) for k in want.files if k.split(".")[0] == name)
    got = dict(changed_outputs(case, arrays_after))
    if sorted(got) != stored:
        return ["changed-set", sorted(got), stored]
    for q, a in got.items():
        w = want[f"{name}.{q}"]
        ok = np.array_equal(w, digest(a)) if w.dtype == np.uint8 else (F.same(w, a))
        if not ok:
            bad.append(q)
    return bad


def generate():
    for grid, dims in GRIDS.items():
        blob = {}
        for name, case in cases(*dims).items():
            for q, a in changed_outputs(case, run_oracle(case)):
                blob[f"{name}.{q}"] = a if grid == "24x20x16" else digest(a)
        np.savez_compressed(path(grid), **blob)
        print("wrote", path(grid), f"{os.path.getsize(path(grid)) / 1e6:.2f} MB")
