"""Deterministic analytic test inputs (no RNG): smooth velocities, near-identity maps, scalars.

All arrays are flat float32 in the reference layout (x fastest): index = i + nx*j + nx*ny*k.
"""
import numpy as np

F = np.float32


def _grid(nx, ny, nz):
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    return i.astype(np.float64), j.astype(np.float64), k.astype(np.float64)


def sizes(ni, nj, nk):
    return ni * nj * nk, (ni + 1) * nj * nk, ni * (nj + 1) * nk, ni * nj * (nk + 1)


def velocity(ni, nj, nk, h, amp=0.35):
    """A swirling, mildly divergent MAC velocity, |u| <= amp."""
    L = np.array([ni, nj, nk], dtype=np.float64) * h
    out = []
    for (nx, ny, nz), (ox, oy, oz), ph in (((ni + 1, nj, nk), (-0.5, 0, 0), 0.0),
                                           ((ni, nj + 1, nk), (0, -0.5, 0), 1.3),
                                           ((ni, nj, nk + 1), (0, 0, -0.5), 2.1)):
        i, j, k = _grid(nx, ny, nz)
        x, y, z = (i + ox) * h / L[0], (j + oy) * h / L[1], (k + oz) * h / L[2]
        f = amp * (np.sin(2 * np.pi * x + ph) * np.cos(2 * np.pi * y - 0.4 * ph) * np.cos(np.pi * z + 0.3)
                   + 0.25 * np.cos(4 * np.pi * z + ph) * np.sin(2 * np.pi * y))
        out.append(np.ascontiguousarray(f.astype(F).ravel()))
    return out


def identity_maps(ni, nj, nk, h):
    i, j, k = _grid(ni, nj, nk)
    h = F(h)
    return [np.ascontiguousarray((c.astype(F) * h).ravel()) for c in (i, j, k)]


def warped_maps(ni, nj, nk, h, amp=0.8, phase=0.0):
    """identity + a smooth displacement of up to `amp` cells, vanishing on the outer two layers
    (like a map produced by forward_kernel / DMC, which leave nodes outside 2..n-3 untouched)."""
    i, j, k = _grid(ni, nj, nk)
    xs = identity_maps(ni, nj, nk, h)
    win = np.ones_like(i)
    for c, n in ((i, ni), (j, nj), (k, nk)):
        win *= ((c > 1) & (c < n - 2))
    x, y, z = i / ni, j / nj, k / nk
    d = [np.sin(2 * np.pi * y + phase) * np.cos(2 * np.pi * z) * np.sin(np.pi * x),
         np.cos(2 * np.pi * x - phase) * np.sin(np.pi * y) * np.sin(2 * np.pi * z + 0.5),
         np.sin(2 * np.pi * x + 0.7) * np.cos(2 * np.pi * y + phase) * np.sin(np.pi * z)]
    out = []
    for base, disp, n in zip(xs, d, (ni, nj, nk)):
        m = base.reshape(nk, nj, ni).astype(np.float64) + amp * h * disp * win
        m = np.clip(m, h, (n - 1) * h) * win + base.reshape(nk, nj, ni) * (1 - win)
        out.append(np.ascontiguousarray(m.astype(F).ravel()))
    return out


def scalar(nx, ny, nz, phase=0.0, amp=1.0):
    i, j, k = _grid(nx, ny, nz)
    x, y, z = i / nx, j / ny, k / nz
    f = amp * (np.sin(2 * np.pi * x + phase) * np.sin(3 * np.pi * y + 0.2) * np.cos(2 * np.pi * z - phase)
               + 0.5 * np.cos(5 * np.pi * x * y + phase) + 0.1 * z)
    return np.ascontiguousarray(f.astype(F).ravel())


def same(a, b):
    """value equality for parity: -0 == +0, NaN == NaN in the same places."""
    return a.shape == b.shape and bool(np.array_equal(a, b, equal_nan=True))


def maxdiff(a, b):
    return float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
