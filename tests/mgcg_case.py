"""Buffers of one gpu_multi_grid_conjugate_gradient call, laid out as BimocqGPUSolver allocates them
(BimocqGPUSolver.cpp:60-90), for the oracle (numpy) and -- through DeviceBuffer64 -- for the HIP path."""
import ctypes as C

import numpy as np

import fields as F
from oracle_lib import CoarseLevel, level_dims


def velocity(ni, nj, nk, h):
    """a smooth, clearly non-solenoidal staggered velocity field"""
    u, v, w = F.velocity(ni, nj, nk, h)
    x = (np.arange(ni + 1, dtype=np.float64) * h)[None, None, :]
    y = (np.arange(nj, dtype=np.float64) * h + 0.5 * h)[None, :, None]
    z = (np.arange(nk, dtype=np.float64) * h + 0.5 * h)[:, None, None]
    bump = 0.4 * np.sin(2.1 * np.pi * x) * np.cos(1.3 * np.pi * y) * np.cos(0.7 * np.pi * z + 0.3)
    u = (u.reshape(nk, nj, ni + 1) + bump).astype(np.float32).ravel()
    return u, v, w


class HostCase:
    """numpy-side buffers + the level table for the oracle"""

    def __init__(self, ni, nj, nk, levels):
        self.dims = level_dims(ni, nj, nk, levels)
        assert all(min(d) >= 1 for d in self.dims), self.dims
        n = ni * nj * nk
        self.n = n
        self.div, self.p, self.dir, self.residual, self.temp0, self.temp1 = (np.zeros(n, np.float64) for _ in range(6))
        self.result = np.zeros(4096, np.float64)
        self.lb, self.lx, self.lr = [], [], []
        self.table = (CoarseLevel * levels)()
        for l, (a, b, c) in enumerate(self.dims):
            m = a * b * c
            self.lb.append(np.zeros(m, np.float64)); self.lx.append(np.zeros(m, np.float64)); self.lr.append(np.zeros(m, np.float64))
            t = self.table[l]
            t.ni, t.nj, t.nk, t.number, t.alpha, t.beta = a, b, c, m, -1.0, 1.0 / 6.0
            t.b, t.x, t.r = self.lb[l].ctypes.data, self.lx[l].ctypes.data, self.lr[l].ctypes.data

    def interior_div_norm(self, u, v, w, hr=1.0):
        ni, nj, nk = self.dims[0]
        U = u.reshape(nk, nj, ni + 1).astype(np.float64); V = v.reshape(nk, nj + 1, ni).astype(np.float64)
        W = w.reshape(nk + 1, nj, ni).astype(np.float64)
        d = hr * ((U[:, :, 1:] - U[:, :, :-1]) + (V[:, 1:, :] - V[:, :-1, :]) + (W[1:] - W[:-1]))
        return float(np.sqrt(np.mean(d[3:-3, 3:-3, 3:-3] ** 2)))
