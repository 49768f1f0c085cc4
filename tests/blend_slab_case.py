"""The two-level advection (blend != 1, GPU_kernel.cu:236-310) on a z-slab rank when its second look-up meets the zeroed border
cells of the previous backward map: arrays of one global case and the local views a slab rank holds of them.
Shared by the oracle-only CPU test and the HIP parity test."""
import numpy as np

import fields as F


def global_case(ni, nj, nk, h):
    """current backward map: identity carried 1.6 / 1.3 / 0.9 cells towards the high walls inside the map window (so that the
    outermost nodes of the operator's window look up the previous map in its border cells); previous map: a smooth warp whose
    border nodes (index <= 1 or >= n - 2 on any axis) are zero in ALL THREE components, as the DMC update leaves them"""
    h = float(np.float32(h))
    k, j, i = np.meshgrid(np.arange(nk), np.arange(nj), np.arange(ni), indexing="ij")
    live = np.ones((nk, nj, ni), bool)
    for c, n in ((i, ni), (j, nj), (k, nk)):
        live &= (c > 1) & (c < n - 2)
    ident = F.identity_maps(ni, nj, nk, h)
    back = []
    for base, shift, n in zip(ident, (1.6, 1.3, 0.9), (ni, nj, nk)):
        m = base.reshape(nk, nj, ni).astype(np.float64) + shift * h * live
        back.append(np.ascontiguousarray(np.clip(m, h, (n - 1) * h).astype(np.float32).ravel()))
    backp = [np.ascontiguousarray((a.reshape(nk, nj, ni) * live).astype(np.float32).ravel())
             for a in F.warped_maps(ni, nj, nk, h, 0.5, 2.0)]
    prev = [F.scalar(ni + 1, nj, nk, 1.1), F.scalar(ni, nj + 1, nk, 1.2), F.scalar(ni, nj, nk + 1, 1.3), F.scalar(ni, nj, nk, 1.9)]
    cur = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3), F.scalar(ni, nj, nk, 0.9)]
    return h, back, backp, prev, cur


PLANES = lambda ni, nj: ((ni + 1) * nj, ni * (nj + 1), ni * nj, ni * nj)     # u, v, w, scalar
EXTRA = (0, 0, 1, 0)


def local_view(a, plane, extra, nkg, own0, own1, G):
    """the planes [own0 - G, own1 + G) (+ the top face of a w-type buffer) of a global array; planes outside the grid are zero"""
    nkl = own1 - own0 + 2 * G
    g = a.reshape(nkg + extra, plane)
    out = np.zeros((nkl + extra, plane), np.float32)
    for kl in range(nkl + extra):
        kg = own0 - G + kl
        if 0 <= kg < nkg + extra:
            out[kl] = g[kg]
    return np.ascontiguousarray(out.ravel())


def owned(a, plane, extra, own0, own1, G, local, last):
    """the planes a rank owns (a w-type buffer's top face goes with the last rank)"""
    n = own1 - own0 + (1 if extra and last else 0)
    k0 = G if local else own0
    return a.reshape(-1, plane)[k0:k0 + n]
