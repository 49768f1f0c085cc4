"""GPU parity of the fp64 multigrid-CG projection: gpu_multi_grid_conjugate_gradient
through the C-ABI vs oracle/mgcg_oracle.c on identical buffers -- bit-exact (value equality) for the
pressure, the residual, the search direction, the CG coefficient history and the projected velocity;
including a second call on the same (now stale) buffers, as the solver issues it every step."""
import ctypes as C

import numpy as np
import pytest

import fields as F
from mgcg_case import HostCase, velocity
from oracle_lib import CoarseLevel, dp, fp, lib as oracle

pytestmark = pytest.mark.gpu


class Dev:
    """a device allocation mirroring a numpy array of any dtype"""

    def __init__(self, lib, host):
        self.lib, self.shape, self.dtype, self.nbytes = lib, host.shape, host.dtype, host.nbytes
        self.ptr = lib.fl_malloc(max(self.nbytes, 8))
        assert self.ptr
        lib.fl_memcpy_h2d(self.ptr, host.ctypes.data, self.nbytes)

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        self.lib.fl_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes)
        return out

    def __del__(self):
        if getattr(self, "ptr", None):
            self.lib.fl_free(self.ptr)
            self.ptr = None


@pytest.fixture(scope="module")
def hip():
    import gpufluidsimulation_amd as bq
    lib = bq.hip_lib()
    assert lib.fl_init(0) == 0, lib.fl_last_error_string()
    yield lib
    bq.check()


def device_case(lib, c, levels):
    d = {name: Dev(lib, getattr(c, name)) for name in ("div", "p", "dir", "residual", "temp0", "temp1", "result")}
    lb, lx, lr = [Dev(lib, a) for a in c.lb], [Dev(lib, a) for a in c.lx], [Dev(lib, a) for a in c.lr]
    table = (CoarseLevel * levels)()
    for l in range(levels):
        t, s = table[l], c.table[l]
        t.ni, t.nj, t.nk, t.number, t.alpha, t.beta = s.ni, s.nj, s.nk, s.number, s.alpha, s.beta
        t.b, t.x, t.r = lb[l].ptr, lx[l].ptr, lr[l].ptr
    return d, (lb, lx, lr), table


@pytest.mark.parametrize("ni,nj,nk,levels,iters,hr", [
    (24, 20, 16, 2, 3, 0.5),          # non-cubic, two levels (11x9x7 coarse)
    (40, 36, 32, 3, 3, 1.0),          # even -> odd -> even level dims (19x17x15, 9x8x7)
    (33, 20, 18, 2, 2, 0.5),          # odd fine dims: no out-of-array coarse reads
    (64, 64, 64, 4, 2, 0.5),          # 31, 15, 7
    (16, 16, 16, 1, 2, 0.5),          # one level: V-cycle = smoothing only
    (24, 20, 16, 2, 0, 0.5),          # no outer iteration: divergence, initial sums, gradient of p = 0
])
def test_mgcg_matches_oracle(hip, ni, nj, nk, levels, iters, hr, stale=None):
    import gpufluidsimulation_amd as bq
    h = 1.0 / ni
    u, v, w = velocity(ni, nj, nk, h)
    c = HostCase(ni, nj, nk, levels)
    if stale is not None:             # the work arrays arrive with arbitrary content (rim cells are never written by the stencils)
        rng = np.random.default_rng(stale)
        for a in (c.p, c.dir, c.residual, c.temp0, c.temp1, c.lb[0], c.lx[0], c.lr[0]):
            a[:] = rng.uniform(-2.0, 2.0, a.size)
    d, lv, table = device_case(hip, c, levels)
    du, dv, dw = Dev(hip, u), Dev(hip, v), Dev(hip, w)
    ou, ov, ow = u.copy(), v.copy(), w.copy()
    for call in range(2):             # the second call starts from the buffers the first one left behind
        oracle().orc_multi_grid_conjugate_gradient(fp(ou), fp(ov), fp(ow), dp(c.div), dp(c.p), dp(c.dir), dp(c.residual),
                                                   dp(c.temp0), dp(c.temp1), dp(c.result), c.table, levels, iters, hr)
        hip.gpu_multi_grid_conjugate_gradient(du.ptr, dv.ptr, dw.ptr, d["div"].ptr, d["p"].ptr, d["dir"].ptr,
                                              d["residual"].ptr, d["temp0"].ptr, d["temp1"].ptr, d["result"].ptr,
                                              C.cast(table, C.c_void_p), levels, iters, hr)
        bq.check()
        for name in ("div", "p", "dir", "residual"):
            assert F.same(getattr(c, name), d[name].numpy()), (call, name)
        got = d["result"].numpy()
        assert F.same(c.result[:2 * iters + 3], got[:2 * iters + 3]), call
        assert F.same(c.result[2000:2001 + iters], got[2000:2001 + iters]), call
        for l in range(levels):
            assert F.same(c.lx[l], lv[1][l].numpy()), (call, "x", l)
            assert F.same(c.lb[l], lv[0][l].numpy()), (call, "b", l)
        for a, b in ((ou, du), (ov, dv), (ow, dw)):
            assert F.same(a, b.numpy()), call
    if iters and stale is None:       # (stale rim cells take part in the reference's max)
        assert c.result[2000 + iters] < c.result[2000]      # the positive residual peak went down


def test_mgcg_with_fused_smoothing_on_every_level(hip):
    """the fused two-sweep smoother is chosen by size (level 0 of big grids only); forced on here so that the
    whole operator runs through it on 64^3, 31^3, 15^3 (one cell per lane, odd rows) and stays bit-identical"""
    import gpufluidsimulation_amd as bq
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
    try:
        test_mgcg_matches_oracle(hip, 64, 64, 64, 4, 2, 0.5)
        test_mgcg_matches_oracle(hip, 40, 36, 32, 3, 2, 1.0)
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)


@pytest.mark.parametrize("ni,nj,nk,levels,iters,stale", [
    (256, 16, 12, 2, 3, None),        # rows of two waves; odd iteration count: the direction ends in its own array
    (256, 16, 12, 2, 4, 7),           # even: moved out of levels[0].b before the last V-cycle; stale rim cells everywhere
    (256, 9, 7, 1, 2, 11),            # odd row count, one z-chunk, one level
    (512, 8, 10, 2, 3, 3),            # rows of four waves: two partials per row
    (256, 64, 64, 3, 2, None),        # 2^20 cells: fused by default
    (256, 37, 21, 2, 3, 5),           # rows that do not fill the last block of eight, z-chunks of uneven length
])
def test_mgcg_vector_updates_inside_the_stencil_passes(hip, ni, nj, nk, levels, iters, stale):
    """FL_OPT_MGCG_FUSE: update_x + residual, add + residual + max + dot, update_dir + A dir + dot as three launches
    (bq_mgcg_fused.hip.inc) -- bit-identical to the oracle's nine passes, the dot products' summation trees included, with
    stale values in every rim cell, twice in a row; and the same with the fusion off"""
    import gpufluidsimulation_amd as bq
    default_on = ni * nj * nk >= 1 << 20
    hip.fl_set_option(bq._lib.FL_OPT_MGCG_FUSE, 1 if default_on else 2)
    try:
        hip.fl_mg_fused_launches()
        test_mgcg_matches_oracle(hip, ni, nj, nk, levels, iters, 0.5, stale)
        assert hip.fl_mg_fused_launches() > 0
        if ni == 256:                 # 3 = the wave-per-row form of the kernels (rows of 256 cells only)
            hip.fl_set_option(bq._lib.FL_OPT_MGCG_FUSE, 3)
            test_mgcg_matches_oracle(hip, ni, nj, nk, levels, iters, 0.5, stale)
            assert hip.fl_mg_fused_launches() > 0
        hip.fl_set_option(bq._lib.FL_OPT_MGCG_FUSE, 0)
        test_mgcg_matches_oracle(hip, ni, nj, nk, levels, iters, 0.5, stale)
        assert hip.fl_mg_fused_launches() == 0
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_MGCG_FUSE, -1)


def test_mgcg_fused_vector_updates_on_random_shapes(hip):
    """seeded sweep for the fused level-0 kernels: rows of 256 and 512 cells, row / plane counts that leave partial row blocks and
    uneven z-chunks (chunk length forced through FL_OPT_JACOBI_KCHUNK2 in the wave-per-row form), 1-2 levels, 2-4 iterations,
    stale rim cells; both kernel forms"""
    import gpufluidsimulation_amd as bq
    rng = np.random.default_rng(20261005)
    try:
        for case in range(10):
            ni = 512 if case % 5 == 4 else 256
            nj, nk = int(rng.integers(5, 34)), int(rng.integers(7, 26))
            levels = 1 if min(nj, nk) < 9 else int(rng.integers(1, 3))
            iters, mode = int(rng.integers(2, 5)), (3 if (ni == 256 and case % 2) else 2)
            hip.fl_set_option(bq._lib.FL_OPT_MGCG_FUSE, mode)
            hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, int(rng.integers(4, 12)) if mode == 3 else 0)
            hip.fl_mg_fused_launches()
            test_mgcg_matches_oracle(hip, ni, nj, nk, levels, iters, 0.5, 100 + case)
            assert hip.fl_mg_fused_launches() > 0, (case, ni, nj, nk, levels, iters, mode)
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 0)
        hip.fl_set_option(bq._lib.FL_OPT_MGCG_FUSE, -1)


@pytest.mark.parametrize("ni,nj,nk,levels", [(64, 64, 64, 4), (32, 32, 32, 3), (16, 16, 16, 2), (30, 14, 9, 2), (33, 31, 12, 3)])
def test_mgcg_bottom_of_the_v_cycle_in_one_launch(hip, ni, nj, nk, levels):
    """FL_OPT_MGCG_BOTTOM (default on): the two coarsest levels -- when they fit 4096 and 512 cells -- run as ONE launch
    (mg_vbottom_kernel: 32 sweeps from a cleared x, residual, restriction, 32 sweeps, prolongation, 4 sweeps in LDS) instead of
    ~22: level pairs (15^3, 7^3) below two finer levels, (15^3, 7^3) with the alpha * 8 of level 1, level 0 itself as the
    upper one (its right-hand side is then the caller's residual), non-cubic and odd dims; bit-identical to the oracle, twice,
    and the same with the fusion off."""
    import gpufluidsimulation_amd as bq
    test_mgcg_matches_oracle(hip, ni, nj, nk, levels, 2, 0.5)
    hip.fl_set_option(bq._lib.FL_OPT_MGCG_BOTTOM, 0)
    try:
        test_mgcg_matches_oracle(hip, ni, nj, nk, levels, 2, 0.5)
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_MGCG_BOTTOM, 1)


def test_smoothing_lds_triples_on_random_shapes(hip):
    """seeded sweep over row lengths 130 .. 256 (even), row / plane counts, chunk lengths and sweep counts for mg_lds3_kernel"""
    import gpufluidsimulation_amd as bq
    rng = np.random.default_rng(20260306)
    for case in range(24):
        ni = int(rng.integers(65, 129)) * 2
        nj, nk = int(rng.integers(4, 30)), int(rng.integers(12, 40))
        kc, iters = int(rng.integers(8, 20)), int(rng.choice([6, 10, 12, 14, 18, 32]))
        n = ni * nj * nk
        b = rng.standard_normal(n)
        x0 = rng.standard_normal(n).reshape(nk, nj, ni)
        t0 = np.zeros_like(x0)
        t0[0], t0[-1], t0[:, 0], t0[:, -1], t0[:, :, 0], t0[:, :, -1] = x0[0], x0[-1], x0[:, 0], x0[:, -1], x0[:, :, 0], x0[:, :, -1]
        x0, t0 = x0.ravel().copy(), t0.ravel().copy()
        xr, tr = x0.copy(), t0.copy()
        oracle().orc_mg_smooth(dp(xr), dp(b), dp(tr), -8.0, 1.0 / 6.0, ni, nj, nk, iters)
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, kc)
        dx, db, dt = Dev(hip, x0), Dev(hip, b), Dev(hip, t0)
        hip.gpu_smoothing_jacobi(dx.ptr, db.ptr, dt.ptr, -8.0, 1.0 / 6.0, ni, nj, nk, iters)
        name = hip.fl_mg_smooth_kernel_name().decode()
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 0)
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
        assert name == "mg_lds3_kernel", (case, ni, nj, nk, iters, name)
        assert F.same(xr, dx.numpy()), (case, ni, nj, nk, kc, iters)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,levels", [(136, 24, 24, 2), (256, 16, 20, 2), (200, 12, 32, 3)])
def test_mgcg_with_three_sweep_lds_smoother(hip, ni, nj, nk, levels):
    """the whole operator with level 0 smoothed by mg_lds3_kernel (forced on small grids through the chunk-length option):
    V_Cycle's 32 sweeps from a cleared x run as 10 triples + 1 pair -- an ODD number of launches, the first of which does not
    read its input and writes straight into x -- its 4 sweeps on the way up as two pairs; bit-identical to the oracle, twice"""
    import gpufluidsimulation_amd as bq
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 12)
    try:
        # (default: that last pair launch also writes the residual the restriction reads -- mg_lds3_kernel<8, 3, false, true>)
        test_mgcg_matches_oracle(hip, ni, nj, nk, levels, 2, 0.5)
        hip.fl_set_option(bq._lib.FL_OPT_MGCG_BOTTOM, 0)     # ... and with the residual as a launch of its own
        test_mgcg_matches_oracle(hip, ni, nj, nk, levels, 2, 0.5)
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_MGCG_BOTTOM, 1)
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 0)


@pytest.mark.parametrize("ni,nj,nk,levels,iters", [(64, 64, 64, 4, 2), (72, 40, 24, 3, 2), (129, 33, 17, 3, 1), (40, 36, 32, 3, 2),
                                                   (24, 20, 16, 2, 2), (16, 16, 16, 1, 1)])
def test_mgcg_tile_smoother_on_and_off(hip, ni, nj, nk, levels, iters):
    """FL_OPT_MGCG_TILE: the V-cycle's coarse-level smoothing through mg_smooth_tile_kernel (4 or 2 sweeps per launch on
    LDS tiles, clears folded in; the default) and through one launch per sweep both reproduce the oracle bit for bit --
    on ragged tile edges (129, 33, 17 -> 64, 16, 8 -> 31, 7, 3), one-tile levels and levels thinner than a tile."""
    import gpufluidsimulation_amd as bq
    assert hip.fl_get_option(bq._lib.FL_OPT_MGCG_TILE) == 1
    try:
        for tile in (2, 1, 0):
            hip.fl_set_option(bq._lib.FL_OPT_MGCG_TILE, tile)
            for graph in (1, 0):
                hip.fl_set_option(bq._lib.FL_OPT_MGCG_GRAPH, graph)
                test_mgcg_matches_oracle(hip, ni, nj, nk, levels, iters, 0.5)
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_MGCG_TILE, 1)
        hip.fl_set_option(bq._lib.FL_OPT_MGCG_GRAPH, 1)


def test_mgcg_rejects_bad_arguments(hip):
    import gpufluidsimulation_amd as bq
    c = HostCase(8, 8, 8, 1)
    d, lv, table = device_case(hip, c, 1)
    u = Dev(hip, np.zeros(9 * 8 * 8, np.float32))
    args = [u.ptr, u.ptr, u.ptr, d["div"].ptr, d["p"].ptr, d["dir"].ptr, d["residual"].ptr, d["temp0"].ptr, d["temp1"].ptr,
            d["result"].ptr, C.cast(table, C.c_void_p)]
    hip.gpu_multi_grid_conjugate_gradient(*args, 7, 1, 0.5)            # more levels than LEVEL_COUNT
    with pytest.raises(bq.BimocqError):
        bq.check()
    hip.gpu_multi_grid_conjugate_gradient(*args, 1, 1200, 0.5)         # history would not fit tempResult
    with pytest.raises(bq.BimocqError):
        bq.check()
    table[0].number += 1                                               # dims and count disagree
    hip.gpu_multi_grid_conjugate_gradient(*args, 1, 1, 0.5)
    with pytest.raises(bq.BimocqError):
        bq.check()


@pytest.mark.parametrize("ni,nj,nk", [(256, 12, 9), (128, 20, 40), (130, 9, 7), (127, 16, 11), (63, 63, 63), (31, 9, 70),
                                      (64, 5, 5), (65, 4, 3), (9, 8, 7), (300, 6, 5), (514, 5, 4)])
@pytest.mark.parametrize("iters", [4, 8, 3, 2])
def test_smoothing_fused_pairs(hip, ni, nj, nk, iters):
    """gpu_smoothing_jacobi (two sweeps per launch where it applies: double2 lanes for even rows, one cell
    per lane for odd rows, rows spanning 1-4 waves with the edge-lane path, too-wide rows and odd leftovers
    on the single-sweep kernel) == the oracle's sweep-by-sweep smoothing, bit for bit, boundary untouched"""
    import gpufluidsimulation_amd as bq
    rng = np.random.default_rng(ni * 7 + nj)
    n = ni * nj * nk
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n).reshape(nk, nj, ni)
    # same boundary layer in both ping-pong buffers (the operator's precondition), non-zero on purpose
    t0 = np.zeros_like(x0)
    for arr in (t0,):
        arr[0], arr[-1], arr[:, 0], arr[:, -1], arr[:, :, 0], arr[:, :, -1] = x0[0], x0[-1], x0[:, 0], x0[:, -1], x0[:, :, 0], x0[:, :, -1]
    x0, t0 = x0.ravel().copy(), t0.ravel().copy()
    for fuse in (2, 0):                                  # 2: fused wherever it applies, whatever the size
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, fuse)
        xr, tr = x0.copy(), t0.copy()
        oracle().orc_mg_smooth(dp(xr), dp(b), dp(tr), -8.0, 1.0 / 6.0, ni, nj, nk, iters)
        dx, db, dt = Dev(hip, x0), Dev(hip, b), Dev(hip, t0)
        hip.gpu_smoothing_jacobi(dx.ptr, db.ptr, dt.ptr, -8.0, 1.0 / 6.0, ni, nj, nk, iters)
        bq.check()
        assert F.same(xr, dx.numpy()), fuse
        if not fuse:
            assert F.same(tr, dt.numpy())           # sweep by sweep the older iterate matches too
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)


@pytest.mark.parametrize("ni,nj,nk,kc", [(256, 12, 30, 10), (130, 9, 24, 8), (200, 21, 40, 13), (254, 8, 16, 16), (256, 64, 48, 24),
                                        (132, 4, 12, 12), (256, 7, 13, 8)])
@pytest.mark.parametrize("iters", [6, 10, 32, 12, 4])
def test_smoothing_lds_triples(hip, ni, nj, nk, kc, iters):
    """mg_lds3_kernel (round 3): THREE fp64 smoothing sweeps per launch, a wave owns one row of 130 .. 256 doubles as two
    coalesced segments (the x-neighbours across the seam by wave rotation), the neighbouring rows of the intermediate levels
    come out of LDS; as many triples as leave an even number of launches, pairs (the same kernel with two levels) for the rest.  Against the oracle's
    sweep-by-sweep smoothing: rows that end inside segment B, row counts that leave the last block partly outside the grid,
    chunks whose warm-up planes reach below plane 0, a cleared input (the ZIN form)."""
    import gpufluidsimulation_amd as bq
    rng = np.random.default_rng(ni * 5 + nk)
    n = ni * nj * nk
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n).reshape(nk, nj, ni)
    t0 = np.zeros_like(x0)
    t0[0], t0[-1], t0[:, 0], t0[:, -1], t0[:, :, 0], t0[:, :, -1] = x0[0], x0[-1], x0[:, 0], x0[:, -1], x0[:, :, 0], x0[:, :, -1]
    x0, t0 = x0.ravel().copy(), t0.ravel().copy()
    xr, tr = x0.copy(), t0.copy()
    oracle().orc_mg_smooth(dp(xr), dp(b), dp(tr), -8.0, 1.0 / 6.0, ni, nj, nk, iters)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, kc)
    dx, db, dt = Dev(hip, x0), Dev(hip, b), Dev(hip, t0)
    hip.gpu_smoothing_jacobi(dx.ptr, db.ptr, dt.ptr, -8.0, 1.0 / 6.0, ni, nj, nk, iters)
    name = hip.fl_mg_smooth_kernel_name().decode()
    # the same call with the triples off: same bits, the pair kernel
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 5)
    dx2, dt2 = Dev(hip, x0), Dev(hip, t0)
    hip.gpu_smoothing_jacobi(dx2.ptr, db.ptr, dt2.ptr, -8.0, 1.0 / 6.0, ni, nj, nk, iters)
    name2 = hip.fl_mg_smooth_kernel_name().decode()
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
    bq.check()
    assert name == "mg_lds3_kernel", name                # (4 sweeps: two pair launches of the same kernel)
    assert name2 == "mg_lean2r_kernel", name2
    assert F.same(xr, dx.numpy())
    assert F.same(xr, dx2.numpy())
