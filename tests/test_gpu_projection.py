"""GPU parity of the pressure projection: divergence, Jacobi sweeps (generic and
LDS-tiled kernels, every tile-edge case), gradient, the full gpu_projection_jacobi with its
off-by-one (Q1) and quarter-strength (Q2) behaviour, residual norms.  Bit-exact vs the oracle."""
import ctypes as C

import numpy as np
import pytest

import fields as F
from oracle_lib import fp, lib as oracle

pytestmark = pytest.mark.gpu

ALPHA, BETA = -1.0, float(np.float32(1.0 / 6.0))


def dev(*arrays):
    from gpufluidsimulation_amd import DeviceBuffer
    return [DeviceBuffer.from_numpy(a) for a in arrays]


@pytest.fixture(scope="module")
def hip():
    import gpufluidsimulation_amd as bq
    lib = bq.hip_lib()
    assert lib.fl_init(0) == 0, lib.fl_last_error_string()
    yield lib
    lib.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, 0)
    lib.fl_set_option(bq._lib.FL_OPT_RESIDUAL_STRIDE, 0)
    bq.check()


# (ni, nj, nk): generic-only (ni%4 != 0 or small), 128-wide tile, 256-wide tile, two x tiles with a
# 4-column remainder (x halo columns), ragged y (nj not a multiple of the tile), k chunks with remainder
JACOBI_GRIDS = [(24, 20, 16), (30, 9, 7), (32, 32, 32), (64, 48, 40), (128, 37, 19), (256, 32, 24),
                (260, 20, 12), (516, 18, 11), (36, 5, 3), (32, 3, 3)]


@pytest.mark.parametrize("ni,nj,nk", JACOBI_GRIDS)
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_jacobi_sweeps(hip, ni, nj, nk, variant):
    import gpufluidsimulation_amd as bq
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, variant)
    n = ni * nj * nk
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    t0 = F.scalar(ni, nj, nk, 2.9)          # p_temp's boundary layer must survive untouched
    for sweeps in (1, 2, 5):
        a, b = p0.copy(), t0.copy()
        for _ in range(sweeps):
            oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
            a, b = b, a
        dp, dd, dt = dev(p0, div, t0)
        where = hip.gpu_jacobi_sweeps(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, sweeps, ALPHA, BETA)
        assert where == sweeps % 2
        newest, older = (dt, dp) if where else (dp, dt)
        assert F.same(a, newest.numpy()), (ni, nj, nk, sweeps)
        assert F.same(b, older.numpy())
    bq.check()


@pytest.mark.parametrize("ni,nj,nk", [(24, 20, 16), (32, 32, 32), (64, 33, 17)])
def test_divergence_gradient(hip, ni, nj, nk):
    import gpufluidsimulation_amd as bq
    h = 1.0 / ni
    u, v, w = F.velocity(ni, nj, nk, h)
    n = ni * nj * nk
    ref = np.zeros(n, np.float32)
    oracle().orc_divergence(fp(u), fp(v), fp(w), fp(ref), ni, nj, nk, 0.5)
    du, dv, dw, dd = dev(u, v, w, np.ones(n, np.float32))
    hip.gpu_divergence(du.ptr, dv.ptr, dw.ptr, dd.ptr, ni, nj, nk, 0.5)
    assert F.same(ref, dd.numpy())
    p = F.scalar(ni, nj, nk, 0.8)
    ru, rv, rw = u.copy(), v.copy(), w.copy()
    oracle().orc_gradient(fp(ru), fp(p), ni + 1, nj, nk, 1, 0, 0, 0.5)
    oracle().orc_gradient(fp(rv), fp(p), ni, nj + 1, nk, 0, 1, 0, 0.5)
    oracle().orc_gradient(fp(rw), fp(p), ni, nj, nk + 1, 0, 0, 1, 0.5)
    (dp,) = dev(p)
    hip.gpu_gradient(du.ptr, dv.ptr, dw.ptr, dp.ptr, ni, nj, nk, 0.5)
    assert F.same(ru, du.numpy()) and F.same(rv, dv.numpy()) and F.same(rw, dw.numpy())
    # KAT 4: divergence of a discrete curl is zero to rounding
    bq.check()


@pytest.mark.parametrize("ni,nj,nk", [(24, 20, 16), (32, 32, 32), (128, 24, 16)])
@pytest.mark.parametrize("iters", [0, 1, 2, 9, 10])
def test_projection_jacobi(hip, ni, nj, nk, iters):
    import gpufluidsimulation_amd as bq
    h = 1.0 / ni
    u, v, w = F.velocity(ni, nj, nk, h)
    n = ni * nj * nk
    ru, rv, rw = u.copy(), v.copy(), w.copy()
    rd, rp, rt = (np.zeros(n, np.float32) for _ in range(3))
    rdbg = np.zeros(4096, np.float32)
    oracle().orc_projection_jacobi(fp(ru), fp(rv), fp(rw), fp(rd), fp(rp), fp(rt), fp(rdbg),
                                   ni, nj, nk, iters, 0.5, ALPHA, BETA)
    m = bq.GpuMapper(ni, nj, nk, h)
    hip.fl_set_option(bq._lib.FL_OPT_RESIDUAL_STRIDE, 1)
    du, dv, dw = dev(u, v, w)
    dd, dp, dt = dev(*(np.full(n, 9.0, np.float32) for _ in range(3)))      # projectionJacobi must zero them
    (ddbg,) = dev(np.zeros(4096, np.float32))
    m.projectionJacobi(du, dv, dw, dd, dp, dt, ddbg, iters, 0.5, ALPHA, BETA)
    assert F.same(ru, du.numpy()) and F.same(rv, dv.numpy()) and F.same(rw, dw.numpy())
    assert F.same(rd, dd.numpy())
    assert F.same(rp, dp.numpy())                       # p holds iterate iters-1 (Q1)
    g = ddbg.numpy()
    np.testing.assert_allclose(g[:max(iters, 0)], rdbg[:max(iters, 0)], rtol=1e-6)
    assert F.same(rdbg[2000:2000 + max(iters, 0)], g[2000:2000 + max(iters, 0)])     # max|r| is exact
    hip.fl_set_option(bq._lib.FL_OPT_RESIDUAL_STRIDE, 0)
    m.check()


@pytest.mark.parametrize("ni,nj,nk", [(32, 32, 32), (64, 20, 12)])
@pytest.mark.parametrize("iters", [0, 1, 2, 5, 6])
@pytest.mark.parametrize("same_shell", [False, True])
def test_projection_jacobi_raw_abi_warm_start(hip, ni, nj, nk, iters, same_shell):
    """The extern "C" symbol called directly, the way a maintainer's binding would, WITHOUT gpuMapper's clears: a
    warm-started p and a p_temp with its own contents.  The reference ping-pongs, so odd iterates carry p_temp's
    boundary shell (GPU_kernel.cu:1860-1875); the fused two-sweep launch may only be taken when both shells are equal
    (checked inside gpu_projection_jacobi); iter == 0 copies p_temp over p (:1876-1879)."""
    import gpufluidsimulation_amd as bq
    h = 1.0 / ni
    u, v, w = F.velocity(ni, nj, nk, h)
    n = ni * nj * nk
    p0 = F.scalar(ni, nj, nk, 0.3)
    t0 = p0.copy() if same_shell else F.scalar(ni, nj, nk, 2.2, amp=0.7)
    if same_shell:                                       # same shell, different interior
        t3 = t0.reshape(nk, nj, ni)
        t3[1:-1, 1:-1, 1:-1] = F.scalar(ni, nj, nk, 2.2, amp=0.7).reshape(nk, nj, ni)[1:-1, 1:-1, 1:-1]
    ru, rv, rw, rp, rt = u.copy(), v.copy(), w.copy(), p0.copy(), t0.copy()
    rd = np.zeros(n, np.float32)
    oracle().orc_projection_jacobi(fp(ru), fp(rv), fp(rw), fp(rd), fp(rp), fp(rt), None, ni, nj, nk, iters, 0.5, ALPHA, BETA)
    du, dv, dw, dd, dp, dt = dev(u, v, w, np.zeros(n, np.float32), p0, t0)
    assert hip.fl_get_option(bq._lib.FL_OPT_JACOBI_FUSE) == 1
    hip.gpu_projection_jacobi(du.ptr, dv.ptr, dw.ptr, dd.ptr, dp.ptr, dt.ptr, None, ni, nj, nk, iters, 0.5, ALPHA, BETA)
    bq.check()
    assert F.same(rd, dd.numpy())
    assert F.same(rp, dp.numpy()), "p"
    assert F.same(ru, du.numpy()) and F.same(rv, dv.numpy()) and F.same(rw, dw.numpy())


def test_residual_norms(hip):
    import gpufluidsimulation_amd as bq
    ni, nj, nk = 64, 40, 24
    div, p = F.scalar(ni, nj, nk, 0.2), F.scalar(ni, nj, nk, 1.2)
    ss, mx = C.c_double(), C.c_float()
    oracle().orc_residual_norms(fp(div), fp(p), ni, nj, nk, C.byref(ss), C.byref(mx))
    dd, dp = dev(div, p)
    gs, gm = C.c_double(), C.c_float()
    hip.gpu_residual_norms(dd.ptr, dp.ptr, ni, nj, nk, C.byref(gs), C.byref(gm))
    assert gm.value == mx.value
    assert abs(gs.value - ss.value) <= 1e-12 * ss.value     # double accumulation, different order
    bq.check()


def test_projection_quarter_strength(hip):
    """SURVEY Q2 known answer: with halfrdx=0.5 on the staggered grid the converged projection
    removes only 1/4 of the divergence (ratio 0.75); with halfrdx=1.0 it removes all of it."""
    import gpufluidsimulation_amd as bq
    ni = nj = nk = 32
    h = 1.0 / ni
    n = ni * nj * nk
    u, v, w = F.velocity(ni, nj, nk, h)
    # compactly supported divergence: zero the velocity near the walls so boundary cells stay div-free
    U, V, W = u.reshape(nk, nj, ni + 1), v.reshape(nk, nj + 1, ni), w.reshape(nk + 1, nj, ni)
    for a in (U, V, W):
        a[:4] = 0; a[-4:] = 0; a[:, :4] = 0; a[:, -4:] = 0; a[:, :, :4] = 0; a[:, :, -4:] = 0
    m = bq.GpuMapper(ni, nj, nk, h)

    def ratio(hr, iters=3000):
        du, dv, dw = dev(u, v, w)
        dd, dp, dt = dev(*(np.zeros(n, np.float32) for _ in range(3)))
        before = np.zeros(n, np.float32)
        oracle().orc_divergence(fp(u), fp(v), fp(w), fp(before), ni, nj, nk, 1.0)
        m.projectionJacobi(du, dv, dw, dd, dp, dt, None, iters, hr, ALPHA, BETA)
        after = np.zeros(n, np.float32)
        oracle().orc_divergence(fp(du.numpy()), fp(dv.numpy()), fp(dw.numpy()), fp(after), ni, nj, nk, 1.0)
        inner = (slice(2, ni - 2),) * 3
        A, B = after.reshape(nk, nj, ni)[inner], before.reshape(nk, nj, ni)[inner]
        return float(np.linalg.norm(A) / np.linalg.norm(B))

    assert abs(ratio(0.5) - 0.75) < 1e-4
    assert ratio(1.0) < 1e-5
    m.check()


@pytest.mark.parametrize("ni,nj,nk", [(32, 32, 32), (64, 48, 40), (128, 37, 19), (256, 32, 24), (200, 20, 12), (36, 5, 3),
                                      (512, 9, 7), (260, 20, 12), (516, 6, 5), (1024, 5, 4), (384, 10, 40)])   # rows of 2-4 waves
@pytest.mark.parametrize("sweeps", [2, 3, 4, 9])
@pytest.mark.parametrize("rows", [1, 2])
def test_fused_two_sweep_kernel(hip, ni, nj, nk, sweeps, rows):
    """jacobi_march2_kernel (two sweeps per launch; rows = 2: jacobi_march2r_kernel, two rows per thread, where a row
    is one wave) against single oracle sweeps; both ping-pong buffers carry the same boundary layer (the kernel's
    precondition), which is otherwise arbitrary."""
    import gpufluidsimulation_amd as bq
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    a, b = p0.copy(), p0.copy()             # same boundary layer in both buffers
    for _ in range(sweeps):
        oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
        a, b = b, a
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, rows)
    dp, dd, dt = dev(p0, div, p0)
    where = hip.gpu_jacobi_sweeps(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, sweeps, ALPHA, BETA)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 0)
    newest = dt if where else dp
    assert F.same(a, newest.numpy()), (ni, nj, nk, sweeps)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk", [(64, 48, 40), (256, 32, 24), (36, 5, 3), (512, 9, 24), (260, 20, 12), (1024, 5, 9)])
@pytest.mark.parametrize("pf", [1, 2])
def test_fused_two_sweep_kernel_prefetch_distances(hip, ni, nj, nk, pf):
    """jacobi_lean2r_kernel<WIDE, PF>: loads 1 or 2 planes ahead (2: with streaming stores) (rings of 3 + PF planes, loop unrolled 3 + PF times);
    every distance gives the bits of single oracle sweeps, on rows of one wave and of 2-4 waves, with chunks shorter
    than the ring."""
    import gpufluidsimulation_amd as bq
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    a, b = p0.copy(), p0.copy()
    for _ in range(4):
        oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
        a, b = b, a
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 4)          # pairs only
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 2)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK, pf)
    dp, dd, dt = dev(p0, div, p0)
    where = hip.gpu_jacobi_sweeps(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, 4, ALPHA, BETA)
    name = hip.fl_jacobi_kernel_name().decode()
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK, 0)
    assert name == "jacobi_lean2r_kernel"
    assert F.same(a, (dt if where else dp).numpy()), (ni, nj, nk, pf)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk", [(24, 20, 16), (64, 48, 40), (256, 32, 24)])
@pytest.mark.parametrize("variant", [0, 1, 3])
def test_jacobi_sweep_range(hip, ni, nj, nk, variant):
    """a sweep split into plane ranges (interior first, boundary planes after the exchange) touches exactly
    the requested planes and composes to the full sweep"""
    import gpufluidsimulation_amd as bq
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, variant)
    p0, div, t0 = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2), F.scalar(ni, nj, nk, 2.9)
    full = t0.copy()
    oracle().orc_jacobi_sweep(fp(p0), fp(div), fp(full), ni, nj, nk, ALPHA, BETA)
    dp, dd, dt = dev(p0, div, t0)
    g = 3
    hip.gpu_jacobi_sweep_range(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, g + 1, nk - g - 1, ALPHA, BETA)
    part = dt.numpy().reshape(nk, nj, ni)
    want = t0.copy().reshape(nk, nj, ni)
    want[g + 1:nk - g - 1] = full.reshape(nk, nj, ni)[g + 1:nk - g - 1]
    assert F.same(want, part)
    hip.gpu_jacobi_sweep_range(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, 0, g + 1, ALPHA, BETA)
    hip.gpu_jacobi_sweep_range(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, nk - g - 1, nk, ALPHA, BETA)
    hip.gpu_jacobi_sweep_range(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, 5, 5, ALPHA, BETA)      # empty range
    assert F.same(full, dt.numpy())
    bq.check()


@pytest.mark.parametrize("ni,nj,nk", [(64, 48, 40), (256, 32, 30), (128, 37, 19), (512, 9, 24), (32, 6, 12)])
@pytest.mark.parametrize("rows", [0, 1, 2])
def test_fused_pair_on_plane_ranges(hip, ni, nj, nk, rows):
    """gpu_jacobi_sweep_pair_ranges: two sweeps in one launch, output restricted to two plane ranges.  The interior
    piece leaves every other plane of `out` untouched; interior + the two ends together equal two oracle sweeps
    (the split a z-slab rank uses around its ghost-plane exchange)."""
    import gpufluidsimulation_amd as bq
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    a, b = p0.copy(), p0.copy()
    for _ in range(2):
        oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
        a, b = b, a
    want = a                                             # after two sweeps the newest iterate is back in `a`
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, rows)
    G = 4
    lo, hi = G + 2, nk - G - 2
    dp, dd = dev(p0, div)
    (dout,) = dev(p0)                                    # same boundary layer; interior values are overwritten
    assert hip.gpu_jacobi_sweep_pair_ranges(dp.ptr, dd.ptr, dout.ptr, ni, nj, nk, lo, hi, 0, 0, ALPHA, BETA) == 1
    part = dout.numpy().reshape(nk, nj, ni)
    w3, p3 = want.reshape(nk, nj, ni), p0.reshape(nk, nj, ni)
    assert F.same(part[lo:hi], w3[lo:hi])
    assert F.same(part[:lo], p3[:lo]) and F.same(part[hi:], p3[hi:])
    assert hip.gpu_jacobi_sweep_pair_ranges(dp.ptr, dd.ptr, dout.ptr, ni, nj, nk, 0, lo, hi, nk, ALPHA, BETA) == 1
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 0)
    assert F.same(dout.numpy(), want)
    assert F.same(dp.numpy(), p0)                        # the input is only read
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,kc", [(64, 48, 40, 10), (256, 32, 30, 8), (128, 37, 19, 8), (32, 12, 12, 12), (36, 25, 50, 9),
                                        (256, 100, 64, 16), (200, 13, 33, 16), (256, 256, 40, 10)])
@pytest.mark.parametrize("sweeps", [3, 7])
@pytest.mark.parametrize("shape", [24, 26, 18, 19])
def test_fused_three_sweep_lds_kernel(hip, ni, nj, nk, kc, sweeps, shape):
    """jacobi_lds3_kernel (round 3): three sweeps per launch, every wave evaluates each level on its own two rows and takes
    the neighbouring rows of the level below from LDS (blocks of 6 row pairs + a halo wave at either end, one barrier per
    plane; shape = 10 R + W: R rows per wave, W output waves per block -- 24 / 26: row pairs, 18 / 19: single rows, 8 / 12 of
    them with two halo waves per side).  Against single oracle sweeps: row counts that leave the last block partly or wholly outside the grid, rows
    shorter than a wave, chunks whose warm-up planes reach below plane 0, an odd remainder swept by the other kernels."""
    import gpufluidsimulation_amd as bq
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    a, b = p0.copy(), p0.copy()
    for _ in range(sweeps):
        oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
        a, b = b, a
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 4)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, kc)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK, shape)
    if nj < (12 if shape in (26, 19) else 8):
        pytest.skip("fewer rows than one block of this shape")
    dp, dd, dt = dev(p0, div, p0)
    where = hip.gpu_jacobi_sweeps(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, 3, ALPHA, BETA)      # one triple: names the kernel
    name = hip.fl_jacobi_kernel_name().decode()
    where2 = hip.gpu_jacobi_sweeps((dt if where else dp).ptr, dd.ptr, (dp if where else dt).ptr, ni, nj, nk, sweeps - 3, ALPHA, BETA)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK, 0)
    assert name == "jacobi_lds3_kernel", name
    newest = (dt if where else dp) if not where2 else (dp if where else dt)
    assert F.same(a, newest.numpy()), (ni, nj, nk, sweeps)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,kc", [(512, 16, 30, 10), (260, 9, 24, 8), (384, 21, 40, 13), (508, 8, 16, 16), (512, 64, 48, 24),
                                        (264, 8, 12, 12), (512, 11, 13, 8)])
@pytest.mark.parametrize("sweeps", [3, 7, 8])
def test_fused_three_sweep_two_segment_kernel(hip, ni, nj, nk, kc, sweeps):
    """jacobi_lds2seg_kernel (round 3): three sweeps per launch on rows of 260 .. 512 floats -- a lane holds two float4 segments
    of its wave's row (cells 4l .. 4l+3 and 256 + 4l ..), the x-neighbours across the seam travel by wave rotation, the
    neighbour rows of all three levels (the input too) come out of LDS.  Against single oracle sweeps: rows that end inside
    segment B, row counts that leave the last block partly outside the grid, chunks whose warm-up planes reach below plane 0,
    remainders swept by the two-sweep kernels."""
    import gpufluidsimulation_amd as bq
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    a, b = p0.copy(), p0.copy()
    for _ in range(sweeps):
        oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
        a, b = b, a
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, kc)
    dp, dd, dt = dev(p0, div, p0)
    where = hip.gpu_jacobi_sweeps(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, 3, ALPHA, BETA)      # one triple: names the kernel
    name = hip.fl_jacobi_kernel_name().decode()
    where2 = hip.gpu_jacobi_sweeps((dt if where else dp).ptr, dd.ptr, (dp if where else dt).ptr, ni, nj, nk, sweeps - 3, ALPHA, BETA)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 0)
    assert name == "jacobi_lds2seg_kernel", name
    newest = (dt if where else dp) if not where2 else (dp if where else dt)
    assert F.same(a, newest.numpy()), (ni, nj, nk, sweeps)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,kc", [(64, 48, 40, 10), (256, 32, 30, 8), (128, 37, 19, 8), (32, 12, 12, 12), (36, 25, 50, 9),
                                        (256, 100, 64, 16), (200, 13, 33, 16), (256, 256, 40, 10)])
@pytest.mark.parametrize("sweeps", [4, 9])
@pytest.mark.parametrize("shape", [24, 18])
def test_fused_four_sweep_lds_kernel(hip, ni, nj, nk, kc, sweeps, shape):
    """jacobi_lds_kernel<W, R, 4>: FOUR sweeps per launch (FL_OPT_JACOBI_ROWS = 6) -- three intermediate levels exchanged
    through LDS, three halo rows per block side (two halo waves of row pairs, or three of single rows), six warm-up planes per
    chunk; a remainder of 1-3 sweeps goes to the other kernels.  Against single oracle sweeps."""
    import gpufluidsimulation_amd as bq
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    a, b = p0.copy(), p0.copy()
    for _ in range(sweeps):
        oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
        a, b = b, a
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_VARIANT, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 6)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, kc)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK, shape)
    dp, dd, dt = dev(p0, div, p0)
    where = hip.gpu_jacobi_sweeps(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, 4, ALPHA, BETA)      # one quad: names the kernel
    name = hip.fl_jacobi_kernel_name().decode()
    where2 = hip.gpu_jacobi_sweeps((dt if where else dp).ptr, dd.ptr, (dp if where else dt).ptr, ni, nj, nk, sweeps - 4, ALPHA, BETA)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 0)
    hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK, 0)
    assert name == "jacobi_lds_kernel<4 sweeps>", name
    newest = (dt if where else dp) if not where2 else (dp if where else dt)
    assert F.same(a, newest.numpy()), (ni, nj, nk, sweeps)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk", [(64, 48, 40), (256, 32, 30), (128, 37, 26), (512, 9, 24), (32, 8, 12), (264, 16, 40)])
def test_fused_triple_on_plane_ranges(hip, ni, nj, nk):
    """gpu_jacobi_sweep_triple_ranges: three sweeps in one launch, output restricted to two plane ranges (the LDS-exchanged
    kernels, rows of one wave and rows of two float4 segments per lane): the ends of a slab chunk first, then its interior --
    together the whole array -- against three oracle sweeps; planes outside the ranges stay untouched, the input is only read."""
    import gpufluidsimulation_amd as bq
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    a, b = p0.copy(), p0.copy()
    for _ in range(3):
        oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
        a, b = b, a
    want = a
    lo, hi = 5, nk - 4
    dp, dd = dev(p0, div)
    (dout,) = dev(p0)
    assert hip.gpu_jacobi_sweep_triple_ranges(dp.ptr, dd.ptr, dout.ptr, ni, nj, nk, 0, lo, hi, nk, ALPHA, BETA) == 1
    part = dout.numpy().reshape(nk, nj, ni)
    w3, p3 = want.reshape(nk, nj, ni), p0.reshape(nk, nj, ni)
    assert F.same(part[:lo], w3[:lo]) and F.same(part[hi:], w3[hi:])
    assert F.same(part[lo:hi], p3[lo:hi])
    assert hip.gpu_jacobi_sweep_triple_ranges(dp.ptr, dd.ptr, dout.ptr, ni, nj, nk, lo, hi, 0, 0, ALPHA, BETA) == 1
    assert F.same(dout.numpy(), want)
    assert F.same(dp.numpy(), p0)
    assert hip.fl_jacobi_kernel_name().decode() == ("jacobi_lds2seg_kernel" if ni > 256 else "jacobi_lds3_kernel")
    bq.check()


def test_three_sweep_kernels_on_random_shapes_and_ranges(hip):
    """seeded sweep over grid shapes, chunk lengths and plane ranges for the two three-sweep LDS kernels (rows of one float4
    segment per lane and of two): whole arrays through gpu_jacobi_sweeps, then the same three sweeps as two range launches
    (ends, interior) through gpu_jacobi_sweep_triple_ranges -- all against three oracle sweeps"""
    import gpufluidsimulation_amd as bq
    rng = np.random.default_rng(20260305)
    for case in range(36):
        wide = case % 3 == 2
        ni = int(rng.integers(65, 129)) * 4 if wide else int(rng.integers(8, 65)) * 4
        nj, nk = int(rng.integers(8, 40)), int(rng.integers(12, 44))
        kc = int(rng.integers(8, 20))
        p0, div = F.scalar(ni, nj, nk, 0.3 + 0.01 * case), F.scalar(ni, nj, nk, 1.1, amp=0.2)
        a, b = p0.copy(), p0.copy()
        for _ in range(3):
            oracle().orc_jacobi_sweep(fp(a), fp(div), fp(b), ni, nj, nk, ALPHA, BETA)
            a, b = b, a
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 2)
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, kc)
        dp, dd, dt = dev(p0, div, p0)
        where = hip.gpu_jacobi_sweeps(dp.ptr, dd.ptr, dt.ptr, ni, nj, nk, 3, ALPHA, BETA)
        name = hip.fl_jacobi_kernel_name().decode()
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_KCHUNK2, 0)
        assert name == ("jacobi_lds2seg_kernel" if wide else "jacobi_lds3_kernel"), (case, ni, nj, nk, name)
        assert F.same(a, (dt if where else dp).numpy()), (case, ni, nj, nk, kc)
        lo = int(rng.integers(0, nk // 2)); hi = int(rng.integers(max(lo, nk // 2), nk + 1))
        dp2, dout2 = dev(p0, p0)
        assert hip.gpu_jacobi_sweep_triple_ranges(dp2.ptr, dd.ptr, dout2.ptr, ni, nj, nk, 0, lo, hi, nk, ALPHA, BETA) == 1
        assert hip.gpu_jacobi_sweep_triple_ranges(dp2.ptr, dd.ptr, dout2.ptr, ni, nj, nk, lo, hi, 0, 0, ALPHA, BETA) == 1
        hip.fl_set_option(bq._lib.FL_OPT_JACOBI_FUSE, 1)
        assert F.same(a, dout2.numpy()), (case, ni, nj, nk, lo, hi)
    bq.check()


def test_fused_pair_ranges_reports_when_it_does_not_apply(hip):
    """rows that are not a multiple of 4 floats cannot take the fused kernel: 0 is returned and nothing is written"""
    import gpufluidsimulation_amd as bq
    ni, nj, nk = 30, 9, 12
    p0, div = F.scalar(ni, nj, nk, 0.3), F.scalar(ni, nj, nk, 1.1, amp=0.2)
    dp, dd, dout = dev(p0, div, p0)
    assert hip.gpu_jacobi_sweep_pair_ranges(dp.ptr, dd.ptr, dout.ptr, ni, nj, nk, 3, 9, 0, 0, ALPHA, BETA) == 0
    assert F.same(dout.numpy(), p0)
    bq.check()
