"""FL_OPT_FIELD_WINDOW (round 4): the nine-point operators as z-marching blocks that read the sampled field out of a rolling LDS
window (csrc/bq_gather_march.hip.h).  The window is a cache, never a semantics change: a tap outside it takes the direct
path under the lane's own predicate.  So every parity test of the one-plane kernels must pass unchanged with the option on --
this module re-runs them (same functions, same oracle comparisons) with several chunk lengths, in the exact arithmetic and in
the one-fma variant, and adds the cases the window itself introduces: displacements beyond the window, chunk seams, z-slab
plane offsets."""
import numpy as np
import pytest

import fields as F
import test_gpu_ops as T
from oracle_lib import OracleSolver, fp, lib as oracle

pytestmark = pytest.mark.gpu

P2_GRIDS = [(32, 32, 32, 1.0 / 32), (1024, 12, 10, 1.0 / 1024), (40, 24, 16, 1.0 / 64), (72, 20, 41, 1.0 / 128)]


@pytest.fixture(scope="module")
def gm():
    import gpufluidsimulation_amd as bq
    cache = {}

    def get(ni, nj, nk, h):
        key = (ni, nj, nk, h)
        if key not in cache:
            cache[key] = bq.GpuMapper(ni, nj, nk, h)
        return cache[key]
    yield get
    bq.check()


@pytest.fixture(params=[1, 3, 11])
def window(request):
    """the option on for the test's duration; value = planes marched per block (1 = auto): 3 and 11 put chunk seams everywhere"""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    hip.fl_set_option(bq._lib.FL_OPT_FIELD_WINDOW, request.param)
    yield request.param
    hip.fl_set_option(bq._lib.FL_OPT_FIELD_WINDOW, -1)      # (the default: the library's choice)


@pytest.mark.parametrize("ni,nj,nk,h", P2_GRIDS)
def test_operator_parity_tests_pass_with_the_window_on(gm, window, ni, nj, nk, h):
    T.test_advect_velocity_and_field(gm, ni, nj, nk, h, False)
    T.test_compensate_velocity_and_field(gm, ni, nj, nk, h)
    T.test_accumulate(gm, ni, nj, nk, h, -0.5)
    T.test_batched_scalar_ops(gm, ni, nj, nk, h, False)
    T.test_accumulate_velocity_batched_and_identity(gm, ni, nj, nk, h)


@pytest.mark.parametrize("ni,nj,nk,h", P2_GRIDS)
def test_wild_maps_with_the_window_on(gm, window, ni, nj, nk, h):
    """zeroed borders, NaN / Inf, positions far outside the window and outside the grid: per-lane fall-back to the direct path"""
    T.test_gather_ops_on_wild_maps(gm, ni, nj, nk, h)


@pytest.mark.parametrize("ni,nj,nk,h", P2_GRIDS[:3])
def test_side_options_with_the_window_on(gm, window, ni, nj, nk, h):
    T.test_fused_housekeeping_bits(gm, ni, nj, nk, h)
    T.test_plane_windows_partition_the_operators(gm, ni, nj, nk, h)
    T.test_map_quarter_fp32_option_changes_no_bit(gm, ni, nj, nk, h, True)


@pytest.mark.parametrize("amp", [0.4, 1.7, 2.6, 4.5])
def test_displacements_inside_and_beyond_the_window(gm, window, amp):
    """smooth maps that displace by up to `amp` cells: below 2 every tap is served by the window, above it a growing share
    of the lanes leaves it (and, at 4.5, most waves run both paths) -- same bits as the oracle throughout"""
    import gpufluidsimulation_amd as bq
    ni, nj, nk, h = 96, 28, 40, float(np.float32(1.0 / 128))
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    vel = F.velocity(ni, nj, nk, h)
    fwd, back = F.warped_maps(ni, nj, nk, h, amp, 0.3), F.warped_maps(ni, nj, nk, h, -amp, 1.1)
    m = gm(ni, nj, nk, h)
    dfwd, dback, dvel = T.dev(*fwd), T.dev(*back), T.dev(*vel)
    ref = [np.zeros(c, np.float32) for c in (nu, nv, nw)]
    oracle().orc_advect_velocity(*map(fp, ref), *map(fp, vel), *map(fp, back), h, ni, nj, nk, 0)
    out = T.dev(*[np.zeros(c, np.float32) for c in (nu, nv, nw)])
    m.advectVelocity(*out, *dvel, *dback, False)
    for r, g in zip(ref, out):
        assert F.same(r, g.numpy())
    cur = [F.scalar(ni + 1, nj, nk, 1.1), F.scalar(ni, nj + 1, nk, 1.2), F.scalar(ni, nj, nk + 1, 1.3)]
    ru, ri, rs = [a.copy() for a in cur], [a.copy() for a in vel], [np.zeros(c, np.float32) for c in (nu, nv, nw)]
    oracle().orc_compensate_velocity(*map(fp, ru), *map(fp, ri), *map(fp, rs), *map(fp, fwd), *map(fp, back), h, ni, nj, nk, 0)
    du, di = T.dev(*cur), T.dev(*vel)
    m.compensateVelocity(*du, *di, *dfwd, *dback, False)
    for r, g in zip(ru + ri, du + di):
        assert F.same(r, g.numpy())
    bq.check()


@pytest.mark.parametrize("fast", [0, 1])
def test_trajectory_with_the_window_on(window, fast):
    """the whole step, 32^3 rising smoke, 24 steps (two DMC sub-steps from step ~10 on): exact arithmetic = the oracle;
    one-fma arithmetic = the oracle in the same mode (orc_set_fast_lerp), both bit for bit"""
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    hip = bq.hip_lib()
    n = 32
    em = [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)]
    o = OracleSolver(n, n, n, 1.0, 0.0, 1.0); o.set_smoke(0.0, 1.0, em); o.set_projection(40, 0.5)
    s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0); s.setSmoke(0.0, 1.0, em); s.setProjection(40, 0.5)
    s.setOption(3, 1)                                   # the reference's full per-step sequence
    hip.fl_set_option(bq._lib.FL_OPT_FAST_LERP, fast)
    try:
        for f in range(24):
            oracle().orc_set_fast_lerp(fast)
            o.advance(f, 2.0 / n)
            oracle().orc_set_fast_lerp(0)
            s.advance(f, 2.0 / n)
            if f % 6 == 5:
                for name in ("rho", "T", "u", "v", "w"):
                    assert F.same(o.field(name), s.field(name)), (f, name, F.maxdiff(o.field(name), s.field(name)))
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 0)
        oracle().orc_set_fast_lerp(0)
    s.close(); o.close()
