"""GPU parity, operator by operator: HIP path (through the C-ABI) vs the CPU oracle on the same
deterministic inputs.  Bar: bit-exact (value equality; the sign of zero is not compared), except
the summed residual norm (1e-6 relative); the emitter's acosf / cosf / hypotf are restated portably on both sides.

Grids: a non-cubic 24x20x16 with h = 1/24 (IEEE-division path, catches index-order errors) and
32^3 with h = 1/32 (power-of-two spacing fast path).
"""
import os

import numpy as np
import pytest

import fields as F
from oracle_lib import fp, lib as oracle

pytestmark = pytest.mark.gpu

# ... and (round 3) BASELINE config 5's rows: 1024 / 1025 floats wide (rows of 16-17 waves: 16 x-blocks per row in the gather
# kernels, the marching limiter's 1024-float limit, 4-wave rows in the fused Jacobi), a few rows and planes deep
GRIDS = [(24, 20, 16, 1.0 / 24), (32, 32, 32, 1.0 / 32), (1024, 12, 10, 1.0 / 1024)]
# BQ_TEST_EXTRA_GRID="ni,nj,nk": one more grid for a diagnostic run (h = 1 / ni), e.g. 1024,1024,18 -- fields beyond 2^24
# elements with config 5's planes; minutes of oracle time, so not part of the default suite
if os.environ.get("BQ_TEST_EXTRA_GRID"):
    _g = tuple(int(x) for x in os.environ["BQ_TEST_EXTRA_GRID"].split(","))
    GRIDS.append(_g + (1.0 / _g[0],))


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


def dev(*arrays):
    from gpufluidsimulation_amd import DeviceBuffer
    return [DeviceBuffer.from_numpy(a) for a in arrays]


def setup(ni, nj, nk, h):
    h = float(np.float32(h))
    u, v, w = F.velocity(ni, nj, nk, h)
    fwd = F.warped_maps(ni, nj, nk, h, 0.8, 0.3)
    back = F.warped_maps(ni, nj, nk, h, -0.7, 1.1)
    backp = F.warped_maps(ni, nj, nk, h, 0.5, 2.0)
    return h, (u, v, w), fwd, back, backp


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
@pytest.mark.parametrize("dtscale", [1.0, -1.0, 2.7])
def test_solve_forward(gm, ni, nj, nk, h, dtscale):
    h, vel, fwd, _, _ = setup(ni, nj, nk, h)
    cfldt, dt = 0.9 * h / 0.35, dtscale * 2 * h
    ref = [a.copy() for a in fwd]
    oracle().orc_solve_forward(*map(fp, vel), *map(fp, ref), h, ni, nj, nk, cfldt, dt)
    m = gm(ni, nj, nk, h)
    d = dev(*vel, *fwd)
    m.solveForward(*d, cfldt, dt)
    for r, g in zip(ref, d[3:]):
        assert F.same(r, g.numpy())
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_solve_backward_dmc(gm, ni, nj, nk, h):
    h, vel, _, back, _ = setup(ni, nj, nk, h)
    sub = 0.8 * h / 0.35
    ref_out = [a.copy() for a in back]
    oracle().orc_solve_backwardDMC(*map(fp, vel), *map(fp, back), *map(fp, ref_out), h, ni, nj, nk, sub)
    m = gm(ni, nj, nk, h)
    d = dev(*vel, *back)
    for o, b in zip((m.x_out, m.y_out, m.z_out), back):
        o.upload(b)                       # nodes outside 2..n-3 keep what the out buffer held
    m.solveBackwardDMC(*d, sub)
    for r, g in zip(ref_out, d[3:]):
        assert F.same(r, g.numpy())
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
@pytest.mark.parametrize("is_point", [False, True])
def test_advect_velocity_and_field(gm, ni, nj, nk, h, is_point):
    h, vel, _, back, _ = setup(ni, nj, nk, h)
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    ref = [np.zeros(c, np.float32) for c in (nu, nv, nw)]
    oracle().orc_advect_velocity(*map(fp, ref), *map(fp, vel), *map(fp, back), h, ni, nj, nk, int(is_point))
    m = gm(ni, nj, nk, h)
    out = dev(*[np.full(c, 7.0, np.float32) for c in (nu, nv, nw)])      # advectVelocity must zero them
    dv, db = dev(*vel), dev(*back)
    m.advectVelocity(*out, *dv, *db, is_point)
    for r, g in zip(ref, out):
        assert F.same(r, g.numpy())
    rho = F.scalar(ni, nj, nk, 0.4)
    rref = np.zeros(n, np.float32)
    oracle().orc_advect_field(fp(rref), fp(rho), *map(fp, back), h, ni, nj, nk, int(is_point))
    (drho,), (dout,) = dev(rho), dev(np.ones(n, np.float32))
    m.advectField(dout, drho, *db, is_point)
    assert F.same(rref, dout.numpy())
    m.check()


@pytest.mark.parametrize("ranks,h", [(2, 0.05), (3, 1.0 / 32)])
def test_advect_double_with_whole_grid_previous_fields_on_a_slab_rank(ranks, h):
    """gpu_advect_vel_double_global / gpu_advect_field_double_global under fl_set_slab: the *_prev fields are the whole grid's,
    everything else the rank's local planes.  The case makes the second look-up meet the zeroed border cells of the previous
    map (tests/blend_slab_case.py), where the local form reads outside the slab; the global form equals the oracle's on every
    local plane and the single-domain operator on the owned planes -- on a spacing that is and one that is not a power of two."""
    import blend_slab_case as B
    import gpufluidsimulation_amd as bq
    lib = bq.hip_lib()
    ni, nj, nk, G, blend = 20, 18, 24, 5, 0.6
    h, back, backp, prev, cur = B.global_case(ni, nj, nk, h)
    o = oracle()
    ref = [a.copy() for a in cur]
    o.orc_advect_vel_double(*map(fp, ref[:3]), *map(fp, prev[:3]), *map(fp, back), *map(fp, backp), h, ni, nj, nk, 0, blend)
    o.orc_advect_field_double(fp(ref[3]), fp(prev[3]), *map(fp, back), *map(fp, backp), h, ni, nj, nk, 0, blend)
    pl = B.PLANES(ni, nj)
    for r in range(ranks):
        own0, own1 = r * nk // ranks, (r + 1) * nk // ranks
        nkl = own1 - own0 + 2 * G
        view = lambda a, c: B.local_view(a, pl[c], B.EXTRA[c], nk, own0, own1, G)
        lb, lbp = [view(a, 3) for a in back], [view(a, 3) for a in backp]
        want = [view(cur[c], c) for c in range(4)]
        o.orc_set_slab(own0 - G, nk, own0, own1, nkl)
        try:
            o.orc_advect_vel_double_global(*map(fp, want[:3]), *map(fp, prev[:3]), *map(fp, lb), *map(fp, lbp), h, ni, nj, nkl, 0, blend)
            o.orc_advect_field_double_global(fp(want[3]), fp(prev[3]), *map(fp, lb), *map(fp, lbp), h, ni, nj, nkl, 0, blend)
        finally:
            o.orc_set_slab(0, 0, 0, 0, 0)
        d = dev(*[view(cur[c], c) for c in range(4)])
        dprev, dlb, dlbp = dev(*prev), dev(*lb), dev(*lbp)
        lib.fl_set_slab(own0 - G, nk, own0, own1, nkl)
        try:
            lib.gpu_advect_vel_double_global(*[x.ptr for x in d[:3]], *[x.ptr for x in dprev[:3]], *[x.ptr for x in dlb], *[x.ptr for x in dlbp],
                                             h, ni, nj, nkl, False, blend)
            lib.gpu_advect_field_double_global(d[3].ptr, dprev[3].ptr, *[x.ptr for x in dlb], *[x.ptr for x in dlbp], h, ni, nj, nkl, False, blend)
        finally:
            lib.fl_set_slab(0, 0, 0, 0, 0)
        bq.check()
        for c in range(4):
            got = d[c].numpy()
            assert F.same(want[c], got), (r, c)
            assert np.array_equal(B.owned(got, pl[c], B.EXTRA[c], own0, own1, G, True, r == ranks - 1),
                                  B.owned(ref[c], pl[c], B.EXTRA[c], own0, own1, G, False, r == ranks - 1)), (r, c)


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
@pytest.mark.parametrize("blend", [1.0, 0.6])
def test_advect_double(gm, ni, nj, nk, h, blend):
    import gpufluidsimulation_amd as bq
    h, vel, _, back, backp = setup(ni, nj, nk, h)
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    cur = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)]
    ref = [a.copy() for a in cur]
    oracle().orc_advect_vel_double(*map(fp, ref), *map(fp, vel), *map(fp, back), *map(fp, backp),
                                   h, ni, nj, nk, 0, blend)
    m = gm(ni, nj, nk, h)
    for skip in (1, 2, 0):                 # no launch, the field+0 kernel and the full kernel must agree
        bq.hip_lib().fl_set_option(bq._lib.FL_OPT_SKIP_UNIT_BLEND, skip)
        d = dev(*cur)
        m.advectVelocityDouble(*d, *dev(*vel), *dev(*back), *dev(*backp), False, blend)
        for r, g in zip(ref, d):
            assert F.same(r, g.numpy())
    bq.hip_lib().fl_set_option(bq._lib.FL_OPT_SKIP_UNIT_BLEND, 1)
    rho, prev = F.scalar(ni, nj, nk, 0.9), F.scalar(ni, nj, nk, 1.9)
    rref = rho.copy()
    oracle().orc_advect_field_double(fp(rref), fp(prev), *map(fp, back), *map(fp, backp), h, ni, nj, nk, 0, blend)
    (dr, dp) = dev(rho, prev)
    m.advectFieldDouble(dr, dp, *dev(*back), *dev(*backp), False, blend)
    assert F.same(rref, dr.numpy())
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_compensate_velocity_and_field(gm, ni, nj, nk, h):
    h, vel, fwd, back, _ = setup(ni, nj, nk, h)
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    cur = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)]
    ru, ri = [a.copy() for a in cur], [a.copy() for a in vel]
    rs = [np.zeros(c, np.float32) for c in (nu, nv, nw)]
    oracle().orc_compensate_velocity(*map(fp, ru), *map(fp, ri), *map(fp, rs), *map(fp, fwd), *map(fp, back),
                                     h, ni, nj, nk, 0)
    m = gm(ni, nj, nk, h)
    du, di = dev(*cur), dev(*vel)
    m.compensateVelocity(*du, *di, *dev(*fwd), *dev(*back), False)
    for r, g in zip(ru + ri, du + di):          # compensated field AND the clobbered init (Q3)
        assert F.same(r, g.numpy())
    for r, g in zip(rs, (m.u_src, m.v_src, m.w_src)):
        assert F.same(r, g.numpy())
    rho, init = F.scalar(ni, nj, nk, 0.9), F.scalar(ni, nj, nk, 1.9)
    rr, rin, rsrc = rho.copy(), init.copy(), np.zeros(n, np.float32)
    oracle().orc_compensate_field(fp(rr), fp(rin), fp(rsrc), *map(fp, fwd), *map(fp, back), h, ni, nj, nk, 0)
    dr, dinit = dev(rho, init)
    m.compensateField(dr, dinit, *dev(*fwd), *dev(*back), False)
    assert F.same(rr, dr.numpy()) and F.same(rin, dinit.numpy())
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
@pytest.mark.parametrize("coeff", [1.0, 2.0, -0.5])
def test_accumulate(gm, ni, nj, nk, h, coeff):
    h, vel, fwd, _, _ = setup(ni, nj, nk, h)
    init = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)]
    ref = [a.copy() for a in init]
    oracle().orc_accumulate_velocity(*map(fp, vel), *map(fp, ref), *map(fp, fwd), h, ni, nj, nk, 0, coeff)
    m = gm(ni, nj, nk, h)
    d = dev(*init)
    m.accumulateVelocity(*dev(*vel), *d, *dev(*fwd), False, coeff)
    for r, g in zip(ref, d):
        assert F.same(r, g.numpy())
    ch, ini = F.scalar(ni, nj, nk, 0.5), F.scalar(ni, nj, nk, 1.5)
    rr = ini.copy()
    oracle().orc_accumulate_field(fp(ch), fp(rr), *map(fp, fwd), h, ni, nj, nk, 0, coeff)
    dch, dini = dev(ch, ini)
    m.accumulateField(dch, dini, *dev(*fwd), False, coeff)
    assert F.same(rr, dini.numpy())
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_estimate_distortion_and_semilag(gm, ni, nj, nk, h):
    h, vel, fwd, back, _ = setup(ni, nj, nk, h)
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    ref = np.zeros(n, np.float32)
    oracle().orc_estimate_distortion(fp(ref), *map(fp, back), *map(fp, fwd), h, ni, nj, nk)
    m = gm(ni, nj, nk, h)
    (dd,) = dev(np.ones(n, np.float32))
    m.estimateDistortionCUDA(dd, *dev(*back), *dev(*fwd))
    assert F.same(ref, dd.numpy())
    cfldt, dt = 0.9 * h / 0.35, -1.5 * h
    src = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)]
    refs = [np.zeros(c, np.float32) for c in (nu, nv, nw)]
    for r, s, d3 in zip(refs, src, ((1, 0, 0), (0, 1, 0), (0, 0, 1))):
        oracle().orc_semilag(fp(r), fp(s), *map(fp, vel), *d3, h, ni, nj, nk, cfldt, dt)
    outs = dev(*[np.ones(c, np.float32) for c in (nu, nv, nw)])
    m.semilagAdvectVelocity(*outs, *dev(*src), *dev(*vel), cfldt, dt)
    for r, g in zip(refs, outs):
        assert F.same(r, g.numpy())
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_streaming_ops(gm, ni, nj, nk, h):
    h, vel, _, _, _ = setup(ni, nj, nk, h)
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    m = gm(ni, nj, nk, h)
    a, b = F.scalar(ni + 1, nj, nk, 0.3), F.scalar(ni + 1, nj, nk, 1.7)
    number = nu - 5                              # not a multiple of 256: the tail must stay untouched
    ra = a.copy(); oracle().orc_add(fp(ra), fp(b), -1.0, number)
    da, db = dev(a, b); m.add(da, db, -1.0, number)
    assert F.same(ra, da.numpy())
    ro = np.full(nu, 3.0, np.float32); oracle().orc_add_field(fp(ro), fp(a), fp(b), -1.0, number)
    (do,) = dev(np.full(nu, 3.0, np.float32)); m.addFields(do, da, db, -1.0, number)
    a2 = da.numpy()
    ro2 = np.full(nu, 3.0, np.float32); oracle().orc_add_field(fp(ro2), fp(a2), fp(b), -1.0, number)
    assert F.same(ro2, do.numpy())
    rm = np.full(nu, 3.0, np.float32); oracle().orc_mad(fp(rm), fp(a2), fp(b), 2.0, -1.0, number)
    (dm,) = dev(np.full(nu, 3.0, np.float32)); m.mad(dm, da, db, 2.0, -1.0, number)
    assert F.same(rm, dm.numpy())
    # buoyancy (Q9-fixed semantics)
    rho, T = F.scalar(ni, nj, nk, 0.2), F.scalar(ni, nj, nk, 2.2)
    rv = vel[1].copy(); oracle().orc_add_buoyancy(fp(rv), fp(rho), fp(T), ni, nj, nk, 0.3, 1.1, 2 * h)
    (dvv, dr, dT) = dev(vel[1], rho, T); m.add_buoyancy(dvv, dr, dT, 0.3, 1.1, 2 * h)
    assert F.same(rv, dvv.numpy())
    # device CFL
    import gpufluidsimulation_amd as bq
    d = dev(*vel)
    got = bq.hip_lib().gpu_max_abs3(d[0].ptr, d[1].ptr, d[2].ptr, ni, nj, nk)
    assert got == oracle().orc_max_abs3(*map(fp, vel), ni, nj, nk)
    z = dev(*[np.zeros_like(x) for x in vel])
    assert bq.hip_lib().gpu_max_abs3(z[0].ptr, z[1].ptr, z[2].ptr, ni, nj, nk) == np.float32(1e-4)
    # identity maps
    mx = dev(*[np.zeros(n, np.float32)] * 3)
    bq.hip_lib().gpu_init_maps(mx[0].ptr, mx[1].ptr, mx[2].ptr, h, ni, nj, nk)
    for r, g in zip(F.identity_maps(ni, nj, nk, h), mx):
        assert F.same(r, g.numpy())
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_emit_smoke(gm, ni, nj, nk, h):
    h = float(np.float32(h))
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    L = ni * h
    args = (0.45 * L, 0.41 * nj * h, 0.5 * nk * h, 0.22 * L, 1.0, 50.0)
    m = gm(ni, nj, nk, h)
    for emiter in (0.0, 1.0, -1.0):
        ref = [np.full(c, 0.5, np.float32) for c in (nu, nv, nw, n, n)]
        oracle().orc_emit_smoke(*map(fp, ref), h, ni, nj, nk, *args, emiter)
        d = dev(*[np.full(c, 0.5, np.float32) for c in (nu, nv, nw, n, n)])
        m.emitSmoke(*d, *args, emiter)
        got = [g.numpy() for g in d]
        assert (ref[3] == 1.0).sum() > 20                       # the sphere really hits voxels
        for r, g in zip(ref, got):
            assert F.same(r, g)             # u too: acosf / cosf / hypotf are restated portably on both sides (round 3)
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
@pytest.mark.parametrize("iters", [1, 6, 7])
def test_diffuse(gm, ni, nj, nk, h, iters):
    h = float(np.float32(h))
    m = gm(ni, nj, nk, h)
    for nb in ((ni + 1, nj, nk), (ni, nj + 1, nk), (ni, nj, nk + 1)):
        f = F.scalar(*nb, 0.6)
        t0 = F.scalar(*nb, 1.6)
        t1 = F.scalar(*nb, 2.6)                                 # stale border of tmp1 leaks into the result (Q7)
        rf, r0, r1 = f.copy(), t0.copy(), t1.copy()
        oracle().orc_diffuse_field(fp(rf), fp(r0), fp(r1), *nb, iters, 0.37)
        df, d0, d1 = dev(f, t0, t1)
        m.diffuseField(df, d0, d1, *nb, iters, 0.37)
        assert F.same(rf, df.numpy()) and F.same(r0, d0.numpy()) and F.same(r1, d1.numpy())
    m.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_clamp_extrema_box(gm, ni, nj, nk, h):
    import gpufluidsimulation_amd as bq
    before, after = F.scalar(ni + 1, nj, nk, 0.2), F.scalar(ni + 1, nj, nk, 0.45, amp=1.4)
    ref = after.copy()
    oracle().orc_clamp_extrema_box(fp(before), fp(ref), ni + 1, nj, nk)
    db, da = dev(before, after)
    bq.hip_lib().gpu_clamp_extrema_box(db.ptr, da.ptr, ni + 1, nj, nk)
    assert F.same(ref, da.numpy())
    assert not F.same(ref, after)
    # idempotent and bounded by the neighbourhood (KAT 7)
    bq.hip_lib().gpu_clamp_extrema_box(db.ptr, da.ptr, ni + 1, nj, nk)
    assert F.same(ref, da.numpy())
    bq.check()


def test_unsupported_and_bad_arguments():
    import gpufluidsimulation_amd as bq
    lib = bq.hip_lib()
    lib.fl_clear_error()
    lib.gpu_conjugate_gradient(None, None, None, None, None, None, None, None, 8, 8, 8, 1, 0.5)   # compiled out in the reference
    assert lib.fl_last_error() == bq._lib.FL_ERR_UNSUPPORTED
    with pytest.raises(bq.BimocqError):
        bq.check()
    lib.gpu_add(None, None, 1.0, 16)
    assert lib.fl_last_error() == bq._lib.FL_ERR_BAD_ARGUMENT
    lib.fl_clear_error()
    lib.gpu_solve_forward(None, None, None, None, None, None, 0.1, 8, 8, 8, 0.1, 0.1)
    assert lib.fl_last_error() == bq._lib.FL_ERR_BAD_ARGUMENT
    lib.fl_clear_error()


def test_rccl_binding_selftest():
    """the dlopen'ed RCCL binding (ids, init, all-reduce dtypes/ops, grouped send/recv on the halo
    stream, destroy) on a one-rank communicator -- the only part of the multi-GPU transport a one-GPU
    box can execute"""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    assert hip.fl_init(0) == 0, hip.fl_last_error_string()
    rc = hip.fl_comm_selftest()
    assert rc == 0, hip.fl_last_error_string()
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS + [(40, 24, 16, 1.0 / 64)])
@pytest.mark.parametrize("is_point", [False, True])
def test_batched_scalar_ops(gm, ni, nj, nk, h, is_point):
    """gpu_advect_field2 / gpu_compensate_error_field2 / gpu_accumulate_field2 == the two single oracle
    calls each of them stands for"""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    h, _, fwd, back, _ = setup(ni, nj, nk, h)
    n = ni * nj * nk
    a_init, b_init = F.scalar(ni, nj, nk, 0.4), F.scalar(ni, nj, nk, 1.9, amp=2.0)
    ra, rb = np.zeros(n, np.float32), np.zeros(n, np.float32)
    oracle().orc_advect_field(fp(ra), fp(a_init), *map(fp, back), h, ni, nj, nk, int(is_point))
    oracle().orc_advect_field(fp(rb), fp(b_init), *map(fp, back), h, ni, nj, nk, int(is_point))
    gm(ni, nj, nk, h)
    da, db, dai, dbi = dev(np.zeros(n, np.float32), np.zeros(n, np.float32), a_init, b_init)
    dback, dfwd = dev(*back), dev(*fwd)
    hip.gpu_advect_field2(da.ptr, dai.ptr, db.ptr, dbi.ptr, *[x.ptr for x in dback], h, ni, nj, nk, is_point)
    assert F.same(ra, da.numpy()) and F.same(rb, db.numpy())
    # error stage
    ea, eb = np.zeros(n, np.float32), np.zeros(n, np.float32)
    oracle().orc_compensate_error_field(fp(ra), fp(a_init), fp(ea), *map(fp, fwd), h, ni, nj, nk, int(is_point))
    oracle().orc_compensate_error_field(fp(rb), fp(b_init), fp(eb), *map(fp, fwd), h, ni, nj, nk, int(is_point))
    dea, deb = dev(np.zeros(n, np.float32), np.zeros(n, np.float32))
    hip.gpu_compensate_error_field2(da.ptr, dai.ptr, dea.ptr, db.ptr, dbi.ptr, deb.ptr, *[x.ptr for x in dfwd], h, ni, nj, nk, is_point)
    assert F.same(ea, dea.numpy()) and F.same(eb, deb.numpy())
    # accumulate, different targets and coefficients
    oracle().orc_accumulate_field(fp(ea), fp(ra), *map(fp, back), h, ni, nj, nk, int(is_point), -0.5)
    oracle().orc_accumulate_field(fp(eb), fp(rb), *map(fp, back), h, ni, nj, nk, int(is_point), 2.0)
    hip.gpu_accumulate_field2(dea.ptr, da.ptr, -0.5, deb.ptr, db.ptr, 2.0, *[x.ptr for x in dback], h, ni, nj, nk, is_point)
    assert F.same(ra, da.numpy()) and F.same(rb, db.numpy())
    # same target twice: applied in order
    oracle().orc_accumulate_field(fp(ea), fp(ra), *map(fp, fwd), h, ni, nj, nk, int(is_point), 1.0)
    oracle().orc_accumulate_field(fp(eb), fp(ra), *map(fp, fwd), h, ni, nj, nk, int(is_point), 0.75)
    hip.gpu_accumulate_field2(dea.ptr, da.ptr, 1.0, deb.ptr, da.ptr, 0.75, *[x.ptr for x in dfwd], h, ni, nj, nk, is_point)
    assert F.same(ra, da.numpy())
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS + [(40, 24, 16, 1.0 / 64)])
def test_accumulate_velocity_batched_and_identity(gm, ni, nj, nk, h):
    """gpu_accumulate_velocity2 == two gpu_accumulate_velocity; gpu_accumulate_velocity_identity == the
    plain operator on the map gpu_init_maps writes (with and without the structured path)"""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    h, vel, fwd, _, _ = setup(ni, nj, nk, h)
    c2 = [F.scalar(ni + 1, nj, nk, 2.1), F.scalar(ni, nj + 1, nk, 2.2), F.scalar(ni, nj, nk + 1, 2.3)]
    init = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)]
    ref = [a.copy() for a in init]
    oracle().orc_accumulate_velocity(*map(fp, vel), *map(fp, ref), *map(fp, fwd), h, ni, nj, nk, 0, 1.0)
    oracle().orc_accumulate_velocity(*map(fp, c2), *map(fp, ref), *map(fp, fwd), h, ni, nj, nk, 0, 2.0)
    gm(ni, nj, nk, h)
    d, dv1, dv2, dfwd = dev(*init), dev(*vel), dev(*c2), dev(*fwd)
    hip.gpu_accumulate_velocity2(*[x.ptr for x in dv1], 1.0, *[x.ptr for x in dv2], 2.0, *[x.ptr for x in d],
                                 *[x.ptr for x in dfwd], h, ni, nj, nk, False)
    for r, g in zip(ref, d):
        assert F.same(r, g.numpy())
    # identity map
    n = ni * nj * nk
    ident = dev(np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32))
    hip.gpu_init_maps(*[x.ptr for x in ident], h, ni, nj, nk)
    imaps = [x.numpy() for x in ident]
    ref = [a.copy() for a in init]
    oracle().orc_accumulate_velocity(*map(fp, vel), *map(fp, ref), *map(fp, imaps), h, ni, nj, nk, 0, 1.5)
    for structured in (1, 0):
        hip.fl_set_option(bq._lib.FL_OPT_STRUCTURED_MAPS, structured)
        d = dev(*init)
        hip.gpu_accumulate_velocity_identity(*[x.ptr for x in dv1], *[x.ptr for x in d], *[x.ptr for x in ident],
                                             h, ni, nj, nk, False, 1.5)
        for r, g in zip(ref, d):
            assert F.same(r, g.numpy()), structured
    hip.fl_set_option(bq._lib.FL_OPT_STRUCTURED_MAPS, 1)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_clamp_extrema_corrected(gm, ni, nj, nk, h):
    """gpu_clamp_extrema (MacCormack limiter, corrected semantics -- include/bimocq_gpu.h) for a scalar and for
    each staggered component; the candidate field overshoots on purpose so that both branches are taken"""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    h, vel, _, _, _ = setup(ni, nj, nk, h)
    gm(ni, nj, nk, h)
    dvel = dev(*vel)
    dt = 1.7 * h / 0.35
    for (dx, dy, dz) in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)):
        bi, bj, bk = ni + dx, nj + dy, nk + dz
        field = F.scalar(bi, bj, bk, 0.7)
        cand = (field + F.scalar(bi, bj, bk, 2.3, amp=0.6)).astype(np.float32)
        ref = cand.copy()
        oracle().orc_clamp_extrema(fp(field), fp(ref), *map(fp, vel), bi, bj, bk, dx, dy, dz, 0.5 * dx, 0.5 * dy, 0.5 * dz, h, dt)
        changed = int((ref != cand).sum())
        assert 0 < changed < ref.size, changed
        df, dc = dev(field, cand)
        hip.gpu_clamp_extrema(df.ptr, dc.ptr, *[x.ptr for x in dvel], bi, bj, bk, dx, dy, dz, 0.5 * dx, 0.5 * dy, 0.5 * dz, h, dt)
        assert F.same(ref, dc.numpy()), (dx, dy, dz)
        assert F.same(field, df.numpy())
    bq.check()


def wild_maps(ni, nj, nk, h, phase):
    """Maps that leave the comfortable range on purpose: the zero border the DMC update leaves behind,
    positions inside the first cell (q < 1: the lerps must take the contract's two-rounding form), exact zeros,
    positions outside the domain on both sides, tiny values next to large ones (the 3/4*a midpoint case of the
    constant-weight lerps), infinities and NaNs.  Deterministic (no RNG)."""
    maps = F.warped_maps(ni, nj, nk, h, 0.9, phase)
    n = ni * nj * nk
    idx = np.arange(n)
    k, j, i = idx // (ni * nj), (idx // ni) % nj, idx % ni
    border = (i <= 1) | (i >= ni - 2) | (j <= 1) | (j >= nj - 2) | (k <= 1) | (k >= nk - 2)
    out = []
    for c, m in enumerate(maps):
        m = m.copy()
        m[border] = 0.0                                                  # Q13
        sel = (idx * 7 + c * 3) % 23
        m[sel == 0] *= np.float32(0.01)                                  # inside the first cell
        m[sel == 1] = np.float32(h) * np.float32(0.999)
        m[sel == 2] = -m[sel == 2]                                       # below the domain
        m[sel == 3] *= np.float32(3.0)                                   # possibly above it
        m[sel == 4] = np.float32(1e-30)                                  # tiny next to O(1)
        m[sel == 5] = np.float32(2.0 ** -60)
        m[(idx % 997) == 5 + c] = np.nan
        m[(idx % 1013) == 7 + c] = np.inf
        m[(idx % 1019) == 11 + c] = -np.inf
        out.append(np.ascontiguousarray(m.astype(np.float32)))
    return out


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS + [(40, 24, 16, 1.0 / 64)])
def test_gather_ops_on_wild_maps(gm, ni, nj, nk, h):
    """advect / compensate-error / accumulate, one and two fields, on maps that hit every special case of the
    structured look-up and of the sample weights (see wild_maps): still bit-identical to the oracle."""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    h = float(np.float32(h))
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    vel = F.velocity(ni, nj, nk, h)
    fwd, back = wild_maps(ni, nj, nk, h, 0.3), wild_maps(ni, nj, nk, h, 1.1)
    m = gm(ni, nj, nk, h)
    dfwd, dback, dvel = dev(*fwd), dev(*back), dev(*vel)
    # advect (backward map), three staggered components + a scalar
    ref = [np.zeros(c, np.float32) for c in (nu, nv, nw)]
    oracle().orc_advect_velocity(*map(fp, ref), *map(fp, vel), *map(fp, back), h, ni, nj, nk, 0)
    out = dev(*[np.zeros(c, np.float32) for c in (nu, nv, nw)])
    m.advectVelocity(*out, *dvel, *dback, False)
    for r, g in zip(ref, out):
        assert F.same(r, g.numpy())
    # accumulate (forward map), coefficient != 1
    init = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)]
    racc = [a.copy() for a in init]
    oracle().orc_accumulate_velocity(*map(fp, vel), *map(fp, racc), *map(fp, fwd), h, ni, nj, nk, 0, -0.5)
    dacc = dev(*init)
    m.accumulateVelocity(*dvel, *dacc, *dfwd, False, -0.5)
    for r, g in zip(racc, dacc):
        assert F.same(r, g.numpy())
    # full compensation of the velocity (error, back-mapped correction, limiter)
    cur = [F.scalar(ni + 1, nj, nk, 1.1), F.scalar(ni, nj + 1, nk, 1.2), F.scalar(ni, nj, nk + 1, 1.3)]
    ru, ri, rs = [a.copy() for a in cur], [a.copy() for a in vel], [np.zeros(c, np.float32) for c in (nu, nv, nw)]
    oracle().orc_compensate_velocity(*map(fp, ru), *map(fp, ri), *map(fp, rs), *map(fp, fwd), *map(fp, back), h, ni, nj, nk, 0)
    du, di = dev(*cur), dev(*vel)
    m.compensateVelocity(*du, *di, *dfwd, *dback, False)
    for r, g in zip(ru + ri, du + di):
        assert F.same(r, g.numpy())
    # the two-field forms
    a_init, b_init = F.scalar(ni, nj, nk, 0.4), F.scalar(ni, nj, nk, 1.9, amp=2.0)
    ra, rb = np.zeros(n, np.float32), np.zeros(n, np.float32)
    oracle().orc_advect_field(fp(ra), fp(a_init), *map(fp, back), h, ni, nj, nk, 0)
    oracle().orc_advect_field(fp(rb), fp(b_init), *map(fp, back), h, ni, nj, nk, 0)
    da, db, dai, dbi = dev(np.zeros(n, np.float32), np.zeros(n, np.float32), a_init, b_init)
    hip.gpu_advect_field2(da.ptr, dai.ptr, db.ptr, dbi.ptr, *[x.ptr for x in dback], h, ni, nj, nk, False)
    assert F.same(ra, da.numpy()) and F.same(rb, db.numpy())
    ea, eb = np.zeros(n, np.float32), np.zeros(n, np.float32)
    oracle().orc_compensate_error_field(fp(ra), fp(a_init), fp(ea), *map(fp, fwd), h, ni, nj, nk, 0)
    oracle().orc_compensate_error_field(fp(rb), fp(b_init), fp(eb), *map(fp, fwd), h, ni, nj, nk, 0)
    dea, deb = dev(np.zeros(n, np.float32), np.zeros(n, np.float32))
    hip.gpu_compensate_error_field2(da.ptr, dai.ptr, dea.ptr, db.ptr, dbi.ptr, deb.ptr, *[x.ptr for x in dfwd], h, ni, nj, nk, False)
    assert F.same(ea, dea.numpy()) and F.same(eb, deb.numpy())
    oracle().orc_accumulate_field(fp(ea), fp(ra), *map(fp, back), h, ni, nj, nk, 0, -0.5)
    oracle().orc_accumulate_field(fp(eb), fp(rb), *map(fp, back), h, ni, nj, nk, 0, 2.0)
    hip.gpu_accumulate_field2(dea.ptr, da.ptr, -0.5, deb.ptr, db.ptr, 2.0, *[x.ptr for x in dback], h, ni, nj, nk, False)
    assert F.same(ra, da.numpy()) and F.same(rb, db.numpy())
    # the structured look-up and the generic one agree on these maps too
    hip.fl_set_option(bq._lib.FL_OPT_STRUCTURED_MAPS, 0)
    out2 = dev(*[np.zeros(c, np.float32) for c in (nu, nv, nw)])
    m.advectVelocity(*out2, *dvel, *dback, False)
    hip.fl_set_option(bq._lib.FL_OPT_STRUCTURED_MAPS, 1)
    for r, g in zip(ref, out2):
        assert F.same(r, g.numpy())
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_fused_housekeeping_bits(gm, ni, nj, nk, h):
    """FL_OPT_FUSED_HOUSEKEEPING, bit by bit, against the plain operator sequence it replaces: outputs that were NOT
    cleared come out with zero borders (1), the error kernels leave the uncompensated field in `init` (2), the DMC
    update writes zero (4) or copied (8) border nodes into an uncleared output."""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    OPT = bq._lib.FL_OPT_FUSED_HOUSEKEEPING
    h, vel, fwd, back, _ = setup(ni, nj, nk, h)
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    gm(ni, nj, nk, h)
    dvel, dfwd, dback = dev(*vel), dev(*fwd), dev(*back)
    junk = lambda c: np.full(c, 7.5, np.float32)
    # bit 1: advect
    ref = dev(*[np.zeros(c, np.float32) for c in (nu, nv, nw)])
    hip.gpu_advect_velocity(*[x.ptr for x in ref], *[x.ptr for x in dvel], *[x.ptr for x in dback], h, ni, nj, nk, False)
    out = dev(junk(nu), junk(nv), junk(nw))
    hip.fl_set_option(OPT, 1)
    hip.gpu_advect_velocity(*[x.ptr for x in out], *[x.ptr for x in dvel], *[x.ptr for x in dback], h, ni, nj, nk, False)
    hip.fl_set_option(OPT, 0)
    for r, g in zip(ref, out):
        assert F.same(r.numpy(), g.numpy())
    # bits 1 | 2: error stage, velocity and two scalars at once
    cur = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)]
    dcur = dev(*cur)
    init_ref, err_ref = dev(*vel), dev(*[np.zeros(c, np.float32) for c in (nu, nv, nw)])
    hip.gpu_compensate_error_velocity(*[x.ptr for x in dcur], *[x.ptr for x in init_ref], *[x.ptr for x in err_ref],
                                      *[x.ptr for x in dfwd], h, ni, nj, nk, False)
    init_f, err_f = dev(*vel), dev(junk(nu), junk(nv), junk(nw))
    hip.fl_set_option(OPT, 3)
    hip.gpu_compensate_error_velocity(*[x.ptr for x in dcur], *[x.ptr for x in init_f], *[x.ptr for x in err_f],
                                      *[x.ptr for x in dfwd], h, ni, nj, nk, False)
    hip.fl_set_option(OPT, 0)
    for r, g in zip(err_ref, err_f):
        assert F.same(r.numpy(), g.numpy())
    for c, v0, r, g in zip(cur, vel, init_ref, init_f):
        assert F.same(v0, r.numpy())                             # the plain operator leaves init alone
        assert F.same(c, g.numpy())                              # fused: init now holds the uncompensated field
    a, b = F.scalar(ni, nj, nk, 0.4), F.scalar(ni, nj, nk, 1.9, amp=2.0)
    ai, bi = F.scalar(ni, nj, nk, 2.4), F.scalar(ni, nj, nk, 0.7)
    da, db = dev(a, b)
    r_ai, r_bi, r_ea, r_eb = dev(ai, bi, np.zeros(n, np.float32), np.zeros(n, np.float32))
    hip.gpu_compensate_error_field2(da.ptr, r_ai.ptr, r_ea.ptr, db.ptr, r_bi.ptr, r_eb.ptr, *[x.ptr for x in dfwd], h, ni, nj, nk, False)
    f_ai, f_bi, f_ea, f_eb = dev(ai, bi, junk(n), junk(n))
    hip.fl_set_option(OPT, 3)
    hip.gpu_compensate_error_field2(da.ptr, f_ai.ptr, f_ea.ptr, db.ptr, f_bi.ptr, f_eb.ptr, *[x.ptr for x in dfwd], h, ni, nj, nk, False)
    hip.fl_set_option(OPT, 0)
    assert F.same(r_ea.numpy(), f_ea.numpy()) and F.same(r_eb.numpy(), f_eb.numpy())
    assert F.same(a, f_ai.numpy()) and F.same(b, f_bi.numpy())
    # bits 4 / 8: DMC border
    xin = dev(*back)
    plain = dev(*[np.zeros(n, np.float32) for _ in range(3)])
    hip.gpu_solve_backwardDMC(*[x.ptr for x in dvel], *[x.ptr for x in xin], *[x.ptr for x in plain], h, ni, nj, nk, 0.4 * h)
    for bits, start in ((4, None), (8, back)):
        out = dev(junk(n), junk(n), junk(n))
        hip.fl_set_option(OPT, bits)
        hip.gpu_solve_backwardDMC(*[x.ptr for x in dvel], *[x.ptr for x in xin], *[x.ptr for x in out], h, ni, nj, nk, 0.4 * h)
        hip.fl_set_option(OPT, 0)
        for c in range(3):
            want = plain[c].numpy().copy()
            if start is not None:                       # what a copy of the input before the update would have left
                idx = np.arange(n); k, j, i = idx // (ni * nj), (idx // ni) % nj, idx % ni
                border = (i <= 1) | (i >= ni - 2) | (j <= 1) | (j >= nj - 2) | (k <= 1) | (k >= nk - 2)
                want[border] = start[c][border]
            assert F.same(want, out[c].numpy()), (bits, c)
    assert hip.fl_get_option(OPT) == 0
    bq.check()


@pytest.mark.parametrize("nx", [32, 33, 64, 65, 129, 256, 257, 36, 37, 260, 384, 512, 513, 1024, 1025, 772])
def test_clamp_extrema_box_row_widths(nx):
    """the limiter's marching kernel on rows of 4m floats and of 4m + 1 floats (the u component: unaligned float4
    accesses, the last column fetched by the row's last lane), and the plain kernel on everything else"""
    import gpufluidsimulation_amd as bq
    ny, nz = 11, 9
    before, after = F.scalar(nx, ny, nz, 0.2), F.scalar(nx, ny, nz, 0.45, amp=1.4)
    ref = after.copy()
    oracle().orc_clamp_extrema_box(fp(before), fp(ref), nx, ny, nz)
    db, da = dev(before, after)
    bq.hip_lib().gpu_clamp_extrema_box(db.ptr, da.ptr, nx, ny, nz)
    assert F.same(ref, da.numpy())
    assert not F.same(ref, after)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,h", GRIDS)
def test_plane_windows_partition_the_operators(gm, ni, nj, nk, h):
    """fl_set_plane_window: a map operator run on three plane windows that partition the grid (with the fused
    housekeeping, so that it is nothing but window-honouring launches) produces what one unrestricted call does --
    the split a z-slab host uses to overlap its ghost-plane exchange with the operator's interior planes."""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    OPT = bq._lib.FL_OPT_FUSED_HOUSEKEEPING
    h, vel, fwd, back, _ = setup(ni, nj, nk, h)
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    gm(ni, nj, nk, h)
    dvel, dfwd, dback = dev(*vel), dev(*fwd), dev(*back)
    cur = [F.scalar(ni + 1, nj, nk, 0.1), F.scalar(ni, nj + 1, nk, 0.2), F.scalar(ni, nj, nk + 1, 0.3)]
    windows = [(5, nk - 4), (0, 5), (nk - 4, nk)]            # interior first, then the two ends

    def both(run, make_outputs):
        """run(outputs) once unrestricted and once per window; returns the two sets of output arrays"""
        full = make_outputs(); run(full)
        split = make_outputs()
        for k0, k1 in windows:
            assert hip.fl_set_plane_window(k0, k1) == 1
            run(split)
        hip.fl_set_plane_window(-1, -1)
        return [x.numpy() for x in full], [x.numpy() for x in split]

    junk = lambda c: np.full(c, 3.25, np.float32)
    hip.fl_set_option(OPT, 1 | 2 | 4)
    try:
        # advect (outputs start as junk: the kernels write the zero border themselves)
        a, b = both(lambda o: hip.gpu_advect_velocity(*[x.ptr for x in o], *[x.ptr for x in dvel], *[x.ptr for x in dback], h, ni, nj, nk, False),
                    lambda: dev(junk(nu), junk(nv), junk(nw)))
        assert all(F.same(x, y) for x, y in zip(a, b))
        # compensate-error: error outputs and the init <- field store
        dcur = dev(*cur)
        a, b = both(lambda o: hip.gpu_compensate_error_velocity(*[x.ptr for x in dcur], *[x.ptr for x in o[:3]], *[x.ptr for x in o[3:]],
                                                                *[x.ptr for x in dfwd], h, ni, nj, nk, False),
                    lambda: dev(*vel) + dev(junk(nu), junk(nv), junk(nw)))
        assert all(F.same(x, y) for x, y in zip(a, b))
        # accumulate (dst += ...): every node exactly once
        a, b = both(lambda o: hip.gpu_accumulate_velocity(*[x.ptr for x in dvel], *[x.ptr for x in o], *[x.ptr for x in dfwd], h, ni, nj, nk, False, -0.5),
                    lambda: dev(*cur))
        assert all(F.same(x, y) for x, y in zip(a, b))
        # DMC sub-step with its border nodes, forward map update in place
        a, b = both(lambda o: hip.gpu_solve_backwardDMC(*[x.ptr for x in dvel], *[x.ptr for x in dback], *[x.ptr for x in o], h, ni, nj, nk, 0.4 * h),
                    lambda: dev(junk(n), junk(n), junk(n)))
        assert all(F.same(x, y) for x, y in zip(a, b))
        a, b = both(lambda o: hip.gpu_solve_forward(*[x.ptr for x in dvel], *[x.ptr for x in o], h, ni, nj, nk, 0.3 * h, 0.9 * h),
                    lambda: dev(*fwd))
        assert all(F.same(x, y) for x, y in zip(a, b))
        # two scalar fields at once
        fa, fb = F.scalar(ni, nj, nk, 0.4), F.scalar(ni, nj, nk, 1.9, amp=2.0)
        dfa, dfb = dev(fa, fb)
        a, b = both(lambda o: hip.gpu_advect_field2(o[0].ptr, dfa.ptr, o[1].ptr, dfb.ptr, *[x.ptr for x in dback], h, ni, nj, nk, False),
                    lambda: dev(junk(n), junk(n)))
        assert all(F.same(x, y) for x, y in zip(a, b))
    finally:
        hip.fl_set_option(OPT, 0)
        hip.fl_set_plane_window(-1, -1)
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,h", [(32, 32, 32, 1.0 / 32), (40, 24, 16, 1.0 / 64)])
@pytest.mark.parametrize("zero_border", [False, True])
def test_map_quarter_fp32_option_changes_no_bit(gm, ni, nj, nk, h, zero_border):
    """FL_OPT_MAP_QUARTER_FP32 (the weight-1/4 lerps of the structured map look-up as one fp32 fma) on maps that pass
    gpu_maps_quarter_safe -- warped coordinates, optionally with the zeroed border the DMC update leaves --
    against the oracle, for every operator that stages map tiles: one and two fields, all four staggerings."""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    h = float(np.float32(h))
    n, nu, nv, nw = F.sizes(ni, nj, nk)
    vel = F.velocity(ni, nj, nk, h)
    fwd, back = F.warped_maps(ni, nj, nk, h, 0.8, 0.3), F.warped_maps(ni, nj, nk, h, -0.7, 1.1)
    if zero_border:
        for a in back:
            a3 = a.reshape(nk, nj, ni)
            keep = a3[2:-2, 2:-2, 2:-2].copy()
            a3[...] = 0.0
            a3[2:-2, 2:-2, 2:-2] = keep
    m = gm(ni, nj, nk, h)
    dfwd, dback, dvel = dev(*fwd), dev(*back), dev(*vel)
    assert hip.gpu_maps_quarter_safe(*[x.ptr for x in dback], h, ni, nj, nk) == 1
    assert hip.gpu_maps_quarter_safe(*[x.ptr for x in dfwd], h, ni, nj, nk) == 1
    hip.fl_set_option(bq._lib.FL_OPT_MAP_QUARTER_FP32, 1)
    try:
        ref = [np.zeros(c, np.float32) for c in (nu, nv, nw)]
        oracle().orc_advect_velocity(*map(fp, ref), *map(fp, vel), *map(fp, back), h, ni, nj, nk, 0)
        out = dev(*[np.zeros(c, np.float32) for c in (nu, nv, nw)])
        m.advectVelocity(*out, *dvel, *dback, False)
        for r, g in zip(ref, out):
            assert F.same(r, g.numpy())
        cur = [F.scalar(ni + 1, nj, nk, 1.1), F.scalar(ni, nj + 1, nk, 1.2), F.scalar(ni, nj, nk + 1, 1.3)]
        ru, ri, rs = [a.copy() for a in cur], [a.copy() for a in vel], [np.zeros(c, np.float32) for c in (nu, nv, nw)]
        oracle().orc_compensate_velocity(*map(fp, ru), *map(fp, ri), *map(fp, rs), *map(fp, fwd), *map(fp, back), h, ni, nj, nk, 0)
        du, di = dev(*cur), dev(*vel)
        m.compensateVelocity(*du, *di, *dfwd, *dback, False)
        for r, g in zip(ru + ri, du + di):
            assert F.same(r, g.numpy())
        a_init, b_init = F.scalar(ni, nj, nk, 0.4), F.scalar(ni, nj, nk, 1.9, amp=2.0)
        ra, rb = np.zeros(n, np.float32), np.zeros(n, np.float32)
        oracle().orc_advect_field(fp(ra), fp(a_init), *map(fp, back), h, ni, nj, nk, 0)
        oracle().orc_advect_field(fp(rb), fp(b_init), *map(fp, back), h, ni, nj, nk, 0)
        da, db, dai, dbi = dev(np.zeros(n, np.float32), np.zeros(n, np.float32), a_init, b_init)
        hip.gpu_advect_field2(da.ptr, dai.ptr, db.ptr, dbi.ptr, *[x.ptr for x in dback], h, ni, nj, nk, False)
        assert F.same(ra, da.numpy()) and F.same(rb, db.numpy())
        ea, eb = np.zeros(n, np.float32), np.zeros(n, np.float32)
        oracle().orc_compensate_error_field(fp(ra), fp(a_init), fp(ea), *map(fp, fwd), h, ni, nj, nk, 0)
        oracle().orc_compensate_error_field(fp(rb), fp(b_init), fp(eb), *map(fp, fwd), h, ni, nj, nk, 0)
        dea, deb = dev(np.zeros(n, np.float32), np.zeros(n, np.float32))
        hip.gpu_compensate_error_field2(da.ptr, dai.ptr, dea.ptr, db.ptr, dbi.ptr, deb.ptr, *[x.ptr for x in dfwd], h, ni, nj, nk, False)
        assert F.same(ea, dea.numpy()) and F.same(eb, deb.numpy())
        oracle().orc_accumulate_field(fp(ea), fp(ra), *map(fp, back), h, ni, nj, nk, 0, -0.5)
        oracle().orc_accumulate_field(fp(eb), fp(rb), *map(fp, back), h, ni, nj, nk, 0, 2.0)
        hip.gpu_accumulate_field2(dea.ptr, da.ptr, -0.5, deb.ptr, db.ptr, 2.0, *[x.ptr for x in dback], h, ni, nj, nk, False)
        assert F.same(ra, da.numpy()) and F.same(rb, db.numpy())
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_MAP_QUARTER_FP32, 0)
    bq.check()


def test_map_value_guards_catch_what_the_fp32_lerps_cannot_take(gm):
    """gpu_maps_quarter_safe and the guard fused into the map updates (fl_map_guard_*): a NaN, an Inf, a negative or a
    tiny positive coordinate anywhere in a map says no; ordinary maps (zeros included) say yes"""
    import ctypes as C
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    ni, nj, nk, h = 32, 32, 32, float(np.float32(1.0 / 32))
    gm(ni, nj, nk, h)
    base = F.warped_maps(ni, nj, nk, h, 0.8, 0.3)
    for bad in (np.nan, np.inf, -1e-3, 1e-30, 40.0):
        m = [a.copy() for a in base]
        m[1][12345] = bad
        d = dev(*m)
        assert hip.gpu_maps_quarter_safe(*[x.ptr for x in d], h, ni, nj, nk) == 0, bad
    d = dev(*base)
    assert hip.gpu_maps_quarter_safe(*[x.ptr for x in d], h, ni, nj, nk) == 1
    # the fused guard: a DMC sub-step and a forward update on ordinary data flag nothing ...
    vel = F.velocity(ni, nj, nk, h)
    dvel, din = dev(*vel), dev(*base)
    dout = dev(*[np.zeros(ni * nj * nk, np.float32) for _ in range(3)])
    ok = (C.c_int * 2)()
    hip.fl_map_guard_reset(0); hip.fl_map_guard_reset(1)
    hip.gpu_solve_backwardDMC(*[x.ptr for x in dvel], *[x.ptr for x in din], *[x.ptr for x in dout], h, ni, nj, nk, 0.5 * h)
    hip.gpu_solve_forward(*[x.ptr for x in dvel], *[x.ptr for x in din], h, ni, nj, nk, 0.5 * h, h)
    hip.fl_map_guard_read(ok)
    assert list(ok) == [1, 1]
    # ... a velocity field with a NaN poisons the backward map it touches, and the guard says so; the forward map cannot
    # be poisoned: traceRK3 clamps every position it returns into [h, (n-1) h] (GPU_kernel.cu:88-89), a NaN included
    v2 = [a.copy() for a in vel]
    v2[0][(ni + 1) * nj * 16 + (ni + 1) * 16 + 16] = np.nan
    dbad = dev(*v2)
    hip.fl_map_guard_reset(0); hip.fl_map_guard_reset(1)
    hip.gpu_solve_backwardDMC(*[x.ptr for x in dbad], *[x.ptr for x in din], *[x.ptr for x in dout], h, ni, nj, nk, 0.5 * h)
    hip.gpu_solve_forward(*[x.ptr for x in dbad], *[x.ptr for x in din], h, ni, nj, nk, 0.5 * h, h)
    hip.fl_map_guard_read(ok)
    assert list(ok) == [0, 1]
    hip.fl_map_guard_reset(-1)
    bq.check()


def test_nonfinite_velocity_is_flagged(gm):
    """gpu_max_abs3 skips NaNs like the reference's host scan, so the CFL of a field that has gone NaN looks calm;
    fl_nonfinite_seen is how a driver finds out (sticky until reset; NaN and Inf, in any of the three components)."""
    import gpufluidsimulation_amd as bq
    lib = bq.hip_lib()
    ni, nj, nk, h = GRIDS[0]
    gm(ni, nj, nk, h)
    u, v, w = F.velocity(ni, nj, nk, float(np.float32(h)))
    lib.fl_nonfinite_seen(1)
    d = dev(u, v, w)
    ref = max(1e-4, float(max(np.abs(u).max(), np.abs(v).max(), np.abs(w).max())))
    assert lib.gpu_max_abs3(d[0].ptr, d[1].ptr, d[2].ptr, ni, nj, nk) == np.float32(ref) and lib.fl_nonfinite_seen(0) == 0
    for which, bad in ((0, np.nan), (1, np.inf), (2, -np.inf)):
        f = [u.copy(), v.copy(), w.copy()]
        f[which][f[which].size // 3] = bad
        d = dev(*f)
        got = lib.gpu_max_abs3(d[0].ptr, d[1].ptr, d[2].ptr, ni, nj, nk)
        assert lib.fl_nonfinite_seen(0) == 1 and lib.fl_nonfinite_seen(1) == 1 and lib.fl_nonfinite_seen(0) == 0
        assert got == (np.float32(ref) if np.isnan(bad) else np.float32(np.inf))
    bq.check()


@pytest.mark.parametrize("ni,nj,nk,h", [(24, 20, 16, 1.0 / 24), (72, 68, 66, 0.002), (130, 24, 20, 0.002), (40, 36, 30, 0.01)])
@pytest.mark.parametrize("fast", [0, 1])
def test_tabled_lookup_on_other_spacings(gm, ni, nj, nk, h, fast):
    """Round 4: on spacings that are not a power of two the structured map look-up runs from per-axis tables of cell and
    weight (csrc/bq_device.hip.h: MapTabs) instead of 27 IEEE divisions per component and node.  h = 0.002f is the reference
    binary's own spacing: there (i h) / h rounds to just below i at i = 63, 125, 126, so the centre tap reads cell i - 1 with
    weight 1 - 2^-24 -- the grids here put index 63 on every axis.  Operators with warped and with wild maps, one and two
    fields, in both arithmetic variants: the oracle's bits, and the generic path's (FL_OPT_STRUCTURED_MAPS = 0)."""
    import gpufluidsimulation_amd as bq
    hip = bq.hip_lib()
    hip.fl_set_option(bq._lib.FL_OPT_FAST_LERP, fast)
    oracle().orc_set_fast_lerp(fast)
    try:
        test_advect_velocity_and_field(gm, ni, nj, nk, h, False)
        test_compensate_velocity_and_field(gm, ni, nj, nk, h)
        test_accumulate(gm, ni, nj, nk, h, -0.5)
        test_batched_scalar_ops(gm, ni, nj, nk, h, False)
        test_gather_ops_on_wild_maps(gm, ni, nj, nk, h)
        test_fused_housekeeping_bits(gm, ni, nj, nk, h)
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 0)
        oracle().orc_set_fast_lerp(0)
