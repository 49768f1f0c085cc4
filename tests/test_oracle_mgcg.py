"""Known-answer tests of the MGCG oracle (oracle/mgcg_oracle.c).  The reference holds no
fixture for this operator: what can be pinned on the CPU is that the restated algorithm does what a
multigrid-corrected CG projection must do, and that its quirks are the documented ones."""
import numpy as np
import pytest

from mgcg_case import HostCase, velocity
from oracle_lib import dp, fp, lib as oracle


def run(ni, nj, nk, levels, iters, hr):
    h = 1.0 / ni
    u, v, w = velocity(ni, nj, nk, h)
    c = HostCase(ni, nj, nk, levels)
    before = c.interior_div_norm(u, v, w)
    oracle().orc_multi_grid_conjugate_gradient(fp(u), fp(v), fp(w), dp(c.div), dp(c.p), dp(c.dir), dp(c.residual),
                                               dp(c.temp0), dp(c.temp1), dp(c.result), c.table, levels, iters, hr)
    return c, before, c.interior_div_norm(u, v, w)


def test_projection_removes_divergence():
    """halfrdx = 1: u -= grad p with lap p = div, so the interior divergence collapses"""
    c, before, after = run(40, 36, 32, 3, 12, 1.0)
    assert after < 1e-3 * before, (before, after)
    hist = c.result[2000:2013]
    assert hist[0] > 0 and hist[-1] < 1e-4 * hist[0]         # max positive residual, per outer iteration
    assert np.all(np.isfinite(c.p))


def test_quarter_strength_with_reference_halfrdx():
    """halfrdx = 0.5 in both divergence and gradient: a converged solve leaves 3/4 of it"""
    c, before, after = run(40, 36, 32, 3, 12, 0.5)
    assert abs(after / before - 0.75) < 2e-3, after / before


def test_dot_product_quirk_is_the_documented_one():
    """M1: block result = float(sum of 12 float-narrowed partials + raw products 3, 7, 11, 15)"""
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal(700), rng.standard_normal(700)
    out = np.zeros(3)
    oracle().orc_mg_dot_partials(dp(a), dp(b), dp(out), 700)
    prod = np.zeros(768); prod[:700] = a * b
    for blk in range(3):
        sh = prod[blk * 256:(blk + 1) * 256]
        part = [np.float64(np.float32(sum_left(sh[t * 16:(t + 1) * 16]))) for t in range(16)]
        terms = [part[t] if t % 4 != 3 else sh[t] for t in range(16)]
        assert out[blk] == np.float64(np.float32(sum_left(terms)))
    exact = float(np.dot(a, b))
    assert abs(out.sum() - exact) > 1e-6 * abs(exact)          # it really is not the dot product


def sum_left(values):
    s = np.float64(values[0])
    for x in values[1:]:
        s = s + np.float64(x)
    return s


def test_restriction_and_prolongation_are_float_lerps():
    """M2: constants survive exactly; a field whose values need more than 24 bits is narrowed"""
    ni = nj = nk = 9
    ci = cj = ck = 4
    fine = np.full(ni * nj * nk, 1.0 + 2.0 ** -40)
    coarse = np.zeros(ci * cj * ck)
    oracle().orc_mg_restrict(dp(fine), dp(coarse), ni, nj, nk, ci, cj, ck)
    assert np.all(coarse == 1.0)                               # 1 + 2^-40 narrows to 1.0f
    # even fine dims (10 -> 4): the last interior plane/row/column sits at coarse coordinate 3.5
    ni = nj = nk = 10
    x = np.zeros(ni * nj * nk)
    oracle().orc_mg_prolong(dp(x), dp(np.full(ci * cj * ck, 3.0)), ni, nj, nk, ci, cj, ck)
    X = x.reshape(nk, nj, ni)
    assert np.all(X[0] == 0) and np.all(X[:, 0] == 0) and np.all(X[:, :, 0] == 0)      # boundary untouched
    assert np.all(X[-1] == 0) and np.all(X[:, -1] == 0) and np.all(X[:, :, -1] == 0)
    assert np.all(X[1:8, 1:8, 1:8] == 3.0)
    # M4: one past a coarse row/plane wraps into the next one (in-allocation: 3.0 here); one past the
    # ARRAY reads 0, so the top interior plane gets half weight on nothing
    assert X[3, 3, 8] == 3.0 and X[3, 8, 3] == 3.0
    assert X[8, 3, 3] == 1.5


def test_smoothing_rounds_odd_counts_up_and_keeps_boundary():
    ni, nj, nk = 8, 7, 6
    rng = np.random.default_rng(5)
    b = rng.standard_normal(ni * nj * nk)
    x1, x2, t = np.zeros_like(b), np.zeros_like(b), np.zeros_like(b)
    oracle().orc_mg_smooth(dp(x1), dp(b), dp(t), -1.0, 1.0 / 6.0, ni, nj, nk, 3)
    t[:] = 0
    oracle().orc_mg_smooth(dp(x2), dp(b), dp(t), -1.0, 1.0 / 6.0, ni, nj, nk, 4)
    assert np.array_equal(x1, x2)
    X = x1.reshape(nk, nj, ni)
    assert np.all(X[0] == 0) and np.all(X[-1] == 0) and np.all(X[:, 0] == 0) and np.all(X[:, :, -1] == 0)
