"""BASELINE's full sizes on the GPU (configs 2, 3 and the 512^3 grid of config 4): what the toy shapes of the other
files cannot reach -- the launcher's auto-chunking at 128^3 (one-row fused Jacobi kernel), 256^3 (two-row kernel, 8
chunks of 32 planes) and 512^3 (three-sweep two-segment kernel, 4 chunks of 128 planes; WIDE two-row kernel for the remainder), and the gather
kernels on >2^24-element fields.

* 128^3 / 256^3: per-step SHA-256 of rho, u, v, w against the CPU oracle's, committed as
  tests/golden/rising_smoke_hashes.json by tests/golden/make_hashes.py (no oracle in the loop here);
* 512^3: no oracle can run it in test time, so size-independent properties -- the production kernels against the
  generic one-thread-per-cell kernels of the same library (bit-identical by contract), finiteness, conservation of
  what advection conserves, the dump's voxel count."""
import json
import os
import sys

import numpy as np
import pytest

import fields as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _hashes():
    with open(os.path.join(HERE, "golden", "rising_smoke_hashes.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", ["128", "256"])
def test_hip_reproduces_full_size_hashes(case):
    from make_hashes import FIELDS, SMOKE, digest_hex
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    rows = _hashes()["cases"][case]
    n = int(case)
    s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0)
    s.setSmoke(0.0, 1.0, [SMOKE])
    s.setProjection(200, 0.5)
    s.setOption(3, 1)                                   # the reference's full per-step sequence, as in bench.py
    for row in rows:
        s.advance(row["step"] - 1, 2.0 / n)
        assert float(np.float32(s.cfldt)) == row["cfldt"], row["step"]
        for k in FIELDS:
            assert digest_hex(s.field(k)) == row[k], (case, row["step"], k)
    s._check()
    s.close()


_LONG_FINAL = {}          # final fields of the 128^3 x 200 runs, exact and one-fma: what the RMS test below compares


@pytest.mark.parametrize("case", ["128_200", "128_200_fast", "256_12", "256_rise8_12"])
def test_hip_reproduces_long_run_hashes(case):
    """Round 4: the north star's own LENGTH.  tests/golden/long_run_hashes.json (tests/golden/make_long_hashes.py, CPU oracle):
    128^3 for 200 steps (two DMC sub-steps per step from step ~41 on) in the exact arithmetic and in the oracle's one-fma mode
    (which pins the library's FL_OPT_FAST_LERP variant with its z-marching window kernels), 256^3 for 12 steps, and 256^3 with
    eight times the buoyancy so that 2^24-element fields reach the two-sub-step regime within 12 steps.  cfldt every step,
    SHA-256 of rho, u, v, w where the fixture holds them (every 10th step of the long runs).  No oracle in the loop here."""
    import gpufluidsimulation_amd as bq
    from make_hashes import FIELDS, digest_hex
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    path = os.path.join(HERE, "golden", "long_run_hashes.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/long_run_hashes.json not generated")
    spec = json.load(open(path))["cases"].get(case)
    if spec is None:
        pytest.skip(f"case {case} not in the fixture")
    n = spec["grid"]
    hip = bq.hip_lib()
    hip.fl_set_option(bq._lib.FL_OPT_FAST_LERP, spec["fast_lerp"])
    try:
        s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0)
        s.setSmoke(0.0, spec["rise"], [tuple(spec["emitter"])])
        s.setProjection(spec["jacobi_iters"], spec["halfrdx"])
        s.setOption(3, 1)
        checked = 0
        for row in spec["rows"]:
            s.advance(row["step"] - 1, 2.0 / n)
            assert float(np.float32(s.cfldt)) == row["cfldt"], (case, row["step"])
            if "rho" in row:
                for k in FIELDS:
                    assert digest_hex(s.field(k)) == row[k], (case, row["step"], k)
                checked += 1
        assert checked >= 2
        if case.startswith("128_200"):
            _LONG_FINAL[case] = {k: s.field(k).astype(np.float64) for k in FIELDS}
        s._check()
        s.close()
    finally:
        hip.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 0)


def test_fast_variant_is_within_the_tolerance_after_200_steps_at_128():
    """SURVEY 8(d)'s parity figure for the fast variant at BASELINE config 2's size: RMS of rho, u, v, w against the exact
    fields after 200 steps <= 1e-5 (the north star's tolerance).  Both trajectories were just checked against the oracle's
    hashes (exact mode / one-fma mode), so this IS the deviation of the oracle's two arithmetic modes."""
    if set(_LONG_FINAL) != {"128_200", "128_200_fast"}:
        pytest.skip("needs both 128^3 x 200 cases of test_hip_reproduces_long_run_hashes in this session")
    worst = {}
    for k in ("rho", "u", "v", "w"):
        a, b = _LONG_FINAL["128_200"][k], _LONG_FINAL["128_200_fast"][k]
        assert not np.array_equal(a, b), k                                 # it really is another arithmetic
        worst[k] = float(np.sqrt(np.mean((a - b) ** 2)))
    assert max(worst.values()) <= 1e-5, worst
    print("RMS fast vs exact, 128^3 after 200 steps:", worst)


@pytest.mark.parametrize("case", ["mgcg_128", "reflection_128", "reflection_mgcg_64", "mgcg_256", "reflection_256"])
def test_hip_reproduces_next_row_hashes(case):
    """SURVEY 8(f) rows N1 / N3 at sizes the toy shapes do not reach: the fp64 multigrid-CG projection at 128^3 and 256^3
    (lean smoother on levels 0 and 1, LDS tile smoother below, block transfer operators, marching residual), the
    MAC_REFLECTION scheme at 128^3 and 256^3 with the Jacobi projection, and the reference binary's default configuration
    (reflection + multigrid-CG) at 64^3 -- per-step SHA-256 of rho, u, v, w against the CPU oracle's, committed as
    tests/golden/next_row_hashes.json by tests/golden/make_next_row_hashes.py (no oracle in the loop here)."""
    from make_hashes import FIELDS, SMOKE, digest_hex
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    with open(os.path.join(HERE, "golden", "next_row_hashes.json")) as f:
        spec = json.load(f)["cases"][case]
    n = spec["grid"]
    s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0, scheme=spec["scheme"])
    s.setSmoke(0.0, 1.0, [SMOKE])
    s.setProjection(spec["iterations"], 0.5, spec["projection_kind"])
    s.setOption(3, 1)
    for row in spec["rows"]:
        s.advance(row["step"] - 1, 2.0 / n)
        assert float(np.float32(s.cfldt)) == row["cfldt"], row["step"]
        for k in FIELDS:
            assert digest_hex(s.field(k)) == row[k], (case, row["step"], k)
    s._check()
    s.close()


def test_512_production_kernels_equal_generic_kernels(tmp_path):
    """512^3 (the grid of BASELINE config 4), 2 steps, 200 Jacobi iterations: default launch configuration (three-sweep
    two-segment Jacobi kernel + the WIDE two-row kernel for the remainder, LDS-staged structured map look-ups, marching limiter) against FL_OPT_JACOBI_VARIANT = 1 /
    FL_OPT_JACOBI_FUSE = 0 / FL_OPT_STRUCTURED_MAPS = 0 (one thread per cell, generic map look-up)."""
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    L_ = bq._lib
    lib = bq.hip_lib()
    n = 512
    dt = 2.0 / n
    em = [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)]
    res = {}
    for mode in ("production", "generic"):
        if mode == "generic":
            lib.fl_set_option(L_.FL_OPT_JACOBI_VARIANT, 1); lib.fl_set_option(L_.FL_OPT_JACOBI_FUSE, 0)
            lib.fl_set_option(L_.FL_OPT_STRUCTURED_MAPS, 0)
        try:
            s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0)
            s.setSmoke(0.0, 1.0, em); s.setProjection(200, 0.5); s.setOption(3, 1)
            for f in range(2):
                s.advance(f, dt)
            s._check()
            res[mode] = {k: s.field(k) for k in ("rho", "v", "p")}
            if mode == "production":
                count = s.outputResult(1, str(tmp_path))
                kernel = (lib.fl_jacobi_kernel_name() or b"").decode()
            s.close()
        finally:
            lib.fl_set_option(L_.FL_OPT_JACOBI_VARIANT, 0); lib.fl_set_option(L_.FL_OPT_JACOBI_FUSE, 1)
            lib.fl_set_option(L_.FL_OPT_STRUCTURED_MAPS, 1)
    assert kernel == "jacobi_lds2seg_kernel", kernel                      # the three-sweep kernel for rows of two float4 segments did run
    for k in res["production"]:
        a, b = res["production"][k], res["generic"][k]
        assert np.isfinite(a).all(), k
        assert F.same(a, b), (k, F.maxdiff(a, b))
    rho = res["production"]["rho"]
    inside = int((np.abs(rho) > 1e-4).sum())
    assert count == inside and inside > 0.9 * 4.0 / 3.0 * np.pi * (0.1 * n) ** 3      # the source sphere, a little smeared
    assert float(np.abs(res["production"]["v"]).max()) > 1e-3                           # buoyancy acted
