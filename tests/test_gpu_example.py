"""The C++ driver of examples/bimocq3d_main.cpp (the reference's main.cpp loop on this library) runs end to end on
the GPU: frames advance, the asynchronous dumps appear and parse, both schemes and both projections start."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "bimocq3d")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "example"], cwd=ROOT)
    return EXE


@pytest.mark.parametrize("scheme,projection", [(0, 0), (0, 1), (3, 0)])
def test_driver_runs_and_dumps(exe, tmp_path, scheme, projection):
    from gpufluidsimulation_amd.solver import read_density_dump
    out = str(tmp_path / "out")
    r = subprocess.run([exe, "48", "4", out, str(scheme), str(projection), "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "Frame 3 Starts !!!" in r.stdout and "[Bimocq GPU Time:" in r.stdout and "last dump ok" in r.stdout
    files = sorted(os.listdir(out))
    assert files == [f"density_render_{i:04d}.bqd" for i in range(1, 5)], files
    hd, rec = read_density_dump(os.path.join(out, files[-1]))
    assert hd["nx"] == 48 and hd["count"] == len(rec) and len(rec) > 50
    assert np.all(rec["value"] > 1e-4)


def test_leapfrog_scene_config5_shape(exe, tmp_path):
    """BASELINE config 5 scaled down: N x N x N/2 box, two coaxial vortex rings from the emitter's velocity formula,
    a density dump every frame (blocking writer)."""
    from gpufluidsimulation_amd.solver import read_density_dump
    out = str(tmp_path / "leap")
    r = subprocess.run([exe, "64", "6", out, "0", "0", "0", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "6 frames of 64x64x32" in r.stdout
    files = sorted(os.listdir(out))
    assert files == [f"density_render_{i:04d}.bqd" for i in range(1, 7)], files
    for f in files:
        hd, rec = read_density_dump(os.path.join(out, f))
        assert (hd["nx"], hd["ny"], hd["nz"]) == (64, 64, 32) and hd["count"] == len(rec) > 100
        assert np.isfinite(rec["value"]).all() and rec["k"].max() < 32
        # two separate puffs of smoke, one around each ring's source (x = 0.15 and 0.35 of 64 cells)
        assert (rec["i"] < 16).any() and (rec["i"] > 19).any()
