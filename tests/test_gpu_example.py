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


def test_cpp_rank_driver_two_ranks_stitch_to_one(tmp_path):
    """examples/bimocq3d_ranks.cpp: the N-rank driver in C++ only (RANK / WORLD_SIZE from the environment, the ncclUniqueId
    through a file).  Two ranks on this one GPU through the stream-ordered stand-in for librccl (BQ_RCCL_LIBRARY) write
    per-slab dumps that stitch to the dumps of the same program run as a single rank, byte for byte."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from build_fake_rccl import build
    from gpufluidsimulation_amd.solver import read_density_dump
    exe2 = os.path.join(ROOT, "build", "bimocq3d_ranks")
    if not os.path.exists(exe2):
        subprocess.check_call(["make", "-s", "example"], cwd=ROOT)
    fake = build("async")
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    args = ["64", "48", "64", "3", None, "1", "8", "60"]            # leapfrog scene: the sources impose velocity
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([exe2] + [one if a is None else a for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "last dump ok" in r.stdout, r.stdout[-2000:]
    procs = [subprocess.Popen([exe2] + [two if a is None else a for a in args],
                              env=dict(env, RANK=str(rk), WORLD_SIZE="2", LOCAL_RANK="0", BQ_JOB_ID="t1", BQ_RCCL_LIBRARY=fake),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for rk in (1, 0)]      # rank 1 first: it has to wait for the id
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "[rank 0/2]" in outs[1] and "ghost exchanges" in outs[1] and "last dump ok" in outs[0]
    for f in sorted(x for x in os.listdir(one) if x.endswith(".bqd")):
        _, rec = read_density_dump(os.path.join(one, f))
        parts = sorted(p for p in os.listdir(two) if p.startswith(f[:-4] + ".k"))
        assert len(parts) == 2, (f, os.listdir(two))
        stitched = np.concatenate([read_density_dump(os.path.join(two, p))[1] for p in parts])
        assert len(rec) > 100 and stitched.tobytes() == rec.tobytes(), f
