"""BASELINE config 5 (bimocq3D 1024 x 1024 x 512 leapfrogging vortex rings, 8 GPUs, density dumped every frame) at its OWN
row width and rank geometry -- rows of 1024 / 1025 floats, planes of 1 M cells, the vortex-ring scene whose sources impose
velocity through the emitter's acosf / cosf:

* one GPU, 1024 x 1024 x 32: per-step SHA-256 of rho, u, v, w against the CPU oracle's, committed as
  tests/golden/config5_hashes.json by tests/golden/make_config5_hashes.py (no oracle in the loop here);
* two z-slab ranks of 1024 x 1024 x 32 + 2 x 8 planes on the stream-ordered RCCL stand-in against one GPU at
  1024 x 1024 x 64: every field bit-identical, the per-slab dumps stitch to the single-GPU dump byte for byte;
* BASELINE config 4's rank geometry too: two ranks of 512 x 512 x (64 + 16) planes against one GPU at 512 x 512 x 128;
* the full 1024 x 1024 x 512 grid on ONE rank is refused with the 2 GiB message (include/bimocq_gpu.h, "Limits").
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_hip_reproduces_config5_row_geometry_hashes():
    from make_hashes import FIELDS, digest_hex
    from gpufluidsimulation_amd.scenes import leapfrog
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    import gpufluidsimulation_amd as bq
    with open(os.path.join(HERE, "golden", "config5_hashes.json")) as f:
        spec = json.load(f)
    nx, ny, nz = spec["grid"]
    assert (nx, ny) == (1024, 1024)
    h = 1.0 / nx
    s = BimocqGPUSolver(nx, ny, nz, 1.0, 0.0, 1.0)
    s.setSmoke(0.0, 0.0, leapfrog(nz, h))
    s.setProjection(spec["scene"]["jacobi_iters"], spec["scene"]["halfrdx"])
    s.setOption(3, 1)                                   # the reference's full per-step sequence, as in bench.py
    for row in spec["rows"]:
        s.advance(row["step"] - 1, 2.0 * h)
        assert float(np.float32(s.cfldt)) == row["cfldt"], row["step"]
        for k in FIELDS:
            assert digest_hex(s.field(k)) == row[k], (row["step"], k)
    # the velocity ring really was imposed (0.06 (1 + 0.01 cos 8 theta)) and the fused Jacobi kernel took rows of four waves
    assert 0.0594 < float(np.abs(s.field("u")).max()) <= 0.0606 + 0.01
    assert (bq.hip_lib().fl_jacobi_kernel_name() or b"").decode() == "jacobi_lean2r_kernel"
    s._check()
    s.close()


def test_two_ranks_of_config5_rows_equal_one_gpu(tmp_path):
    """2 x (1024 x 1024 x 32 owned + 16 ghost planes) on the stream-ordered stand-in for RCCL (device mailboxes over hipIpc,
    stream memory operations: a missing stream dependency shows as a mismatch) against one GPU at 1024 x 1024 x 64, two
    steps of the leapfrog scene with 200 Jacobi iterations: RMS of rho, u, v, w exactly 0, dumps stitch byte for byte."""
    from build_fake_rccl import build
    from gpufluidsimulation_amd.solver import read_density_dump
    fake = build("async")               # (a ghost exchange of nine 1025 x 1024 planes is 38 MB: mailbox slots of 48 MB below)
    ref, one, two = str(tmp_path / "ref"), str(tmp_path / "one"), str(tmp_path / "two")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BQ_FAKE_RCCL_DELAY_MB"):
        env.pop(k, None)
    worker = os.path.join(HERE, "slab_deviation_worker.py")
    common = ["--grid", "1024", "1024", "64", "--scene", "leapfrog", "--steps", "2", "--iters", "200", "--checkpoints", "1", "2"]
    r = subprocess.run([sys.executable, worker, "--make-reference", ref, "--dump", one, *common], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    js = str(tmp_path / "dev.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), worker,
                        "--reference", ref, "--dump", two, *common, "--rms-tol", "1e-5", "--json", js], cwd=ROOT,
                       env=dict(env, BQ_RCCL_LIBRARY=fake, BQ_FAKE_RCCL_SLOT_MB="48"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:]
    out = json.load(open(js))
    assert out["grid"] == [1024, 1024, 64] and out["ranks"] == 2 and "RCCL branch" in out["transport"]
    assert out["keep_dmc_border"] == 0                      # the reference-faithful mode bench.py runs (wall sheets travel)
    assert out["worst_rms"] == 0.0, out["checkpoints"][-1]
    files = sorted(os.listdir(one))
    assert files == ["density_render_0001.bqd", "density_render_0002.bqd"], files
    for f in files:
        hd, rec = read_density_dump(os.path.join(one, f))
        parts = sorted(p for p in os.listdir(two) if p.startswith(f[:-4] + ".k"))
        assert parts == [f[:-4] + ".k00000.bqd", f[:-4] + ".k00032.bqd"], parts
        stitched = np.concatenate([read_density_dump(os.path.join(two, p))[1] for p in parts])
        assert hd["nx"] == 1024 and len(rec) > 100000 and stitched.tobytes() == rec.tobytes(), f


def test_two_ranks_of_config4_rank_geometry_equal_one_gpu(tmp_path):
    """BASELINE config 4's RANK geometry -- 512 x 512 x (64 owned + 16 ghost) planes, what each of 8 ranks holds of the 512^3
    grid: rows of two waves, the plane-range Jacobi launches with the ends-first schedule, ghost exchanges of 2 MB planes -- as
    two such ranks on the stream-ordered RCCL stand-in against one GPU at 512 x 512 x 128: the rising-smoke scene bench.py
    runs, 200 Jacobi iterations, every field bit-identical after each of three steps."""
    from build_fake_rccl import build
    fake = build("async")
    ref = str(tmp_path / "ref")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BQ_FAKE_RCCL_DELAY_MB"):
        env.pop(k, None)
    worker = os.path.join(HERE, "slab_deviation_worker.py")
    common = ["--grid", "512", "512", "128", "--steps", "3", "--iters", "200", "--checkpoints", "1", "2", "3"]
    r = subprocess.run([sys.executable, worker, "--make-reference", ref, *common], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    js = str(tmp_path / "dev.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), worker,
                        "--reference", ref, *common, "--rms-tol", "1e-5", "--json", js], cwd=ROOT,
                       env=dict(env, BQ_RCCL_LIBRARY=fake, BQ_FAKE_RCCL_SLOT_MB="24"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:]
    out = json.load(open(js))
    assert out["grid"] == [512, 512, 128] and out["ranks"] == 2 and "RCCL branch" in out["transport"]
    assert out["keep_dmc_border"] == 0
    assert out["worst_rms"] == 0.0, out["checkpoints"][-1]


def test_full_grid_on_one_rank_is_refused():
    """1024 x 1024 x 512 is 2.0 GiB per field: the operators address fields through 32-bit buffer descriptors, so one rank
    refuses the grid BEFORE allocating anything and says how many z-slab ranks it takes; two ranks' worth of planes pass
    the same check (only the check: nothing of that size is allocated here)."""
    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd import _lib
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    lib = bq.hip_lib()
    lib.fl_clear_error()
    with pytest.raises(_lib.BimocqError) as e:
        BimocqGPUSolver(1024, 1024, 512, 1.0, 0.0, 1.0)
    msg = str(e.value)
    assert "2 GiB" in msg and "at least 2 z-slab ranks" in msg, msg
    lib.fl_clear_error()
    # the operator ABI itself latches the same limit when called directly with such dims
    d = lib.fl_malloc(64)
    lib.gpu_solve_forward(d, d, d, d, d, d, 1.0 / 1024, 1024, 1024, 512, 0.001, 0.001)
    assert lib.fl_last_error() == _lib.FL_ERR_BAD_ARGUMENT and b"2 GiB" in lib.fl_last_error_string()
    lib.fl_clear_error()
    lib.fl_free(d)
