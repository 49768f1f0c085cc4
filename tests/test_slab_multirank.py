"""Multi-rank (z-slab) path: one process per rank under torch.distributed.run with the gloo backend,
127.0.0.1 rendezvous.  Each rank runs the C++ host solver in slab mode, exchanges ghost planes through
the transport hook, and compares the planes it owns with the single-domain CPU oracle after every step
(tests/slab_worker.py).

CPU tests (no GPU): the host solver on the test-only CPU stand-in of the C-ABI -- they validate the
slab bookkeeping (ghost validity tracking, exchange plane ranges, chunked Jacobi, owned-plane
reductions), which is backend independent.
GPU test: the same with the HIP kernels, all ranks sharing GPU 0 (host-staged transport instead of
RCCL, which needs one GPU per rank and is exercised by the driver's multi-GPU bench).
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(nproc, *args, threads=2, timeout=900):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "slab_worker.py"), *map(str, args)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("[rank")]
    return r.returncode, "\n".join(lines[-30:]) or r.stdout[-3000:]


@pytest.fixture(scope="module", autouse=True)
def built_cpu_host():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from build_cpu_host import build
    build()
    import oracle_lib
    oracle_lib.build()


def test_two_ranks_bit_exact_cpu():
    """2 slabs of 16 planes, 6 ghost planes, 30 Jacobi iterations (5 chunks), blend 0.8, two sources (one
    straddling the slab boundary): every owned plane equals the single-domain oracle bit for bit."""
    rc, out = launch(2, "--backend", "cpu", "--steps", 4)
    assert rc == 0, out
    assert "mismatches=0" in out and "exchanges=" in out


def test_two_ranks_without_exchange_overlap_cpu():
    """BQ_OPT_OVERLAP_EXCHANGES = 0 (exchange, then the whole operator) against the default, in which the map operators
    run on the planes out of reach of the ghost planes first and on the two ends after the exchange (the stand-in
    implements fl_set_plane_window and the fused housekeeping like the HIP library)"""
    rc, out = launch(2, "--backend", "cpu", "--steps", 3, "--overlap", 0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2


def test_three_ranks_middle_rank_cpu():
    """a middle rank has two neighbours; 36 planes -> 12 owned each, 6 ghost planes"""
    rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 36, "--ghost", 6, "--steps", 3, "--iters", 16, "--dt-cells", 1.0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_distortion_driven_reinit_on_slabs_cpu():
    """BQ_OPT_REINIT_POLICY = 1 on z-slab ranks (round 3): maps live for several steps, their z-travel is measured after every
    update (gpu_map_travel_z, all-reduced) and a map set is re-initialised before it would outgrow the ghost zone
    (BQ_OPT_REINIT_MAX_TRAVEL = G); the single-domain oracle applies the same rule (option 4), so fields, re-initialisation
    counts and distortions must agree bit for bit.  blend = 1: the two-level look-up of blend < 1 is exact on slabs only with
    the kept DMC border (second run)."""
    env = dict(os.environ)
    os.environ["SLAB_TEST_BLEND"] = "1.0"
    try:
        rc, out = launch(2, "--backend", "cpu", "--steps", 14, "--iters", 20, "--policy", 1, "--dt-cells", 1.0)
    finally:
        os.environ.clear(); os.environ.update(env)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2 and "policy 1" in out
    rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 36, "--ghost", 6, "--steps", 8, "--iters", 12, "--dt-cells", 0.8,
                     "--policy", 1, "--keep-dmc-border", 1)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_blend_below_one_reference_faithful_on_slabs_cpu():
    """blend 0.6 WITHOUT BQ_OPT_KEEP_DMC_BORDER: the second look-up of the two-level advection can
    land anywhere between the origin and the node once it meets the zeroed border cells of the previous backward map
    (tests/test_oracle_kat.py::test_two_level_advection_on_slab_ranks_needs_the_whole_previous_fields shows the operator
    needing it), so every rank assembles whole-grid copies of the *Prev fields at each re-initialisation
    (BQ_OPT_WHOLE_GRID_PREV, default on) and samples those.  Sources blowing at the x-high and y-high walls inside the upper
    ranks' slabs; 2 and 3 ranks; the copies are checked to be in use and every owned plane equals the single-domain oracle."""
    env = dict(os.environ)
    os.environ["SLAB_TEST_BLEND"] = "0.6"
    try:
        rc, out = launch(2, "--backend", "cpu", "--steps", 6, "--scene", "wall", "--expect-whole-grid-prev", 1)
        assert rc == 0, out
        assert out.count("mismatches=0") == 2 and out.count("whole-grid copies") == 2
        rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 36, "--ghost", 6, "--steps", 4, "--iters", 16, "--dt-cells", 1.0,
                         "--scene", "wall", "--expect-whole-grid-prev", 1)
        assert rc == 0, out
        assert out.count("mismatches=0") == 3
        # the local *Prev fields (option 0) stay available: exact while no look-up meets the border cells, as here
        rc, out = launch(2, "--backend", "cpu", "--steps", 3, "--scene", "wall", "--whole-grid-prev", 0, "--expect-whole-grid-prev", 0)
        assert rc == 0, out
        assert out.count("mismatches=0") == 2 and out.count("local *Prev fields") == 2
    finally:
        os.environ.clear(); os.environ.update(env)


def test_multigrid_projection_replicated_on_slabs_cpu():
    """the fp64 multigrid-CG projection (what the reference's binary ships) on z-slab ranks: every rank assembles the global
    velocity (one point-to-point message per peer and component), runs the single-domain solver on it and takes its planes
    back -- bit-identical to the single-domain oracle, with the BiMocq scheme (2 ranks) and the reflection scheme, the
    reference binary's default configuration (3 ranks: two projections per step)"""
    rc, out = launch(2, "--backend", "cpu", "--steps", 3, "--iters", 4, "--projection-kind", 1)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 36, "--ghost", 6, "--steps", 2, "--iters", 3, "--dt-cells", 1.0,
                     "--projection-kind", 1, "--scheme", 3)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_viscous_diffusion_on_slabs_cpu():
    """nu != 0: the 20 diffusion sweeps per component run in chunks of G with ghost refreshes in between
    (gpu_diffuse_sweeps), including the reference's buffer aliasing; bit-exact on 2 and 3 ranks"""
    rc, out = launch(2, "--backend", "cpu", "--steps", 3, "--viscosity", 2e-3)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 36, "--ghost", 6, "--steps", 2, "--iters", 16,
                     "--dt-cells", 1.0, "--viscosity", 2e-3)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_kept_border_mode_cpu():
    """BQ_OPT_KEEP_DMC_BORDER = 1 (the backward map keeps its border through the DMC update: no far reads, no wall
    sheets): still bit-exact against the oracle in the same mode"""
    rc, out = launch(2, "--backend", "cpu", "--steps", 4, "--keep-dmc-border", 1)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2 and "p2p=0msgs" in out


def test_wall_sheets_are_what_makes_the_default_exact_cpu():
    """The default IS the reference-faithful zeroed border: the wall layers of the compensation sample the
    error field at 1/4..3/4 of their position, far outside a slab.  Every other test of this file passes bit for bit in
    that mode because the ranks exchange those sheets point to point (csrc/host/wall_sheets.*); here the traffic is
    checked to exist, on 2 and on 3 ranks (where rank 2 also pulls from rank 0, not a z-neighbour)."""
    rc, out = launch(2, "--backend", "cpu", "--steps", 3)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2 and "p2p=0msgs" not in out.split("[rank 1/2]")[1]
    rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 36, "--ghost", 6, "--steps", 3, "--iters", 16, "--dt-cells", 1.0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_tall_slabs_run_the_ends_first_chunks_cpu():
    """slabs of at least 2 G + 8 owned planes: the last two fused pairs of every pressure chunk run ends first and the
    exchange for the next chunk starts before their interiors (fluid_solver.cpp: projection); G = 6 (three pairs per
    chunk) on two ranks, G = 8 (four pairs) on three"""
    rc, out = launch(2, "--backend", "cpu", "--dims", 24, 20, 64, "--ghost", 6, "--steps", 3, "--iters", 40)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 96, "--ghost", 8, "--steps", 2, "--iters", 36, "--dt-cells", 1.0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3
    rc, out = launch(2, "--backend", "cpu", "--dims", 24, 20, 64, "--ghost", 6, "--steps", 2, "--iters", 40, "--ends-first", 0)
    assert rc == 0, out                                  # BQ_OPT_JACOBI_ENDS_FIRST = 0: the plain chunk order stays tested
    assert out.count("mismatches=0") == 2
    # G = 8: the six sweeps after a chunk's overlapped pair run as two triples (BQ_OPT_JACOBI_TRIPLES, default) -- ends first
    # above; here with the ends-first order off, and with the triples off (the pair schedule stays tested)
    rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 96, "--ghost", 8, "--steps", 2, "--iters", 36, "--dt-cells", 1.0, "--ends-first", 0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3
    rc, out = launch(3, "--backend", "cpu", "--dims", 24, 20, 96, "--ghost", 8, "--steps", 2, "--iters", 36, "--dt-cells", 1.0, "--triples", 0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_shallow_blocking_exchanges_cpu():
    """BQ_OPT_SHALLOW_BLOCKING_EXCHANGE = 1: blocking refreshes move only the planes asked for; same values, fewer planes"""
    rc, deep = launch(2, "--backend", "cpu", "--steps", 3)
    assert rc == 0, deep
    import re
    planes = lambda text: int(re.search(r"planes=(\d+)", text).group(1))
    for mode in (1, 2):                                  # 2: the overlapped exchanges move only the operator's reach as well
        rc, out = launch(2, "--backend", "cpu", "--steps", 3, "--shallow", mode)
        assert rc == 0, out
        assert out.count("mismatches=0") == 2
        assert planes(out) < planes(deep), (mode, planes(out), planes(deep))


def test_reflection_scheme_on_slabs_cpu():
    """MAC_REFLECTION (the reference binary's default scheme) with the Jacobi projection on two and three
    z-slab ranks: MacCormack semilag pairs, the corrected limiter, two projections per step -- every owned plane equals
    the single-domain oracle bit for bit; also with viscosity (20 diffusion sweeps per component and half step)"""
    rc, out = launch(2, "--backend", "cpu", "--scheme", 3, "--steps", 3, "--iters", 16, "--dt-cells", 1.0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch(3, "--backend", "cpu", "--scheme", 3, "--dims", 24, 20, 36, "--steps", 2, "--iters", 12, "--dt-cells", 1.5,
                     "--viscosity", 2e-3)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_ghost_zone_too_shallow_is_refused_cpu():
    """a time step that moves data further than the ghost zone must fail loudly, not silently diverge"""
    rc, out = launch(2, "--backend", "cpu", "--steps", 4, "--ghost", 3, "--dt-cells", 2.0)
    assert rc != 0


@pytest.mark.gpu
def test_two_ranks_bit_exact_gpu():
    rc, out = launch(2, "--backend", "gpu", "--steps", 4, threads=4)
    assert rc == 0, out
    assert "mismatches=0" in out


@pytest.mark.gpu
def test_two_ranks_without_exchange_overlap_gpu():
    """BQ_OPT_OVERLAP_EXCHANGES = 0: exchange first, then the whole operator (the default splits every map operator
    that needs ghost planes into the part that cannot reach them, run during the exchange, and the two ends)"""
    rc, out = launch(2, "--backend", "gpu", "--steps", 4, "--overlap", 0, threads=4)
    assert rc == 0, out
    assert "mismatches=0" in out


@pytest.mark.gpu
def test_three_ranks_gpu():
    """a middle rank exchanges with two neighbours; 36 planes -> 12 owned each, 6 ghost planes: too thin for the
    operator split at the larger reaches, so both forms occur in one run"""
    rc, out = launch(3, "--backend", "gpu", "--dims", 24, 20, 36, "--ghost", 6, "--steps", 3, "--iters", 16, "--dt-cells", 1.0, threads=4)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


@pytest.mark.gpu
def test_tall_slabs_run_the_ends_first_chunks_gpu():
    rc, out = launch(2, "--backend", "gpu", "--dims", 32, 32, 96, "--L", 1.0, "--ghost", 8, "--steps", 3, "--iters", 60, threads=4)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch(2, "--backend", "gpu", "--dims", 32, 32, 96, "--L", 1.0, "--ghost", 8, "--steps", 2, "--iters", 60, "--triples", 0, threads=4)
    assert rc == 0, out                                  # the pair schedule (BQ_OPT_JACOBI_TRIPLES = 0)
    assert out.count("mismatches=0") == 2
    rc, out = launch(2, "--backend", "gpu", "--dims", 32, 32, 96, "--L", 1.0, "--ghost", 8, "--steps", 2, "--iters", 60, "--ends-first", 0, threads=4)
    assert rc == 0, out                                  # triples without the ends-first order
    assert out.count("mismatches=0") == 2
    rc, out = launch(3, "--backend", "gpu", "--dims", 24, 20, 96, "--ghost", 6, "--steps", 2, "--iters", 40, "--dt-cells", 1.0, threads=4)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


@pytest.mark.gpu
def test_shallow_blocking_exchanges_gpu():
    for mode in (1, 2):
        rc, out = launch(2, "--backend", "gpu", "--steps", 3, "--shallow", mode, threads=4)
        assert rc == 0, out
        assert out.count("mismatches=0") == 2


@pytest.mark.gpu
def test_reflection_scheme_on_slabs_gpu():
    rc, out = launch(2, "--backend", "gpu", "--scheme", 3, "--steps", 3, "--iters", 16, "--dt-cells", 1.0, threads=4)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch(3, "--backend", "gpu", "--scheme", 3, "--dims", 24, 20, 36, "--steps", 2, "--iters", 12, "--dt-cells", 1.5,
                     "--viscosity", 2e-3, threads=4)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


@pytest.mark.gpu
def test_two_ranks_viscous_gpu():
    rc, out = launch(2, "--backend", "gpu", "--steps", 3, "--viscosity", 2e-3, threads=4)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2


@pytest.mark.gpu
def test_two_ranks_pow2_grid_gpu():
    """32x32x64 with h = 1/32 (power-of-two fast path) and the tiled Jacobi kernels, 2 ranks"""
    rc, out = launch(2, "--backend", "gpu", "--dims", 32, 32, 64, "--L", 1.0, "--ghost", 6, "--steps", 3,
                     "--iters", 40, "--dt-cells", 1.5, threads=4)
    assert rc == 0, out


@pytest.mark.gpu
def test_bench_mode_slabs_equal_one_gpu_at_128(tmp_path):
    """Exactly what `bench.py --gpus N` runs (library defaults: zeroed DMC border as in the reference, full state, 200
    Jacobi iterations, G = 8, overlapped exchanges), 128^3 rising smoke, 24 steps, 2 ranks sharing the GPU through the
    host-staged transport, against the single-GPU run of the same library (itself bit-identical to the CPU oracle):
    RMS of rho, u, v, w over the whole grid must be <= 1e-5 (north star) -- and is 0, the wall sheets make it exact."""
    ref = str(tmp_path / "ref")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    worker = os.path.join(ROOT, "tests", "slab_deviation_worker.py")
    common = ["--size", "128", "--steps", "24", "--iters", "200", "--checkpoints", "1", "8", "16", "24"]
    r = subprocess.run([sys.executable, worker, "--make-reference", ref, *common], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    js = str(tmp_path / "dev.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), worker,
                        "--reference", ref, *common, "--rms-tol", "1e-5", "--json", js], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    import json
    out = json.load(open(js))
    assert out["keep_dmc_border"] == 0 and out["worst_rms"] <= 1e-5
    assert out["worst_rms"] == 0.0, out["checkpoints"][-1]
