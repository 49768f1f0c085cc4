"""The library's RCCL code path with MORE THAN ONE RANK on a one-GPU box.

Real RCCL refuses two ranks on one device, so the neighbour exchange (fl_halo_exchange), the wall-sheet messages between
arbitrary ranks (fl_p2p_exchange) and the in-stream all-reduces of csrc/bq_halo.hip could otherwise only be run with a
single rank (fl_comm_selftest).  tests/fake_rccl is a stand-in for librccl.so over POSIX shared memory (it checks that
every receive meets a send of the same size from the same peer in issue order, and times out loudly on a deadlock);
BQ_RCCL_LIBRARY makes the library load it.  What these tests establish: the bootstrap (unique id over gloo ->
fl_comm_init on every rank), the grouped send/receive protocol of 2-, 3- and 4-rank runs, and `bench.py --gpus N` on
the RCCL branch -- all bit-identical to the single-domain oracle / the single-GPU run.  What they cannot: xGMI, real
asynchrony (the stand-in synchronises the stream it is given)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module", params=["sync", "async", "async-delayed"])
def fake(request):
    """sync: blocking copies through host memory (checks matching, sizes, deadlocks); async: stream-ordered transfers
    into device mailboxes shared over hipIpc (a missing stream dependency on the caller's side shows as a mismatch);
    async-delayed: the same with a 256 MB fill in front of every transfer, so that it starts ~100 us after the call"""
    from build_fake_rccl import build
    import oracle_lib
    oracle_lib.build()
    os.environ.pop("BQ_FAKE_RCCL_DELAY_MB", None)
    if request.param == "async-delayed":
        os.environ["BQ_FAKE_RCCL_DELAY_MB"] = "256"
    yield build("sync" if request.param == "sync" else "async")
    os.environ.pop("BQ_FAKE_RCCL_DELAY_MB", None)


def skip_plain_async(fake):
    """the three heaviest tests run on the blocking stand-in and on the DELAYED stream-ordered one (the stronger of the two
    stream-ordered forms: a transfer that starts ~100 us after the call exposes a missing stream dependency), not on both"""
    if "async" in os.path.basename(fake) and not os.environ.get("BQ_FAKE_RCCL_DELAY_MB"):
        pytest.skip("covered by the delayed stream-ordered stand-in")


def launch_worker(fake, nproc, *args, timeout=900):
    env = dict(os.environ, OMP_NUM_THREADS="4", MASTER_ADDR="127.0.0.1", BQ_RCCL_LIBRARY=fake)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gpu", "--transport", "rccl", *map(str, args)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("[rank") or "fake_rccl" in l]
    return r.returncode, "\n".join(lines[-30:]) or r.stdout[-3000:]


def test_two_ranks_over_the_rccl_branch(fake):
    """default mode (zeroed DMC border -> wall sheets travel through fl_p2p_exchange), overlapped exchanges"""
    rc, out = launch_worker(fake, 2, "--steps", 4)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    # round 4: two communicators (exchanges / scalar all-reduces), every step's calls cross-checked, a falsified ledger caught
    assert out.count("communicators=2 clean rc=0 falsified rc=5") == 2, out


def test_single_communicator_fallback(fake):
    """BQ_SINGLE_COMM=1: the one-communicator set-up of rounds 1-3 (what a RCCL without ncclCommSplit gets) still works"""
    os.environ["BQ_SINGLE_COMM"] = "1"
    try:
        rc, out = launch_worker(fake, 2, "--steps", 2)
    finally:
        os.environ.pop("BQ_SINGLE_COMM", None)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2 and out.count("communicators=1 clean rc=0 falsified rc=5") == 2, out


def test_two_ranks_on_a_spacing_that_is_not_a_power_of_two(fake):
    """L = 1 on 24 cells: h = 1/24 -- the tabled structured look-up (round 4) with its z tables indexed by GLOBAL plane"""
    rc, out = launch_worker(fake, 2, "--L", 1.0, "--steps", 3)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2


def test_blend_below_one_with_whole_grid_previous_fields_over_the_rccl_branch(fake):
    """blend 0.6 in the reference-faithful mode (zeroed map border, no BQ_OPT_KEEP_DMC_BORDER): at every re-initialisation the
    ranks assemble whole-grid copies of the *Prev fields with one point-to-point group (fl_p2p_exchange) and the two-level
    advection samples those (gpu_advect_*_double_global); sources blowing at the walls inside the upper rank's slab; 2 and 3
    ranks, bit-identical to the single-domain oracle"""
    if os.environ.get("BQ_FAKE_RCCL_DELAY_MB"):
        pytest.skip("the delayed transport adds nothing here: the gather is one blocking group per re-initialisation")
    env = dict(os.environ)
    os.environ["SLAB_TEST_BLEND"] = "0.6"
    try:
        rc, out = launch_worker(fake, 2, "--steps", 5, "--scene", "wall", "--expect-whole-grid-prev", 1)
        assert rc == 0, out
        assert out.count("mismatches=0") == 2 and out.count("whole-grid copies") == 2
        rc, out = launch_worker(fake, 3, "--dims", 24, 20, 36, "--ghost", 6, "--steps", 4, "--iters", 16, "--dt-cells", 1.0,
                                "--scene", "wall", "--expect-whole-grid-prev", 1)
        assert rc == 0, out
        assert out.count("mismatches=0") == 3
    finally:
        os.environ.clear(); os.environ.update(env)


def test_three_ranks_over_the_rccl_branch(fake):
    """a middle rank with two neighbours; wall sheets between non-neighbours (rank 2 needs planes of rank 0)"""
    rc, out = launch_worker(fake, 3, "--dims", 24, 20, 36, "--ghost", 6, "--steps", 3, "--iters", 16, "--dt-cells", 1.0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_four_ranks_without_overlap_and_with_viscosity(fake):
    rc, out = launch_worker(fake, 4, "--dims", 24, 20, 48, "--ghost", 6, "--steps", 3, "--iters", 12, "--dt-cells", 1.0,
                            "--overlap", 0, "--viscosity", 2e-3)
    assert rc == 0, out
    assert out.count("mismatches=0") == 4


def test_tall_slabs_ends_first_chunks_over_the_rccl_branch(fake):
    """slabs tall enough for the ends-first pressure chunks (the exchange for the next chunk starts while two pair interiors
    of this one are still to run): on the stream-ordered stand-ins a sweep that touched planes in flight would show"""
    skip_plain_async(fake)
    rc, out = launch_worker(fake, 2, "--dims", 32, 32, 96, "--L", 1.0, "--ghost", 8, "--steps", 3, "--iters", 60)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch_worker(fake, 3, "--dims", 24, 20, 96, "--ghost", 6, "--steps", 2, "--iters", 40, "--dt-cells", 1.0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3
    # G = 8 runs the chunk's last six sweeps as two ends-first triples (default, above); the pair schedule stays tested, and
    # rows of two float4 segments per lane (jacobi_lds2seg_kernel on plane ranges)
    rc, out = launch_worker(fake, 2, "--dims", 32, 32, 96, "--L", 1.0, "--ghost", 8, "--steps", 2, "--iters", 60, "--triples", 0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch_worker(fake, 2, "--dims", 264, 16, 96, "--ghost", 8, "--steps", 2, "--iters", 44, "--dt-cells", 1.0)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2


def test_reserved_cus_change_no_value_over_the_rccl_branch(fake):
    """FL_OPT_RESERVE_CUS = 8 / 16: the compute stream is recreated with a CU mask (248 / 240 of the 256 compute units, the
    Jacobi launchers size their grids for them) so that RCCL's send / recv kernels find free CUs; tall slabs, so that the
    ends-first pressure chunks and the fused plane-range launches run under the mask.  Every field equals the oracle's."""
    skip_plain_async(fake)
    rc, out = launch_worker(fake, 2, "--dims", 32, 32, 96, "--L", 1.0, "--ghost", 8, "--steps", 3, "--iters", 60, "--reserve-cus", 8)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2 and "comm profile" in out
    rc, out = launch_worker(fake, 3, "--dims", 24, 20, 96, "--ghost", 6, "--steps", 2, "--iters", 40, "--dt-cells", 1.0, "--reserve-cus", 16)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


def test_multigrid_projection_on_slabs_over_the_rccl_branch(fake):
    """the fp64 multigrid-CG projection on z-slab ranks (replicated solve: the global velocity assembled with one
    point-to-point group, csrc/host/fluid_solver.cpp projectionMgcgSlabs): BiMocq on 2 ranks, the reference binary's default
    configuration (MAC_REFLECTION + multigrid-CG) on 3 -- every field equals the single-domain oracle's"""
    rc, out = launch_worker(fake, 2, "--steps", 3, "--iters", 4, "--projection-kind", 1)
    assert rc == 0, out
    assert out.count("mismatches=0") == 2
    rc, out = launch_worker(fake, 3, "--dims", 24, 20, 36, "--ghost", 6, "--steps", 2, "--iters", 3, "--dt-cells", 1.0,
                            "--projection-kind", 1, "--scheme", 3)
    assert rc == 0, out
    assert out.count("mismatches=0") == 3


@pytest.mark.parametrize("nranks,dims", [(2, (32, 32, 96)), (3, (32, 32, 96)), (4, (32, 16, 128))])
def test_multigrid_levels_shared_between_the_ranks(fake, nranks, dims):
    """Round 4: the fp64 multigrid-CG projection with the grid's fine levels SHARED between the slab ranks
    (gpu_multi_grid_conjugate_gradient_slab, csrc/bq_mgcg_slab.hip.inc): owned + ghost planes per level, the single-GPU
    launchers on plane ranges, ghost planes refreshed every 8 sweeps, block dot products all-gathered, thin levels gathered
    and solved replicated.  Planes of 1024 / 512 cells (multiples of the 256-cell dot blocks), 48 / 32 planes per rank, three
    shared levels.  Every field equals the single-domain oracle's bit for bit, BiMocq and the reference binary's default
    (MAC_REFLECTION + multigrid-CG); with the option off the replicated solve runs and gives the same."""
    skip_plain_async(fake)
    common = ["--dims", *dims, "--L", 1.0, "--ghost", 8, "--steps", 2, "--iters", 3, "--dt-cells", 1.0, "--projection-kind", 1]
    rc, out = launch_worker(fake, nranks, *common, "--expect-shared", 1)
    assert rc == 0, out
    assert out.count("mismatches=0") == nranks and out.count("levels SHARED between the ranks") == nranks, out
    if nranks == 2:
        rc, out = launch_worker(fake, nranks, *common, "--scheme", 3, "--expect-shared", 1)
        assert rc == 0, out
        assert out.count("mismatches=0") == nranks
        rc, out = launch_worker(fake, nranks, *common, "--mgcg-shared", 0, "--expect-shared", 0)
        assert rc == 0, out
        assert out.count("mismatches=0") == nranks and out.count("replicated solve") == nranks, out


@pytest.mark.parametrize("nranks", [2, 4])
def test_slab_ranks_reproduce_the_mgcg_128_hashes(fake, nranks):
    """BASELINE-size evidence for the same: 2 and 4 z-slab ranks of the 128^3 rising-smoke run with the multigrid-CG projection
    (50 outer iterations, 6 levels), their owned planes stitched on rank 0, hash like the CPU oracle's global fields
    (tests/golden/next_row_hashes.json: mgcg_128 -- no oracle in the loop)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", BQ_RCCL_LIBRARY=fake)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "tests", "slab_deviation_worker.py"), "--hash-case", "mgcg_128", "--rms-tol", "0"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("[slab-hash] step") >= 2 and "MISMATCH" not in r.stdout


def test_bench_mode_at_128_over_the_rccl_branch(fake, tmp_path):
    """what `bench.py --gpus N` runs (library defaults, 200 Jacobi iterations, G = 8, wall sheets), 128^3, 12 steps, two
    ranks on the RCCL branch against the single-GPU run: RMS of rho, u, v, w exactly 0"""
    ref = str(tmp_path / "ref")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    worker = os.path.join(ROOT, "tests", "slab_deviation_worker.py")
    common = ["--size", "128", "--steps", "12", "--iters", "200", "--checkpoints", "1", "6", "12"]
    r = subprocess.run([sys.executable, worker, "--make-reference", ref, *common], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    js = str(tmp_path / "dev.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), worker,
                        "--reference", ref, *common, "--rms-tol", "1e-5", "--json", js], cwd=ROOT,
                       env=dict(env, BQ_RCCL_LIBRARY=fake), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    out = json.load(open(js))
    assert "RCCL branch" in out["transport"] and out["keep_dmc_border"] == 0
    assert out["worst_rms"] == 0.0, out["checkpoints"][-1]


def test_bench_gpus_2_runs_the_rccl_branch(fake, tmp_path):
    """`python bench.py --gpus 2` (self-launched ranks, default --transport rccl): the line says RCCL saw two ranks, and
    the per-slab dumps of the run stitch to the single-GPU dump byte for byte"""
    import numpy as np
    from gpufluidsimulation_amd.solver import read_density_dump
    env = dict(os.environ, OMP_NUM_THREADS="4", BQ_RCCL_LIBRARY=fake)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    common = ["--size", "64", "--steps", "3", "--warmup", "0", "--jacobi-iters", "40", "--no-cpu-baseline", "--no-extra"]

    def run(*args):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0]), r.stderr

    a, _ = run(*common, "--dump", one)
    b, err = run("--gpus", "2", *common, "--dump", two)
    assert "falling back" not in err, err[-2000:]
    assert b["n_gpus"] == 2 and b["config"]["comm_size"] == 2 and "rccl" in b["config"]["parallelism"].lower()
    assert "FALLBACK" not in b["config"]["parallelism"]
    # the line diagnoses itself: exposed communication and per-phase times of every rank, the knobs as extra legs
    d = b["diagnostics"]
    assert d["steps"] >= 1 and d["comm_exposed_ms_per_step"] >= 0.0 and len(d["per_rank"]) == 2
    for r in d["per_rank"]:
        assert r["comm_waits_per_step"] > 0 and r["ghost_MB_sent_per_step"] > 0
        assert set(r["phase_ms_per_step"]) == {"maps", "advect_compensate", "forces", "projection", "accumulate_reinit"}
        assert r["phase_ms_per_step"]["projection"] > 0 and sum(r["phase_ms_per_step"].values()) <= 1.05 * r["ms_per_step"]
    assert {"shallow_exchange_2", "ends_first_off", "jacobi_triples_off", "reserve_cus_8", "reserve_cus_16"} <= set(b["extra"])
    assert all(b["extra"][k]["value"] > 0 for k in ("shallow_exchange_2", "ends_first_off", "reserve_cus_8", "reserve_cus_16"))
    assert b["config"]["rccl_version"] is None or b["config"]["rccl_version"] > 0       # (the stand-in exports no ncclGetVersion)
    for f in sorted(os.listdir(one)):
        _, rec = read_density_dump(os.path.join(one, f))
        parts = sorted(p for p in os.listdir(two) if p.startswith(f[:-4] + ".k"))
        assert len(parts) == 2
        stitched = np.concatenate([read_density_dump(os.path.join(two, p))[1] for p in parts])
        assert len(rec) > 100 and stitched.tobytes() == rec.tobytes(), f


def test_bench_watchdog_hands_out_the_headline_of_a_run_that_hangs_afterwards(fake):
    """Real RCCL has no deadlock detection: a collective that never returns would leave `bench.py --gpus N` waiting for the
    launcher's kill.  N > 1 runs carry a watchdog (BENCH_WATCHDOG_S): when it fires every rank prints its Python stacks, and if
    the timed region had finished on every rank, rank 0 prints the headline's contract fields and the run exits with 0.
    BENCH_HANG_AFTER_TIMED=1 parks the ranks right after the timed region."""
    if "async" in os.path.basename(fake):
        pytest.skip("one transport is enough")
    env = dict(os.environ, OMP_NUM_THREADS="4", BQ_RCCL_LIBRARY=fake, BENCH_HANG_AFTER_TIMED="1", BENCH_WATCHDOG_S="300")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--steps", "3", "--warmup", "1",
                        "--jacobi-iters", "40", "--no-cpu-baseline", "--no-extra"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 3 and "WATCHDOG LINE" in d["config"]["note"]
    assert "WATCHDOG" in r.stderr and "timed region done" in r.stderr and "bench.py\", line" in r.stderr      # the stacks name the place


def test_bench_gpus_4_on_the_rccl_branch(fake):
    """four self-launched ranks (interior ranks with two neighbours, wall sheets between all pairs), strong scaling of one
    64^3 grid: rc 0, one line, four ranks in the communicator"""
    env = dict(os.environ, OMP_NUM_THREADS="2", BQ_RCCL_LIBRARY=fake)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--size", "64", "--steps", "3", "--warmup", "1",
                        "--jacobi-iters", "40", "--no-cpu-baseline", "--no-extra"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert "falling back" not in r.stderr
    assert line["n_gpus"] == 4 and line["config"]["comm_size"] == 4 and line["config"]["grid_per_gpu"] == [64, 64, 16]
    assert line["scaling"] == "strong" and line["value"] > 0
