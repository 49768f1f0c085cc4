"""One z-slab rank of the multi-rank parity test (launched by torch.distributed.run, gloo).

    --backend cpu : the C++ host solver on the test-only CPU stand-in of the C-ABI (oracle kernels)
    --backend gpu : the product (HIP kernels); all ranks share GPU 0, ghost planes travel through the
                    host-staged transport (gpufluidsimulation_amd/transport.py) instead of RCCL

Every rank also runs the single-domain CPU oracle on the GLOBAL grid and checks, after every step,
that the planes it owns are bit-identical to the oracle's.  Exit code 0 = parity on this rank.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

BLEND = float(os.environ.get("SLAB_TEST_BLEND", "0.8"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["cpu", "gpu"], required=True)
    ap.add_argument("--dims", type=int, nargs=3, default=[24, 20, 32])
    ap.add_argument("--L", type=float, default=0.75)
    ap.add_argument("--ghost", type=int, default=6)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dt-cells", type=float, default=1.5)
    ap.add_argument("--keep-dmc-border", type=int, default=0,
                    help="0 (the library's default): the reference's zeroed map border -- slab ranks fetch the wall sheets "
                         "(csrc/host/wall_sheets.hpp); 1: border kept, no far reads.  Bit-exact against the oracle in the same mode either way")
    ap.add_argument("--rms-tol", type=float, default=0.0, help="> 0: compare by RMS instead of bit for bit (debug)")
    ap.add_argument("--viscosity", type=float, default=0.0)
    ap.add_argument("--overlap", type=int, default=1, help="BQ_OPT_OVERLAP_EXCHANGES")
    ap.add_argument("--shallow", type=int, default=0, help="BQ_OPT_SHALLOW_BLOCKING_EXCHANGE")
    ap.add_argument("--ends-first", type=int, default=1, help="BQ_OPT_JACOBI_ENDS_FIRST")
    ap.add_argument("--triples", type=int, default=1, help="BQ_OPT_JACOBI_TRIPLES")
    ap.add_argument("--reserve-cus", type=int, default=0, help="FL_OPT_RESERVE_CUS (gpu backend): CU-masked compute stream")
    ap.add_argument("--policy", type=int, default=0, help="BQ_OPT_REINIT_POLICY: 1 = distortion-driven re-initialisation (maps live for several steps)")
    ap.add_argument("--travel-limit", type=int, default=0, help="BQ_OPT_REINIT_MAX_TRAVEL / oracle option 4 (policy 1): 0 = the rank's ghost depth")
    ap.add_argument("--projection-kind", type=int, default=0, help="0 Jacobi (--iters sweeps), 1 fp64 multigrid-CG (--iters outer iterations; "
                    "replicated solve on slab ranks)")
    ap.add_argument("--scheme", type=int, default=0, help="0: BiMocq, 3: MAC_REFLECTION (BQ_SCHEME_*)")
    ap.add_argument("--mgcg-shared", type=int, default=1, help="BQ_OPT_MGCG_SHARED: 1 = the multigrid levels shared between the ranks "
                    "where the decomposition allows it, 0 = the replicated solve")
    ap.add_argument("--expect-shared", type=int, default=-1, help="1 / 0: fail unless the multigrid projection did / did not take the shared path")
    ap.add_argument("--whole-grid-prev", type=int, default=1, help="BQ_OPT_WHOLE_GRID_PREV: whole-grid copies of the *Prev fields for blend != 1")
    ap.add_argument("--expect-whole-grid-prev", type=int, default=-1, help="1 / 0: fail unless the copies were / were not in use at the end")
    ap.add_argument("--scene", choices=["default", "wall"], default="default",
                    help="wall: velocity sources blowing at the x-high and y-high walls inside the upper ranks' slabs, so that with blend < 1 "
                         "the second look-up of the two-level advection lands in the zeroed border cells of the previous backward map")
    ap.add_argument("--transport", choices=["host", "rccl"], default="host",
                    help="rccl (gpu backend): the library's own RCCL code path (fl_comm_init + ncclSend/ncclRecv); with several "
                         "ranks on one GPU that needs BQ_RCCL_LIBRARY = the tests' stand-in (tests/fake_rccl)")
    a = ap.parse_args()

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.set_num_threads(1)

    import fields as F
    from gpufluidsimulation_amd import solver, transport
    from oracle_lib import OracleSolver

    if a.backend == "cpu":
        from build_cpu_host import SO
        lib = solver.bind_host(C.CDLL(SO, mode=C.RTLD_LOCAL))
        for name, res, args in (("fl_last_error", C.c_int, []), ("fl_last_error_string", C.c_char_p, []),
                                ("fl_clear_error", None, []),
                                ("fl_memcpy_d2h", None, [C.c_void_p, C.c_void_p, C.c_size_t]),
                                ("fl_memcpy_h2d", None, [C.c_void_p, C.c_void_p, C.c_size_t])):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        hostlib, abilib = lib, lib
    else:
        import gpufluidsimulation_amd as bq
        abilib = bq.hip_lib()
        hostlib = solver.host_lib()
        assert abilib.fl_init(0) == 0
        if a.reserve_cus:
            abilib.fl_set_option(bq._lib.FL_OPT_RESERVE_CUS, a.reserve_cus)
            assert abilib.fl_get_option(bq._lib.FL_OPT_RESERVE_CUS) == a.reserve_cus and abilib.fl_last_error() == 0
        abilib.fl_set_option(bq._lib.FL_OPT_PROFILE_COMM, 1)
    if a.transport == "rccl":
        assert a.backend == "gpu"
        assert abilib.fl_comm_selftest() == 0, abilib.fl_last_error_string()
        transport.init_rccl(abilib, dist)
        assert abilib.fl_comm_size() == world and abilib.fl_comm_rank() == rank
        # round 4: the in-stream scalar all-reduces have a communicator of their own (ncclCommSplit), and every step's
        # communicator calls are checked across the ranks (FL_OPT_COMM_CHECK: a mismatch latches FL_ERR_COMM -> s._check())
        want_comms = 1 if (world == 1 or os.environ.get("BQ_SINGLE_COMM", "0") not in ("", "0")) else 2
        assert abilib.fl_comm_count() == (0 if world == 1 else want_comms), abilib.fl_comm_count()
        abilib.fl_set_option(bq._lib.FL_OPT_COMM_CHECK, 1)

        class _Stats:                               # the RCCL path keeps no Python-side counters
            exchanges, planes_moved, p2p_messages, p2p_floats, trace = -1, -1, -1, -1, None
        tr = _Stats()
    else:
        tr = transport.HostStagedTransport(abilib, dist)

    ni, nj, nk = a.dims
    h = a.L / ni
    # two sources: one straddling the slab boundary (z in the middle of the global grid), one inside rank 0
    zmid = 0.5 * nk * h
    em = [(0.5 * ni * h, 0.3 * nj * h, zmid + 0.3 * h, 0.16 * ni * h, 1.0, 2.0, 0.0, 2),
          (0.4 * ni * h, 0.35 * nj * h, 0.22 * nk * h, 0.12 * ni * h, 0.7, 1.0, 0.0, 1)]
    if a.scene == "wall":
        zup = (nk - 0.3 * nk / world) * h            # inside the last rank's slab
        em = [((ni - 5.2) * h, 0.5 * nj * h + 0.37 * h, zup + 0.29 * h, 0.14 * ni * h, 1.0, 2.0, 1.0, 3),
              (0.45 * ni * h, (nj - 4.6) * h, zmid + 0.31 * h, 0.14 * ni * h, 0.8, 1.0, 1.0, 2)]
    s = solver.BimocqGPUSolver(ni, nj, nk, a.L, a.viscosity, BLEND, lib=hostlib, errlib=abilib, rank=rank, nranks=world, ghost=a.ghost,
                               scheme=a.scheme)
    s.setSmoke(0.05, 1.0, em)
    s.setProjection(a.iters, 0.5, a.projection_kind)
    limit = a.travel_limit or a.ghost
    if a.policy:
        s.setOption(2, a.policy)
        s.setOption(9, limit)
        assert s.getOption(9) == limit
    s.setOption(1, a.keep_dmc_border)
    s.setOption(5, a.overlap)
    s.setOption(6, a.shallow)
    s.setOption(7, a.ends_first)
    s.setOption(10, a.triples)
    s.setOption(11, a.mgcg_shared)
    s.setOption(13, a.whole_grid_prev)
    if a.backend == "cpu":
        # the oracle library inside the CPU stand-in carries the slab context; the reference run below
        # uses the separately loaded liboracle.so, which stays single-domain
        pass
    o = OracleSolver(ni, nj, nk, a.L, a.viscosity, BLEND)
    o.set_smoke(0.05, 1.0, em)
    o.set_projection(a.iters, 0.5, a.projection_kind)
    o.set_option(1, a.keep_dmc_border)
    if a.policy:
        o.set_option(2, a.policy)
        o.set_option(4, limit)
    if a.scheme:
        o.set_option(3, a.scheme)
    dt = a.dt_cells * h
    names = ["rho", "T", "div", "p", "u", "v", "w", "uinit", "vinit", "winit", "rhoinit", "Tinit"]
    if a.projection_kind:
        names = [n_ for n_ in names if n_ not in ("div", "p")]      # the multigrid solver keeps its own fp64 div / p
    if a.scheme:
        names = [n_ for n_ in ("rho", "T", "div", "p", "u", "v", "w") if n_ in names]        # the reflection scheme keeps no map state
    plane = {"u": (ni + 1) * nj, "uinit": (ni + 1) * nj, "v": ni * (nj + 1), "vinit": ni * (nj + 1)}
    bad = 0
    if os.environ.get("SLAB_TEST_TRACE") == "1":
        tr.trace = []
    for f in range(a.steps):
        if tr.trace is not None:
            tr.trace.append(("step", f, 0))
        o.advance(f, dt)
        s.advance(f, dt)
        s._check()
        if s.cfldt != o.cfldt:
            print(f"[rank {rank}] step {f}: cfldt {s.cfldt} != {o.cfldt}", flush=True)
            bad += 1
        for name in names:
            pe = plane.get(name, ni * nj)
            ref = o.field(name)
            mine = s.owned(name)
            want = ref[pe * s.own0: pe * s.own0 + mine.size]
            if a.rms_tol > 0:           # (debug: compare by RMS instead of bit for bit)
                rms = float(np.sqrt(np.mean((want.astype(np.float64) - mine.astype(np.float64)) ** 2)))
                if not (rms <= a.rms_tol):
                    print(f"[rank {rank}] step {f}: {name} RMS {rms:.3e} > {a.rms_tol}", flush=True)
                    bad += 1
                continue
            if not F.same(want, mine):
                d = np.abs(want.astype(np.float64) - mine.astype(np.float64))
                planes = sorted(set((np.nonzero(d)[0] // pe + s.own0).tolist()))
                print(f"[rank {rank}] step {f}: {name} differs, max|diff| {d.max():.3e} in global planes {planes[:12]}{'...' if len(planes) > 12 else ''}", flush=True)
                bad += 1
    if a.policy:
        mine = s.reinitCounts() + (s.forcedReinits(),) + s.lastDistortion()
        want = o.reinit_counts() + (o.l.orc_solver_reinit_counts(o.s, 2),) + o.last_distortion()
        print(f"[rank {rank}] policy {a.policy}, travel limit {limit}: re-initialisations (velocity, scalar, forced) + distortions {mine}, oracle {want}", flush=True)
        if mine != want or not (0 < mine[0] < a.steps):
            bad += 1
    if a.backend == "gpu":
        # FL_OPT_PROFILE_COMM was on: every wait of the compute stream on the halo stream was timed
        ms2, n2 = (C.c_double * 2)(), (C.c_longlong * 2)()
        abilib.fl_comm_profile(ms2, n2, 1)
        print(f"[rank {rank}] comm profile: {n2[0]} waits {ms2[0]:.3f} ms exposed, {n2[1]} in-stream all-reduces {ms2[1]:.3f} ms", flush=True)
        if world > 1 and not (n2[0] > 0 and ms2[0] >= 0.0):
            bad += 1
    if a.transport == "rccl" and world > 1:
        # the check itself: clean ledgers pass, a falsified one is seen by EVERY rank (the comparison is an all-reduce)
        rc_clean = abilib.fl_comm_check(0)
        rc_bad = abilib.fl_comm_check(1 if rank == world - 1 else 0)
        text = abilib.fl_last_error_string().decode(errors="replace")
        abilib.fl_clear_error()
        print(f"[rank {rank}] comm check: communicators={abilib.fl_comm_count()} clean rc={rc_clean} falsified rc={rc_bad} ({text[:60]})", flush=True)
        if rc_clean != 0 or rc_bad != bq._lib.FL_ERR_COMM:
            bad += 1
    if a.projection_kind and a.backend == "gpu":
        took = s.getOption(11) == 2
        print(f"[rank {rank}] multigrid projection on slabs: {'levels SHARED between the ranks' if took else 'replicated solve'}", flush=True)
        if a.expect_shared >= 0 and took != bool(a.expect_shared):
            bad += 1
    if a.expect_whole_grid_prev >= 0:
        in_use = s.getOption(13) == 2
        print(f"[rank {rank}] two-level advection: {'whole-grid copies of the *Prev fields' if in_use else 'local *Prev fields'}", flush=True)
        if in_use != bool(a.expect_whole_grid_prev):
            bad += 1
    moved = np.abs(o.field("v")).max()
    print(f"[rank {rank}/{world}] backend={a.backend} steps={a.steps} exchanges={tr.exchanges} planes={tr.planes_moved} "
          f"p2p={tr.p2p_messages}msgs/{tr.p2p_floats}floats max|v|={moved:.4f} mismatches={bad}", flush=True)
    if tr.trace is not None and rank == 0:
        print(f"[rank 0] exchange trace (fields, depth, bytes per neighbour): {[t for t in tr.trace if not (t[0] == 1 and t[1] == a.ghost)]}", flush=True)
    ok = torch.tensor([bad])
    dist.all_reduce(ok)
    s.close()
    dist.destroy_process_group()
    sys.exit(0 if int(ok.item()) == 0 and moved > 0.01 and (tr.exchanges != 0 or world == 1) else 1)


if __name__ == "__main__":
    main()
