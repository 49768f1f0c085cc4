#!/usr/bin/env python3
"""bench.py -- bimocq3D per-step throughput on MI355X.

Metric (BASELINE.json): Mvoxels/s per step of the 256^3 rising-smoke scene (L=1, h=1/N,
dt=2h, nu=0, blend=1, alpha=0, beta=1, one spherical source at step 0, Jacobi 200 iterations,
halfrdx=0.5 = the reference's value), plus the HBM roofline fraction of the dominant kernel (the
Jacobi sweep, 12 algorithmic bytes per voxel per sweep) and the CPU oracle timed beside it.

A "step" is one BimocqGPUSolver::advance(): map update, advection with error compensation, forces,
projection (divergence, Jacobi sweeps, gradient), accumulation, re-initialisation -- all resident in
HBM; nothing crosses PCIe inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--jacobi-iters 200]

N > 1: launched by torch.distributed.run, one rank per GPU (see DESIGN.md "Multi-GPU").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
JACOBI_BYTES_PER_VOXEL = 12.0   # read p + read div + write p'  


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=180)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", dest="n", type=int, default=256, help="grid is size^3 (per rank when --gpus > 1)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (BASELINE config 4: 512^3 across 8 GPUs): the global grid is size^3 and each of the N "
                         "ranks owns size/N planes; default is weak scaling, size^3 per GPU")
    ap.add_argument("--jacobi-iters", type=int, default=200)
    ap.add_argument("--halfrdx", type=float, default=0.5)
    ap.add_argument("--projection", choices=["jacobi", "mgcg"], default="jacobi",
                    help="jacobi: BASELINE's headline config; mgcg: the fp64 multigrid-CG projection the reference's "
                         "shipped binary runs (SURVEY 8f N1), --mg-iters outer iterations, single GPU")
    ap.add_argument("--mg-iters", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=128, help="grid of the bounded CPU sample")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--ghost", type=int, default=8, help="ghost planes per side of a z-slab rank (N > 1)")
    ap.add_argument("--emulate-slab", action="store_true",
                    help="debug, one GPU: run the z-slab code path of rank 0 of 2 (size^3 owned planes + ghost planes, chunked "
                         "Jacobi, split operators) with exchanges that move nothing -- the compute-side cost of the slab path; "
                         "the line is labelled and is not a benchmark result")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="N > 1: ghost planes over RCCL/xGMI (default) or staged through the host over gloo (debug)")
    return ap.parse_args()


def usable_cores():
    """CPU threads this job may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box
    hands a one-GPU job a share of the host's cores; more OpenMP threads than that only thrash)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = float(f.read().split()[0])
            if quota not in ("max", "-1"):
                cores = min(cores, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(cores, 64))


def cpu_baseline(args):
    """The CPU oracle (a port: the reference has no runnable CPU path for bimocq3D) on this box's
    host cores, on a bounded sample of the same scene: smaller grid, same algorithm and settings."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = int(os.environ.get("BENCH_CPU_THREADS", usable_cores()))
    os.environ["OMP_NUM_THREADS"] = str(cores)                  # read by libgomp when the oracle is loaded
    import oracle_lib
    oracle_lib.lib(march="native", out="_build_native")        # rebuilt for this host's ISA
    n = args.cpu_n
    s = oracle_lib.OracleSolver(n, n, n, 1.0, 0.0, 1.0)
    s.set_smoke(0.0, 1.0, [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)])
    mg = args.projection == "mgcg"
    s.set_projection(args.mg_iters if mg else args.jacobi_iters, args.halfrdx, 1 if mg else 0)
    dt = 2.0 / n
    s.advance(0, dt)                                            # untimed: first-touch + emission
    t0 = time.perf_counter()
    for f in range(1, 1 + args.cpu_steps):
        s.advance(f, dt)
    el = time.perf_counter() - t0
    s.close()
    return {"value": round(n ** 3 * args.cpu_steps / el / 1e6, 4), "unit": "Mvoxels/s", "cores": cores,
            "kind": "port",
            "sample": f"{args.cpu_steps} steps of {n}^3 rising smoke ("
                      + (f"fp64 multigrid-CG, {args.mg_iters} outer iterations" if mg else f"{args.jacobi_iters} Jacobi iters")
                      + f"), OpenMP C oracle (-O2 -march=native), {el:.1f} s"}


def pmc_traffic(n, sweeps_per_launch=1.0, kernel=""):
    """HBM bytes per Jacobi launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "jacobi_pmc_traffic.json")
    key = str(n) if sweeps_per_launch < 1.5 else (f"{n}_fused2r" if kernel == "jacobi_march2r_kernel" else f"{n}_fused2")
    try:
        with open(path) as f:
            return json.load(f).get(key, {}).get("bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
    force_fail = os.environ.get("BENCH_FORCE_RCCL_FAILURE") == "1"      # test hook for the fallback below
    if local_rank >= torch.cuda.device_count():
        if args.transport != "host" and not force_fail:
            sys.exit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
        local_rank %= torch.cuda.device_count()         # debug transport: several ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.transport == "rccl" and not force_fail:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:                                   # (the fallback test hook shares one GPU: no NCCL group possible)
            dist.init_process_group("gloo")

    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd import transport
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    lib = bq.hip_lib()
    n = args.n
    if lib.fl_init(local_rank) != 0:
        bq.check()
    # N > 1: weak scaling -- the grid grows along z, n x n x (n*N), one z-slab of n planes per GPU, one
    # source per slab (same work everywhere); N = 1 is exactly BASELINE's n^3 workload
    keep = None
    side = None                                 # gloo side channel: agreement on the transport, fallback exchange
    if world > 1:
        if args.transport == "rccl":
            # RCCL neighbour exchange is the product path.  If its set-up fails on any rank (binding self-test,
            # unique id, communicator), all ranks agree over gloo to fall back to the host-staged transport, so that
            # a broken fabric yields a slow, clearly labelled number instead of none.
            side = dist.new_group(backend="gloo")
            failure = ""
            try:
                if force_fail:
                    raise RuntimeError("BENCH_FORCE_RCCL_FAILURE=1")
                if lib.fl_comm_selftest() != 0:
                    raise RuntimeError(lib.fl_last_error_string().decode(errors="replace"))
                transport.init_rccl(lib, dist)
            except Exception as e:              # noqa: BLE001 -- any set-up failure takes the same exit
                failure = str(e) or type(e).__name__
            ok = torch.tensor([0 if failure else 1], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=side)
            if int(ok.item()) == 0:
                if rank == 0:
                    print(f"[bench] RCCL set-up failed ({failure or 'on another rank'}); falling back to the host-staged "
                          f"transport over gloo", file=sys.stderr, flush=True)
                lib.fl_comm_destroy()
                lib.fl_clear_error()
                keep = transport.HostStagedTransport(lib, dist, group=side)
                args.transport = "host"
                args.transport_note = "host-staged over gloo (FALLBACK: RCCL set-up failed)"
        else:
            keep = transport.HostStagedTransport(lib, dist)
        if args.strong and n % world:
            sys.exit("--strong needs size divisible by the number of ranks")
        nz_global = n if args.strong else n * world
        s = BimocqGPUSolver(n, n, nz_global, 1.0, 0.0, 1.0, device=local_rank, rank=rank, nranks=world, ghost=args.ghost)
    elif args.emulate_slab:
        keep = transport.NullTransport(lib, 0, 2)
        nz_global = n
        s = BimocqGPUSolver(n, n, 2 * n, 1.0, 0.0, 1.0, device=local_rank, rank=0, nranks=2, ghost=args.ghost)
    else:
        nz_global = n
        s = BimocqGPUSolver(n, n, n, 1.0, 0.0, 1.0, device=local_rank)
    if args.strong:
        s.setSmoke(0.0, 1.0, [(0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)])         # BASELINE's one source in the n^3 box
    else:
        s.setSmoke(0.0, 1.0, [(0.5, 0.2, 0.5 + r, 0.1, 1.0, 1.0, 0.0, 1) for r in range(world)])
    mg = args.projection == "mgcg"
    if mg and world > 1:
        sys.exit("--projection mgcg is single-GPU (the z-slab path runs the Jacobi projection)")
    s.setProjection(args.mg_iters if mg else args.jacobi_iters, args.halfrdx, 1 if mg else 0)
    # The headline number is measured with the reference's FULL per-step sequence (BQ_OPT_FULL_STATE = 1): every
    # buffer the reference updates is updated, including the *Prev state that nothing reads when blend == 1.  The
    # library's default elides that dead state; its rate is reported next to the headline as "extra".
    s.setOption(3, 1)
    dt = 2.0 / n

    def barrier():
        lib.fl_sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=side) if (side is not None and args.transport == "host") else dist.barrier()
        lib.fl_sync()

    frame = 0
    for _ in range(args.warmup):
        s.advance(frame, dt)
        frame += 1
    lib.fl_set_option(bq._lib.FL_OPT_PROFILE_JACOBI, 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s.advance(frame, dt)
        frame += 1
    barrier()
    el = time.perf_counter() - t0
    lib.fl_set_option(bq._lib.FL_OPT_PROFILE_JACOBI, 0)
    bq.check()
    # the same scene continued with the library's default (dead *Prev state not computed), a shorter untimed-in-the-
    # headline leg: reported as extra information only
    extra_steps = max(1, min(40, args.steps))
    s.setOption(3, 0)
    barrier()
    t1 = time.perf_counter()
    for _ in range(extra_steps):
        s.advance(frame, dt)
        frame += 1
    barrier()
    el_extra = time.perf_counter() - t1
    # ... and with the fast arithmetic variant on top (FL_OPT_FAST_LERP: one fp32 fma per lerp; within 1e-6 RMS of
    # the exact fields after 200 steps, tests/test_gpu_solver.py::test_fast_lerp_variant) -- extra information too
    lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 1)
    barrier()
    t2 = time.perf_counter()
    for _ in range(extra_steps):
        s.advance(frame, dt)
        frame += 1
    barrier()
    el_fast = time.perf_counter() - t2
    lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 0)
    ms, launches, sweeps = C.c_double(0.0), C.c_longlong(0), C.c_longlong(0)
    lib.fl_jacobi_profile(C.byref(ms), C.byref(launches), C.byref(sweeps))
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.transport == "rccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=side if (side is not None and args.transport == "host") else None)
        el = float(t.item())

    voxels = n * n * nz_global                  # weak scaling: every rank advances its own n^3 grid; strong: n^3 in all
    ms_per_step = el / args.steps * 1e3
    value = voxels * args.steps / el / 1e6
    line = {
        "metric": "Mvoxels/s per step (bimocq3D rising smoke" + (", multigrid-CG projection)" if mg else ")"),
        "value": round(value, 2), "unit": "Mvoxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64" if mg else "f32", "data": "synthetic",
        "config": {"workload": f"bimocq3D {n}^3 rising smoke, "
                               + (f"fp64 multigrid-CG projection ({args.mg_iters} outer iterations, 6 levels), fp32 advection, "
                                  if mg else f"{args.jacobi_iters} Jacobi iters, fp32, ")
                               + f"halfrdx {args.halfrdx}, reinit every step",
                   "grid_per_gpu": [n, n, nz_global // world], "global_grid": [n, n, nz_global], "dt": dt,
                   "parallelism": "1 GPU" if world == 1 else
                   f"{world} z-slabs of {nz_global // world} planes, {args.ghost} ghost planes, neighbour exchange over "
                   + getattr(args, "transport_note", args.transport)},
    }
    if args.emulate_slab:
        line["metric"] += " [EMULATED z-slab rank 0 of 2 on one GPU, exchanges move nothing: compute-side cost only]"
        line["config"]["parallelism"] = f"emulated slab: {n} owned + 2 x {args.ghost} ghost planes, no data exchanged"
    line["extra"] = {"dead_state_elision": {"value": round(voxels * extra_steps / el_extra / 1e6, 2), "unit": "Mvoxels/s",
                                            "ms_per_step": round(el_extra / extra_steps * 1e3, 3), "steps": extra_steps,
                                            "note": "library default: with blend == 1 and a re-initialisation every frame the "
                                                    "*Prev fields are never read, so the accumulation that only feeds them is "
                                                    "skipped; every observable field is identical (DESIGN.md section 3)"},
                     "fast_lerp_variant": {"value": round(voxels * extra_steps / el_fast / 1e6, 2), "unit": "Mvoxels/s",
                                           "ms_per_step": round(el_fast / extra_steps * 1e3, 3), "steps": extra_steps,
                                           "note": "FL_OPT_FAST_LERP = 1 on top of the elision: every lerp of the gather kernels is "
                                                   "one fp32 fma; NOT bit-identical to the reference arithmetic, within 1e-6 RMS "
                                                   "after 200 steps (tolerance 1e-5; DESIGN.md section 12)"}}
    if launches.value > 0 and mg:
        # dominant kernel: the level-0 fp64 smoothing sweep, two per launch of mg_smooth2_kernel:
        # 24 B/cell/sweep (x, rhs in, x' out)
        us = ms.value * 1e3 / launches.value
        spl = sweeps.value / launches.value
        alg = 24 * n ** 3 * spl
        achieved = alg / (us * 1e-6) / 1e9
        line["roofline"] = {"bound": "hbm", "kernel": "mg_smooth2_kernel (level 0)", "achieved": round(achieved, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                            "us_per_launch": round(us, 3), "launches_timed": int(launches.value),
                            "sweeps_per_launch": round(spl, 3), "us_per_sweep": round(ms.value * 1e3 / sweeps.value, 3),
                            "algorithmic_bytes_per_launch": int(alg)}
    elif launches.value > 0:
        # dominant kernel: the Jacobi sweep.  A launch of jacobi_march2_kernel performs two sweeps, so the
        # algorithmic bytes of a launch are 12 B/voxel x voxels x sweeps-per-launch.
        us = ms.value * 1e3 / launches.value
        spl = sweeps.value / launches.value
        # a z-slab rank sweeps its ghost planes too (communication-avoiding chunks): n x n x (n + 2G) cells per sweep;
        # the overlapped first sweep of a chunk (three range launches) is not inside the timed spans
        cells = n ** 3 if (world == 1 and not args.emulate_slab) else n * n * (nz_global // world + 2 * args.ghost)
        alg = JACOBI_BYTES_PER_VOXEL * cells * spl
        achieved = alg / (us * 1e-6) / 1e9
        kname = (lib.fl_jacobi_kernel_name() or b"").decode() or "jacobi_march2_kernel"
        line["roofline"] = {"bound": "hbm", "kernel": kname if spl > 1.5 else "jacobi_march_kernel",
                            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(n, spl, kname) if (world == 1 and not args.emulate_slab) else None,
                            "us_per_launch": round(us, 3), "launches_timed": int(launches.value),
                            "sweeps_per_launch": round(spl, 3), "us_per_sweep": round(ms.value * 1e3 / sweeps.value, 3),
                            "algorithmic_bytes_per_launch": int(alg)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:                  # never lose the GPU numbers to a host-side hiccup
            line["cpu_baseline"] = {"value": None, "unit": "Mvoxels/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    s.close()
    if dist is not None:
        dist.barrier(group=side) if (side is not None and args.transport == "host") else dist.barrier()
        lib.fl_comm_destroy()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
