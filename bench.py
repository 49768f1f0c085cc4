#!/usr/bin/env python3
"""bench.py -- bimocq3D per-step throughput on MI355X.

Metric (BASELINE.json): Mvoxels/s per step of the rising-smoke scene (L=1, h=1/N,
dt=2h, nu=0, blend=1, alpha=0, beta=1, one spherical source at step 0, Jacobi 200 iterations,
halfrdx=0.5 = the reference's value), plus the HBM roofline fraction of the dominant kernel (the
Jacobi sweep) and the CPU oracle timed beside it.

A "step" is one BimocqGPUSolver::advance(): map update, advection with error compensation, forces,
projection (divergence, Jacobi sweeps, gradient), accumulation, re-initialisation -- all resident in
HBM; nothing crosses PCIe inside the timed region (with --dump the density download of frame f
overlaps frame f+1 on a third stream).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--jacobi-iters 200]

Workloads:
    N = 1 (default)   BASELINE config 3: 256^3, 200 Jacobi iterations, one MI355X
    N > 1 (default)   BASELINE config 4: ONE 512^3 grid split into N z-slabs (strong scaling), ghost planes
                      exchanged with RCCL send/recv over xGMI; --weak gives size^3 per GPU instead
    --scene leapfrog --grid 1024 1024 512 --dump DIR     BASELINE config 5 (two coaxial vortex rings, density dumped
                      every frame, one file per slab and frame)

`python bench.py --gpus N` (N > 1) started on its own launches `python -m torch.distributed.run` with N ranks as a
CHILD process before anything touches the GPU and relays rank 0's JSON line and the exit code; started by
torch.distributed.run (RANK/WORLD_SIZE in the environment) it is one rank of that job.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20
JACOBI_BYTES_PER_VOXEL = 12.0   # read p + read div + write p'  

from gpufluidsimulation_amd.scenes import SMOKE, collision, leapfrog, rising_smoke     # noqa: E402  (pure Python, no GPU touched)

WATCHDOG = {}                # N > 1: the headline of a finished timed region, for the watchdog (main)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=180)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", dest="n", type=int, default=None,
                    help="grid is size^3; default 256 on one GPU (BASELINE config 3), 512 across N > 1 GPUs (config 4)")
    ap.add_argument("--grid", type=int, nargs=3, default=None, metavar=("NX", "NY", "NZ"),
                    help="non-cubic global grid (config 5: 1024 1024 512); h = 1/NX")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: weak scaling, size^3 (default 256) PER GPU, the grid grows along z; the default for N > 1 is "
                         "strong scaling of one global grid (BASELINE config 4)")
    ap.add_argument("--strong", action="store_true", help="(default for N > 1; kept for round-1 command lines)")
    ap.add_argument("--length", type=float, default=1.0, help="domain length along x (h = length / NX); the reference's scene: 0.2")
    ap.add_argument("--dt", type=float, default=None, help="time step (default 2 h); the reference's scene: 0.08")
    ap.add_argument("--viscosity", type=float, default=0.0, help="kinematic viscosity (> 0: 20 diffusion sweeps per component and step); "
                    "the reference's scene: 1e-6")
    ap.add_argument("--reference-scene", action="store_true",
                    help="shorthand for the reference binary's own configuration (src/bimocq3D/main.cpp:28-80): --grid 100 200 200 "
                         "--length 0.2 --dt 0.08 --viscosity 1e-6 --scene collision (spacing 0.002: NOT a power of two; add --scheme "
                         "reflection --projection mgcg for the scheme and projection it ships)")
    ap.add_argument("--scene", choices=["smoke", "leapfrog", "collision"], default="smoke",
                    help="smoke: SURVEY 8d rising smoke; leapfrog: two coaxial vortex rings blown along x by the reference's "
                         "emitter formula (src/bimocq3D/main.cpp:52-78), no buoyancy (BASELINE config 5)")
    ap.add_argument("--dump", default=None, metavar="DIR",
                    help="write the density of every frame (outputResultAsync: download on a third stream, writer thread; "
                         "one density_render_%%04d[.k%%05d].bqd per frame [and slab]) INSIDE the timed region")
    ap.add_argument("--jacobi-iters", type=int, default=200)
    ap.add_argument("--halfrdx", type=float, default=0.5)
    ap.add_argument("--fl-opt", action="append", default=[], metavar="ID=VALUE",
                    help="fl_set_option(ID, VALUE) before the run (A/B timing of library options; repeatable)")
    ap.add_argument("--bq-opt", action="append", default=[], metavar="ID=VALUE",
                    help="bq_solver_set_option(ID, VALUE) on the solver before the run (A/B timing of host-solver options; repeatable)")
    ap.add_argument("--scheme", choices=["bimocq", "reflection"], default="bimocq",
                    help="bimocq: BASELINE's headline solver; reflection: the MacCormack + reflection scheme the reference's binary "
                         "ships as its default (main.cpp:51); z-slab ranks with the Jacobi projection")
    ap.add_argument("--projection", choices=["jacobi", "mgcg"], default="jacobi",
                    help="jacobi: BASELINE's headline config; mgcg: the fp64 multigrid-CG projection the reference's "
                         "shipped binary runs (SURVEY 8f N1), --mg-iters outer iterations, single GPU")
    ap.add_argument("--mg-iters", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the two 'extra' legs (dead-state elision, fast lerps)")
    ap.add_argument("--no-measure-traffic", action="store_true",
                    help="N = 1: do not run the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure the Jacobi "
                         "launch's HBM traffic in this run; roofline.traffic then comes from the committed passes (--no-extra implies it)")
    ap.add_argument("--cpu-n", type=int, default=128, help="grid of the bounded CPU sample")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--shallow-exchange", type=int, nargs="?", const=1, default=0,
                    help="BQ_OPT_SHALLOW_BLOCKING_EXCHANGE: 1 = blocking ghost refreshes, 2 = also the overlapped ones move only "
                         "the planes asked for (N > 1)")
    ap.add_argument("--no-ends-first", action="store_true", help="BQ_OPT_JACOBI_ENDS_FIRST = 0 (N > 1)")
    ap.add_argument("--reinit-policy", type=int, choices=[0, 1], default=0,
                    help="BQ_OPT_REINIT_POLICY: 0 = both map sets re-initialised every frame (the reference's GPU solver, the headline); "
                         "1 = distortion-driven re-initialisation with the CPU solver's thresholds (SURVEY 8f N2)")
    ap.add_argument("--reserve-cus", type=int, default=0, metavar="K",
                    help="FL_OPT_RESERVE_CUS: the compute stream leaves K compute units (K / 8 per XCD) to the halo stream's "
                         "RCCL kernels")
    ap.add_argument("--diag-steps", type=int, default=None,
                    help="N > 1 (or --emulate-slab): steps of the diagnostic leg after the timed region (exposed communication, "
                         "per-phase times) and of each extra leg (--shallow-exchange 2, --no-ends-first, --reserve-cus 8/16); "
                         "default min(steps, 10), 0 = none")
    ap.add_argument("--cpu-256", action="store_true", help="CPU baseline on BASELINE.md section 4's second sample too: 256^3 x 3 steps (minutes)")
    ap.add_argument("--ghost", type=int, default=8, help="ghost planes per side of a z-slab rank (N > 1)")
    ap.add_argument("--keep-dmc-border", type=int, default=None,
                    help="N > 1: BQ_OPT_KEEP_DMC_BORDER (see DESIGN.md section 7); default = the library's slab default")
    ap.add_argument("--emulate-slab", type=int, nargs="?", const=2, default=0, metavar="R",
                    help="debug, one GPU: run the z-slab code path of ONE rank of R (own planes + ghost planes, chunked "
                         "Jacobi, split operators) with exchanges that move nothing -- the compute-side cost of the slab "
                         "path; the line is labelled and is not a benchmark result")
    ap.add_argument("--emulate-rank", type=int, default=None, help="which rank to emulate (default: a middle one)")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="N > 1: ghost planes over RCCL/xGMI (default) or staged through the host over gloo (debug)")
    return ap.parse_args()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: become the launcher.  Nothing in this process
    has touched HIP or torch.cuda (torch is not even imported), and the ranks are CHILD processes, never an exec."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


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
    oracle_lib.lib(march="native", out="_build_native")        # rebuilt for this host's ISA (oracle/Makefile: -O3 -fopenmp)
    mg = args.projection == "mgcg"

    def sample(n, steps):
        s = oracle_lib.OracleSolver(n, n, n, 1.0, 0.0, 1.0)
        s.set_smoke(0.0, 1.0, [SMOKE])
        s.set_projection(args.mg_iters if mg else args.jacobi_iters, args.halfrdx, 1 if mg else 0)
        dt = 2.0 / n
        s.advance(0, dt)                                        # untimed: first-touch + emission
        t0 = time.perf_counter()
        for f in range(1, 1 + steps):
            s.advance(f, dt)
        el = time.perf_counter() - t0
        s.close()
        return round(n ** 3 * steps / el / 1e6, 4), el

    n = args.cpu_n
    value, el = sample(n, args.cpu_steps)
    out = {"value": value, "unit": "Mvoxels/s", "cores": cores, "kind": "port",
           "sample": f"{args.cpu_steps} steps of {n}^3 rising smoke ("
                     + (f"fp64 multigrid-CG, {args.mg_iters} outer iterations" if mg else f"{args.jacobi_iters} Jacobi iters")
                     + f"), OpenMP C oracle (gcc -O3 -march=native -fopenmp -ffp-contract=off), {el:.1f} s; BASELINE.md section 4 "
                       "names 128^3 x 20 and 256^3 x 3 steps -- bounded here to keep the default run within minutes "
                       "(--cpu-steps 20 / --cpu-256 run those)"}
    if args.cpu_256:
        v256, el256 = sample(256, 3)
        out["sample_256"] = {"value": v256, "unit": "Mvoxels/s", "sample": f"3 steps of 256^3, {el256:.1f} s"}
    return out


def pmc_traffic(dims, kernel):
    """Fabric bytes per Jacobi launch from the committed rocprofv3 PMC passes (profiles/jacobi_pmc_traffic.json:
    FETCH_SIZE doubled as the microarchitecture guide prescribes for gfx950 + WRITE_SIZE, separate passes), or None
    when no pass exists for this grid and kernel."""
    path = os.path.join(ROOT, "profiles", "jacobi_pmc_traffic.json")
    nx, ny, nz = dims
    if not (nx == ny == nz):
        return None
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None
    keys = [f"{nx}_{kernel}"]
    if kernel in ("jacobi_march2r_kernel", "jacobi_march2_kernel"):     # (the round-1 passes were filed under these names)
        keys.append(f"{nx}_fused2r" if kernel == "jacobi_march2r_kernel" else f"{nx}_fused2")
    for key in keys:
        if key in table and table[key].get("bytes_per_launch"):
            return table[key]["bytes_per_launch"]
    return None


def pmc_pass(counters, child_args, timeout_s=150):
    """One child process under `rocprofv3 --pmc <counters> --kernel-trace`: returns ({kernel: {counter: [values], "ns": [durations]}},
    None) or (None, reason).  The child runs in its own process group and is ended with it on a timeout; the program follows
    `--` directly."""
    import csv, glob, shutil, signal, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler"
    env = dict(os.environ, TMPDIR="/tmp")       # (the children run the product's stream set-up: CU-masked copy stream, released at exit by the library itself)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    work = tempfile.mkdtemp(prefix="bq_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", work, "-o", "run", "--", sys.executable, *child_args]
        r = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            r.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(r.pid, signal.SIGKILL)
            except OSError:
                pass
            r.wait()
            return None, f"rocprofv3 --pmc {' '.join(counters)} timed out after {timeout_s} s"
        files = glob.glob(os.path.join(work, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None, f"rocprofv3 --pmc {' '.join(counters)} left no counter file (rc {r.returncode})"
        out = {}
        for row in csv.DictReader(open(files[0])):
            name = row.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("bq::exact::", "").replace("bq::", "")
            d = out.setdefault(name, {})
            d.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if row["Counter_Name"] == counters[0]:
                d.setdefault("ns", []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        return out, None
    finally:
        shutil.rmtree(work, ignore_errors=True)


GATHER_BYTES_PER_VOXEL = {"advect_kernel": 20, "compensate_kernel": 24, "cumulate_kernel": 24}     # per sampled field


def measure_gather_counters(n):
    """The nine-point gather family against ITS bounds, measured now: three child passes over a few full steps at n^3
    (tools/step_child.py) -- SQ_INSTS_VALU, TA_BUSY_avr, TCP_TCC_READ_REQ_sum, one counter block per pass.
    valu_issue_busy = instructions x 4 cycles / (1024 SIMDs x duration x 2.4 GHz); ta_busy = TA_BUSY_avr / (duration x 2.4 GHz);
    L1 -> L2 bytes = requests x 64 B over the kernel's algorithmic bytes.  Single-field and two-field launches apart, the
    identity-map accumulate (a light kernel of its own kind) left out."""
    child = [os.path.join(ROOT, "tools", "step_child.py"), "--n", str(n), "--steps", "3", "--warmup", "2", "--jacobi-iters", "6"]
    res = {}
    for counter in ("SQ_INSTS_VALU", "TA_BUSY_avr", "TCP_TCC_READ_REQ_sum"):
        got, why = pmc_pass([counter], child)
        if got is None:
            return None, why
        res[counter] = got
    GHZ = 2.4
    fam = {"single_field": {"valu": [], "ta": [], "l2": []}, "two_field": {"valu": [], "ta": [], "l2": []}}
    for name, d in res["SQ_INSTS_VALU"].items():
        base = name.split("<")[0]
        if base not in GATHER_BYTES_PER_VOXEL or "fast::" in name:
            continue
        tpl = [t.strip() for t in name[name.index("<") + 1:name.rindex(">")].split(",")]
        nf = int(tpl[3])
        if base == "cumulate_kernel" and tpl[4] == "true":      # ID: the identity-map accumulate
            continue
        ns = sum(d["ns"]) / len(d["ns"])
        key = "two_field" if nf == 2 else "single_field"
        fam[key]["valu"].append(sum(d["SQ_INSTS_VALU"]) / len(d["SQ_INSTS_VALU"]) * 4.0 / (1024.0 * ns * GHZ))
        ta = res["TA_BUSY_avr"].get(name)
        if ta:
            fam[key]["ta"].append((sum(ta["TA_BUSY_avr"]) / len(ta["TA_BUSY_avr"])) / ((sum(ta["ns"]) / len(ta["ns"])) * GHZ))
        l2 = res["TCP_TCC_READ_REQ_sum"].get(name)
        if l2:
            alg = GATHER_BYTES_PER_VOXEL[base] * nf * float(n) ** 3
            fam[key]["l2"].append(sum(l2["TCP_TCC_READ_REQ_sum"]) / len(l2["TCP_TCC_READ_REQ_sum"]) * 64.0 / alg)
    if not fam["single_field"]["valu"]:
        return None, "no gather kernel in the counter passes"
    rng = lambda v: [round(min(v), 3), round(max(v), 3)] if v else None
    return {"valu_issue_busy": {k: rng(v["valu"]) for k, v in fam.items()}, "ta_busy": {k: rng(v["ta"]) for k, v in fam.items()},
            "l1_to_l2_bytes_over_algorithmic": {k: rng(v["l2"]) for k, v in fam.items()},
            "kernels_seen": {k: len(v["valu"]) for k, v in fam.items()}, "clock_assumed_ghz": GHZ}, None


def measure_traffic(n, kernel_hint, timeout_s=120):
    """HBM bytes per Jacobi launch measured NOW: two child processes under rocprofv3 --pmc (FETCH_SIZE, then WRITE_SIZE --
    separate passes, as the microarchitecture guide prescribes) run tools/jacobi_tune.py on an n^3 grid with the library's
    default launch configuration; FETCH_SIZE is doubled (gfx950 counts 128-byte requests at 64 bytes), both are in KB.
    Returns (bytes_per_launch, {details}) or (None, {reason}).  The parent (this process) is idle meanwhile."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, {"reason": "rocprofv3 not on PATH"}
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, {"reason": "this process already runs under a profiler"}
    env = dict(os.environ, TMPDIR="/tmp")       # (the children run the product's stream set-up: CU-masked copy stream, released at exit by the library itself)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    means = {}
    work = tempfile.mkdtemp(prefix="bq_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "run", "--",
                   sys.executable, os.path.join(ROOT, "tools", "jacobi_tune.py"), "--n", str(n), "--variants", "4:0:0", "--sweeps", "21", "--reps", "1"]
            # its own process group, so that a pass that overruns is ended together with the program rocprofv3 started
            r = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                r.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(r.pid, signal.SIGKILL)
                except OSError:
                    pass
                r.wait()
                return None, {"reason": f"rocprofv3 --pmc {counter} pass timed out after {timeout_s} s"}
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, {"reason": f"rocprofv3 --pmc {counter} left no counter file (rc {r.returncode})"}
            by = {}
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") == counter and "jacobi" in row.get("Kernel_Name", ""):
                    by.setdefault(row["Kernel_Name"].split("(")[0].replace("void ", "").replace("bq::", ""), []).append(float(row["Counter_Value"]))
            if not by:
                return None, {"reason": f"no Jacobi kernel in the {counter} pass"}
            # the fused kernel the default configuration launches: the one with the most launches in the child's loop
            name = max(by, key=lambda k_: len(by[k_]))
            means[counter] = (name, sum(by[name]) / len(by[name]), len(by[name]))
        if means["FETCH_SIZE"][0] != means["WRITE_SIZE"][0]:
            return None, {"reason": "the two passes saw different kernels"}
        fetch_b, write_b = means["FETCH_SIZE"][1] * 1024.0 * 2.0, means["WRITE_SIZE"][1] * 1024.0
        return int(fetch_b + write_b), {"kernel": means["FETCH_SIZE"][0], "fetch_bytes": int(fetch_b), "write_bytes": int(write_b),
                                        "launches": means["FETCH_SIZE"][2], "hint": kernel_hint}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    args = parse()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
    force_fail = os.environ.get("BENCH_FORCE_RCCL_FAILURE") == "1"      # test hook for the fallback below
    if local_rank >= torch.cuda.device_count():
        # (BQ_RCCL_LIBRARY: the tests' multi-process stand-in for RCCL lets ranks share a GPU, real RCCL does not)
        if args.transport != "host" and not force_fail and not os.environ.get("BQ_RCCL_LIBRARY"):
            sys.exit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
        local_rank %= torch.cuda.device_count()         # debug transport: several ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dist = None
    t_start = time.perf_counter()

    def stage(msg):
        # N > 1: where every rank is, on stderr -- a collective that never returns (real RCCL has no deadlock detection) then
        # shows as the last line of the rank that entered it and the missing line of the rank that did not
        if world > 1:
            sys.stderr.write(f"[bench rank {rank}/{world} +{time.perf_counter() - t_start:6.1f} s] {msg}\n")      # (one write: ranks share the pipe)
            sys.stderr.flush()

    if world > 1:
        # ... and a hung run ends itself with every thread's Python stack on stderr (BENCH_WATCHDOG_S seconds, 0 = never)
        # instead of waiting for the launcher's kill, which says nothing
        wd = int(os.environ.get("BENCH_WATCHDOG_S", "480"))
        if wd > 0:
            import faulthandler
            import threading

            def watchdog_fired():
                print(f"[bench rank {rank}/{world}] WATCHDOG: no result after {wd} s -- the Python stacks of this rank follow", file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                fb = WATCHDOG.get("headline")
                if rank == 0 and fb is not None and not WATCHDOG.get("printed"):
                    # the timed region had finished on every rank: hand out its number rather than nothing
                    print(json.dumps(fb), flush=True)
                os._exit(0 if (WATCHDOG.get("printed") or fb is not None) else 3)      # (the headline exists on every rank)
            timer = threading.Timer(wd, watchdog_fired)
            timer.daemon = True
            timer.start()
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the CONTROL plane only (rendezvous, the unique id, barriers, the max over ranks of the
        # elapsed time): gloo.  The DATA plane -- ghost planes, wall sheets, the scalar all-reduces inside a step -- is
        # the library's own RCCL communicator over xGMI (csrc/bq_halo.hip), created below from an id that travels over
        # this group.  (One RCCL instance per process: torch's bundled copy is never initialised.)
        dist.init_process_group("gloo")
        stage("control plane up (gloo)")

    import gpufluidsimulation_amd as bq
    from gpufluidsimulation_amd import transport
    from gpufluidsimulation_amd.solver import BimocqGPUSolver
    lib = bq.hip_lib()
    if lib.fl_init(local_rank) != 0:
        bq.check()
    if args.reserve_cus:
        lib.fl_set_option(bq._lib.FL_OPT_RESERVE_CUS, args.reserve_cus)
        bq.check()

    # ---- the grid -------------------------------------------------------------------------------------
    emul = args.emulate_slab if world == 1 else 0
    multi = world > 1 or emul > 1
    nslabs = world if world > 1 else (emul if emul > 1 else 1)
    weak = args.weak and not args.strong
    if args.reference_scene:
        args.grid = args.grid or [100, 200, 200]
        args.length, args.scene = 0.2, "collision"
        args.dt = args.dt if args.dt is not None else 0.08
        args.viscosity = args.viscosity or 1e-6
        args.no_extra = True
    if args.grid:
        nx, ny, nz_global = args.grid
        weak = False
    else:
        n = args.n if args.n else (256 if (not multi or weak) else 512)
        nx = ny = n
        nz_global = n * nslabs if (multi and weak) else n
    if multi and nz_global % nslabs:
        sys.exit("the global plane count must be divisible by the number of ranks")
    h = args.length / nx
    dt = args.dt if args.dt is not None else 2.0 * h

    keep = None
    side = None                                 # gloo side channel: agreement on the transport, fallback exchange
    if world > 1:
        if args.transport == "rccl":
            # RCCL neighbour exchange is the product path.  If its set-up fails on any rank (binding self-test,
            # unique id, communicator), all ranks agree over gloo to fall back to the host-staged transport, so that
            # a broken fabric yields a slow, clearly labelled number instead of none.
            side = None                         # (the default group is gloo already)
            failure = ""
            try:
                if force_fail:
                    raise RuntimeError("BENCH_FORCE_RCCL_FAILURE=1")
                stage("RCCL binding self-test (one-rank communicator)")
                if lib.fl_comm_selftest() != 0:
                    raise RuntimeError(lib.fl_last_error_string().decode(errors="replace"))
                stage("fl_comm_init: unique id over gloo, ncclCommInitRank, second communicator")
                transport.init_rccl(lib, dist)
                stage(f"communicators up: {lib.fl_comm_count()}")
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
        s = BimocqGPUSolver(nx, ny, nz_global, args.length, args.viscosity, 1.0, device=local_rank, rank=rank, nranks=world, ghost=args.ghost,
                            scheme=3 if args.scheme == "reflection" else 0)
    elif emul > 1:
        erank = args.emulate_rank if args.emulate_rank is not None else emul // 2
        keep = transport.NullTransport(lib, erank, emul)
        s = BimocqGPUSolver(nx, ny, nz_global, args.length, args.viscosity, 1.0, device=local_rank, rank=erank, nranks=emul, ghost=args.ghost,
                            scheme=3 if args.scheme == "reflection" else 0)
    else:
        s = BimocqGPUSolver(nx, ny, nz_global, args.length, args.viscosity, 1.0, device=local_rank, scheme=3 if args.scheme == "reflection" else 0)
    comm_size = int(lib.fl_comm_size())

    # ---- the scene ------------------------------------------------------------------------------------
    if args.scene == "collision":
        s.setSmoke(0.0, 0.0, collision(h))                  # the reference binary's own scene (gpufluidsimulation_amd/scenes.py)
    elif args.scene == "leapfrog":
        s.setSmoke(0.0, 0.0, leapfrog(nz_global, h))        # gpufluidsimulation_amd/scenes.py
    elif multi and weak:
        s.setSmoke(0.0, 1.0, [(0.5, 0.2, 0.5 + r, 0.1, 1.0, 1.0, 0.0, 1) for r in range(nslabs)])    # one source per slab
    else:
        s.setSmoke(0.0, 1.0, rising_smoke(nz_global, h))
    mg = args.projection == "mgcg"
    # (N > 1 with --projection mgcg: every rank assembles the global velocity and runs the single-GPU solver on it --
    #  replicated, bit-identical, not scalable: csrc/host/fluid_solver.cpp projectionMgcgSlabs)
    if args.scheme == "reflection":
        args.no_extra = True                    # the extra legs compare BiMocq state-elision variants
    s.setProjection(args.mg_iters if mg else args.jacobi_iters, args.halfrdx, 1 if mg else 0)
    # The headline number is measured with the reference's FULL per-step sequence (BQ_OPT_FULL_STATE = 1): every
    # buffer the reference updates is updated, including the *Prev state that nothing reads when blend == 1.  The
    # library's default elides that dead state; its rate is reported next to the headline as "extra".
    s.setOption(3, 1)
    if args.reinit_policy:
        s.setOption(2, 1)                       # BQ_OPT_REINIT_POLICY (before the first advance)
        args.no_extra = True                    # the extra legs belong to the every-frame policy
    if args.shallow_exchange:
        s.setOption(6, args.shallow_exchange)
    if args.no_ends_first:
        s.setOption(7, 0)
    if args.keep_dmc_border is not None:
        s.setOption(1, args.keep_dmc_border)
    for kv in args.bq_opt:
        k_, v_ = kv.split("=")
        s.setOption(int(k_), int(v_))
    if args.dump:
        os.makedirs(args.dump, exist_ok=True)

    def barrier():
        lib.fl_sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        lib.fl_sync()

    frame = 0

    def run(k):
        nonlocal frame
        for _ in range(k):
            s.advance(frame, dt)
            if args.dump:
                s.outputResultAsync(frame, args.dump)        # joins the previous frame's dump, starts this one
            frame += 1
        if args.dump:
            s.waitOutput()

    for kv in args.fl_opt:
        k, v = kv.split("=")
        lib.fl_set_option(int(k), int(v))
    stage(f"solver built, {args.warmup} warm-up steps")
    run(args.warmup)
    lib.fl_set_option(bq._lib.FL_OPT_PROFILE_JACOBI, 1)
    lib.fl_comm_stats(None, 1)
    barrier()
    stage(f"timed region: {args.steps} steps")
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    el = time.perf_counter() - t0
    stage(f"timed region done: {el / max(1, args.steps) * 1e3:.3f} ms per step on this rank; diagnostics and knob legs follow")
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        # (test hook: BENCH_HANG_AFTER_TIMED=1 parks every rank here, as a collective that never returns would)
        # what the watchdog prints if a later leg never returns (the contract's fields only; the full line replaces it)
        WATCHDOG["headline"] = {
            "metric": "Mvoxels/s per step (bimocq3D)", "value": round(nx * ny * nz_global * args.steps / el / 1e6, 2), "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"bimocq3D {nx}x{ny}x{nz_global}, {args.jacobi_iters} Jacobi iters, fp32", "parallelism": f"{world} z-slabs",
                       "note": "WATCHDOG LINE: the timed region finished on every rank, a diagnostics / knob leg after it did not return "
                               "(stacks on stderr); roofline, cpu_baseline and diagnostics are missing from this line"}}
        if os.environ.get("BENCH_HANG_AFTER_TIMED") == "1":
            if wd > 0:                          # (the test does not wait out the whole period: the same handler, three seconds from here)
                timer.cancel()
                threading.Timer(3.0, watchdog_fired).start()
            time.sleep(10 ** 6)
    comm_stats = (C.c_longlong * 4)()
    lib.fl_comm_stats(comm_stats, 0)
    lib.fl_set_option(bq._lib.FL_OPT_PROFILE_JACOBI, 0)
    bq.check()
    ms, launches, sweeps = C.c_double(0.0), C.c_longlong(0), C.c_longlong(0)
    lib.fl_jacobi_profile(C.byref(ms), C.byref(launches), C.byref(sweeps))
    extra = None
    if not args.no_extra and args.scene == "smoke" and not args.dump:
        # the same scene continued with the library's default (dead *Prev state not computed), a shorter leg outside
        # the headline: reported as extra information only
        extra_steps = max(1, min(40, args.steps))
        s.setOption(3, 0)
        barrier()
        t1 = time.perf_counter()
        run(extra_steps)
        barrier()
        el_extra = time.perf_counter() - t1
        # ... and with the fast arithmetic variant on top (FL_OPT_FAST_LERP: one fp32 fma per lerp; NOT the reference
        # arithmetic) -- extra information too
        lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 1)
        barrier()
        t2 = time.perf_counter()
        run(extra_steps)
        barrier()
        el_fast = time.perf_counter() - t2
        lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 0)
        extra = (extra_steps, el_extra, el_fast)
    # ---- one GPU: where the step's time goes, and the second kernel family against its own bounds (outside the timed region) ----
    phase_ms, quad_leg = None, None
    if world == 1 and args.scheme == "bimocq" and mg and not args.no_extra:
        # multigrid-CG mode (one GPU, or an emulated slab rank): where the step's time goes, nothing else
        n_ph = max(1, min(3, args.steps))
        s.setOption(8, 1)
        s.phaseMs(reset=True)
        barrier()
        run(n_ph)
        barrier()
        phases, psteps = s.phaseMs(reset=True)
        s.setOption(8, 0)
        phase_ms = {k: round(v / max(1, psteps), 3) for k, v in phases.items()}
    if world == 1 and not emul and args.scheme == "bimocq" and not mg and not args.no_extra:
        n_ph = max(1, min(10, args.steps))
        s.setOption(3, 1)                                   # the headline's full sequence
        s.setOption(8, 1)                                   # BQ_OPT_PROFILE_PHASES
        s.phaseMs(reset=True)
        barrier()
        run(n_ph)
        barrier()
        phases, psteps = s.phaseMs(reset=True)
        s.setOption(8, 0)
        phase_ms = {k: round(v / max(1, psteps), 3) for k, v in phases.items()}
        # ... and the same sequence with FOUR Jacobi sweeps per launch where the LDS-exchanged kernel applies
        # (FL_OPT_JACOBI_ROWS = 6): fewer microseconds per sweep, but a launch that moves the same bytes takes longer --
        # the per-launch roofline fraction falls while the loop gets faster.  Extra information; the headline keeps the default.
        n_q = max(1, min(20, args.steps))
        lib.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 6)
        run(2)
        lib.fl_set_option(bq._lib.FL_OPT_PROFILE_JACOBI, 1)
        barrier()
        tq = time.perf_counter()
        run(n_q)
        barrier()
        el_q = time.perf_counter() - tq
        lib.fl_set_option(bq._lib.FL_OPT_PROFILE_JACOBI, 0)
        lib.fl_set_option(bq._lib.FL_OPT_JACOBI_ROWS, 0)
        qms, ql, qs = C.c_double(0.0), C.c_longlong(0), C.c_longlong(0)
        lib.fl_jacobi_profile(C.byref(qms), C.byref(ql), C.byref(qs))
        if ql.value > 0 and qs.value > 0:
            quad_leg = {"value": round(nx * ny * nz_global * n_q / el_q / 1e6, 2), "unit": "Mvoxels/s", "ms_per_step": round(el_q / n_q * 1e3, 3),
                        "steps": n_q, "sweeps_per_launch": round(qs.value / ql.value, 3), "us_per_sweep": round(qms.value * 1e3 / qs.value, 3),
                        "us_per_launch": round(qms.value * 1e3 / ql.value, 3),
                        "note": "FL_OPT_JACOBI_ROWS = 6: four sweeps per launch (jacobi_lds_kernel<4, 2, 4>) where it applies, same results bit for bit; "
                                "these are LATER steps of the scene than the timed region's (more DMC sub-steps per step): compare us_per_sweep with "
                                "roofline.us_per_sweep, not `value` with the headline"}
        else:
            quad_leg = None
    # ---- diagnostics for z-slab runs: what the timed region alone cannot tell (outside it: the event pairs cost ~1 %) ----
    # One leg with the run's own settings -- communication no kernel hid (the compute stream's waits on the halo stream),
    # the in-stream all-reduces, milliseconds per phase of the step, on EVERY rank -- and one short leg per knob whose best
    # setting only real links can decide.  A single 8-GPU run then says where the time goes and which knob to turn.
    diag, legs = None, {}
    dsteps = args.diag_steps if args.diag_steps is not None else max(1, min(10, args.steps))
    if multi and dsteps > 0 and args.scheme == "bimocq" and not mg:
        L_ = bq._lib
        own_voxels = nx * ny * (nz_global // nslabs)
        voxels = nx * ny * nz_global

        def leg(setup=None, restore=None):
            if setup:
                setup()
            lib.fl_set_option(L_.FL_OPT_PROFILE_COMM, 1)
            s.setOption(8, 1)                                   # BQ_OPT_PROFILE_PHASES
            lib.fl_comm_profile(None, None, 1)
            s.phaseMs(reset=True)
            lib.fl_comm_stats(None, 1)
            barrier()
            t_ = time.perf_counter()
            run(dsteps)
            barrier()
            e_ = time.perf_counter() - t_
            ms2, n2, st4 = (C.c_double * 2)(), (C.c_longlong * 2)(), (C.c_longlong * 4)()
            lib.fl_comm_profile(ms2, n2, 1)
            phases, psteps = s.phaseMs(reset=True)
            lib.fl_comm_stats(st4, 0)
            lib.fl_set_option(L_.FL_OPT_PROFILE_COMM, 0)
            s.setOption(8, 0)
            if restore:
                restore()
            bq.check()
            mine = {"rank": rank, "ms_per_step": round(e_ / dsteps * 1e3, 3),
                    "comm_exposed_ms_per_step": round(ms2[0] / dsteps, 4), "comm_waits_per_step": round(n2[0] / dsteps, 1),
                    "allreduce_ms_per_step": round(ms2[1] / dsteps, 4), "allreduces_per_step": round(n2[1] / dsteps, 1),
                    "phase_ms_per_step": {k: round(v / max(1, psteps), 3) for k, v in phases.items()},
                    "ghost_MB_sent_per_step": round(st4[1] / dsteps / 1e6, 1), "wall_sheet_MB_sent_per_step": round(st4[3] / dsteps / 1e6, 2)}
            everyone = [mine]
            if dist is not None:
                everyone = [None] * world
                dist.all_gather_object(everyone, mine)
            slow = max(everyone, key=lambda r_: r_["ms_per_step"])
            out = {"steps": dsteps, "ms_per_step": slow["ms_per_step"],
                   "value": round((voxels if world > 1 else own_voxels) / (slow["ms_per_step"] * 1e-3) / 1e6, 2),
                   "comm_exposed_ms_per_step": max(r_["comm_exposed_ms_per_step"] for r_ in everyone),
                   "allreduce_ms_per_step": max(r_["allreduce_ms_per_step"] for r_ in everyone),
                   "phase_ms_per_step_slowest_rank": slow["phase_ms_per_step"], "slowest_rank": slow["rank"]}
            return out, everyone

        diag, everyone = leg()
        diag["per_rank"] = everyone
        diag["note"] = ("comm_exposed_ms_per_step: time the compute stream spent blocked on the halo stream (event pairs around "
                        "every wait; max over ranks) -- communication that no kernel hid; phase times include the waits that fall "
                        "into the phase; measured in a separate leg after the timed region")
        sh, ef, rc = args.shallow_exchange, not args.no_ends_first, args.reserve_cus
        alt_sh = 0 if sh == 2 else 2
        legs[f"shallow_exchange_{alt_sh}"] = leg(lambda: s.setOption(6, alt_sh), lambda: s.setOption(6, sh))[0]
        legs["ends_first_" + ("off" if ef else "on")] = leg(lambda: s.setOption(7, 0 if ef else 1), lambda: s.setOption(7, 1 if ef else 0))[0]
        legs["jacobi_triples_off"] = leg(lambda: s.setOption(10, 0), lambda: s.setOption(10, 1))[0]      # BQ_OPT_JACOBI_TRIPLES: the pair schedule
        for k_ in ([8, 16] if rc == 0 else [0]):
            legs[f"reserve_cus_{k_}"] = leg(lambda: lib.fl_set_option(L_.FL_OPT_RESERVE_CUS, k_), lambda: lib.fl_set_option(L_.FL_OPT_RESERVE_CUS, rc))[0]
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    voxels = nx * ny * nz_global                # the whole job's grid: every rank advances its slab of it once per step
    own_planes = nz_global // nslabs
    ms_per_step = el / args.steps * 1e3
    value = voxels * args.steps / el / 1e6
    grid_txt = f"{nx}^3" if nx == ny == nz_global else f"{nx}x{ny}x{nz_global}"
    scene_txt = {"smoke": "rising smoke", "leapfrog": "leapfrogging vortex rings",
                 "collision": f"vortex-ring collision (the reference binary's scene: L {args.length}, dt {dt:g}, viscosity {args.viscosity:g})"}[args.scene]
    border_txt = ""
    if multi:
        kb = s.getOption(1) if hasattr(s, "getOption") else None
        border_txt = {1: ", DMC map border kept (BQ_OPT_KEEP_DMC_BORDER = 1: slab ranks bit-identical to one GPU in this mode)",
                      0: ", DMC map border zeroed as in the reference (BQ_OPT_KEEP_DMC_BORDER = 0)"}.get(kb, "")
    line = {
        "metric": f"Mvoxels/s per step (bimocq3D {scene_txt}" + (", MacCormack-reflection scheme" if args.scheme == "reflection" else "")
                  + (", multigrid-CG projection)" if mg else ")"),
        "value": round(value, 2), "unit": "Mvoxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None,
        "dtype": "f64" if mg else "f32", "data": "synthetic",
        "config": {"workload": f"bimocq3D {grid_txt} {scene_txt}, " + ("MAC_REFLECTION scheme (two projections per step), " if args.scheme == "reflection" else "")
                               + (f"fp64 multigrid-CG projection ({args.mg_iters} outer iterations, 6 levels), fp32 advection, "
                                  if mg else f"{args.jacobi_iters} Jacobi iters, fp32, ")
                               + f"halfrdx {args.halfrdx}, " + ("distortion-driven reinit (BQ_OPT_REINIT_POLICY = 1)" if args.reinit_policy else "reinit every step")
                               + (", density dumped every frame (async, per slab)" if args.dump else ""),
                   "grid_per_gpu": [nx, ny, own_planes], "global_grid": [nx, ny, nz_global], "dt": dt,
                   "nonfinite_velocity_seen": bool(lib.fl_nonfinite_seen(0)),
                   "map_reinitialisations": ({"velocity": s.reinitCounts()[0], "scalar": s.reinitCounts()[1], "forced_by_travel_limit": s.forcedReinits(),
                                              "steps_run": frame} if args.reinit_policy else None),
                   "comm_size": comm_size, "rccl_version": (int(lib.fl_comm_rccl_version()) or None) if world > 1 else None,
                   "communicators": int(lib.fl_comm_count()) if world > 1 else 0,
                   "reserved_cus": args.reserve_cus,
                   "comm_per_step_rank0": None if not multi else {
                       "ghost_exchanges": round(comm_stats[0] / args.steps, 1), "ghost_MB_sent": round(comm_stats[1] / args.steps / 1e6, 1),
                       "wall_sheet_groups": round(comm_stats[2] / args.steps, 1), "wall_sheet_MB_sent": round(comm_stats[3] / args.steps / 1e6, 2)},
                   "parallelism": "1 GPU" if world == 1 else
                   f"{world} z-slabs of {own_planes} planes, {args.ghost} ghost planes, neighbour exchange over "
                   + getattr(args, "transport_note", args.transport + (f" ({comm_size} ranks in the communicator)" if args.transport == "rccl" else ""))
                   + border_txt},
    }
    if emul > 1:
        line["metric"] += f" [EMULATED z-slab rank {s.rank} of {emul} on one GPU, exchanges move nothing: compute-side cost only]"
        line["config"]["parallelism"] = (f"emulated slab rank {s.rank} of {emul}: {own_planes} owned + 2 x {args.ghost} ghost planes, "
                                         f"no data exchanged" + border_txt)
        # what the rank advances per step is its own slab
        line["value"] = round(nx * ny * own_planes * args.steps / el / 1e6, 2)
        line["config"]["value_counts"] = "owned voxels of the emulated rank only"
    if diag:
        line["diagnostics"] = diag
        line.setdefault("extra", {}).update(legs)
    if extra:
        extra_steps, el_extra, el_fast = extra
        line.setdefault("extra", {}).update({"dead_state_elision": {"value": round(voxels * extra_steps / el_extra / 1e6, 2), "unit": "Mvoxels/s",
                                                "ms_per_step": round(el_extra / extra_steps * 1e3, 3), "steps": extra_steps,
                                                "note": "library default: with blend == 1 and a re-initialisation every frame the "
                                                        "*Prev fields are never read, so the accumulation that only feeds them is "
                                                        "skipped; every observable field is identical (DESIGN.md section 3)"},
                         "fast_lerp_variant": {"value": round(voxels * extra_steps / el_fast / 1e6, 2), "unit": "Mvoxels/s",
                                               "ms_per_step": round(el_fast / extra_steps * 1e3, 3), "steps": extra_steps,
                                               "note": "FL_OPT_FAST_LERP = 1 on top of the elision: every lerp of the gather kernels is "
                                                       "one fp32 fma; NOT the reference arithmetic (DESIGN.md section 12: deviation "
                                                       "from the exact fields measured per grid size, tolerance 1e-5 RMS)"}})
    if phase_ms and quad_leg:
        line.setdefault("extra", {})["jacobi_four_sweeps_per_launch"] = quad_leg
    if launches.value > 0 and mg:
        # dominant kernel: the level-0 fp64 smoothing sweep, two per launch of mg_lean2r_kernel (mg_smooth2_kernel with FL_OPT_JACOBI_ROWS = 3): one launch reads x and
        # rhs and writes x' once (24 B/cell compulsory), which is 24 B/cell/sweep x 2 sweeps in SURVEY 8(d)'s per-sweep
        # accounting
        us = ms.value * 1e3 / launches.value
        spl = sweeps.value / launches.value
        cells = nx * ny * nz_global
        compulsory = 24.0 * cells
        alg = compulsory * spl
        ach = compulsory / (us * 1e-6) / 1e9
        mgk = (lib.fl_mg_smooth_kernel_name() or b"").decode() or "mg_lean2r_kernel"
        mg_traffic = pmc_traffic((nx, ny, nz_global), mgk)
        line["roofline"] = {"bound": "hbm", "kernel": mgk + " (level 0 of the V-cycle: three / two fp64 smoothing sweeps per launch)", "achieved": round(ach, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": mg_traffic,
                            "traffic_source": ("profiles/jacobi_pmc_traffic.json: committed rocprofv3 --pmc passes of this kernel at this grid "
                                               "(FETCH_SIZE x 2 for gfx950 + WRITE_SIZE, separate passes) -- NOT measured in this run") if mg_traffic else None,
                            "frac_traffic": round(mg_traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if mg_traffic else None,
                            "compulsory_bytes_per_launch": int(compulsory),
                            "algorithmic_equiv": {"bytes_per_launch": int(alg), "achieved": round(alg / (us * 1e-6) / 1e9, 1),
                                                  "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                  "note": "24 B/cell/sweep x sweeps per launch: what unfused sweeps would move"},
                            "us_per_launch": round(us, 3), "launches_timed": int(launches.value),
                            "sweeps_per_launch": round(spl, 3), "us_per_sweep": round(ms.value * 1e3 / sweeps.value, 3)}
    elif launches.value > 0:
        # Dominant kernel: the Jacobi sweep.  A fused launch performs two sweeps but reads p and div and writes p' ONCE:
        # its compulsory bytes are 12 B/voxel x voxels, whatever the number of sweeps.  `achieved`/`frac` are those
        # bytes over the launch time -- a true fraction of the HBM peak.  SURVEY 8(d)'s per-sweep accounting
        # (12 B/voxel/sweep x sweeps per launch = what unfused sweeps would move) is kept as `algorithmic_equiv`.
        us = ms.value * 1e3 / launches.value
        spl = sweeps.value / launches.value
        # a z-slab rank sweeps its ghost planes too (communication-avoiding chunks); the overlapped first launches of
        # a chunk (plane ranges) are not inside the timed spans
        planes = nz_global if not multi else own_planes + 2 * args.ghost
        cells = nx * ny * planes
        compulsory = JACOBI_BYTES_PER_VOXEL * cells
        alg = compulsory * spl
        ach = compulsory / (us * 1e-6) / 1e9
        kname = (lib.fl_jacobi_kernel_name() or b"").decode() or "jacobi_march2_kernel"
        kname = kname if spl > 1.5 else "jacobi_march_kernel"
        traffic = pmc_traffic((nx, ny, nz_global), kname) if not multi else None
        traffic_source = ("profiles/jacobi_pmc_traffic.json: committed rocprofv3 --pmc passes of this kernel at this "
                          "grid (FETCH_SIZE x 2 as the microarchitecture guide prescribes for gfx950 + WRITE_SIZE, "
                          "separate passes, tools/jacobi_pmc.sh) -- NOT measured in this run") if traffic else None
        if rank == 0 and world == 1 and not emul and not args.no_extra and not args.no_measure_traffic and nx == ny == nz_global and spl > 1.5:
            live, how = measure_traffic(nx, kname)
            if live:
                traffic_committed = traffic
                traffic = live
                traffic_source = (f"MEASURED IN THIS RUN: two child passes of rocprofv3 --pmc (FETCH_SIZE x 2 for gfx950, WRITE_SIZE) over "
                                  f"{how['launches']} launches of {how['kernel']} in tools/jacobi_tune.py --n {nx} (same library, same launch "
                                  f"configuration); the committed passes say {traffic_committed}")
            else:
                traffic_source = (traffic_source or "none") + f" [live measurement not available: {how['reason']}]"
        resident = compulsory <= INFINITY_CACHE_BYTES
        line["roofline"] = {"bound": "hbm", "kernel": kname,
                            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                            "traffic": traffic,
                            "traffic_source": traffic_source,
                            "frac_traffic": round(traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                            "compulsory_bytes_per_launch": int(compulsory),
                            "working_set": ("infinity-cache" if resident else "hbm"),
                            "bound_note": ((f"p, p', div = {compulsory / 1e6:.0f} MB fit the 256 MiB Infinity Cache: the launch is bound "
                                            "by the fabric/Infinity Cache, not by DRAM; the HBM peak is the yardstick BASELINE names"
                                            if resident else f"p, p', div = {compulsory / 1e6:.0f} MB: HBM-resident")
                                           + (f"; {spl:.2f} sweeps per launch: a fused launch moves ONE sweep's bytes, so `frac` falls as "
                                              "more sweeps are fused while the time per sweep (us_per_sweep) improves"
                                              + (" -- the LDS-exchanged three-sweep kernel keeps VALU, LDS and the L1 path each 25-40 % busy "
                                                 "between one barrier per plane (DESIGN.md section 4), it is not bound by memory" if resident else
                                                 " -- HBM-resident arrays: the three-sweep launch streams them once, near the rate a plain "
                                                 "triad reaches at this size") if spl > 2.5 else "")),
                            "algorithmic_equiv": {"bytes_per_launch": int(alg), "achieved": round(alg / (us * 1e-6) / 1e9, 1),
                                                  "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                  "note": "SURVEY 8(d): 12 B/voxel/SWEEP x sweeps per launch -- the bytes unfused "
                                                          "sweeps would move; above 1 only because a fused launch moves one sweep's bytes"},
                            "us_per_launch": round(us, 3), "launches_timed": int(launches.value),
                            "sweeps_per_launch": round(spl, 3), "us_per_sweep": round(ms.value * 1e3 / sweeps.value, 3)}
    if phase_ms and (mg or emul):
        line["phase_ms_per_step"] = phase_ms
        line["config"]["mgcg_levels_shared"] = bool(emul > 1 and s.getOption(11) == 2) if mg else None
    elif phase_ms:
        # The nine-point gather family (advect / compensate / cumulate + the limiter: the advection phase) is NOT bound by
        # HBM: algorithmic bytes per SURVEY 8(d) -- 20 B/voxel per advected component + ~60 B per compensate chain, 5
        # components -- over the phase time give a small fraction of the HBM peak; what binds it is VALU instruction issue
        # (the reference's double-rounded lerp: 5 instructions each, 63 per sampled field and node) together with the
        # texture-addresser path of its 36 dwordx2 corner loads per field, both ~80 % busy (committed counter passes).
        vox = nx * ny * nz_global
        alg = (5 * 20.0 + 5 * 60.0) * vox
        t_ms = phase_ms["advect_compensate"]
        line["phase_ms_per_step"] = phase_ms
        line["roofline_gather"] = {
            "bound": "valu-issue + texture-addresser (not hbm)", "phase": "advect_compensate",
            "launches": "12 nine-point gathers (advect, compensate-error, cumulate; 1- and 2-field) + 5 limiter launches per step",
            "ms_per_step": t_ms, "algorithmic_bytes_per_step": int(alg),
            "achieved": round(alg / (t_ms * 1e-3) / 1e9, 1) if t_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if t_ms > 0 else None,
            "counters": {"source": "committed rocprofv3 --pmc passes at 256^3, NOT this run: profiles/r01_n_gather_sq_counters.json "
                                   "(SQ_INSTS_VALU, SQ_BUSY_CYCLES), profiles/r02_j_gather_ta_tcp_counters.txt (TA_BUSY, TCP_TCC_READ_REQ)",
                         "valu_issue_busy": {"single_field": [0.92, 1.0], "two_field": [0.81, 0.89]},
                         "ta_busy": [0.74, 0.80], "l1_to_l2_bytes_over_algorithmic": 3.2,
                         "issue_floor_cycles_per_node_wave": 3800},
            "note": "measured in a separate leg of min(steps, 10) steps after the timed region (event pairs per phase)"}
        # ... and the family's counters measured in THIS run (three rocprofv3 --pmc child passes over a few full steps on this
        # grid); the committed ones above stay beside them
        if rank == 0 and not args.no_measure_traffic and nx == ny == nz_global and nx <= 256:
            live, why = measure_gather_counters(nx)
            if live:
                live["source"] = ("MEASURED IN THIS RUN: three child passes of rocprofv3 --pmc (SQ_INSTS_VALU; TA_BUSY_avr; TCP_TCC_READ_REQ_sum) "
                                  f"over 5 steps of tools/step_child.py --n {nx} (same library; 6 Jacobi iterations per step -- the gather "
                                  "launches are the same)")
                line["roofline_gather"]["counters_live"] = live
            else:
                line["roofline_gather"]["counters_live"] = {"source": f"not available: {why}"}
    # ---- the one-GPU point of the 512^3 strong-scaling curve (N > 1 runs ONE 512^3 grid in N slabs; this run's headline is
    # config 3's 256^3): a short leg of the same scene at 512^3 on this GPU, so that SCALE's N >= 2 values have their anchor ----
    if rank == 0 and world == 1 and not emul and not args.no_extra and args.scene == "smoke" and args.scheme == "bimocq" and not mg \
            and (nx, ny, nz_global) == (256, 256, 256) and not args.dump:
        # ---- SURVEY 8(d)'s own window: "Mvoxels/s per step averaged over steps 20-200" of a FRESH run of the scene.  The
        # headline above is whatever --steps / --warmup asked for (the driver's command times steps 5-25, each with one DMC
        # sub-step; later steps take two); this leg is the metric as the survey defines it, whatever the command line was.
        try:
            s.close()
            sm = BimocqGPUSolver(nx, ny, nz_global, 1.0, 0.0, 1.0, device=local_rank, scheme=0)
            sm.setSmoke(0.0, 1.0, rising_smoke(nx, 1.0 / nx))
            sm.setProjection(args.jacobi_iters, args.halfrdx, 0)
            sm.setOption(3, 1)
            for f_ in range(20):
                sm.advance(f_, dt)
            barrier()
            tm = time.perf_counter()
            for f_ in range(20, 200):
                sm.advance(f_, dt)
            barrier()
            elm = time.perf_counter() - tm
            sm.close()
            line.setdefault("extra", {})["survey_metric"] = {
                "value": round(voxels * 180 / elm / 1e6, 2), "unit": "Mvoxels/s", "ms_per_step": round(elm / 180 * 1e3, 3),
                "steps": "20-200 of a fresh run", "note": "SURVEY 8(d): Mvoxels/s per step averaged over steps 20-200 (full state, exact "
                "arithmetic, 200 Jacobi iterations); the headline `value` times the steps the command line names"}
        except Exception as e:                  # extra information only
            line.setdefault("extra", {})["survey_metric"] = {"value": None, "note": f"failed: {e}"}
        # ---- the two arithmetic variants side by side ("report separately for the exact and the fast variant"):
        # the one-fma variant with the z-marching field-window kernels, steps 20-60 of a fresh run, and its deviation from the
        # exact fields where the north star states its tolerance: 128^3 after 200 steps (both trajectories on this GPU; the
        # exact one is the trajectory tests/golden pins against the CPU oracle)
        try:
            import numpy as np
            L_ = bq._lib
            legs_v = {}
            for tag, fast_, win_ in (("exact", 0, 0), ("fast", 1, 0), ("fast_window", 1, 1)):
                lib.fl_set_option(L_.FL_OPT_FAST_LERP, fast_)
                lib.fl_set_option(L_.FL_OPT_FIELD_WINDOW, win_)
                sv = BimocqGPUSolver(nx, ny, nz_global, 1.0, 0.0, 1.0, device=local_rank, scheme=0)
                sv.setSmoke(0.0, 1.0, rising_smoke(nx, 1.0 / nx))
                sv.setProjection(args.jacobi_iters, args.halfrdx, 0)
                sv.setOption(3, 1)
                for f_ in range(20):
                    sv.advance(f_, dt)
                sv.setOption(8, 1)
                sv.phaseMs(reset=True)
                barrier()
                tv = time.perf_counter()
                for f_ in range(20, 60):
                    sv.advance(f_, dt)
                barrier()
                elv = time.perf_counter() - tv
                ph, pst = sv.phaseMs(reset=True)
                sv.close()
                legs_v[tag] = {"value": round(voxels * 40 / elv / 1e6, 2), "ms_per_step": round(elv / 40 * 1e3, 3),
                               "phase_ms_per_step": {k: round(v / max(1, pst), 3) for k, v in ph.items()}}
            n1 = 128
            fields_ = {}
            for tag, fast_, win_ in (("exact", 0, 0), ("fast_window", 1, 1)):
                lib.fl_set_option(L_.FL_OPT_FAST_LERP, fast_)
                lib.fl_set_option(L_.FL_OPT_FIELD_WINDOW, win_)
                sv = BimocqGPUSolver(n1, n1, n1, 1.0, 0.0, 1.0, device=local_rank, scheme=0)
                sv.setSmoke(0.0, 1.0, rising_smoke(n1, 1.0 / n1))
                sv.setProjection(200, 0.5, 0)
                sv.setOption(3, 1)
                for f_ in range(200):
                    sv.advance(f_, 2.0 / n1)
                fields_[tag] = {k: sv.field(k).astype(np.float64) for k in ("rho", "u", "v", "w")}
                sv.close()
            rms = {k: float(np.sqrt(np.mean((fields_["exact"][k] - fields_["fast_window"][k]) ** 2))) for k in ("rho", "u", "v", "w")}
            line.setdefault("extra", {})["arithmetic_variants"] = {
                "unit": "Mvoxels/s", "grid": [nx, ny, nz_global], "steps": "20-60 of a fresh run each, full state, phases timed",
                "exact": legs_v["exact"], "fast": legs_v["fast"], "fast_window": legs_v["fast_window"],
                "rms_fast_vs_exact_128_cubed_after_200_steps": rms, "rms_tolerance": 1e-5,
                "note": "exact: the reference's double-rounded lerp (the headline's arithmetic); fast: FL_OPT_FAST_LERP = 1, one fp32 fma per "
                        "lerp, bit-identical to the oracle in ITS fast mode, within the north star's 1e-5 RMS of the exact fields; "
                        "fast_window: the same arithmetic with FL_OPT_FIELD_WINDOW = 1 (z-marching gather kernels, sampled field in an LDS "
                        "window: same bits as `fast`)"}
        except Exception as e:                  # extra information only
            line.setdefault("extra", {})["arithmetic_variants"] = {"note": f"failed: {e}"}
        finally:
            lib.fl_set_option(bq._lib.FL_OPT_FAST_LERP, 0)
            lib.fl_set_option(bq._lib.FL_OPT_FIELD_WINDOW, -1)
        try:
            big = 512
            s5 = BimocqGPUSolver(big, big, big, 1.0, 0.0, 1.0, device=local_rank, scheme=0)
            s5.setSmoke(0.0, 1.0, rising_smoke(big, 1.0 / big))
            s5.setProjection(args.jacobi_iters, args.halfrdx, 0)
            s5.setOption(3, 1)
            n5, w5 = max(1, min(8, args.steps)), max(1, min(4, args.warmup))
            f5 = 0
            for _ in range(w5):
                s5.advance(f5, 2.0 / big); f5 += 1
            lib.fl_jacobi_profile(None, None, None)         # (reset)
            lib.fl_set_option(bq._lib.FL_OPT_PROFILE_JACOBI, 1)
            barrier()
            t5 = time.perf_counter()
            for _ in range(n5):
                s5.advance(f5, 2.0 / big); f5 += 1
            barrier()
            el5 = time.perf_counter() - t5
            lib.fl_set_option(bq._lib.FL_OPT_PROFILE_JACOBI, 0)
            ms5, l5, sw5 = C.c_double(0.0), C.c_longlong(0), C.c_longlong(0)
            lib.fl_jacobi_profile(C.byref(ms5), C.byref(l5), C.byref(sw5))
            k5 = (lib.fl_jacobi_kernel_name() or b"").decode()
            s5.close()
            roof5 = None
            if l5.value > 0 and sw5.value > 0:
                us5 = ms5.value * 1e3 / l5.value
                comp5 = JACOBI_BYTES_PER_VOXEL * big ** 3
                roof5 = {"bound": "hbm", "kernel": k5, "achieved": round(comp5 / (us5 * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(comp5 / (us5 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "compulsory_bytes_per_launch": int(comp5), "working_set": "hbm (p, p', div = 1.6 GB, six times the Infinity Cache)",
                         "us_per_launch": round(us5, 3), "launches_timed": int(l5.value), "sweeps_per_launch": round(sw5.value / l5.value, 3),
                         "us_per_sweep": round(ms5.value * 1e3 / sw5.value, 3),
                         "note": "the HBM-resident counterpart of the headline's roofline object: compulsory bytes of a fused launch (p and "
                                 "div read once, p' written once) over its event-timed duration inside these steps"}
            line.setdefault("extra", {})["single_gpu_512_anchor"] = {
                "roofline": roof5,
                "value": round(big ** 3 * n5 / el5 / 1e6, 2), "unit": "Mvoxels/s", "ms_per_step": round(el5 / n5 * 1e3, 3), "steps": n5,
                "warmup": w5, "note": "BASELINE config 4's grid (512^3 rising smoke, 200 Jacobi iterations, full state) on ONE GPU: the N = 1 "
                                      "point of the strong-scaling curve that `bench.py --gpus N` (N > 1) measures on the same grid; early "
                                      "steps of the scene (steps " + f"{w5}-{w5 + n5}" + ")"}
        except Exception as e:                  # extra information only
            line.setdefault("extra", {})["single_gpu_512_anchor"] = {"value": None, "note": f"failed: {e}"}
    if rank == 0 and world == 1 and not emul and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:                  # never lose the GPU numbers to a host-side hiccup
            line["cpu_baseline"] = {"value": None, "unit": "Mvoxels/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        # (printed BEFORE the communicators are torn down: a teardown that never returns must not cost the measured line)
        if line["config"]["nonfinite_velocity_seen"]:
            print("[bench] WARNING: a NaN or an Inf appeared in the velocity field during this run (fl_nonfinite_seen): the "
                  "timings are those of a broken simulation", file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    WATCHDOG["printed"] = True             # (every rank: a watchdog that fires during the teardown exits with 0)
    stage("result printed; closing the solver and the communicators")
    s.close()
    if dist is not None:
        dist.barrier()
        lib.fl_comm_destroy()
        dist.destroy_process_group()
    stage("done")
    # (no explicit fl_shutdown: the library releases its streams, events and cached graphs itself at process exit --
    # fl_shutdown_all, registered by fl_init and by gpufluidsimulation_amd._lib)


if __name__ == "__main__":
    main()
