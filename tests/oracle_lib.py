"""ctypes binding of the CPU oracle (oracle/bimocq_oracle.c) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

FP = C.POINTER(C.c_float)
c_f, c_i = C.c_float, C.c_int


def usable_cores():
    """CPU threads this job may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job 16 cores of a much bigger host)."""
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
    return max(1, cores)


def limit_openmp_threads():
    """libgomp starts one thread per visible core; beyond the job's CPU share every barrier of the oracle's small
    parallel loops crawls (measured on a GPU box: 35 s instead of 0.3 s per 32^3 step).  Must run before the first
    OpenMP library is loaded; an explicit OMP_NUM_THREADS wins."""
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, usable_cores())))


SANITIZE = os.environ.get("BQ_SANITIZE", "0") not in ("", "0")     # see tests/build_cpu_host.py, `make sanitize`


def build(march="x86-64", out="_build"):
    """(re)build liboracle.so with gcc; returns its path."""
    extra = []
    if SANITIZE:
        out, extra = out + "_san", ["SANITIZE=1", "OPT=-O1"]
    so = os.path.join(ORACLE_DIR, out, "liboracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("bimocq_oracle.c", "mgcg_oracle.c", "bimocq_oracle.h", "Makefile")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, f"MARCH={march}", f"OUT={out}", *extra])
    return so


class CoarseLevel(C.Structure):
    """SCoarseLevelInfo (GPU_Advection.h:15-24)"""
    _fields_ = [("ni", c_i), ("nj", c_i), ("nk", c_i), ("number", c_i), ("alpha", C.c_double), ("beta", C.c_double),
                ("b", C.c_void_p), ("x", C.c_void_p), ("r", C.c_void_p)]


DP = C.POINTER(C.c_double)


def dp(a):
    """double* of a contiguous float64 numpy array"""
    assert a.dtype.name == "float64" and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(DP)


def level_dims(ni, nj, nk, count):
    """the reference's level pyramid (BimocqGPUSolver.cpp:68-90): n -> (n - 1) / 2"""
    dims = [(ni, nj, nk)]
    for _ in range(1, count):
        a, b, c = dims[-1]
        dims.append(((a - 1) // 2, (b - 1) // 2, (c - 1) // 2))
    return dims


class Emitter(C.Structure):
    _fields_ = [("cx", c_f), ("cy", c_f), ("cz", c_f), ("radius", c_f), ("density", c_f),
                ("temperature", c_f), ("emiter", c_f), ("emit_frames", c_i)]


_SIGS = {
    "orc_expf": (c_f, [c_f]),
    "orc_lerp": (c_f, [c_f, c_f, c_f]),
    "orc_set_fast_lerp": (None, [c_i]),
    "orc_sample": (c_f, [FP, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "orc_solve_forward": (None, [FP] * 6 + [c_f, c_i, c_i, c_i, c_f, c_f]),
    "orc_solve_backwardDMC": (None, [FP] * 9 + [c_f, c_i, c_i, c_i, c_f]),
    "orc_advect_velocity": (None, [FP] * 9 + [c_f, c_i, c_i, c_i, c_i]),
    "orc_advect_vel_double": (None, [FP] * 12 + [c_f, c_i, c_i, c_i, c_i, c_f]),
    "orc_advect_vel_double_global": (None, [FP] * 12 + [c_f, c_i, c_i, c_i, c_i, c_f]),
    "orc_advect_field_double_global": (None, [FP] * 8 + [c_f, c_i, c_i, c_i, c_i, c_f]),
    "orc_advect_field": (None, [FP] * 5 + [c_f, c_i, c_i, c_i, c_i]),
    "orc_advect_field_double": (None, [FP] * 8 + [c_f, c_i, c_i, c_i, c_i, c_f]),
    "orc_accumulate_velocity": (None, [FP] * 9 + [c_f, c_i, c_i, c_i, c_i, c_f]),
    "orc_accumulate_field": (None, [FP] * 5 + [c_f, c_i, c_i, c_i, c_i, c_f]),
    "orc_estimate_distortion": (None, [FP] * 7 + [c_f, c_i, c_i, c_i]),
    "orc_add": (None, [FP, FP, c_f, c_i]),
    "orc_compensate_velocity": (None, [FP] * 15 + [c_f, c_i, c_i, c_i, c_i]),
    "orc_compensate_field": (None, [FP] * 9 + [c_f, c_i, c_i, c_i, c_i]),
    "orc_compensate_error_velocity": (None, [FP] * 12 + [c_f, c_i, c_i, c_i, c_i]),
    "orc_compensate_error_field": (None, [FP] * 6 + [c_f, c_i, c_i, c_i, c_i]),
    "orc_jacobi_sweep_range": (None, [FP] * 3 + [c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "orc_multi_grid_conjugate_gradient": (None, [FP] * 3 + [DP] * 7 + [C.POINTER(CoarseLevel), c_i, c_i, C.c_double]),
    "orc_mg_divergence": (None, [FP] * 3 + [DP, c_i, c_i, c_i, C.c_double]),
    "orc_mg_poisson": (None, [DP, DP, c_i, c_i, c_i]),
    "orc_mg_residual": (None, [DP, DP, DP, c_i, c_i, c_i]),
    "orc_mg_dot_partials": (None, [DP, DP, DP, C.c_long]),
    "orc_mg_calc_sum": (None, [DP, DP, C.c_long, C.c_long, c_i]),
    "orc_mg_calc_max": (None, [DP, DP, C.c_long, c_i]),
    "orc_mg_smooth": (None, [DP, DP, DP, C.c_double, C.c_double, c_i, c_i, c_i, c_i]),
    "orc_mg_restrict": (None, [DP, DP] + [c_i] * 6),
    "orc_mg_prolong": (None, [DP, DP] + [c_i] * 6),
    "orc_clamp_extrema": (None, [FP] * 5 + [c_i] * 6 + [c_f] * 5),
    "orc_semilag": (None, [FP] * 5 + [c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_f, c_f]),
    "orc_emit_smoke": (None, [FP] * 5 + [c_f, c_i, c_i, c_i] + [c_f] * 7),
    "orc_add_buoyancy": (None, [FP] * 3 + [c_i, c_i, c_i, c_f, c_f, c_f]),
    "orc_diffuse_field": (None, [FP] * 3 + [c_i, c_i, c_i, c_i, c_f]),
    "orc_add_field": (None, [FP, FP, FP, c_f, c_i]),
    "orc_mad": (None, [FP, FP, FP, c_f, c_f, c_i]),
    "orc_clamp_extrema_box": (None, [FP, FP, c_i, c_i, c_i]),
    "orc_divergence": (None, [FP] * 4 + [c_i, c_i, c_i, c_f]),
    "orc_jacobi_sweep": (None, [FP] * 3 + [c_i, c_i, c_i, c_f, c_f]),
    "orc_gradient": (None, [FP, FP, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "orc_residual_norms": (None, [FP, FP, c_i, c_i, c_i, C.POINTER(C.c_double), FP]),
    "orc_projection_jacobi": (None, [FP] * 7 + [c_i, c_i, c_i, c_i, c_f, c_f, c_f]),
    "orc_max_abs3": (c_f, [FP] * 3 + [c_i, c_i, c_i]),
    "orc_solver_create": (C.c_void_p, [c_i, c_i, c_i, c_f, c_f, c_f]),
    "orc_solver_destroy": (None, [C.c_void_p]),
    "orc_solver_set_smoke": (None, [C.c_void_p, c_f, c_f, C.POINTER(Emitter), c_i]),
    "orc_solver_set_projection": (None, [C.c_void_p, c_i, c_f]),
    "orc_solver_reinit_counts": (c_i, [C.c_void_p, c_i]),
    "orc_solver_last_distortion": (c_f, [C.c_void_p, c_i]),
    "orc_solver_set_projection_kind": (None, [C.c_void_p, c_i, c_i]),
    "orc_solver_mg_levels": (c_i, [C.c_void_p]),
    "orc_solver_mg_history": (DP, [C.c_void_p]),
    "orc_solver_set_option": (None, [C.c_void_p, c_i, c_i]),
    "orc_solver_advance": (None, [C.c_void_p, c_i, c_f]),
    "orc_solver_field": (FP, [C.c_void_p, c_i, C.POINTER(C.c_long)]),
    "orc_solver_last_cfldt": (c_f, [C.c_void_p]),
}

_lib = None


def lib(march="x86-64", out="_build"):
    global _lib
    if _lib is None:
        limit_openmp_threads()
        _lib = C.CDLL(build(march, out))
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def fp(a):
    """float32 C-contiguous ndarray -> float* (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(FP)


FIELD_IDS = {"rho": 0, "T": 1, "u": 2, "v": 3, "w": 4, "uinit": 5, "vinit": 6, "winit": 7,
             "rhoinit": 8, "Tinit": 9, "fx": 10, "fy": 11, "fz": 12, "bx": 13, "by": 14, "bz": 15, "p": 16, "div": 17}


class OracleSolver:
    """BimocqGPUSolver surface (advance + field access) on the CPU oracle."""

    def __init__(self, ni, nj, nk, L=1.0, viscosity=0.0, blend=1.0):
        self.l = lib()
        self.ni, self.nj, self.nk = ni, nj, nk
        self.h = np.float32(L) / np.float32(ni)
        self.s = self.l.orc_solver_create(ni, nj, nk, L, viscosity, blend)

    def set_smoke(self, drop, rise, emitters):
        arr = (Emitter * max(1, len(emitters)))()
        for i, e in enumerate(emitters):
            arr[i] = Emitter(*e)
        self.l.orc_solver_set_smoke(self.s, drop, rise, arr, len(emitters))

    def set_projection(self, iters, halfrdx, kind=0):
        self.l.orc_solver_set_projection(self.s, iters if kind == 0 else 100, halfrdx)
        self.l.orc_solver_set_projection_kind(self.s, kind, iters)

    def mg_history(self):
        p = self.l.orc_solver_mg_history(self.s)
        return np.ctypeslib.as_array(p, shape=(4096,)).copy() if p else None

    def set_option(self, option, value):
        self.l.orc_solver_set_option(self.s, option, value)

    def reinit_counts(self):
        return self.l.orc_solver_reinit_counts(self.s, 0), self.l.orc_solver_reinit_counts(self.s, 1)

    def last_distortion(self):
        return self.l.orc_solver_last_distortion(self.s, 0), self.l.orc_solver_last_distortion(self.s, 1)

    def advance(self, frame, dt):
        self.l.orc_solver_advance(self.s, frame, dt)

    def field(self, name):
        cnt = C.c_long(0)
        p = self.l.orc_solver_field(self.s, FIELD_IDS[name], C.byref(cnt))
        return np.ctypeslib.as_array(p, shape=(cnt.value,)).copy()

    @property
    def cfldt(self):
        return self.l.orc_solver_last_cfldt(self.s)

    def close(self):
        if self.s:
            self.l.orc_solver_destroy(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
