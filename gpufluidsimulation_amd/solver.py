"""Python handle on the C++ host solver (csrc/host/, C view in include/bimocq_solver.h).

Same surface as the reference's BimocqGPUSolver (src/bimocq3D/BimocqGPUSolver.h:27-56):
construct, setSmoke, advance(framenum, dt), outputResult(frame, path).
"""
import ctypes as C
import os

import numpy as np

from . import _lib

FIELD_IDS = {"rho": 0, "T": 1, "u": 2, "v": 3, "w": 4, "uinit": 5, "vinit": 6, "winit": 7,
             "rhoinit": 8, "Tinit": 9, "fx": 10, "fy": 11, "fz": 12, "bx": 13, "by": 14, "bz": 15, "p": 16, "div": 17}


class Emitter(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("radius", C.c_float),
                ("density", C.c_float), ("temperature", C.c_float), ("emiter", C.c_float),
                ("emit_frames", C.c_int)]


HOST_SIGS = {
    "bq_solver_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]),
    "bq_solver_create_slab": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                          C.c_int, C.c_int, C.c_int]),
    "bq_solver_slab_info": (None, [C.c_void_p, C.POINTER(C.c_int)]),
    "bq_solver_destroy": (None, [C.c_void_p]),
    "bq_solver_set_smoke": (None, [C.c_void_p, C.c_float, C.c_float, C.POINTER(Emitter), C.c_int]),
    "bq_solver_set_projection": (None, [C.c_void_p, C.c_int, C.c_int, C.c_float]),
    "bq_solver_set_option": (None, [C.c_void_p, C.c_int, C.c_int]),
    "bq_solver_get_option": (C.c_int, [C.c_void_p, C.c_int]),
    "bq_solver_advance": (None, [C.c_void_p, C.c_int, C.c_float]),
    "bq_solver_output_result": (C.c_long, [C.c_void_p, C.c_uint, C.c_char_p]),
    "bq_solver_output_result_async": (C.c_int, [C.c_void_p, C.c_uint, C.c_char_p]),
    "bq_solver_output_wait": (C.c_long, [C.c_void_p]),
    "bq_solver_download": (C.c_long, [C.c_void_p, C.c_int, C.c_void_p, C.c_long]),
    "bq_solver_reinit_counts": (C.c_int, [C.c_void_p, C.c_int]),
    "bq_solver_last_distortion": (C.c_float, [C.c_void_p, C.c_int]),
    "bq_solver_mg_history": (C.c_long, [C.c_void_p, C.POINTER(C.c_double), C.c_long]),
    "bq_solver_last_cfldt": (C.c_float, [C.c_void_p]),
    "bq_solver_last_ms": (C.c_float, [C.c_void_p]),
    "bq_solver_reinit_count": (C.c_int, [C.c_void_p]),
    "bq_solver_phase_ms": (C.c_longlong, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
}
PHASES = ("maps", "advect_compensate", "forces", "projection", "accumulate_reinit")

_host = None


def bind_host(lib):
    for name, (res, args) in HOST_SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def host_lib():
    """libbimocq_host.so (C++ solver); pulls in libbimocq_hip.so through its rpath."""
    global _host
    if _host is None:
        _lib.hip_lib()
        if not os.path.exists(_lib.HOST_SO):
            raise _lib.BimocqLibraryMissing(f"{_lib.HOST_SO} not found: run `make`")
        _host = bind_host(C.CDLL(_lib.HOST_SO))
    return _host


class BimocqGPUSolver:
    """advance()/outputResult() on the MI355X; `lib`/`errlib` are injectable so the CPU-only tests
    can drive the very same host code linked against a CPU stand-in of the C-ABI."""

    def __init__(self, nx, ny, nz, L=1.0, viscosity=0.0, blend=1.0, device=0, lib=None, errlib=None,
                 rank=0, nranks=1, ghost=0, scheme=0):
        """nz is the GLOBAL plane count; with nranks > 1 (or ghost > 0) this object is one z-slab rank
        (set the communicator up first: gpufluidsimulation_amd.transport)."""
        self.lib = lib or host_lib()
        self.errlib = errlib or (_lib.hip_lib() if lib is None else lib)
        self.nx, self.ny, self.nz = nx, ny, nz
        self.h = float(np.float32(L) / np.float32(nx))
        self.s = self.lib.bq_solver_create_slab(device, nx, ny, nz, L, viscosity, blend, scheme, rank, nranks, ghost)
        if not self.s:
            self._check()
            raise _lib.BimocqError("bq_solver_create failed")
        info = (C.c_int * 8)()
        self.lib.bq_solver_slab_info(self.s, info)
        (self.slab_on, self.rank, self.nranks, _, self.own0, self.own1, self.ghost, self.nk_local) = list(info)

    def _check(self):
        code = self.errlib.fl_last_error()
        if code != 0:
            text = self.errlib.fl_last_error_string()
            text = text.decode(errors="replace") if isinstance(text, bytes) else str(text)
            self.errlib.fl_clear_error()
            raise _lib.BimocqError(f"bimocq error {code}: {text}")

    def setSmoke(self, drop, rise, emitters):
        """emitters: iterable of (cx, cy, cz, radius, density, temperature, emiter, emit_frames)"""
        emitters = list(emitters)
        arr = (Emitter * max(1, len(emitters)))()
        for i, e in enumerate(emitters):
            arr[i] = Emitter(*e)
        self.lib.bq_solver_set_smoke(self.s, drop, rise, arr, len(emitters))

    def setProjection(self, iters, halfrdx, kind=0):
        """kind 0: Jacobi, iters sweeps; kind 1: fp64 multigrid-CG, iters outer iterations (reference: 50)"""
        self.lib.bq_solver_set_projection(self.s, kind, iters, halfrdx)
        self._check()

    def mgHistory(self):
        """tempResult of the last multigrid-CG projection (4096 doubles), or None before the first one"""
        n = self.lib.bq_solver_mg_history(self.s, None, 0)
        if not n:
            return None
        out = np.zeros(n, dtype=np.float64)
        self.lib.bq_solver_mg_history(self.s, out.ctypes.data_as(C.POINTER(C.c_double)), n)
        return out

    def setOption(self, option, value):
        """option 1 = BQ_OPT_KEEP_DMC_BORDER, 2 = BQ_OPT_REINIT_POLICY (0 every frame, 1 distortion-driven),
        3 = BQ_OPT_FULL_STATE, 4 = BQ_OPT_FUSED_HOUSEKEEPING, 5 = BQ_OPT_OVERLAP_EXCHANGES, 6 = BQ_OPT_SHALLOW_BLOCKING_EXCHANGE,
        7 = BQ_OPT_JACOBI_ENDS_FIRST, 8 = BQ_OPT_PROFILE_PHASES, 9 = BQ_OPT_REINIT_MAX_TRAVEL, 10 = BQ_OPT_JACOBI_TRIPLES (include/bimocq_solver.h)"""
        self.lib.bq_solver_set_option(self.s, option, value)
        self._check()

    def getOption(self, option):
        return self.lib.bq_solver_get_option(self.s, option)

    def phaseMs(self, reset=True):
        """BQ_OPT_PROFILE_PHASES (option 8): ({phase: ms summed over the profiled steps}, steps)"""
        ms = (C.c_double * len(PHASES))()
        steps = self.lib.bq_solver_phase_ms(self.s, ms, 1 if reset else 0)
        return dict(zip(PHASES, list(ms))), int(steps)

    def reinitCounts(self):
        """(velocity map re-initialisations, scalar map re-initialisations) so far"""
        return self.lib.bq_solver_reinit_counts(self.s, 0), self.lib.bq_solver_reinit_counts(self.s, 1)

    def forcedReinits(self):
        """re-initialisations caused by BQ_OPT_REINIT_MAX_TRAVEL (option 9) rather than by the policy's thresholds"""
        return self.lib.bq_solver_reinit_counts(self.s, 2)

    def lastDistortion(self):
        return self.lib.bq_solver_last_distortion(self.s, 0), self.lib.bq_solver_last_distortion(self.s, 1)

    def advance(self, framenum, dt):
        self.lib.bq_solver_advance(self.s, framenum, dt)

    def outputResult(self, frame, path=None):
        n = self.lib.bq_solver_output_result(self.s, frame, path.encode() if path else None)
        self._check()
        return n

    def outputResultAsync(self, frame, path=None):
        """start the dump of the current density without stalling the simulation; waitOutput() joins it"""
        ok = self.lib.bq_solver_output_result_async(self.s, frame, path.encode() if path else None)
        self._check()
        return bool(ok)

    def waitOutput(self):
        return self.lib.bq_solver_output_wait(self.s)

    def field(self, name):
        which = FIELD_IDS[name]
        count = self.lib.bq_solver_download(self.s, which, None, 0)
        out = np.empty(count, dtype=np.float32)
        self.lib.bq_solver_download(self.s, which, out.ctypes.data, count)
        self._check()
        return out

    def owned(self, name):
        """the planes this rank owns, as a flat array in global plane order (w on the last rank also
        carries its top plane) -- concatenating owned() over the ranks gives the single-GPU field"""
        a = self.field(name)
        if not self.slab_on:
            return a
        kind = {"u": (self.nx + 1) * self.ny, "uinit": (self.nx + 1) * self.ny,
                "v": self.nx * (self.ny + 1), "vinit": self.nx * (self.ny + 1)}.get(name, self.nx * self.ny)
        extra = 1 if name in ("w", "winit") and self.own1 == self.nz else 0
        return a[kind * self.ghost: kind * (self.ghost + self.own1 - self.own0 + extra)]

    @property
    def cfldt(self):
        return self.lib.bq_solver_last_cfldt(self.s)

    @property
    def last_ms(self):
        return self.lib.bq_solver_last_ms(self.s)

    @property
    def reinit_count(self):
        return self.lib.bq_solver_reinit_count(self.s)

    def close(self):
        if getattr(self, "s", None):
            self.lib.bq_solver_destroy(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_density_dump(path):
    """Reader of the BQDENS01 container written by outputResult (density_dump.cpp)."""
    hdr = np.dtype([("magic", "S8"), ("frame", "<u4"), ("nx", "<i4"), ("ny", "<i4"), ("nz", "<i4"),
                    ("k_offset", "<i4"), ("nz_local", "<i4"), ("voxel_size", "<f4"), ("threshold", "<f4"),
                    ("grid_name", "S16"), ("grid_class", "<u4"), ("count", "<u8")])
    rec = np.dtype([("i", "<i4"), ("j", "<i4"), ("k", "<i4"), ("value", "<f4")])
    with open(path, "rb") as f:
        h = np.frombuffer(f.read(hdr.itemsize), dtype=hdr)[0]
        assert h["magic"] == b"BQDENS01", h["magic"]
        r = np.frombuffer(f.read(), dtype=rec)
    assert len(r) == h["count"]
    return h, r
