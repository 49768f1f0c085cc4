import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import gpufluidsimulation_amd as bq
from gpufluidsimulation_amd import DeviceBuffer
lib = bq.hip_lib(); assert lib.fl_init(0) == 0
L = bq._lib
ok = True
for (nx, ny, nz, sweeps) in [(64, 37, 29, 6), (256, 64, 40, 7), (32, 8, 9, 4), (128, 5, 17, 3), (256, 256, 64, 10), (96, 33, 50, 8)]:
    n = nx * ny * nz
    rng = np.random.default_rng(nx + ny)
    p0 = rng.standard_normal(n, dtype=np.float32); d0 = rng.standard_normal(n, dtype=np.float32)
    # both ping-pong buffers must carry the same boundary layer
    res = []
    for rows in (1, 2):
        for kc in (0, 8, 5):
            lib.fl_set_option(L.FL_OPT_JACOBI_VARIANT, 0); lib.fl_set_option(L.FL_OPT_JACOBI_FUSE, 2)
            lib.fl_set_option(L.FL_OPT_JACOBI_ROWS, rows); lib.fl_set_option(L.FL_OPT_JACOBI_KCHUNK2, kc)
            p = DeviceBuffer.from_numpy(p0); t = DeviceBuffer.from_numpy(p0); d = DeviceBuffer.from_numpy(d0)
            where = lib.gpu_jacobi_sweeps(p.ptr, d.ptr, t.ptr, nx, ny, nz, sweeps, -1.0, 1.0 / 6.0)
            res.append(((rows, kc), (t if where else p).numpy().copy()))
    lib.fl_set_option(L.FL_OPT_JACOBI_FUSE, 0); lib.fl_set_option(L.FL_OPT_JACOBI_VARIANT, 1)
    p = DeviceBuffer.from_numpy(p0); t = DeviceBuffer.from_numpy(p0); d = DeviceBuffer.from_numpy(d0)
    where = lib.gpu_jacobi_sweeps(p.ptr, d.ptr, t.ptr, nx, ny, nz, sweeps, -1.0, 1.0 / 6.0)
    ref = (t if where else p).numpy().copy()
    for key, a in res:
        same = np.array_equal(a, ref)
        ok &= same
        print((nx, ny, nz, sweeps), key, "OK" if same else f"MISMATCH {np.abs(a-ref).max()} at {np.count_nonzero(a!=ref)}")
bq.check()
print("ALL OK" if ok else "FAILED")
