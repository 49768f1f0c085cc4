"""Python mirror of the reference's gpuMapper (src/bimocq3D/GPU_Advection.h:110-627).

Same method names, argument order and pre-zeroing behaviour as the reference class, so the
parity tests read like calls into the reference.  It owns nothing but device buffers and calls
only the C-ABI; numpy arrays cross the boundary through DeviceBuffer.
"""
import numpy as np

from . import _lib


class DeviceBuffer:
    """A zero-filled device allocation of `count` fp32 (allocGPUBuffer, GPU_Advection.h:322-326)."""

    def __init__(self, count, lib=None):
        self.lib = lib or _lib.hip_lib()
        self.count = int(count)
        self.nbytes = self.count * 4
        self.ptr = self.lib.fl_malloc(self.nbytes)
        if not self.ptr:
            _lib.check(self.lib)
            raise MemoryError(f"fl_malloc({self.nbytes}) failed")

    @classmethod
    def from_numpy(cls, a, lib=None):
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        b = cls(a.size, lib)
        b.upload(a)
        return b

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        assert a.size == self.count, (a.size, self.count)
        self.lib.fl_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes)

    def numpy(self):
        out = np.empty(self.count, dtype=np.float32)
        self.lib.fl_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes)
        return out

    def zero(self):
        self.lib.fl_memset(self.ptr, 0, self.nbytes)

    def free(self):
        if self.ptr:
            self.lib.fl_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GpuMapper:
    """gpuMapper: scratch buffers + one thin method per gpu_* operator."""

    def __init__(self, nx, ny, nz, h, device=0):
        self.lib = _lib.hip_lib()
        rc = self.lib.fl_init(device)                       # cudaInit(), GPU_Advection.h:214-226
        if rc != _lib.FL_OK:
            _lib.check(self.lib)
        self.ni, self.nj, self.nk, self.h = nx, ny, nz, float(np.float32(h))
        n, nu, nv, nw = self.sizes()
        # GPU_Advection.h:122-136: compensation scratch and the DMC ping buffers
        self.u_src, self.v_src, self.w_src = DeviceBuffer(nu), DeviceBuffer(nv), DeviceBuffer(nw)
        self.x_out, self.y_out, self.z_out = DeviceBuffer(n), DeviceBuffer(n), DeviceBuffer(n)

    def sizes(self):
        ni, nj, nk = self.ni, self.nj, self.nk
        return ni * nj * nk, (ni + 1) * nj * nk, ni * (nj + 1) * nk, ni * nj * (nk + 1)

    def _g(self):
        return (self.h, self.ni, self.nj, self.nk)

    def check(self):
        _lib.check(self.lib)

    # GPU_Advection.h:453-458
    def solveForward(self, u, v, w, x_fwd, y_fwd, z_fwd, cfldt, dt):
        self.lib.gpu_solve_forward(u.ptr, v.ptr, w.ptr, x_fwd.ptr, y_fwd.ptr, z_fwd.ptr, *self._g(), cfldt, dt)

    # GPU_Advection.h:460-470: kernel into x_out.., then copy back into the in/out maps
    def solveBackwardDMC(self, u, v, w, x, y, z, substep):
        self.lib.gpu_solve_backwardDMC(u.ptr, v.ptr, w.ptr, x.ptr, y.ptr, z.ptr,
                                       self.x_out.ptr, self.y_out.ptr, self.z_out.ptr, *self._g(), substep)
        for dst, src in ((x, self.x_out), (y, self.y_out), (z, self.z_out)):
            self.lib.fl_memcpy_d2d(dst.ptr, src.ptr, dst.nbytes)

    # GPU_Advection.h:472-482
    def advectVelocity(self, u, v, w, u_init, v_init, w_init, bx, by, bz, is_point=False):
        for f in (u, v, w):
            f.zero()
        self.lib.gpu_advect_velocity(u.ptr, v.ptr, w.ptr, u_init.ptr, v_init.ptr, w_init.ptr,
                                     bx.ptr, by.ptr, bz.ptr, *self._g(), is_point)

    # GPU_Advection.h:484-491
    def advectVelocityDouble(self, u, v, w, ut, vt, wt, bx, by, bz, bxp, byp, bzp, is_point, blend):
        self.lib.gpu_advect_vel_double(u.ptr, v.ptr, w.ptr, ut.ptr, vt.ptr, wt.ptr, bx.ptr, by.ptr, bz.ptr,
                                       bxp.ptr, byp.ptr, bzp.ptr, *self._g(), is_point, blend)

    # GPU_Advection.h:493-503
    def compensateVelocity(self, u, v, w, du, dv, dw, fx, fy, fz, bx, by, bz, is_point=False):
        for f in (self.u_src, self.v_src, self.w_src):
            f.zero()
        self.lib.gpu_compensate_velocity(u.ptr, v.ptr, w.ptr, du.ptr, dv.ptr, dw.ptr,
                                         self.u_src.ptr, self.v_src.ptr, self.w_src.ptr,
                                         fx.ptr, fy.ptr, fz.ptr, bx.ptr, by.ptr, bz.ptr, *self._g(), is_point)

    # GPU_Advection.h:505-511 (the reference zeroes (ni+1)*nj*nk floats of an ni*nj*nk buffer; not replicated)
    def advectField(self, field, field_init, bx, by, bz, is_point=False):
        field.zero()
        self.lib.gpu_advect_field(field.ptr, field_init.ptr, bx.ptr, by.ptr, bz.ptr, *self._g(), is_point)

    # GPU_Advection.h:513-519
    def advectFieldDouble(self, field, field_prev, bx, by, bz, bxp, byp, bzp, is_point, blend):
        self.lib.gpu_advect_field_double(field.ptr, field_prev.ptr, bx.ptr, by.ptr, bz.ptr,
                                         bxp.ptr, byp.ptr, bzp.ptr, *self._g(), is_point, blend)

    # GPU_Advection.h:521-528 (u_src doubles as the scalar scratch, sized for the u buffer)
    def compensateField(self, f, df, fx, fy, fz, bx, by, bz, is_point=False):
        self.u_src.zero()
        self.lib.gpu_compensate_field(f.ptr, df.ptr, self.u_src.ptr, fx.ptr, fy.ptr, fz.ptr,
                                      bx.ptr, by.ptr, bz.ptr, *self._g(), is_point)

    # GPU_Advection.h:530-542
    def semilagAdvectVelocity(self, uo, vo, wo, us, vs, ws, u, v, w, cfldt, dt):
        for f in (uo, vo, wo):
            f.zero()
        for out, src, d in ((uo, us, (1, 0, 0)), (vo, vs, (0, 1, 0)), (wo, ws, (0, 0, 1))):
            self.lib.gpu_semilag(out.ptr, src.ptr, u.ptr, v.ptr, w.ptr, *d, *self._g(), cfldt, dt)

    # GPU_Advection.h:544-551
    def semilagAdvectField(self, field, field_src, u, v, w, dx, dy, dz, cfldt, dt):
        field.zero()
        self.lib.gpu_semilag(field.ptr, field_src.ptr, u.ptr, v.ptr, w.ptr, dx, dy, dz, *self._g(), cfldt, dt)

    # GPU_Advection.h:553-558
    def emitSmoke(self, u, v, w, rho, T, cx, cy, cz, radius, density, temperature, emiter):
        self.lib.gpu_emit_smoke(u.ptr, v.ptr, w.ptr, rho.ptr, T.ptr, *self._g(),
                                cx, cy, cz, radius, density, temperature, emiter)

    # GPU_Advection.h:560-564
    def add_buoyancy(self, v, rho, T, alpha, beta, dt):
        self.lib.gpu_add_buoyancy(v.ptr, rho.ptr, T.ptr, self.ni, self.nj, self.nk, alpha, beta, dt)

    # GPU_Advection.h:566-571 (ni,nj,nk here are BUFFER dims)
    def diffuseField(self, field, tmp0, tmp1, ni, nj, nk, iters, coef):
        self.lib.gpu_diffuse_field(field.ptr, tmp0.ptr, tmp1.ptr, ni, nj, nk, iters, coef)

    # GPU_Advection.h:573-576, 410-418
    def addFields(self, out, f1, f2, coeff, number):
        self.lib.gpu_add_field(out.ptr, f1.ptr, f2.ptr, coeff, number)

    def add(self, f1, f2, coeff, number):
        self.lib.gpu_add(f1.ptr, f2.ptr, coeff, number)

    def mad(self, out, f1, f2, c1, c2, number):
        self.lib.gpu_mad(out.ptr, f1.ptr, f2.ptr, c1, c2, number)

    # GPU_Advection.h:578-585
    def estimateDistortionCUDA(self, dist, xb, yb, zb, xf, yf, zf):
        dist.zero()
        self.lib.gpu_estimate_distortion(dist.ptr, xb.ptr, yb.ptr, zb.ptr, xf.ptr, yf.ptr, zf.ptr, *self._g())

    # GPU_Advection.h:587-600
    def accumulateVelocity(self, uc, vc, wc, dui, dvi, dwi, fx, fy, fz, is_point, coeff):
        self.lib.gpu_accumulate_velocity(uc.ptr, vc.ptr, wc.ptr, dui.ptr, dvi.ptr, dwi.ptr,
                                         fx.ptr, fy.ptr, fz.ptr, *self._g(), is_point, coeff)

    def accumulateField(self, fc, dfi, fx, fy, fz, is_point, coeff):
        self.lib.gpu_accumulate_field(fc.ptr, dfi.ptr, fx.ptr, fy.ptr, fz.ptr, *self._g(), is_point, coeff)

    # GPU_Advection.h:602-608
    def projectionJacobi(self, u, v, w, div, p, p_temp, debug, iters, halfrdx, alpha, beta):
        for f in (div, p, p_temp):
            f.zero()
        self.lib.gpu_projection_jacobi(u.ptr, v.ptr, w.ptr, div.ptr, p.ptr, p_temp.ptr,
                                       debug.ptr if debug is not None else None,
                                       self.ni, self.nj, self.nk, iters, halfrdx, alpha, beta)
