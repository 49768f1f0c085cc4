// gpu_mapper.cpp -- see gpu_mapper.hpp.  Reference: src/bimocq3D/GPU_Advection.h:110-627.
#include "gpu_mapper.hpp"

namespace bqhost {

gpuMapper::gpuMapper(int device, int nx, int ny, int nz, float h)
{
    g.ni = nx; g.nj = ny; g.nk = nz; g.h = h;
    if (fl_init(device) != FL_OK) return;                 // cudaInit(), GPU_Advection.h:214-226
    ok_ = u_src.alloc(g.nu()) && v_src.alloc(g.nv()) && w_src.alloc(g.nw())
       && x_out.alloc(g.n()) && y_out.alloc(g.n()) && z_out.alloc(g.n())
       && x_out2.alloc(g.n()) && y_out2.alloc(g.n()) && z_out2.alloc(g.n());
}

void gpuMapper::startEventRecord()
{
    if (!ev_start_) ev_start_ = fl_event_create();
    if (!ev_stop_) ev_stop_ = fl_event_create();
    fl_event_record(ev_start_);
}

float gpuMapper::endEventRecord()
{
    fl_event_record(ev_stop_);
    return fl_event_elapsed_ms(ev_start_, ev_stop_);
}

// GPU_Advection.h:472-482: the kernels write an interior window only, so zero first
void gpuMapper::advectVelocity(float *u, float *v, float *w, float *ui, float *vi, float *wi,
                               float *bx, float *by, float *bz, bool is_point) const
{
    fl_memset(u, 0, g.nu() * sizeof(float));
    fl_memset(v, 0, g.nv() * sizeof(float));
    fl_memset(w, 0, g.nw() * sizeof(float));
    gpu_advect_velocity(u, v, w, ui, vi, wi, bx, by, bz, g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:493-503
void gpuMapper::compensateVelocity(float *u, float *v, float *w, float *du, float *dv, float *dw,
                                   float *fx, float *fy, float *fz, float *bx, float *by, float *bz,
                                   bool is_point) const
{
    u_src.zero(); v_src.zero(); w_src.zero();
    gpu_compensate_velocity(u, v, w, du, dv, dw, u_src, v_src, w_src, fx, fy, fz, bx, by, bz,
                            g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:505-511 (sized ni*nj*nk, not the reference's (ni+1)*nj*nk overrun)
void gpuMapper::advectField(float *f, float *fi, float *bx, float *by, float *bz, bool is_point) const
{
    fl_memset(f, 0, g.n() * sizeof(float));
    gpu_advect_field(f, fi, bx, by, bz, g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:521-528: u_src doubles as the scalar error scratch
void gpuMapper::compensateField(float *f, float *df, float *fx, float *fy, float *fz,
                                float *bx, float *by, float *bz, bool is_point) const
{
    fl_memset(u_src, 0, g.n() * sizeof(float));
    gpu_compensate_field(f, df, u_src, fx, fy, fz, bx, by, bz, g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:602-608
void gpuMapper::projectionJacobi(float *u, float *v, float *w, float *div, float *p, float *p_temp, float *debug,
                                 int iter, float halfrdx, float alpha, float beta) const
{
    fl_memset(div, 0, g.n() * sizeof(float));
    fl_memset(p, 0, g.n() * sizeof(float));
    fl_memset(p_temp, 0, g.n() * sizeof(float));
    gpu_projection_jacobi(u, v, w, div, p, p_temp, debug, g.ni, g.nj, g.nk, iter, halfrdx, alpha, beta);
}

} // namespace bqhost
