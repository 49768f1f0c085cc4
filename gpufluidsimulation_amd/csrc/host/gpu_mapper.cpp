// gpu_mapper.cpp -- see gpu_mapper.hpp.  Reference: src/bimocq3D/GPU_Advection.h:110-627.
#include "gpu_mapper.hpp"
#include <cstdio>

#include <cstdlib>

namespace bqhost {

void (*g_trace_hook)(const char *stage) = nullptr;

gpuMapper::gpuMapper(int device, int nx, int ny, int nz, float h, const SlabCtx &sl)
{
    slab = sl;
    g.ni = nx; g.nj = ny; g.nk = slab.on ? slab.nk_local() : nz; g.h = h;
    if (fl_init(device) != FL_OK) return;                 // cudaInit(), GPU_Advection.h:214-226
    if (slab.on) fl_set_slab(slab.koff(), slab.nkg, slab.own0, slab.own1, slab.nk_local());
    else fl_set_slab(0, 0, 0, 0, 0);
    ok_ = allocField(u_src, FIELD_U) && allocField(v_src, FIELD_V) && allocField(w_src, FIELD_W)
       && allocField(x_out, FIELD_S) && allocField(y_out, FIELD_S) && allocField(z_out, FIELD_S)
       && allocField(x_out2, FIELD_S) && allocField(y_out2, FIELD_S) && allocField(z_out2, FIELD_S);
}

bool gpuMapper::allocField(DeviceField &f, FieldKind kind) const
{
    size_t count = kind == FIELD_U ? g.nu() : kind == FIELD_V ? g.nv() : kind == FIELD_W ? g.nw() : g.n();
    if (!f.alloc(count)) return false;
    f.plane = kind == FIELD_U ? (size_t)(g.ni + 1) * g.nj : kind == FIELD_V ? (size_t)g.ni * (g.nj + 1) : (size_t)g.ni * g.nj;
    f.extra = kind == FIELD_W ? 1 : 0;
    f.valid = slab.on ? slab.G : DeviceField::kAlwaysValid;      // zero-filled: trivially consistent
    return true;
}

void gpuMapper::require(std::initializer_list<DeviceField *> fields, int depth)
{
    if (!slab.on || slab.nranks <= 1) return;
    if (depth > slab.G) {
        // the caller asked for more ghost planes than exist: the time step moved data further than the
        // slab overlap (CFL too large for G); refuse rather than compute from missing planes
        fl_report_error(FL_ERR_BAD_ARGUMENT, "z-slab ghost zone too shallow for this time step (CFL travel + stencil reach > G)");
        return;
    }
    // BQ_HALO_ALWAYS=1 (debug): refresh on every request, whatever the bookkeeping says
    static const bool always = getenv("BQ_HALO_ALWAYS") && atoi(getenv("BQ_HALO_ALWAYS")) != 0;
    float *ptrs[16]; size_t planes[16]; int extras[16]; int n = 0;
    for (DeviceField *f : fields)
        if ((always || f->valid < depth) && n < 16) { ptrs[n] = f->get(); planes[n] = f->plane; extras[n] = f->extra; n++; }
    if (!n) return;
    // BQ_TRACE_REQUIRE=1 (debug): what triggered the exchange -- requested depth and the shallowest valid depth found
    static const bool trace = getenv("BQ_TRACE_REQUIRE") && atoi(getenv("BQ_TRACE_REQUIRE")) != 0;
    if (trace && slab.rank == 0) {
        int vmin = 1 << 20;
        for (DeviceField *f : fields) if (f->valid < vmin) vmin = f->valid;
        fprintf(stderr, "[require] %d of %d fields, depth %d requested, valid %d\n", n, (int)fields.size(), depth, vmin);
    }
    // All G planes move, whatever depth was asked for: moving only `depth` planes was measured on two ranks
    // (tests/slab_worker.py, 32x32x64, G = 8): 26 % fewer planes outside the Jacobi loop but 4 more exchanges per step
    // (the deeper validity is what spares later operators their own exchange) -- no gain.
    fl_halo_exchange(n, ptrs, planes, extras, g.nk, slab.G, slab.G, 1);
    for (DeviceField *f : fields)
        if (always || f->valid < depth) f->valid = slab.G;
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
    FusedScope fused(fuse_housekeeping, 1);             // the kernels write the zeros outside their window themselves
    if (!fused.on) {
        fl_memset(u, 0, g.nu() * sizeof(float));
        fl_memset(v, 0, g.nv() * sizeof(float));
        fl_memset(w, 0, g.nw() * sizeof(float));
    }
    gpu_advect_velocity(u, v, w, ui, vi, wi, bx, by, bz, g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:493-503
void gpuMapper::compensateVelocity(float *u, float *v, float *w, float *du, float *dv, float *dw,
                                   float *fx, float *fy, float *fz, float *bx, float *by, float *bz,
                                   bool is_point) const
{
    fl_memset(u_src, 0, u_src.bytes()); fl_memset(v_src, 0, v_src.bytes()); fl_memset(w_src, 0, w_src.bytes());
    gpu_compensate_velocity(u, v, w, du, dv, dw, u_src, v_src, w_src, fx, fy, fz, bx, by, bz,
                            g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:505-511 (sized ni*nj*nk, not the reference's (ni+1)*nj*nk overrun)
void gpuMapper::advectField(float *f, float *fi, float *bx, float *by, float *bz, bool is_point) const
{
    FusedScope fused(fuse_housekeeping, 1);
    if (!fused.on) fl_memset(f, 0, g.n() * sizeof(float));
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
