// fluid_solver.cpp -- see fluid_solver.hpp.  Reference: src/bimocq3D/BimocqGPUSolver.cpp.
//
// The step restates BimocqGPUSolver::advanceBimocq (:129-230) operator by operator.  Work that the
// reference performs but that provably cannot change any value is dropped, each case documented
// where it happens (shared map sets, the identically-zero scalar "extern" deltas, buffer swaps in
// place of copies whose source is dead).  Everything else -- including the reference's quirks that
// DO change values -- is kept, and tests/ checks the trajectory against
// the CPU oracle bit for bit.
//
// z-slab ranks run the same code: gpuMapper::require()/produced() keep track of how many ghost
// planes of every field are correct and exchange them with the z-neighbours exactly when an operator
// reaches further than that (no-ops on a single GPU).
#include "fluid_solver.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bqhost {

// BQ_TRACE=1 (debug): after each stage print the sum of squares of the planes this rank owns, so a
// slab run can be compared stage by stage with a single-GPU run (sum the ranks' lines).
static void trace_stage(BimocqGPUSolver &s, const char *stage, int frame)
{
    static const bool on = getenv("BQ_TRACE") && atoi(getenv("BQ_TRACE")) != 0;
    if (!on) return;
    const SlabCtx &sl = s.GpuSolver->slab;
    const int G = sl.on ? sl.G : 0, own = sl.on ? sl.own1 - sl.own0 : s.g.nk;
    const bool top = !sl.on || sl.own1 == sl.nkg;
    struct Item { const char *name; DeviceField *f; } items[] = {
        { "U", &s.VelocityU }, { "V", &s.VelocityV }, { "W", &s.VelocityW }, { "Ui", &s.VelocityUInit }, { "Wi", &s.VelocityWInit },
        { "rho", &s.Density }, { "bz", &s.VelocityAdvector.maps->BackwardZ }, { "fz", &s.VelocityAdvector.maps->ForwardZ },
        { "dUp", &s.duProj }, { "dWp", &s.dwProj }, { "p", &s.p }, { "div", &s.div },
        { "bx", &s.VelocityAdvector.maps->BackwardX }, { "by", &s.VelocityAdvector.maps->BackwardY },
        { "fx", &s.VelocityAdvector.maps->ForwardX }, { "usrc", &s.GpuSolver->u_src }, { "vsrc", &s.GpuSolver->v_src }, { "wsrc", &s.GpuSolver->w_src } };
    std::vector<float> host;
    char fname[128];
    snprintf(fname, sizeof fname, "/tmp/bqtrace_%dof%d.txt", sl.rank, sl.nranks);
    FILE *fo = fopen(fname, "a");
    if (!fo) return;
    fprintf(fo, "[trace r%d f%d %-14s]", sl.rank, frame, stage);
    for (const Item &it : items) {
        host.resize(it.f->count());
        it.f->download(host.data());
        const size_t pe = it.f->plane ? it.f->plane : (size_t)s.g.ni * s.g.nj;
        const size_t lo = pe * (size_t)G, hi = pe * (size_t)(G + own + ((it.f->extra && top) ? 1 : 0));
        double acc = 0;
        for (size_t a = lo; a < hi && a < host.size(); a++) acc += (double)host[a] * (double)host[a];
        fprintf(fo, " %s=%.17g", it.name, acc);
    }
    fprintf(fo, "\n");
    fclose(fo);
    const char *dump = getenv("BQ_DUMP");
    if (dump && frame == 1 && strcmp(dump, stage) == 0) {
        for (const Item &it : items) {
            host.resize(it.f->count());
            it.f->download(host.data());
            snprintf(fname, sizeof fname, "/tmp/bqdump_%dof%d_%s.bin", sl.rank, sl.nranks, it.name);
            FILE *fd = fopen(fname, "wb");
            if (fd) { fwrite(host.data(), sizeof(float), host.size(), fd); fclose(fd); }
        }
    }
}

BimocqGPUSolver::BimocqGPUSolver(unsigned nx, unsigned ny, unsigned nz, float L, float vis_coeff, float blend_coeff,
                                 Scheme inScheme, gpuMapper *mymapper)
    : myscheme(inScheme), GpuSolver(mymapper)
{
    CellSize = L / nx;                                   // :10
    Viscosity = vis_coeff;
    (void)ny; (void)nz;
    if (!mymapper || !mymapper->ok()) return;
    g = mymapper->g;                                     // local dims (a slab rank holds owned + ghost planes)
    g.h = CellSize;

    DeviceField *ub[] = { &VelocityU, &VelocityUInit, &VelocityUPrev, &VelocityUTemp, &duProj, &duExtern, &TempSrcU };
    DeviceField *vb[] = { &VelocityV, &VelocityVInit, &VelocityVPrev, &VelocityVTemp, &dvProj, &dvExtern, &TempSrcV };
    DeviceField *wb[] = { &VelocityW, &VelocityWInit, &VelocityWPrev, &VelocityWTemp, &dwProj, &dwExtern, &TempSrcW };
    for (int a = 0; a < 7; a++)
        if (!mymapper->allocField(*ub[a], FIELD_U) || !mymapper->allocField(*vb[a], FIELD_V) || !mymapper->allocField(*wb[a], FIELD_W)) return;
    DeviceField *sb[] = { &Density, &DensityInit, &DensityPrev, &Temperature, &TemperatureInit, &TemperaturePrev,
                          &div, &p, &p_temp };
    for (DeviceField *f : sb)
        if (!mymapper->allocField(*f, FIELD_S)) return;
    if (!debugParam.alloc(4096)) return;

    if (!VelocityAdvector.init(g.ni, g.nj, g.nk, CellSize, blend_coeff, mymapper)) return;      // :92
    if (!ScalarAdvector.init(g.ni, g.nj, g.nk, CellSize, blend_coeff, mymapper)) return;        // :93
    // Both advectors are updated with the same velocity and re-initialised on the same frames
    // (`if (1)`, :218-229), so their map sets are bit-identical: keep one.
    ScalarAdvector.shareMapsOf(VelocityAdvector);

    host_density.assign(g.n(), 0.f);
    host_u.assign(g.nu(), 0.f); host_v.assign(g.nv(), 0.f); host_w.assign(g.nw(), 0.f);
    ok_ = fl_last_error() == FL_OK;
}

void BimocqGPUSolver::setSmoke(float drop, float raise, const std::vector<Emitter> &emitters)
{
    _alpha = drop;                                       // :531
    _beta = raise;                                       // :532
    sim_emitter = emitters;
}

// :108-127
void BimocqGPUSolver::advance(int framenum, float dt)
{
    GpuSolver->startEventRecord();
    switch (myscheme) {
    case BIMOCQ: advanceBimocq(framenum, dt); break;
    case MAC_REFLECTION: advanceReflection(framenum, dt); break;
    default: break;                                      // the reference's GPU solver has no other scheme (:112-122)
    }
    // FL_OPT_COMM_CHECK (debugging aid for the first runs on real links): every rank has issued this step's communicator
    // calls; compare the ledgers (include/bimocq_gpu.h: fl_comm_check)
    if (fl_comm_size() > 1 && fl_get_option(FL_OPT_COMM_CHECK) > 0) (void)fl_comm_check(0);
    last_ms = GpuSolver->endEventRecord();
    if (verbose) printf("[Bimocq GPU Time: %gms ]\n", last_ms);
}

// :232-337 advanceReflection: MacCormack advection of rho, T and of the velocity (half step),
// sources, projection, reflection 2 u_proj - u advected another half step, sources, projection.  The
// limiter is the corrected gpu_clamp_extrema (include/bimocq_gpu.h).
// z-slab ranks (Jacobi projection): every look-up states how far it reaches along z -- the trace travels at most
// |t| * max|u| (getCFL) and the sample adds its cell and the half-cell stagger -- so gpuMapper::require refreshes ghost
// planes exactly when an input's correct depth no longer covers that reach, and produced() records what is left of it.
void BimocqGPUSolver::advanceReflection(int framenum, float dt)
{
    gpuMapper &gs = *GpuSolver;
    const bool slabs = gs.slab.on && gs.slab.nranks > 1;
    DeviceField *extra[] = { &DensityTemp, &TemperatureTemp };
    for (DeviceField *f : extra)
        if (!f->get() && !gs.allocField(*f, FIELD_S)) return;
    const float cfldt = getCFL();                        // :234
    last_cfldt = cfldt;
    const float h = CellSize;
    const int ni = g.ni, nj = g.nj, nk = g.nk;
    const size_t n = g.n(), nu = g.nu(), nv = g.nv(), nw = g.nw();
    // ghost planes a look-up after a trace over time t can touch
    // (vmax: the speed bound of the velocity field the traces run through -- max|u| of getCFL() until the first
    // projection has changed the field, a fresh maximum after it; single-GPU runs never look at it)
    float vmax = MaxVelocity;
    auto reach = [&](float t) { return (int)std::ceil(std::fabs((double)t) * (double)vmax / (double)h) + 3; };
    auto vel_valid = [&]() { return gpuMapper::minValid({ &VelocityU, &VelocityV, &VelocityW }); };

    // semilagAdvectField / semilagAdvectVelocity (GPU_Advection.h:530-551) clear their outputs first
    auto semilagScalar = [&](DeviceField &dst, DeviceField &src, float t) {
        const int r = reach(t);
        gs.require({ &src, &VelocityU, &VelocityV, &VelocityW }, r);
        fl_memset(dst, 0, n * sizeof(float));
        gpu_semilag(dst, src, VelocityU, VelocityV, VelocityW, 0, 0, 0, h, ni, nj, nk, cfldt, t);
        gs.produced(dst, std::min(src.valid, vel_valid()) - r);
    };
    auto semilagVelocity = [&](DeviceField &uo, DeviceField &vo, DeviceField &wo, DeviceField &us, DeviceField &vs, DeviceField &ws, float t) {
        const int r = reach(t);
        gs.require({ &us, &vs, &ws, &VelocityU, &VelocityV, &VelocityW }, r);
        uo.zero(); vo.zero(); wo.zero();
        gpu_semilag(uo, us, VelocityU, VelocityV, VelocityW, 1, 0, 0, h, ni, nj, nk, cfldt, t);
        gpu_semilag(vo, vs, VelocityU, VelocityV, VelocityW, 0, 1, 0, h, ni, nj, nk, cfldt, t);
        gpu_semilag(wo, ws, VelocityU, VelocityV, VelocityW, 0, 0, 1, h, ni, nj, nk, cfldt, t);
        const int vv = vel_valid();
        gs.produced(uo, std::min(us.valid, vv) - r); gs.produced(vo, std::min(vs.valid, vv) - r); gs.produced(wo, std::min(ws.valid, vv) - r);
    };
    // the limiter reads `field` around the departure point of a trace over dtc and rewrites `temp` at the node itself
    auto limiter = [&](DeviceField &field, DeviceField &temp, int bi, int bj, int bk, int dx, int dy, int dz, float dtc) {
        const int r = reach(dtc);
        gs.require({ &field, &VelocityU, &VelocityV, &VelocityW }, r);
        gpu_clamp_extrema(field, temp, VelocityU, VelocityV, VelocityW, bi, bj, bk, dx, dy, dz, 0.5f * dx, 0.5f * dy, 0.5f * dz, h, dtc);
        gs.produced(temp, std::min(temp.valid, std::min(field.valid, vel_valid()) - r));
    };
    auto clampVelocity = [&]() {                         // :283-285, :323-325: source field = VelocityU/V/W
        limiter(VelocityU, VelocityUTemp, ni + 1, nj, nk, 1, 0, 0, 0.5f * dt);
        limiter(VelocityV, VelocityVTemp, ni, nj + 1, nk, 0, 1, 0, 0.5f * dt);
        limiter(VelocityW, VelocityWTemp, ni, nj, nk + 1, 0, 0, 1, 0.5f * dt);
    };
    // f1 += coeff * f2
    auto axpy = [&](DeviceField &f1, DeviceField &f2, float coeff, size_t count) {
        gs.add(f1, f2, coeff, count);
        gs.produced(f1, std::min(f1.valid, f2.valid));
    };
    auto sources = [&](bool emit) {                      // :289-297 / :330-337
        if (emit) emitSmoke(framenum, dt);
        addBuoyancy(0.5f * dt);
        if (Viscosity) {
            if (slabs) {
                diffuseFieldSlab(VelocityU, VelocityUTemp, TempSrcU, ni + 1, nj, nk, 20, Viscosity, 0.5f * dt);
                diffuseFieldSlab(VelocityV, VelocityVTemp, TempSrcV, ni, nj + 1, nk, 20, Viscosity, 0.5f * dt);
                diffuseFieldSlab(VelocityW, VelocityWTemp, TempSrcW, ni, nj, nk + 1, 20, Viscosity, 0.5f * dt);
            } else {
                diffuseField(VelocityU, VelocityUTemp, TempSrcU, ni + 1, nj, nk, 20, Viscosity, 0.5f * dt);
                diffuseField(VelocityV, VelocityVTemp, TempSrcV, ni, nj + 1, nk, 20, Viscosity, 0.5f * dt);
                diffuseField(VelocityW, VelocityWTemp, TempSrcW, ni, nj, nk + 1, 20, Viscosity, 0.5f * dt);
            }
        }
    };

    DeviceField *scal[2] = { &Density, &Temperature }, *tmp[2] = { &DensityTemp, &TemperatureTemp };
    for (int a = 0; a < 2; a++) {                        // :237-263
        semilagScalar(*tmp[a], *scal[a], -dt);
        semilagScalar(TempSrcU, *tmp[a], dt);
        axpy(*tmp[a], TempSrcU, -0.5f, n);
        axpy(*tmp[a], *scal[a], 0.5f, n);
        limiter(*scal[a], *tmp[a], ni, nj, nk, 0, 0, 0, dt);
        fl_memcpy_d2d(*scal[a], *tmp[a], n * sizeof(float));
        gs.produced(*scal[a], tmp[a]->valid);
    }
    // :267-287
    semilagVelocity(VelocityUTemp, VelocityVTemp, VelocityWTemp, VelocityU, VelocityV, VelocityW, -0.5f * dt);
    semilagVelocity(TempSrcU, TempSrcV, TempSrcW, VelocityUTemp, VelocityVTemp, VelocityWTemp, 0.5f * dt);
    axpy(VelocityUTemp, TempSrcU, -0.5f, nu); axpy(VelocityVTemp, TempSrcV, -0.5f, nv); axpy(VelocityWTemp, TempSrcW, -0.5f, nw);
    axpy(VelocityUTemp, VelocityU, 0.5f, nu); axpy(VelocityVTemp, VelocityV, 0.5f, nv); axpy(VelocityWTemp, VelocityW, 0.5f, nw);
    clampVelocity();
    VelocityU.copy_from(VelocityUTemp); VelocityV.copy_from(VelocityVTemp); VelocityW.copy_from(VelocityWTemp);

    sources(true);
    VelocityUTemp.copy_from(VelocityU); VelocityVTemp.copy_from(VelocityV); VelocityWTemp.copy_from(VelocityW);   // :299-303
    projection();                                        // :305
    if (slabs) vmax = std::max(vmax, gpu_max_abs3(VelocityU, VelocityV, VelocityW, g.ni, g.nj, g.nk));
    gpu_mad(duProj, VelocityU, VelocityUTemp, 2.f, -1.f, (int)nu);      // :307-309
    gpu_mad(dvProj, VelocityV, VelocityVTemp, 2.f, -1.f, (int)nv);
    gpu_mad(dwProj, VelocityW, VelocityWTemp, 2.f, -1.f, (int)nw);
    gs.produced(duProj, std::min(VelocityU.valid, VelocityUTemp.valid));
    gs.produced(dvProj, std::min(VelocityV.valid, VelocityVTemp.valid));
    gs.produced(dwProj, std::min(VelocityW.valid, VelocityWTemp.valid));
    semilagVelocity(VelocityUTemp, VelocityVTemp, VelocityWTemp, duProj, dvProj, dwProj, -0.5f * dt);             // :311
    semilagVelocity(TempSrcU, TempSrcV, TempSrcW, VelocityUTemp, VelocityVTemp, VelocityWTemp, 0.5f * dt);        // :313
    axpy(VelocityUTemp, TempSrcU, -0.5f, nu); axpy(VelocityVTemp, TempSrcV, -0.5f, nv); axpy(VelocityWTemp, TempSrcW, -0.5f, nw);
    axpy(VelocityUTemp, duProj, 0.5f, nu); axpy(VelocityVTemp, dvProj, 0.5f, nv); axpy(VelocityWTemp, dwProj, 0.5f, nw);
    clampVelocity();
    VelocityU.copy_from(VelocityUTemp); VelocityV.copy_from(VelocityVTemp); VelocityW.copy_from(VelocityWTemp);
    sources(false);
    projection();
    steps_taken++;
}

// :348-373.  The reference scans host copies that outputResult() refreshed after the previous
// frame; the same numbers are on the device at this point, so reduce them there (slab ranks reduce
// their owned planes and all-reduce the maximum).
float BimocqGPUSolver::getCFL()
{
    MaxVelocity = gpu_max_abs3(VelocityU, VelocityV, VelocityW, g.ni, g.nj, g.nk);    // includes the 1e-4 floor
    return CellSize / MaxVelocity;
}

// :376-392 with the hard-coded scene constants replaced by the emitter list (pointwise in z)
void BimocqGPUSolver::emitSmoke(int framenum, float /*dt*/)
{
    for (const Emitter &e : sim_emitter)
        if (framenum < e.emitFrame)
            GpuSolver->emitSmoke(VelocityU, VelocityV, VelocityW, Density, Temperature,
                                 e.e_pos[0], e.e_pos[1], e.e_pos[2], e.radius, e.emit_density, e.emit_temperature, e.emiter);
}

// :394-397 (reads rho/T at j and j-1 of the same plane: no reach along z)
void BimocqGPUSolver::addBuoyancy(float dt)
{
    GpuSolver->add_buoyancy(VelocityV, Density, Temperature, _alpha, _beta, dt);
    GpuSolver->produced(VelocityV, gpuMapper::minValid({ &VelocityV, &Density, &Temperature }));
}

// :399-404
void BimocqGPUSolver::diffuseField(float *field, float *t0, float *t1, int ni, int nj, int nk, int iter, float nu, float dt)
{
    float coef = nu * (dt / (CellSize * CellSize));
    GpuSolver->diffuseField(field, t0, t1, ni, nj, nk, iter, coef);
}

// gpu_diffuse_field (GPU_kernel.cu:855-876) on a z-slab rank: t0 <- field, `iter` sweeps in chunks of at most G
// (one exchange of the newest iterate's ghost planes buys G sweeps, each sweep reaching one plane), field <-
// the iterate BEFORE the last one.  bk: buffer planes (nk + 1 for w).
void BimocqGPUSolver::diffuseFieldSlab(DeviceField &field, DeviceField &t0, DeviceField &t1, int bi, int bj, int bk,
                                       int iter, float nu, float dt)
{
    gpuMapper &gs = *GpuSolver;
    const int G = gs.slab.G;
    const float coef = nu * (dt / (CellSize * CellSize));
    gs.require({ &field }, G);
    t0.copy_from(field);
    DeviceField *in = &t0, *out = &t1;
    gs.produced(*out, 0);
    int left = iter;
    while (left > 0) {
        const int chunk = std::min(left, G);
        gs.require({ in }, chunk);
        const int v0 = in->valid;
        const int where = gpu_diffuse_sweeps(field, *in, *out, bi, bj, bk, chunk, coef);
        // sweep s of the chunk leaves min(v0 - s, field.valid) correct ghost planes in its output
        const int newest = std::min(v0 - chunk, field.valid), older = std::min(v0 - chunk + 1, field.valid);
        if (where) std::swap(in, out);                     // `in` = newest iterate, `out` = the one before it
        gs.produced(*in, newest);
        gs.produced(*out, chunk == 1 ? v0 : older);
        left -= chunk;
    }
    field.copy_from(*out);
}

// :406-467, the Jacobi branch (:409-410): alpha = -1, beta = 1/6
// Distortion-driven re-initialisation (BQ_OPT_REINIT_POLICY = 1): the two map sets follow different
// schedules, so the scalar advector needs its own; the scalar snapshot/delta buffers the every-frame
// policy never needs are allocated here.
bool BimocqGPUSolver::setReinitPolicy(int policy)
{
    if (policy == reinit_policy) return true;
    if (steps_taken != 0) { fl_report_error(FL_ERR_BAD_ARGUMENT, "BQ_OPT_REINIT_POLICY must be set before the first advance()"); return false; }
    if (policy != 0 && policy != 1) { fl_report_error(FL_ERR_BAD_ARGUMENT, "BQ_OPT_REINIT_POLICY: 0 or 1"); return false; }
    if (policy == 1) {
        if (!ScalarAdvector.unshareMaps()) return false;
        // z-slab ranks: maps that live for many steps must not outgrow the ghost zone -- their z-travel is measured after
        // every update and a map set is re-initialised before it would (setTravelLimit)
        if (GpuSolver->slab.on && GpuSolver->slab.nranks > 1 && travel_limit == 0) setTravelLimit(GpuSolver->slab.G);
        DeviceField *sb[] = { &DensityTemp, &TemperatureTemp, &DensityExtern, &TemperatureExtern };
        for (DeviceField *f : sb)
            if (!f->get() && !GpuSolver->allocField(*f, FIELD_S)) return false;
    } else {
        ScalarAdvector.shareMapsOf(VelocityAdvector);
    }
    reinit_policy = policy;
    return true;
}

void BimocqGPUSolver::setTravelLimit(int cells)
{
    const SlabCtx &sl = GpuSolver->slab;
    if (cells < 0) cells = 0;
    if (sl.on && sl.nranks > 1 && (cells == 0 || cells > sl.G)) cells = sl.G;      // a slab rank cannot serve more than its ghost planes
    travel_limit = cells;
    VelocityAdvector.measureTravel = ScalarAdvector.measureTravel = cells > 0;
}

// BimocqGPUSolver.cpp:60-90: p, dir, residual, div, temp0, temp1 (N doubles each), tempResult (4096), and
// LEVEL_COUNT levels of b/x/r with dims n -> (n - 1) / 2.  Levels without a single cell are left out (the
// reference launches empty grids for them).
bool BimocqGPUSolver::allocMgcg()
{
    if (mg.ready) return true;
    // temp1 and levels[0].b below are full-size arrays: the operator may keep its fused level-0 vectors in them
    // (include/bimocq_gpu.h, FL_OPT_MGCG_FUSE; -1 = nobody has set the option)
    if (fl_get_option(FL_OPT_MGCG_FUSE) < 0) fl_set_option(FL_OPT_MGCG_FUSE, 1);
    // a z-slab rank solves on the WHOLE grid (projectionMgcgSlabs): global plane count
    const bool slabs = GpuSolver->slab.on && GpuSolver->slab.nranks > 1;
    const int nk_all = slabs ? GpuSolver->slab.nkg : g.nk;
    const size_t n = (size_t)g.ni * g.nj * nk_all;
    DeviceBytes *full[] = { &mg.div, &mg.p, &mg.dir, &mg.residual, &mg.temp0, &mg.temp1 };
    for (DeviceBytes *f : full)
        if (!f->alloc(n * sizeof(double))) return false;
    if (!mg.result.alloc(4096 * sizeof(double))) return false;
    int ni = g.ni, nj = g.nj, nk = nk_all;
    for (int l = 0; l < LEVEL_COUNT; l++) {
        if (l) { ni = (ni - 1) / 2; nj = (nj - 1) / 2; nk = (nk - 1) / 2; }
        if (ni < 1 || nj < 1 || nk < 1) break;
        SCoarseLevelInfo L{};
        L.ni = ni; L.nj = nj; L.nk = nk; L.number = ni * nj * nk;
        L.alpha = -1.0; L.beta = 1.0 / 6.0;
        mg.b.emplace_back(); mg.x.emplace_back(); mg.r.emplace_back();
        const size_t bytes = (size_t)L.number * sizeof(double);
        if (!mg.b.back().alloc(bytes) || !mg.x.back().alloc(bytes) || !mg.r.back().alloc(bytes)) return false;
        L.b = mg.b.back().f64(); L.x = mg.x.back().f64(); L.r = mg.r.back().f64();
        mg.levels.push_back(L);
    }
    if (slabs) {
        if (!mg.gu.alloc((size_t)(g.ni + 1) * g.nj * nk_all) || !mg.gv.alloc((size_t)g.ni * (g.nj + 1) * nk_all) ||
            !mg.gw.alloc((size_t)g.ni * g.nj * (nk_all + 1))) return false;
    }
    mg.ready = true;
    return true;
}

// The multigrid-CG projection on z-slab ranks, REPLICATED: every rank assembles the velocity of the whole grid (its own planes
// + one point-to-point message per peer and component, a single RCCL group: xGMI is a full mesh), runs the very solver a
// single GPU runs on it (slab context off) and takes its own planes, ghost planes included, back.  Bit-identical to one GPU
// by construction -- the reference's float-narrowed dot products, its level pyramid n -> (n - 1) / 2 and its interpolation
// quirks all see the global arrays -- and the first step towards a slab-decomposed V-cycle: the advection
// scales with the rank count, the solve does not (it costs what it costs on one GPU, plus ~1 ms of gather).
bool BimocqGPUSolver::projectionMgcgSlabs()
{
    gpuMapper &gs = *GpuSolver;
    const SlabCtx &sl = gs.slab;
    const int G = sl.G, nkg = sl.nkg, R = sl.nranks;
    const bool last = sl.own1 == nkg;
    struct Comp { DeviceField *loc, *glob; size_t plane; int extra; } comps[3] = {
        { &VelocityU, &mg.gu, (size_t)(g.ni + 1) * g.nj, 0 }, { &VelocityV, &mg.gv, (size_t)g.ni * (g.nj + 1), 0 },
        { &VelocityW, &mg.gw, (size_t)g.ni * g.nj, 1 } };
    // planes rank r owns of a component: its cell planes, the w face on top of the grid goes to the last rank
    const auto own0_of = [&](int r) { return r * (nkg / R); };
    const auto planes_of = [&](int r, int extra) { return nkg / R + ((extra && r == R - 1) ? 1 : 0); };
    std::vector<int> peers; std::vector<float *> send, recv; std::vector<size_t> send_count, recv_count;
    for (const Comp &c : comps) {
        const size_t mine = c.plane * (size_t)planes_of(sl.rank, c.extra);
        fl_memcpy_d2d(c.glob->get() + c.plane * (size_t)sl.own0, c.loc->get() + c.plane * (size_t)G, mine * sizeof(float));
    }
    for (int r = 0; r < R; r++) {
        if (r == sl.rank) continue;
        for (const Comp &c : comps) {
            peers.push_back(r);
            send.push_back(c.loc->get() + c.plane * (size_t)G);
            send_count.push_back(c.plane * (size_t)planes_of(sl.rank, c.extra));
            recv.push_back(c.glob->get() + c.plane * (size_t)own0_of(r));
            recv_count.push_back(c.plane * (size_t)planes_of(r, c.extra));
        }
    }
    fl_p2p_exchange((int)peers.size(), peers.data(), send.data(), send_count.data(), recv.data(), recv_count.data());
    (void)last;
    fl_set_slab(0, 0, 0, 0, 0);                                         // the solve sees one global grid
    gpu_multi_grid_conjugate_gradient(mg.gu, mg.gv, mg.gw, mg.div.f64(), mg.p.f64(), mg.dir.f64(),
                                      mg.residual.f64(), mg.temp0.f64(), mg.temp1.f64(), mg.result.f64(),
                                      mg.levels.data(), (int)mg.levels.size(), mg_iters, (double)halfrdx);
    fl_set_slab(sl.koff(), sl.nkg, sl.own0, sl.own1, sl.nk_local());
    // the projected velocity on this rank's planes, ghost planes included (those outside the grid stay zero)
    const int k0 = std::max(sl.own0 - G, 0), k1 = std::min(sl.own1 + G, nkg);
    for (const Comp &c : comps) {
        const int np = k1 - k0 + ((c.extra && k1 == nkg) ? 1 : 0);
        fl_memcpy_d2d(c.loc->get() + c.plane * (size_t)(k0 - (sl.own0 - G)), c.glob->get() + c.plane * (size_t)k0,
                      c.plane * (size_t)np * sizeof(float));
        gs.produced(*c.loc, G);
    }
    return fl_last_error() == FL_OK;
}

// The multigrid-CG projection on z-slab ranks with the fine levels SHARED (round 4: include/bimocq_gpu.h,
// gpu_multi_grid_conjugate_gradient_slab; csrc/bq_mgcg_slab.hip.inc): nothing global is allocated here -- the operator owns its
// per-level slab arrays -- and the velocity needs its ghost planes only as deep as level 0's (8).  false when the decomposition
// is not covered (planes that are not a multiple of 256 cells, thin or unaligned slabs, the CPU stand-in): the caller falls
// back to the replicated solve, which is bit-identical.
bool BimocqGPUSolver::projectionMgcgShared()
{
    gpuMapper &gs = *GpuSolver;
    const SlabCtx &sl = gs.slab;
    mgcg_shared_ran = false;
    if (!mgcg_shared || !gpu_mgcg_slab_supported(g.ni, g.nj, sl.nkg, sl.own0, sl.own1, sl.G, sl.rank, sl.nranks)) return false;
    if (mg.result.bytes() < 4096 * sizeof(double) && !mg.result.alloc(4096 * sizeof(double))) return false;
    gs.require({ &VelocityU, &VelocityV, &VelocityW }, sl.G);
    gpu_multi_grid_conjugate_gradient_slab(VelocityU, VelocityV, VelocityW, mg.result.f64(), g.ni, g.nj, sl.nkg, sl.own0, sl.own1, sl.G,
                                           mg_iters, (double)halfrdx);
    // projected on the owned planes and on the 7 ghost planes next to them (the gradient needs p one plane below)
    const int depth = std::min(sl.G, 8) - 1;
    gs.produced(VelocityU, depth); gs.produced(VelocityV, depth); gs.produced(VelocityW, depth);
    mgcg_shared_ran = true;
    return true;
}

std::vector<double> BimocqGPUSolver::mgHistory() const
{
    std::vector<double> h(4096, 0.0);
    if (mg.result.bytes() >= h.size() * sizeof(double)) fl_memcpy_d2h(h.data(), mg.result.f64(), h.size() * sizeof(double));
    return h;
}

// with_delta: d*Proj = (projected - unprojected) velocity comes out of the gradient pass itself
// (gpu_gradient_delta) instead of a snapshot before and a subtraction after; Jacobi branch only (returns
// whether it did).
bool BimocqGPUSolver::projection(bool with_delta)
{
    gpuMapper &gs = *GpuSolver;
    const float alpha = -1.f, beta = (float)(1.0 / 6.0);
    if (projection_kind == BQ_PROJECTION_MGCG) {            // :443-446
        if (gs.slab.on && gs.slab.nranks > 1) {
            if (projectionMgcgShared()) return false;
            if (!allocMgcg()) return false;
            projectionMgcgSlabs();
            return false;
        }
        if (!allocMgcg()) return false;
        gpu_multi_grid_conjugate_gradient(VelocityU, VelocityV, VelocityW, mg.div.f64(), mg.p.f64(), mg.dir.f64(),
                                          mg.residual.f64(), mg.temp0.f64(), mg.temp1.f64(), mg.result.f64(),
                                          mg.levels.data(), (int)mg.levels.size(), mg_iters, (double)halfrdx);
        return false;
    }
    if (!gs.slab.on || gs.slab.nranks <= 1) {
        if (!with_delta) {
            gs.projectionJacobi(VelocityU, VelocityV, VelocityW, div, p, p_temp, debugParam, jacobi_iters, halfrdx, alpha, beta);
            return false;
        }
        // gpu_projection_jacobi (GPU_kernel.cu:1839-1895) in its pieces: divergence, iter-1 sweeps, the
        // newest iterate made `p` by a buffer swap instead of the copy-back, gradient with the change handed out
        div.zero(); p.zero(); p_temp.zero();                             // GPU_Advection.h:604-606
        gpu_divergence(VelocityU, VelocityV, VelocityW, div, g.ni, g.nj, g.nk, halfrdx);
        const int fuse_was = fl_get_option(FL_OPT_JACOBI_FUSE);
        if (fuse_was == 1) fl_set_option(FL_OPT_JACOBI_FUSE, 2);        // p and p_temp carry the same (zero) boundary layer
        const int where = jacobi_iters > 1 ? gpu_jacobi_sweeps(p, div, p_temp, g.ni, g.nj, g.nk, jacobi_iters - 1, alpha, beta) : 0;
        fl_set_option(FL_OPT_JACOBI_FUSE, fuse_was);
        if (where) p.swap(p_temp);
        gpu_gradient_delta(VelocityU, VelocityV, VelocityW, p, duProj, dvProj, dwProj, g.ni, g.nj, g.nk, halfrdx);
        return true;
    }
    // z-slab form of gpu_projection_jacobi (GPU_kernel.cu:1839-1895): same three kernels, with the
    // sweeps issued in chunks of G: one exchange of G ghost planes of p buys G sweeps, because each
    // sweep shrinks the correct ghost depth by one plane (the ghost planes are swept redundantly,
    // which yields the very values the neighbour computes).
    const int G = gs.slab.G;
    // The redundant ghost sweeps need div on all G ghost planes.  Rather than refreshing three velocity components to
    // depth G and evaluating div there, div is evaluated where the velocities are valid (the owned planes need one ghost
    // plane of w) and its own ghost planes are fetched: one field instead of three, the same values (a neighbour's
    // owned-plane divergence is the very expression this rank would evaluate).
    gs.require({ &VelocityU, &VelocityV, &VelocityW }, 1);
    div.zero(); p.zero(); p_temp.zero();                                 // GPU_Advection.h:604-606
    gpu_divergence(VelocityU, VelocityV, VelocityW, div, g.ni, g.nj, g.nk, halfrdx);
    gs.produced(div, std::min({ VelocityU.valid, VelocityV.valid, VelocityW.valid - 1 }));
    gs.require({ &div }, G);
    DeviceField *cur = &p, *oth = &p_temp;
    gs.produced(*cur, G);                                                // zeros everywhere
    // p and p_temp carry the same (zero) boundary layer: gpu_jacobi_sweeps may fuse sweeps pairwise
    const int fuse_was = fl_get_option(FL_OPT_JACOBI_FUSE);
    if (fuse_was == 1) fl_set_option(FL_OPT_JACOBI_FUSE, 2);
    int left = jacobi_iters - 1;                                         // iterate iter-1 is applied
    const int own0 = G, own1 = g.nk - G;                                 // local owned planes [own0, own1)
    bool pair_split = G >= 2 && own1 - own0 >= 5;                        // until the operator library says it cannot
    bool triples_ok = gs.jacobi_triples && G == 8;                       // likewise
    // The exchange a chunk starts with is hidden by the interior part of its first pair only -- one launch (40 us at
    // 512 x 512 x 80) against 8 planes per direction (130 us at 64 GB/s).  So the last two pairs of a chunk that another
    // chunk follows run ENDS FIRST: the G + 2 and then the G owned planes at either end (which need nothing beyond what
    // the pair before produced), the exchange of the now final boundary planes starts, and the two interiors plus the
    // interior of the next chunk's first pair run while it travels (three launches instead of one).
    const bool ends_first_ok = gs.overlap_exchanges && gs.jacobi_ends_first && G >= 6 && G % 2 == 0 && own1 - own0 >= 2 * G + 8;
    bool in_flight = false;                                              // the exchange of `cur`'s ghost planes has been started
    while (left > 0) {
        int where;
        int chunk;
        if (!in_flight && cur->valid >= std::min(left, G)) {
            chunk = std::min(left, G);
            where = gpu_jacobi_sweeps(*cur, div, *oth, g.ni, g.nj, g.nk, chunk, alpha, beta);
        } else {
            // The ghost planes of `cur` travel on the halo stream while the compute stream sweeps what does not
            // depend on them; the planes next to them follow the exchange.
            if (!in_flight) {
                float *ptr = cur->get(); size_t pe = cur->plane; int ex = 0;
                fl_halo_exchange(1, &ptr, &pe, &ex, g.nk, G, G, /*wait=*/0);
            }
            in_flight = false;
            int done = 0;
            chunk = std::min(left, G);
            if (pair_split && chunk >= 2) {
                // first TWO sweeps of the chunk as one fused launch per piece: output planes whose two-sweep stencil
                // stays inside the owned planes now, the two ends after the exchange (one launch for both)
                if (gpu_jacobi_sweep_pair_ranges(*cur, div, *oth, g.ni, g.nj, g.nk, own0 + 2, own1 - 2, 0, 0, alpha, beta)) {
                    fl_halo_wait();
                    gs.produced(*cur, G);
                    gpu_jacobi_sweep_pair_ranges(*cur, div, *oth, g.ni, g.nj, g.nk, 0, own0 + 2, own1 - 2, g.nk, alpha, beta);
                    done = 2;
                } else {
                    pair_split = false;                                  // fused kernel does not apply to this grid
                }
            }
            if (!done) {
                // one sweep in three plane ranges; an odd chunk so that everything after it runs as fused pairs
                chunk = std::min(left, (G % 2 == 0 && G > 1) ? G - 1 : G);
                gpu_jacobi_sweep_range(*cur, div, *oth, g.ni, g.nj, g.nk, own0 + 1, own1 - 1, alpha, beta);
                fl_halo_wait();
                gs.produced(*cur, G);
                gpu_jacobi_sweep_range(*cur, div, *oth, g.ni, g.nj, g.nk, 0, own0 + 1, alpha, beta);
                gpu_jacobi_sweep_range(*cur, div, *oth, g.ni, g.nj, g.nk, own1 - 1, g.nk, alpha, beta);
                done = 1;
            }
            // The remaining sweeps of the chunk start from `oth`.  A sweep leaves one ghost plane less correct on each
            // side, and nothing reads the planes beyond: every further fused pair is restricted to the planes that will
            // still be correct after it (done = 2 of G = 8: 264, 260, 256 of 272 planes).
            int rest = chunk - done;
            DeviceField *in = oth, *out = cur;
            int depth = G - done;                                        // correct ghost planes of `in`
            const bool ends_first = ends_first_ok && pair_split && done == 2 && chunk == G && left > chunk;
            // The six sweeps after the overlapped pair of a full chunk as TWO triples (BQ_OPT_JACOBI_TRIPLES, G = 8): triple A
            // produces [own0 - 3, own1 + 3) from the pair's result (correct to depth 6), triple B the owned planes.  Ends
            // first: A and B on the planes next to the slab ends (A: G + 6 planes per end, B: the G planes the neighbours
            // need), the exchange starts, then the two interiors -- A's interior reads `in` from plane own0 + G on, which is
            // where B's end stopped writing it.
            if (triples_ok && pair_split && done == 2 && rest == 6 && depth == 6) {
                bool ran;
                if (ends_first) {
                    ran = gpu_jacobi_sweep_triple_ranges(*in, div, *out, g.ni, g.nj, g.nk, own0 - 3, own0 + G + 3, own1 - G - 3, own1 + 3, alpha, beta) != 0;
                    // (all four launches must apply or none: the same kernel on the same rows either takes every one of these
                    // ranges or refused the first; a later refusal would leave `in` half-updated, so it is an error, not a fall-back)
                    int later = 1;
                    if (ran) {
                        later &= gpu_jacobi_sweep_triple_ranges(*out, div, *in, g.ni, g.nj, g.nk, own0, own0 + G, own1 - G, own1, alpha, beta) != 0;
                        { float *ptr = in->get(); size_t pe = in->plane; int ex = 0; fl_halo_exchange(1, &ptr, &pe, &ex, g.nk, G, G, /*wait=*/0); }
                        in_flight = true;
                        later &= gpu_jacobi_sweep_triple_ranges(*in, div, *out, g.ni, g.nj, g.nk, own0 + G + 3, own1 - G - 3, 0, 0, alpha, beta) != 0;
                        later &= gpu_jacobi_sweep_triple_ranges(*out, div, *in, g.ni, g.nj, g.nk, own0 + G, own1 - G, 0, 0, alpha, beta) != 0;
                    }
                    if (!later) fl_report_error(FL_ERR_UNSUPPORTED, "projection on slabs: a follow-up three-sweep launch was refused after the first one ran");
                } else {
                    ran = gpu_jacobi_sweep_triple_ranges(*in, div, *out, g.ni, g.nj, g.nk, own0 - 3, own1 + 3, 0, 0, alpha, beta) != 0;
                    if (ran && !gpu_jacobi_sweep_triple_ranges(*out, div, *in, g.ni, g.nj, g.nk, own0, own1, 0, 0, alpha, beta))
                        fl_report_error(FL_ERR_UNSUPPORTED, "projection on slabs: the second three-sweep launch was refused after the first one ran");
                }
                if (ran) { rest = 0; depth = 0; }                        // (two launches per piece: the newest iterate is back in `in`)
                else triples_ok = false;                                 // the kernels do not apply to this grid: pairs from now on
            }
            while (pair_split && rest >= 2) {
                if (ends_first && rest == 4) {
                    // the last two pairs, ends first (depth is 4 here: this pair covers [own0 - 2, own1 + 2), the last [own0, own1))
                    if (!gpu_jacobi_sweep_pair_ranges(*in, div, *out, g.ni, g.nj, g.nk, own0 - 2, own0 + G + 2, own1 - G - 2, own1 + 2, alpha, beta)) break;
                    gpu_jacobi_sweep_pair_ranges(*out, div, *in, g.ni, g.nj, g.nk, own0, own0 + G, own1 - G, own1, alpha, beta);
                    // `in` now holds the chunk's result on the planes the neighbours need: send them, receive theirs
                    { float *ptr = in->get(); size_t pe = in->plane; int ex = 0; fl_halo_exchange(1, &ptr, &pe, &ex, g.nk, G, G, /*wait=*/0); }
                    in_flight = true;
                    gpu_jacobi_sweep_pair_ranges(*in, div, *out, g.ni, g.nj, g.nk, own0 + G + 2, own1 - G - 2, 0, 0, alpha, beta);
                    gpu_jacobi_sweep_pair_ranges(*out, div, *in, g.ni, g.nj, g.nk, own0 + G, own1 - G, 0, 0, alpha, beta);
                    rest -= 4; depth -= 4;                               // (two pairs: the newest iterate is back in `in`)
                    break;
                }
                if (!gpu_jacobi_sweep_pair_ranges(*in, div, *out, g.ni, g.nj, g.nk, own0 - (depth - 2), own1 + (depth - 2), 0, 0, alpha, beta)) break;
                std::swap(in, out); rest -= 2; depth -= 2;
            }
            const int w2 = rest > 0 ? gpu_jacobi_sweeps(*in, div, *out, g.ni, g.nj, g.nk, rest, alpha, beta) : 0;
            DeviceField *newest = w2 ? out : in;
            where = newest == cur ? 0 : 1;
        }
        int v = cur->valid;
        for (int s = 0; s < chunk; s++) v = std::min(v - 1, div.valid);     // each sweep reaches one plane
        if (where) std::swap(cur, oth);
        gs.produced(*cur, v);
        gs.produced(*oth, 0);
        left -= chunk;
    }
    fl_set_option(FL_OPT_JACOBI_FUSE, fuse_was);
    if (cur != &p) p.swap(p_temp);                                       // the newest iterate becomes `p` (no copy-back)
    gs.require({ &p }, 1);
    const int vu = std::min(VelocityU.valid, p.valid), vv = std::min(VelocityV.valid, p.valid), vw = std::min(VelocityW.valid, p.valid - 1);
    if (with_delta) {
        gpu_gradient_delta(VelocityU, VelocityV, VelocityW, p, duProj, dvProj, dwProj, g.ni, g.nj, g.nk, halfrdx);
        gs.produced(duProj, vu); gs.produced(dvProj, vv); gs.produced(dwProj, vw);
    } else {
        gpu_gradient(VelocityU, VelocityV, VelocityW, p, g.ni, g.nj, g.nk, halfrdx);
    }
    gs.produced(VelocityU, vu);
    gs.produced(VelocityV, vv);
    gs.produced(VelocityW, vw);
    return with_delta;
}

// The two-level advection of blend != 1 (Mapping.cpp:383-390) samples the *Prev fields through the previous backward map.  With
// the reference's zeroed map border a look-up that meets border cells returns s * q, s in [0, 1]: anywhere between
// the origin and the node -- on a z-slab rank, on any other rank's planes.  The *Prev fields change at a re-initialisation and
// nowhere else, so that is when every rank assembles whole-grid copies of them (gpuMapper::assembleGlobal: one message per peer
// and field); the advectors find the copies through gpuMapper::globalTwin.  BQ_OPT_KEEP_DMC_BORDER = 1 has no zeroed cells and
// needs none of this.
bool BimocqGPUSolver::wholeGridPrev() const
{
    const gpuMapper &gs = *GpuSolver;
    return whole_grid_prev && gs.slab.on && gs.slab.nranks > 1 && VelocityAdvector.BlendCoeff != 1.f && !VelocityAdvector.keepDmcBorder;
}

// :503-516.  UPrev <- UInit by swap (UInit is refilled right after), UInit <- U by copy.
void BimocqGPUSolver::velocityReinitialize()
{
    VelocityUPrev.swap(VelocityUInit); VelocityVPrev.swap(VelocityVInit); VelocityWPrev.swap(VelocityWInit);
    VelocityUInit.copy_from(VelocityU); VelocityVInit.copy_from(VelocityV); VelocityWInit.copy_from(VelocityW);
    if (wholeGridPrev())
        GpuSolver->assembleGlobal({ { &VelocityUPrev, &VelocityUPrevAll }, { &VelocityVPrev, &VelocityVPrevAll }, { &VelocityWPrev, &VelocityWPrevAll } });
    else GpuSolver->dropGlobalTwins();
}

// :518-527
void BimocqGPUSolver::scalarReinitialize()
{
    DensityPrev.swap(DensityInit); TemperaturePrev.swap(TemperatureInit);
    DensityInit.copy_from(Density); TemperatureInit.copy_from(Temperature);
    if (wholeGridPrev())
        GpuSolver->assembleGlobal({ { &DensityPrev, &DensityPrevAll }, { &TemperaturePrev, &TemperaturePrevAll } });
    else GpuSolver->dropGlobalTwins();
}

static BimocqGPUSolver *g_trace_solver = nullptr;
static int g_trace_frame = 0;
static void trace_hook_fn(const char *stage) { if (g_trace_solver) trace_stage(*g_trace_solver, stage, g_trace_frame); }

// BQ_OPT_PROFILE_PHASES: one event per phase boundary on the compute stream (it closes the running span and opens the
// next); the spans are summed in phaseTotals()
void BimocqGPUSolver::phaseMark(int phase)
{
    if (!profile_phases && phase_open_ < 0) return;
    void *ev = fl_event_create();
    if (!ev) return;
    fl_event_record(ev);
    bool used = false;
    if (phase_open_ >= 0) { phase_spans_.push_back(PhaseSpan{ phase_open_ev_, ev, phase_open_ }); used = true; }
    if (profile_phases && phase >= 0 && phase < PH_COUNT) { phase_open_ev_ = ev; phase_open_ = phase; used = true; }
    else { phase_open_ev_ = nullptr; phase_open_ = -1; }
    if (!used) fl_event_destroy(ev);
}

void BimocqGPUSolver::phaseTotals(double ms[PH_COUNT], long long *steps, bool reset)
{
    for (int a = 0; a < PH_COUNT; a++) ms[a] = 0.0;
    for (const PhaseSpan &sp : phase_spans_) {
        const float t = fl_event_elapsed_ms(sp.a, sp.b);
        if (t > 0.f) ms[sp.phase] += (double)t;
    }
    if (steps) *steps = phase_steps_;
    if (reset) {
        std::vector<void *> evs;
        for (const PhaseSpan &sp : phase_spans_) { evs.push_back(sp.a); evs.push_back(sp.b); }
        std::sort(evs.begin(), evs.end());
        evs.erase(std::unique(evs.begin(), evs.end()), evs.end());
        for (void *e : evs) if (e != phase_open_ev_) fl_event_destroy(e);
        phase_spans_.clear();
        phase_steps_ = 0;
    }
}

// :129-230
void BimocqGPUSolver::advanceBimocq(int framenum, float dt)
{
    gpuMapper &gs = *GpuSolver;
    g_trace_solver = this; g_trace_frame = framenum; g_trace_hook = trace_hook_fn;
    if (framenum == 0) MaxVelocity = CellSize;           // :131 (overwritten by getCFL, kept for the record)
    float proj_coeff = 2.f;
    phaseMark(PH_MAPS);
    const float cfldt = getCFL();                        // :136
    last_cfldt = cfldt;
    // how many cells anything can travel this step (bounds the reach of the gather kernels on slab ranks)
    const int dcells = (int)std::ceil((double)dt * (double)MaxVelocity / (double)CellSize) + 1;

    // :138-139.  One update serves both advectors (shared map set).
    VelocityAdvector.updateMapping(VelocityU, VelocityV, VelocityW, cfldt, dt, dcells);
    if (!ScalarAdvector.sharesMaps()) ScalarAdvector.updateMapping(VelocityU, VelocityV, VelocityW, cfldt, dt, dcells);
    trace_stage(*this, "maps", framenum);
    phaseMark(PH_ADVECT);

    // :143-145
    VelocityAdvector.advectVelocity(VelocityU, VelocityV, VelocityW, VelocityUInit, VelocityVInit, VelocityWInit,
                                    VelocityUPrev, VelocityVPrev, VelocityWPrev);
    // density and temperature live on the same nodes and use the same maps: batched (one map look-up)
    ScalarAdvector.advectFields2(Density, DensityInit, DensityPrev, Temperature, TemperatureInit, TemperaturePrev);
    const bool policy1 = reinit_policy == 1;
    trace_stage(*this, "advect", framenum);
    phaseMark(PH_FORCES);

    // Dead state: what :213-214 accumulates into *Init before a re-initialisation moves to *Prev (:503-511), and *Prev
    // is read by the two-level advection only when blend != 1 (Mapping.cpp:383-390).  With blend == 1 and a
    // re-initialisation every frame (the reference's GPU solver) nothing ever reads it: the force delta, its
    // snapshot and that accumulation are not executed (BQ_OPT_FULL_STATE = 1 executes them; every field a caller
    // can observe is the same either way, tests/test_gpu_solver.py).
    const bool prev_dead = !keep_full_state && reinit_policy == 0 && VelocityAdvector.BlendCoeff == 1.f;
    // Buoyancy acts on v only: unless a source imposes its velocity ring this frame or viscosity smooths all three
    // components, u and w stay what they are until the projection.
    bool forces_touch_uw = Viscosity != 0.f;
    for (const Emitter &e : sim_emitter) forces_touch_uw = forces_touch_uw || framenum < e.emitFrame;
    // :157-159 snapshots for the force delta (:175-177).  The u and w snapshots are only ever read by that delta
    // (with the Jacobi projection the gradient pass hands out d*Proj, :179-193 needs no snapshot) and by the
    // viscous diffusion, which uses them as work arrays: not taken when nothing will touch u and w.
    if (!prev_dead) {
        if (forces_touch_uw || projection_kind != BQ_PROJECTION_JACOBI) { VelocityUTemp.copy_from(VelocityU); VelocityWTemp.copy_from(VelocityW); }
        VelocityVTemp.copy_from(VelocityV);
    }
    // policy 1 follows the CPU solver (BimocqSolver.cpp:129-133): scalar snapshots BEFORE the sources act
    if (policy1) { DensityTemp.copy_from(Density); TemperatureTemp.copy_from(Temperature); }

    emitSmoke(framenum, dt);                             // :164
    addBuoyancy(dt);                                     // :165

    if (Viscosity) {                                     // :167-172, with the reference's buffer aliasing
        if (gs.slab.on && gs.slab.nranks > 1) {
            diffuseFieldSlab(VelocityU, VelocityUTemp, TempSrcU, g.ni + 1, g.nj, g.nk, 20, Viscosity, dt);
            diffuseFieldSlab(VelocityV, VelocityVTemp, TempSrcV, g.ni, g.nj + 1, g.nk, 20, Viscosity, dt);
            diffuseFieldSlab(VelocityW, VelocityWTemp, TempSrcW, g.ni, g.nj, g.nk + 1, 20, Viscosity, dt);
        } else {
            diffuseField(VelocityU, VelocityUTemp, TempSrcU, g.ni + 1, g.nj, g.nk, 20, Viscosity, dt);
            diffuseField(VelocityV, VelocityVTemp, TempSrcV, g.ni, g.nj + 1, g.nk, 20, Viscosity, dt);
            diffuseField(VelocityW, VelocityWTemp, TempSrcW, g.ni, g.nj, g.nk + 1, 20, Viscosity, dt);
        }
    }

    // :175-177 velocity change due to external forces.  When nothing touched u and w they still equal their
    // snapshots, U - UTemp is identically +0, and accumulating blend9 of a zero field (:213) changes no value --
    // neither the difference nor its look-up is executed for u and w then.
    if (prev_dead) {
        // nothing: d*Extern only feeds the accumulation that is dead
    } else if (forces_touch_uw) {
        gs.addFields(duExtern, VelocityU, VelocityUTemp, -1.f, g.nu());
        gs.addFields(dwExtern, VelocityW, VelocityWTemp, -1.f, g.nw());
        gs.produced(duExtern, std::min(VelocityU.valid, VelocityUTemp.valid));
        gs.produced(dwExtern, std::min(VelocityW.valid, VelocityWTemp.valid));
    }
    if (!prev_dead) {
        gs.addFields(dvExtern, VelocityV, VelocityVTemp, -1.f, g.nv());
        gs.produced(dvExtern, std::min(VelocityV.valid, VelocityVTemp.valid));
    }
    // :179-193: UTemp <- U, projection, dProj = U - UTemp.  With the Jacobi projection the gradient pass hands the
    // change out itself (same floats, same subtraction): no snapshot, no subtraction pass.
    const bool delta_from_gradient = projection_kind == BQ_PROJECTION_JACOBI;
    if (!delta_from_gradient) { VelocityUTemp.copy_from(VelocityU); VelocityVTemp.copy_from(VelocityV); VelocityWTemp.copy_from(VelocityW); }

    trace_stage(*this, "forces", framenum);
    phaseMark(PH_PROJECTION);
    const bool have_delta = projection(delta_from_gradient);             // :183
    trace_stage(*this, "projection", framenum);
    phaseMark(PH_ACCUMULATE);

    if (!have_delta) {
        // :188-193 dProj = U - UTemp.  The reference copies U into dProj and then adds -1*UTemp in
        // place; out = U + (-1)*UTemp is the same expression in one pass.
        gs.addFields(duProj, VelocityU, VelocityUTemp, -1.f, g.nu());
        gs.addFields(dvProj, VelocityV, VelocityVTemp, -1.f, g.nv());
        gs.addFields(dwProj, VelocityW, VelocityWTemp, -1.f, g.nw());
        gs.produced(duProj, std::min(VelocityU.valid, VelocityUTemp.valid));
        gs.produced(dvProj, std::min(VelocityV.valid, VelocityVTemp.valid));
        gs.produced(dwProj, std::min(VelocityW.valid, VelocityWTemp.valid));
    }
    // :185-186,195-198: DensityExtern = Density - DensityTemp right after DensityTemp <- Density is
    // identically zero, and accumulating a zero field (:215-216) adds 0: not executed.

    bool velReinit = true, scalarReinit = true;
    if (policy1) {
        // BimocqSolver.cpp:160-185: what the sources added to the scalars, the distortion of each map set in
        // units of the step's travel, and the CPU solver's thresholds
        gs.addFields(DensityExtern, Density, DensityTemp, -1.f, g.n());
        gs.addFields(TemperatureExtern, Temperature, TemperatureTemp, -1.f, g.n());
        gs.produced(DensityExtern, std::min(Density.valid, DensityTemp.valid));
        gs.produced(TemperatureExtern, std::min(Temperature.valid, TemperatureTemp.valid));
        last_vel_distortion = VelocityAdvector.estimateDistortion() / (MaxVelocity * dt);
        last_scalar_distortion = ScalarAdvector.estimateDistortion() / (MaxVelocity * dt);
        velReinit = scalarReinit = false;
        if (last_vel_distortion > 1.f || framenum - vel_lastReinit > 10) { velReinit = true; vel_lastReinit = framenum; proj_coeff = 1.f; }
        if (last_scalar_distortion > 5.f || framenum - scalar_lastReinit > 30) { scalarReinit = true; scalar_lastReinit = framenum; }
        if (travel_limit > 0) {
            // BQ_OPT_REINIT_MAX_TRAVEL: the next update samples the velocity up to Dfwd + dcells + 2 planes away and the
            // next advection its fields up to Dback + dcells + 2 (mapping.cpp: reachField), dcells taken as this step's.
            // A set that would not fit is re-initialised now -- the same decision on every rank (the travel is all-reduced)
            // and on one GPU given the same limit.  Should the flow speed up beyond that estimate the next step's require()
            // refuses ("ghost zone too shallow") rather than compute from missing planes.
            const auto outgrows = [&](const MapSet &ms) { return std::max(ms.Dback, ms.Dfwd) + dcells + 2 > travel_limit; };
            if (!velReinit && outgrows(*VelocityAdvector.maps)) { velReinit = true; vel_lastReinit = framenum; proj_coeff = 1.f; forced_reinits++; }
            if (!scalarReinit && outgrows(*ScalarAdvector.maps)) { scalarReinit = true; scalar_lastReinit = framenum; forced_reinits++; }
        }
    } else {
        if (framenum - vel_lastReinit > 10) {                // :200-205
            vel_lastReinit = framenum;
            proj_coeff = 1.f;
        }
        if (framenum - scalar_lastReinit > 30) {             // :207-211
            scalar_lastReinit = framenum;
        }
    }

    // :213-214
    if (!prev_dead)
        VelocityAdvector.accumulateVelocity2(VelocityUInit, VelocityVInit, VelocityWInit,
                                             duExtern, dvExtern, dwExtern, 1.f, duProj, dvProj, dwProj, proj_coeff, !forces_touch_uw);
    if (policy1) {                                       // BimocqSolver.cpp:191-192 (:215-216 here adds zeros)
        ScalarAdvector.accumulateField(DensityInit, DensityExtern);
        ScalarAdvector.accumulateField(TemperatureInit, TemperatureExtern);
    }
    trace_stage(*this, "accumulate", framenum);

    // :218-223 `if (1)`: re-initialise every frame; policy 1: when the rules above say so
    if (velReinit) {
        vel_reinits++;
        VelocityAdvector.reinitializeMapping();
        velocityReinitialize();
        VelocityAdvector.accumulateVelocity(VelocityUInit, VelocityVInit, VelocityWInit, duProj, dvProj, dwProj, 1.f);
    }
    // :225-229
    if (scalarReinit) {
        scalar_reinits++;
        if (ScalarAdvector.sharesMaps()) ScalarAdvector.noteSharedReinit();
        else ScalarAdvector.reinitializeMapping();
        scalarReinitialize();
    }
    steps_taken++;
    if (phase_open_ >= 0) phase_steps_++;
    phaseMark(PH_COUNT);
    trace_stage(*this, "reinit", framenum);
}

bool BimocqGPUSolver::outputResultAsync(unsigned frame, const std::string &filepath)
{
    waitOutput();
    if (!dump_host_) dump_host_ = static_cast<float *>(fl_malloc_host(g.n() * sizeof(float)));
    if (!dump_host_) return false;
    // The next advance() rewrites Density in place (and swaps buffers) long before a 67 MB (256^3) or 2 GB download
    // has left the device, and the copy stream's download is ordered only against compute work queued BEFORE it.  So
    // the frame is first snapshot on the compute stream (a device-to-device copy at HBM rate: 25 us at 256^3) and the
    // copy stream downloads the snapshot; the simulation never waits for PCIe.  waitOutput() above has retired the
    // previous download, so the snapshot buffer is free.
    if (dump_dev_.count() != g.n() && !dump_dev_.alloc(g.n())) return false;
    fl_memcpy_d2d(dump_dev_.get(), Density.get(), g.n() * sizeof(float));
    void *ticket = fl_download_begin(dump_host_, dump_dev_.get(), g.n() * sizeof(float));
    if (!ticket) return false;
    const SlabCtx sl = GpuSolver->slab;
    const GridDims gd = g;
    const float h = CellSize;
    float *host = dump_host_;
    dump_thread_ = std::thread([this, ticket, frame, filepath, sl, gd, h, host] {
        if (fl_download_wait(ticket) != FL_OK) { dump_result_ = -1; return; }
        if (filepath.empty()) { dump_result_ = 0; return; }
        if (!sl.on) { dump_result_ = write_density_dump(frame + 1, filepath, h, host, gd.ni, gd.nj, gd.nk, 0, gd.nk); return; }
        const size_t plane = (size_t)gd.ni * gd.nj;
        dump_result_ = write_density_dump(frame + 1, filepath, h, host + plane * sl.G, gd.ni, gd.nj, sl.own1 - sl.own0, sl.own0, sl.nkg);
    });
    return true;
}

long BimocqGPUSolver::waitOutput()
{
    if (dump_thread_.joinable()) dump_thread_.join();
    return dump_result_;
}

BimocqGPUSolver::~BimocqGPUSolver()
{
    waitOutput();
    if (dump_host_) fl_free_host(dump_host_);
}

// :536-543.  A slab rank downloads its local planes and writes the planes it owns.
long BimocqGPUSolver::outputResult(unsigned frame, const std::string &filepath)
{
    Density.download(host_density.data());
    VelocityU.download(host_u.data());
    VelocityV.download(host_v.data());
    VelocityW.download(host_w.data());
    if (fl_last_error() != FL_OK) return -1;
    if (filepath.empty()) return 0;
    const SlabCtx &sl = GpuSolver->slab;
    if (!sl.on)
        return write_density_dump(frame + 1, filepath, CellSize, host_density.data(), g.ni, g.nj, g.nk, 0, g.nk);
    const size_t plane = (size_t)g.ni * g.nj;
    return write_density_dump(frame + 1, filepath, CellSize, host_density.data() + plane * sl.G, g.ni, g.nj,
                              sl.own1 - sl.own0, sl.own0, sl.nkg);
}

} // namespace bqhost
