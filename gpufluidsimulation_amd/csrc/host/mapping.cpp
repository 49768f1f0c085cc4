// mapping.cpp -- see mapping.hpp.  Reference: src/bimocq3D/Mapping.cpp:276-447.
#include "mapping.hpp"

#include <algorithm>
#include <cmath>

namespace bqhost {

// How far (in planes) the kernels reach along z, as a function of the map displacement bound D:
//   a map sampled at the 9 sub-voxel points of a node touches planes k-1..k+1          -> 1
//   a field sampled at a mapped position that sits <= D cells from the node, +-0.25h
//   offset, trilinear footprint                                                          -> D + 2
//   DMC sub-step: velocity at x and x -+ h, map at a point <= 1 cell away (CFL sub-step)  -> 3
static const int kReachMap = 1;
static inline int reachField(int D) { return D + 2; }
static const int kReachDMC = 3;

// FL_OPT_MAP_QUARTER_FP32 for the duration of one operator call on a map set whose values have been checked
struct QuarterScope {
    explicit QuarterScope(bool on) : on_(on) { if (on_) fl_set_option(FL_OPT_MAP_QUARTER_FP32, 1); }
    ~QuarterScope() { if (on_) fl_set_option(FL_OPT_MAP_QUARTER_FP32, 0); }
    QuarterScope(const QuarterScope &) = delete;
    QuarterScope &operator=(const QuarterScope &) = delete;
private:
    bool on_;
};

bool MapSet::alloc(const gpuMapper &m)
{
    DeviceField *all[] = { &ForwardX, &ForwardY, &ForwardZ, &BackwardX, &BackwardY, &BackwardZ,
                           &BackwardXPrev, &BackwardYPrev, &BackwardZPrev, &InitX, &InitY, &InitZ };
    for (DeviceField *f : all)
        if (!m.allocField(*f, FIELD_S)) return false;
    return true;
}

// Mapping.cpp:276-345: all twelve map fields start as the identity (i*h, j*h, k*h)
bool MapperBaseGPU::init(int ni, int nj, int nk, float h, float coeff, gpuMapper *mymapper)
{
    g.ni = ni; g.nj = nj; g.nk = nk; g.h = h;
    BlendCoeff = coeff;
    TotalReinitCount = 0;
    gpuSolver = mymapper;
    maps = std::make_shared<MapSet>();
    if (!maps->alloc(*mymapper)) return false;
    MapSet &m = *maps;
    gpu_init_maps(m.InitX, m.InitY, m.InitZ, h, ni, nj, nk);    // host loop + H2D in the reference (:306-328)
    m.ForwardX.copy_from(m.InitX); m.ForwardY.copy_from(m.InitY); m.ForwardZ.copy_from(m.InitZ);
    m.BackwardX.copy_from(m.InitX); m.BackwardY.copy_from(m.InitY); m.BackwardZ.copy_from(m.InitZ);
    m.BackwardXPrev.copy_from(m.InitX); m.BackwardYPrev.copy_from(m.InitY); m.BackwardZPrev.copy_from(m.InitZ);
    return true;
}

// Mapping.cpp:347-352
void MapperBaseGPU::updateMapping(DeviceField &U, DeviceField &V, DeviceField &W, float cfldt, float dt, int dcells)
{
    // both updates flag map values outside tile_value_ok's range as they store them (fl_map_guard_*): one 8-byte
    // read-back afterwards tells whether the 9-point operators may run their weight-1/4 map lerps in fp32
    fl_map_guard_reset(0);
    fl_map_guard_reset(1);
    // One GPU, BQ_OPT_CONCURRENT_MAPS (off by default): the forward update (RK3 particle trace, Mapping.cpp:370-373) and the
    // backward one's DMC sub-steps (:354-368) both read the velocity and touch disjoint map arrays, so the forward kernel can run
    // on the library's auxiliary stream BESIDE the sub-steps (fl_aux_*).  Measured in round 4: no gain -- each kernel fills the
    // chip on its own.  Slab ranks keep the order in any case: their exchanges are ordered against one compute stream.
    const bool beside = gpuSolver->concurrent_maps && !gpuSolver->slab.on;
    if (beside) {
        fl_aux_begin();
        updateForward(U, V, W, cfldt, dt, dcells);
        fl_aux_end();
        updateBackward(U, V, W, cfldt, dt, dcells);
        fl_aux_join();
    } else {
        updateBackward(U, V, W, cfldt, dt, dcells);
        updateForward(U, V, W, cfldt, dt, dcells);
    }
    int ok[2] = { 0, 0 };
    fl_map_guard_read(ok);
    fl_map_guard_reset(-1);
    maps->backQ4 = maps->backQ4 && ok[0] == 1;       // (border nodes and stale planes keep older values: stay conservative
    maps->fwdQ4 = maps->fwdQ4 && ok[1] == 1;         //  until the next re-initialisation once a value has failed)
    if (measureTravel) {
        // the displacement bounds become what the maps really do along z -- the only axis a z-slab's ghost depth cares about
        // (on a single GPU nothing reads them; the measurement still runs so that the policy decides alike on any rank count)
        gpu_map_travel_z(maps->BackwardZ, maps->ForwardZ, g.h, g.ni, g.nj, g.nk, lastTravel);
        const auto cells = [](float t) { return t < 1.0e6f ? (int)std::ceil((double)t) : 1000000; };   // (inf: a NaN in the map)
        maps->Dback = cells(lastTravel[0]);
        maps->Dfwd = cells(lastTravel[1]);
    }
}

// Mapping.cpp:354-368.  The reference copies x_out -> Backward after every sub-step
// (GPU_Advection.h:466-468); here sub-steps ping-pong between the mapper's two scratch sets and
// only the final result is copied into Backward.  Both scratch sets keep zero border nodes,
// exactly like the reference's x_out, so the copied-in border is the same.
void MapperBaseGPU::updateBackward(DeviceField &U, DeviceField &V, DeviceField &W, float cfldt, float dt, int dcells)
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    float T = 0.f;
    float substep = cfldt;
    DeviceField *in[3] = { &m.BackwardX, &m.BackwardY, &m.BackwardZ };
    DeviceField *ping[2][3] = { { &gs.x_out, &gs.y_out, &gs.z_out }, { &gs.x_out2, &gs.y_out2, &gs.z_out2 } };
    int which = 0;
    bool any = false, swap_result = false;
    while (T < dt) {
        if (T + substep > dt) substep = dt - T;
        DeviceField **out = ping[which];
        // border nodes of the output: zeros (the reference's cleared scratch) or, with keepDmcBorder, the input's
        // values -- written by the kernel itself when the operator library can do that, else prepared here (the
        // scratch sets then have to keep their zero border, so the result is copied out rather than swapped)
        FusedScope fused(gs.fuse_housekeeping, keepDmcBorder ? 8 : 4);
        swap_result = fused.on;
        if (keepDmcBorder && !fused.on) { out[0]->copy_from(*in[0]); out[1]->copy_from(*in[1]); out[2]->copy_from(*in[2]); }
        gs.withGhosts({ { { &U, &V, &W, in[0], in[1], in[2] }, kReachDMC } }, [&] {
            gs.solveBackwardDMC(U, V, W, *in[0], *in[1], *in[2], *out[0], *out[1], *out[2], substep);
        }, gs.validAfter({ &U, &V, &W, in[0], in[1], in[2] }, kReachDMC) - kReachDMC, { out[0], out[1], out[2] });
        const int v = gpuMapper::minValid({ &U, &V, &W, in[0], in[1], in[2] }) - kReachDMC;
        gs.producedAll({ out[0], out[1], out[2] }, v);
        in[0] = out[0]; in[1] = out[1]; in[2] = out[2];
        which ^= 1;
        any = true;
        T += substep;
    }
    if (any && swap_result) {
        // every node of the newest scratch set was written by the last sub-step: it becomes the map, the old map's
        // buffers become scratch (their content is dead: a sub-step writes all of its output)
        m.BackwardX.swap(*in[0]); m.BackwardY.swap(*in[1]); m.BackwardZ.swap(*in[2]);
    } else if (any) {
        m.BackwardX.copy_from(*in[0]); m.BackwardY.copy_from(*in[1]); m.BackwardZ.copy_from(*in[2]);
    }
    m.Dback += dcells;
}

// Mapping.cpp:370-373.  In place; a node's trace starts at its own map value and samples the
// velocity along a path that stays within Dfwd + dcells cells of the node.
void MapperBaseGPU::updateForward(DeviceField &U, DeviceField &V, DeviceField &W, float cfldt, float dt, int dcells)
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    const int reach = reachField(m.Dfwd + dcells);
    gs.withGhosts({ { { &U, &V, &W }, reach } }, [&] {
        gs.solveForward(U, V, W, m.ForwardX, m.ForwardY, m.ForwardZ, cfldt, dt);
    }, std::min(gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }), gs.validAfter({ &U, &V, &W }, reach) - reach),
       { &m.ForwardX, &m.ForwardY, &m.ForwardZ });
    const int v = std::min(gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }),
                           gpuMapper::minValid({ &U, &V, &W }) - reach);
    gs.producedAll({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }, v);
    m.Dfwd += dcells;
    m.fwdIdentity = false;
}

// Mapping.cpp:375-391.  gpu_compensate_velocity is issued as its four stages (GPU_kernel.cu:652-665)
// so that a slab rank can refresh ghost planes between them; the arithmetic is the same.
void MapperBaseGPU::advectVelocity(DeviceField &U, DeviceField &V, DeviceField &W,
                                   DeviceField &Ui, DeviceField &Vi, DeviceField &Wi,
                                   DeviceField &Up, DeviceField &Vp, DeviceField &Wp)
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    const float h = g.h;
    const int ni = g.ni, nj = g.nj, nk = g.nk;

    // advect: U(x) = blend9(Ui(psi_back(x)))
    gs.withGhosts({ { { &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap }, { { &Ui, &Vi, &Wi }, reachField(m.Dback) } }, [&] {
        QuarterScope q4(m.backQ4);
        gs.advectVelocity(U, V, W, Ui, Vi, Wi, m.BackwardX, m.BackwardY, m.BackwardZ, false);
    }, std::min(gs.validAfter({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap) - kReachMap,
                gs.validAfter({ &Ui, &Vi, &Wi }, reachField(m.Dback)) - reachField(m.Dback)), { &U, &V, &W });
    gs.producedAll({ &U, &V, &W }, std::min(gpuMapper::minValid({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }) - kReachMap,
                                            gpuMapper::minValid({ &Ui, &Vi, &Wi }) - reachField(m.Dback)));

    trace_point("v.advect");
    // stage 1: error at time 0, u_src = blend9(U(psi_fwd(x))) - Ui(x)      (GPU_Advection.h:499-501 zeroes u_src)
    bool stage2_done;
    // The error itself is correct on fewer ghost planes, but with the fused stage 2 the same launch leaves the copy of U in
    // Ui, and that copy is wanted wherever U is correct (the limiter reads it): the kernels run on all of those planes.
    const int err_valid = std::max(0, gs.validAfter({ &U, &V, &W }, reachField(m.Dfwd)));
    {
        // fused: the kernels zero the error outside their window and store U into Ui after reading it (stage 2)
        FusedScope fused(gs.fuse_housekeeping, 1 | 2);
        stage2_done = fused.on;
        if (!fused.on) { gs.u_src.zero(); gs.v_src.zero(); gs.w_src.zero(); }
        gs.withGhosts({ { { &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap }, { { &U, &V, &W }, reachField(m.Dfwd) } }, [&] {
            QuarterScope q4(m.fwdQ4);
            gpu_compensate_error_velocity(U, V, W, Ui, Vi, Wi, gs.u_src, gs.v_src, gs.w_src,
                                          m.ForwardX, m.ForwardY, m.ForwardZ, h, ni, nj, nk, false);
        }, err_valid, { &gs.u_src, &gs.v_src, &gs.w_src, &Ui, &Vi, &Wi });
    }
    gs.producedAll({ &gs.u_src, &gs.v_src, &gs.w_src },
                   std::min({ gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }) - kReachMap,
                              gpuMapper::minValid({ &U, &V, &W }) - reachField(m.Dfwd),
                              gpuMapper::minValid({ &Ui, &Vi, &Wi }) }));
    trace_point("v.stage1");
    // stage 2: Ui <- uncompensated U (clobbers the caller's init)
    // (the fused copy exists on the planes the error kernels ran on: owned + err_valid ghost planes)
    if (stage2_done) { Ui.valid = std::min(U.valid, err_valid); Vi.valid = std::min(V.valid, err_valid); Wi.valid = std::min(W.valid, err_valid); }
    else { Ui.copy_from(U); Vi.copy_from(V); Wi.copy_from(W); }
    // stage 3: U += blend9(-0.5 * u_src(psi_back(x)))
    // z-slab ranks, zeroed map border (Q13): the wall layers sample the error far outside the slab -> wall_sheets.hpp.  The
    // sheets other ranks need are cut from the owned planes of the error field now, so that the messages travel on the
    // halo stream while the operator runs; what arrives corrects the wall layers afterwards
    if (!keepDmcBorder)
        gs.wallFixupBegin({ { &gs.u_src, &Ui, &U, FIELD_U }, { &gs.v_src, &Vi, &V, FIELD_V }, { &gs.w_src, &Wi, &W, FIELD_W } },
                          m.Dback, reachField(m.Dback));
    gs.withGhosts({ { { &gs.u_src, &gs.v_src, &gs.w_src }, reachField(m.Dback) }, { { &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap } }, [&] {
        QuarterScope q4(m.backQ4);
        gs.accumulateVelocity(gs.u_src, gs.v_src, gs.w_src, U, V, W, m.BackwardX, m.BackwardY, m.BackwardZ, false, -0.5f);
    }, std::min({ gpuMapper::minValid({ &U, &V, &W }), gs.validAfter({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap) - kReachMap,
                  gs.validAfter({ &gs.u_src, &gs.v_src, &gs.w_src }, reachField(m.Dback)) - reachField(m.Dback) }), { &U, &V, &W });
    const int stage3_valid = std::min({ gpuMapper::minValid({ &U, &V, &W }),
                                        gpuMapper::minValid({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }) - kReachMap,
                                        gpuMapper::minValid({ &gs.u_src, &gs.v_src, &gs.w_src }) - reachField(m.Dback) });
    if (!keepDmcBorder) gs.wallFixupEnd(m.BackwardX, m.BackwardY, m.BackwardZ, -0.5f, stage3_valid);
    gs.producedAll({ &U, &V, &W }, stage3_valid);
    trace_point("v.stage3");
    // stage 4: limiter against the 3x3x3 box of the uncompensated field
    gs.require({ &Ui, &Vi, &Wi }, 1);
    gpu_clamp_extrema_box(Ui, U, ni + 1, nj, nk);
    gpu_clamp_extrema_box(Vi, V, ni, nj + 1, nk);
    gpu_clamp_extrema_box_w(Wi, W, ni, nj, nk + 1);
    gs.producedAll({ &U, &V, &W }, std::min(gpuMapper::minValid({ &U, &V, &W }), gpuMapper::minValid({ &Ui, &Vi, &Wi }) - 1));

    const float blend = (TotalReinitCount != 0) ? BlendCoeff : 1.f;
    // z-slab ranks, zeroed map border: the second look-up may land anywhere below the node (include/bimocq_gpu.h,
    // gpu_advect_vel_double_global) -- the solver keeps whole-grid copies of the *Prev fields for it
    const float *ug = blend != 1.f ? gs.globalTwin(&Up) : nullptr, *vg = blend != 1.f ? gs.globalTwin(&Vp) : nullptr,
                *wg = blend != 1.f ? gs.globalTwin(&Wp) : nullptr;
    const bool whole = ug && vg && wg;
    if (blend != 1.f) {
        gs.require({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap);
        gs.require({ &m.BackwardXPrev, &m.BackwardYPrev, &m.BackwardZPrev }, reachField(m.Dback));
        if (!whole) gs.require({ &Up, &Vp, &Wp }, reachField(m.Dback + m.DbackPrev));
    }
    if (whole)
        gs.advectVelocityDoubleGlobal(U, V, W, ug, vg, wg, m.BackwardX, m.BackwardY, m.BackwardZ,
                                      m.BackwardXPrev, m.BackwardYPrev, m.BackwardZPrev, false, blend);
    else
        gs.advectVelocityDouble(U, V, W, Up, Vp, Wp, m.BackwardX, m.BackwardY, m.BackwardZ,
                                m.BackwardXPrev, m.BackwardYPrev, m.BackwardZPrev, false, blend);
    if (blend != 1.f)
        gs.producedAll({ &U, &V, &W }, std::min({ gpuMapper::minValid({ &U, &V, &W }),
                                                  gpuMapper::minValid({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }) - kReachMap,
                                                  gpuMapper::minValid({ &m.BackwardXPrev, &m.BackwardYPrev, &m.BackwardZPrev }) - reachField(m.Dback),
                                                  whole ? (int)DeviceField::kAlwaysValid
                                                        : gpuMapper::minValid({ &Up, &Vp, &Wp }) - reachField(m.Dback + m.DbackPrev) }));
}

// Mapping.cpp:393-407, same four-stage expansion of gpu_compensate_field (GPU_kernel.cu:676-681)
void MapperBaseGPU::advectField(DeviceField &f, DeviceField &fInit, DeviceField &fPrev)
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    const float h = g.h;
    const int ni = g.ni, nj = g.nj, nk = g.nk;

    gs.require({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap);
    gs.require({ &fInit }, reachField(m.Dback));
    { QuarterScope q4(m.backQ4); gs.advectField(f, fInit, m.BackwardX, m.BackwardY, m.BackwardZ, false); }
    gs.produced(f, std::min(gpuMapper::minValid({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }) - kReachMap,
                            fInit.valid - reachField(m.Dback)));

    gs.require({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap);
    gs.require({ &f }, reachField(m.Dfwd));
    const int fInit_valid = fInit.valid;
    bool stage2_done;
    {
        FusedScope fused(gs.fuse_housekeeping, 1 | 2);
        stage2_done = fused.on;
        if (!fused.on) fl_memset(gs.u_src, 0, g.n() * sizeof(float));     // GPU_Advection.h:526 (u_src doubles as scalar scratch)
        QuarterScope q4(m.fwdQ4);
        gpu_compensate_error_field(f, fInit, gs.u_src, m.ForwardX, m.ForwardY, m.ForwardZ, h, ni, nj, nk, false);
    }
    // u_src is the u-sized scratch: as a scalar field its planes are ni*nj wide
    const size_t saved_plane = gs.u_src.plane;
    gs.u_src.plane = (size_t)ni * nj;
    gs.produced(gs.u_src, std::min({ gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }) - kReachMap,
                                     f.valid - reachField(m.Dfwd), fInit_valid }));
    if (stage2_done) fInit.valid = f.valid; else fInit.copy_from(f);
    gs.require({ &gs.u_src }, reachField(m.Dback));
    gs.require({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap);
    { QuarterScope q4(m.backQ4); gs.accumulateField(gs.u_src, f, m.BackwardX, m.BackwardY, m.BackwardZ, false, -0.5f); }
    const int stage3_valid = std::min({ f.valid, gpuMapper::minValid({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }) - kReachMap,
                                        gs.u_src.valid - reachField(m.Dback) });
    if (!keepDmcBorder)
        gs.wallFixup({ { &gs.u_src, &fInit, &f, FIELD_S } }, m.BackwardX, m.BackwardY, m.BackwardZ, m.Dback, reachField(m.Dback), -0.5f,
                     DeviceField::kAlwaysValid);           // (this form runs the operator on every local plane)
    gs.produced(f, stage3_valid);
    gs.u_src.plane = saved_plane;
    gs.require({ &fInit }, 1);
    gpu_clamp_extrema_box(fInit, f, ni, nj, nk);
    gs.produced(f, std::min(f.valid, fInit.valid - 1));

    const float blend = (TotalReinitCount != 0) ? BlendCoeff : 1.f;
    const float *fg = blend != 1.f ? gs.globalTwin(&fPrev) : nullptr;       // (advectVelocity: whole-grid copy of the *Prev field)
    if (blend != 1.f) {
        gs.require({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap);
        gs.require({ &m.BackwardXPrev, &m.BackwardYPrev, &m.BackwardZPrev }, reachField(m.Dback));
        if (!fg) gs.require({ &fPrev }, reachField(m.Dback + m.DbackPrev));
    }
    if (fg)
        gs.advectFieldDoubleGlobal(f, fg, m.BackwardX, m.BackwardY, m.BackwardZ,
                                   m.BackwardXPrev, m.BackwardYPrev, m.BackwardZPrev, false, blend);
    else
        gs.advectFieldDouble(f, fPrev, m.BackwardX, m.BackwardY, m.BackwardZ,
                             m.BackwardXPrev, m.BackwardYPrev, m.BackwardZPrev, false, blend);
    if (blend != 1.f)
        gs.produced(f, std::min({ f.valid, gpuMapper::minValid({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }) - kReachMap,
                                  gpuMapper::minValid({ &m.BackwardXPrev, &m.BackwardYPrev, &m.BackwardZPrev }) - reachField(m.Dback),
                                  fg ? (int)DeviceField::kAlwaysValid : fPrev.valid - reachField(m.Dback + m.DbackPrev) }));
}

// Two scalar fields through the same sequence as advectField, stage by stage, with the batched
// operators (one map look-up per stage for both fields).  u_src / v_src are the two error scratches.
void MapperBaseGPU::advectFields2(DeviceField &f1, DeviceField &f1Init, DeviceField &f1Prev,
                                  DeviceField &f2, DeviceField &f2Init, DeviceField &f2Prev)
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    const float h = g.h;
    const int ni = g.ni, nj = g.nj, nk = g.nk;
    const auto back = [&] { return gpuMapper::minValid({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }); };
    const auto fwd = [&] { return gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }); };

    {
        FusedScope fused(gs.fuse_housekeeping, 1);
        if (!fused.on) { f1.zero(); f2.zero(); }                            // GPU_Advection.h:507
        gs.withGhosts({ { { &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap }, { { &f1Init, &f2Init }, reachField(m.Dback) } }, [&] {
            QuarterScope q4(m.backQ4);
            gpu_advect_field2(f1, f1Init, f2, f2Init, m.BackwardX, m.BackwardY, m.BackwardZ, h, ni, nj, nk, false);
        }, std::min(gs.validAfter({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap) - kReachMap,
                    gs.validAfter({ &f1Init, &f2Init }, reachField(m.Dback)) - reachField(m.Dback)), { &f1, &f2 });
    }
    gs.produced(f1, std::min(back() - kReachMap, f1Init.valid - reachField(m.Dback)));
    gs.produced(f2, std::min(back() - kReachMap, f2Init.valid - reachField(m.Dback)));

    DeviceField &e1 = gs.u_src, &e2 = gs.v_src;                             // scalar-sized use of the scratches
    const int f1Init_valid = f1Init.valid, f2Init_valid = f2Init.valid;
    bool stage2_done;
    {
        FusedScope fused(gs.fuse_housekeeping, 1 | 2);
        stage2_done = fused.on;
        if (!fused.on) { fl_memset(e1, 0, g.n() * sizeof(float)); fl_memset(e2, 0, g.n() * sizeof(float)); }
        gs.withGhosts({ { { &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap }, { { &f1, &f2 }, reachField(m.Dfwd) } }, [&] {
            QuarterScope q4(m.fwdQ4);
            gpu_compensate_error_field2(f1, f1Init, e1, f2, f2Init, e2, m.ForwardX, m.ForwardY, m.ForwardZ, h, ni, nj, nk, false);
        }, std::max(0, gs.validAfter({ &f1, &f2 }, reachField(m.Dfwd))),       // (wherever f is correct: the fused copy into fInit)
           { &e1, &e2, &f1Init, &f2Init });
    }
    const size_t plane1 = e1.plane, plane2 = e2.plane;
    e1.plane = e2.plane = (size_t)ni * nj;
    gs.produced(e1, std::min({ fwd() - kReachMap, f1.valid - reachField(m.Dfwd), f1Init_valid }));
    gs.produced(e2, std::min({ fwd() - kReachMap, f2.valid - reachField(m.Dfwd), f2Init_valid }));
    if (stage2_done) { f1Init.valid = f1.valid; f2Init.valid = f2.valid; }
    else { f1Init.copy_from(f1); f2Init.copy_from(f2); }
    if (!keepDmcBorder)                                  // (as for the velocity: the wall-sheet messages start before the operator)
        gs.wallFixupBegin({ { &e1, &f1Init, &f1, FIELD_S }, { &e2, &f2Init, &f2, FIELD_S } }, m.Dback, reachField(m.Dback));
    gs.withGhosts({ { { &e1, &e2 }, reachField(m.Dback) }, { { &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap } }, [&] {
        QuarterScope q4(m.backQ4);
        gpu_accumulate_field2(e1, f1, -0.5f, e2, f2, -0.5f, m.BackwardX, m.BackwardY, m.BackwardZ, h, ni, nj, nk, false);
    }, std::min({ f1.valid, f2.valid, gs.validAfter({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap) - kReachMap,
                  gs.validAfter({ &e1, &e2 }, reachField(m.Dback)) - reachField(m.Dback) }), { &f1, &f2 });
    const int s3v1 = std::min({ f1.valid, back() - kReachMap, e1.valid - reachField(m.Dback) });
    const int s3v2 = std::min({ f2.valid, back() - kReachMap, e2.valid - reachField(m.Dback) });
    if (!keepDmcBorder) gs.wallFixupEnd(m.BackwardX, m.BackwardY, m.BackwardZ, -0.5f, std::min(s3v1, s3v2));
    gs.produced(f1, s3v1);
    gs.produced(f2, s3v2);
    e1.plane = plane1; e2.plane = plane2;
    gs.require({ &f1Init, &f2Init }, 1);
    gpu_clamp_extrema_box(f1Init, f1, ni, nj, nk);
    gpu_clamp_extrema_box(f2Init, f2, ni, nj, nk);
    gs.produced(f1, std::min(f1.valid, f1Init.valid - 1));
    gs.produced(f2, std::min(f2.valid, f2Init.valid - 1));

    const float blend = (TotalReinitCount != 0) ? BlendCoeff : 1.f;
    DeviceField *fs[2] = { &f1, &f2 }, *ps[2] = { &f1Prev, &f2Prev };
    for (int a = 0; a < 2; a++) {
        DeviceField &f = *fs[a], &fPrev = *ps[a];
        const float *fg = blend != 1.f ? gs.globalTwin(&fPrev) : nullptr;   // (advectVelocity: whole-grid copy of the *Prev field)
        if (blend != 1.f) {
            gs.require({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, kReachMap);
            gs.require({ &m.BackwardXPrev, &m.BackwardYPrev, &m.BackwardZPrev }, reachField(m.Dback));
            if (!fg) gs.require({ &fPrev }, reachField(m.Dback + m.DbackPrev));
        }
        if (fg)
            gs.advectFieldDoubleGlobal(f, fg, m.BackwardX, m.BackwardY, m.BackwardZ,
                                       m.BackwardXPrev, m.BackwardYPrev, m.BackwardZPrev, false, blend);
        else
            gs.advectFieldDouble(f, fPrev, m.BackwardX, m.BackwardY, m.BackwardZ,
                                 m.BackwardXPrev, m.BackwardYPrev, m.BackwardZPrev, false, blend);
        if (blend != 1.f)
            gs.produced(f, std::min({ f.valid, back() - kReachMap,
                                      gpuMapper::minValid({ &m.BackwardXPrev, &m.BackwardYPrev, &m.BackwardZPrev }) - reachField(m.Dback),
                                      fg ? (int)DeviceField::kAlwaysValid : fPrev.valid - reachField(m.Dback + m.DbackPrev) }));
    }
}

// Mapping.cpp:420-428 (note the Init/Change order swap towards the gpuMapper)
void MapperBaseGPU::accumulateVelocity(DeviceField &dUi, DeviceField &dVi, DeviceField &dWi,
                                       DeviceField &Uc, DeviceField &Vc, DeviceField &Wc, float coeff)
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    gs.withGhosts({ { { &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap }, { { &Uc, &Vc, &Wc }, reachField(m.Dfwd) } }, [&] {
        QuarterScope q4(m.fwdQ4);
        if (m.fwdIdentity)
            gpu_accumulate_velocity_identity(Uc, Vc, Wc, dUi, dVi, dWi, m.ForwardX, m.ForwardY, m.ForwardZ, g.h, g.ni, g.nj, g.nk, false, coeff);
        else
            gs.accumulateVelocity(Uc, Vc, Wc, dUi, dVi, dWi, m.ForwardX, m.ForwardY, m.ForwardZ, false, coeff);
    }, std::min({ gpuMapper::minValid({ &dUi, &dVi, &dWi }), gs.validAfter({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap) - kReachMap,
                  gs.validAfter({ &Uc, &Vc, &Wc }, reachField(m.Dfwd)) - reachField(m.Dfwd) }), { &dUi, &dVi, &dWi });
    gs.producedAll({ &dUi, &dVi, &dWi }, std::min({ gpuMapper::minValid({ &dUi, &dVi, &dWi }),
                                                    gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }) - kReachMap,
                                                    gpuMapper::minValid({ &Uc, &Vc, &Wc }) - reachField(m.Dfwd) }));
}

void MapperBaseGPU::accumulateVelocity2(DeviceField &dUi, DeviceField &dVi, DeviceField &dWi,
                                        DeviceField &Uc1, DeviceField &Vc1, DeviceField &Wc1, float coeff1,
                                        DeviceField &Uc2, DeviceField &Vc2, DeviceField &Wc2, float coeff2, bool first_uw_zero)
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    if (first_uw_zero) {
        gs.withGhosts({ { { &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap }, { { &Vc1, &Uc2, &Vc2, &Wc2 }, reachField(m.Dfwd) } }, [&] {
            QuarterScope q4(m.fwdQ4);
            gpu_accumulate_component(Uc2, coeff2, nullptr, 0.f, dUi, m.ForwardX, m.ForwardY, m.ForwardZ, g.h, g.ni, g.nj, g.nk, 0, false);
            gpu_accumulate_component(Vc1, coeff1, Vc2, coeff2, dVi, m.ForwardX, m.ForwardY, m.ForwardZ, g.h, g.ni, g.nj, g.nk, 1, false);
            gpu_accumulate_component(Wc2, coeff2, nullptr, 0.f, dWi, m.ForwardX, m.ForwardY, m.ForwardZ, g.h, g.ni, g.nj, g.nk, 2, false);
        }, std::min({ gpuMapper::minValid({ &dUi, &dVi, &dWi }), gs.validAfter({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap) - kReachMap,
                      gs.validAfter({ &Vc1, &Uc2, &Vc2, &Wc2 }, reachField(m.Dfwd)) - reachField(m.Dfwd) }), { &dUi, &dVi, &dWi });
        gs.producedAll({ &dUi, &dVi, &dWi }, std::min({ gpuMapper::minValid({ &dUi, &dVi, &dWi }),
                                                        gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }) - kReachMap,
                                                        gpuMapper::minValid({ &Vc1, &Uc2, &Vc2, &Wc2 }) - reachField(m.Dfwd) }));
        return;
    }
    gs.withGhosts({ { { &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap }, { { &Uc1, &Vc1, &Wc1, &Uc2, &Vc2, &Wc2 }, reachField(m.Dfwd) } }, [&] {
        QuarterScope q4(m.fwdQ4);
        gpu_accumulate_velocity2(Uc1, Vc1, Wc1, coeff1, Uc2, Vc2, Wc2, coeff2, dUi, dVi, dWi,
                                 m.ForwardX, m.ForwardY, m.ForwardZ, g.h, g.ni, g.nj, g.nk, false);
    }, std::min({ gpuMapper::minValid({ &dUi, &dVi, &dWi }), gs.validAfter({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap) - kReachMap,
                  gs.validAfter({ &Uc1, &Vc1, &Wc1, &Uc2, &Vc2, &Wc2 }, reachField(m.Dfwd)) - reachField(m.Dfwd) }), { &dUi, &dVi, &dWi });
    gs.producedAll({ &dUi, &dVi, &dWi }, std::min({ gpuMapper::minValid({ &dUi, &dVi, &dWi }),
                                                    gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }) - kReachMap,
                                                    gpuMapper::minValid({ &Uc1, &Vc1, &Wc1, &Uc2, &Vc2, &Wc2 }) - reachField(m.Dfwd) }));
}

void MapperBaseGPU::accumulateField(DeviceField &dfInit, DeviceField &fChange)
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    gs.require({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }, kReachMap);
    gs.require({ &fChange }, reachField(m.Dfwd));
    { QuarterScope q4(m.fwdQ4); gs.accumulateField(fChange, dfInit, m.ForwardX, m.ForwardY, m.ForwardZ, false, 1.0f); }
    gs.produced(dfInit, std::min({ dfInit.valid, gpuMapper::minValid({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }) - kReachMap,
                                   fChange.valid - reachField(m.Dfwd) }));
}

// Mapping.cpp:430-447.  BackwardPrev <- Backward is a buffer swap (the old BackwardPrev content is
// dead and Backward is refilled right after); the two identity refills stay copies.
void MapperBaseGPU::reinitializeMapping()
{
    MapSet &m = *maps;
    TotalReinitCount++;
    m.BackwardXPrev.swap(m.BackwardX); m.BackwardYPrev.swap(m.BackwardY); m.BackwardZPrev.swap(m.BackwardZ);
    // the two identity refills (:438-443 copy Init*): written by the same kernel that filled Init*, which costs
    // a write instead of a read and a write per array
    gpu_init_maps(m.BackwardX, m.BackwardY, m.BackwardZ, g.h, g.ni, g.nj, g.nk);
    gpu_init_maps(m.ForwardX, m.ForwardY, m.ForwardZ, g.h, g.ni, g.nj, g.nk);
    gpuSolver->producedAll({ &m.BackwardX, &m.BackwardY, &m.BackwardZ, &m.ForwardX, &m.ForwardY, &m.ForwardZ }, DeviceField::kAlwaysValid);
    m.DbackPrev = m.Dback;
    m.Dback = 0;
    m.Dfwd = 0;
    m.fwdIdentity = true;
    m.backQ4 = m.fwdQ4 = true;          // identity maps: 0 or n*h with 1 <= n <= 1024 (planes outside the global grid: 0)
}

// The reference scans gpuSolver->du on the host (Mapping.cpp:497-516); that scratch still holds older data
// outside the window estimate_kernel writes, so it is cleared here: the distortion is the maximum over the
// cells the kernel evaluates.
float MapperBaseGPU::estimateDistortion()
{
    MapSet &m = *maps;
    gpuMapper &gs = *gpuSolver;
    // z-slab ranks: the forward map is sampled at backward-mapped positions and the other way round
    gs.require({ &m.ForwardX, &m.ForwardY, &m.ForwardZ }, reachField(m.Dback));
    gs.require({ &m.BackwardX, &m.BackwardY, &m.BackwardZ }, reachField(m.Dfwd));
    fl_memset(gs.u_src, 0, g.n() * sizeof(float));
    gpu_estimate_distortion(gs.u_src, m.BackwardX, m.BackwardY, m.BackwardZ, m.ForwardX, m.ForwardY, m.ForwardZ,
                            g.h, g.ni, g.nj, g.nk);
    // (owned planes, all-reduced over the slab ranks; single GPU: the whole field)
    return std::sqrt(gpu_max_field_owned(gs.u_src, g.ni, g.nj, g.nk));
}

bool MapperBaseGPU::unshareMaps()
{
    if (!shared_) return true;
    shared_ = false;
    return init(g.ni, g.nj, g.nk, g.h, BlendCoeff, gpuSolver);
}

} // namespace bqhost
