// mapping.cpp -- see mapping.hpp.  Reference: src/bimocq3D/Mapping.cpp:276-447.
#include "mapping.hpp"

namespace bqhost {

bool MapSet::alloc(const GridDims &g)
{
    DeviceField *all[] = { &ForwardX, &ForwardY, &ForwardZ, &BackwardX, &BackwardY, &BackwardZ,
                           &BackwardXPrev, &BackwardYPrev, &BackwardZPrev, &InitX, &InitY, &InitZ };
    for (DeviceField *f : all)
        if (!f->alloc(g.n())) return false;
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
    if (!maps->alloc(g)) return false;
    MapSet &m = *maps;
    gpu_init_maps(m.InitX, m.InitY, m.InitZ, h, ni, nj, nk);    // host loop + H2D in the reference (:306-328)
    m.ForwardX.copy_from(m.InitX); m.ForwardY.copy_from(m.InitY); m.ForwardZ.copy_from(m.InitZ);
    m.BackwardX.copy_from(m.InitX); m.BackwardY.copy_from(m.InitY); m.BackwardZ.copy_from(m.InitZ);
    m.BackwardXPrev.copy_from(m.InitX); m.BackwardYPrev.copy_from(m.InitY); m.BackwardZPrev.copy_from(m.InitZ);
    return true;
}

// Mapping.cpp:347-352
void MapperBaseGPU::updateMapping(float *U, float *V, float *W, float cfldt, float dt)
{
    updateBackward(U, V, W, cfldt, dt);
    updateForward(U, V, W, cfldt, dt);
}

// Mapping.cpp:354-368.  The reference copies x_out -> Backward after every sub-step
// (GPU_Advection.h:466-468); here sub-steps ping-pong between the mapper's two scratch sets and
// only the final result is copied into Backward.  Both scratch sets keep zero border nodes,
// exactly like the reference's x_out, so the copied-in border is the same.
void MapperBaseGPU::updateBackward(float *U, float *V, float *W, float cfldt, float dt)
{
    MapSet &m = *maps;
    float T = 0.f;
    float substep = cfldt;
    float *in[3] = { m.BackwardX, m.BackwardY, m.BackwardZ };
    float *ping[2][3] = { { gpuSolver->x_out, gpuSolver->y_out, gpuSolver->z_out },
                          { gpuSolver->x_out2, gpuSolver->y_out2, gpuSolver->z_out2 } };
    int which = 0;
    bool any = false;
    while (T < dt) {
        if (T + substep > dt) substep = dt - T;
        float **out = ping[which];
        gpuSolver->solveBackwardDMC(U, V, W, in[0], in[1], in[2], out[0], out[1], out[2], substep);
        in[0] = out[0]; in[1] = out[1]; in[2] = out[2];
        which ^= 1;
        any = true;
        T += substep;
    }
    if (any) {
        fl_memcpy_d2d(m.BackwardX, in[0], m.BackwardX.bytes());
        fl_memcpy_d2d(m.BackwardY, in[1], m.BackwardY.bytes());
        fl_memcpy_d2d(m.BackwardZ, in[2], m.BackwardZ.bytes());
    }
}

// Mapping.cpp:370-373
void MapperBaseGPU::updateForward(float *U, float *V, float *W, float cfldt, float dt)
{
    MapSet &m = *maps;
    gpuSolver->solveForward(U, V, W, m.ForwardX, m.ForwardY, m.ForwardZ, cfldt, dt);
}

// Mapping.cpp:375-391
void MapperBaseGPU::advectVelocity(float *U, float *V, float *W, float *Ui, float *Vi, float *Wi,
                                   float *Up, float *Vp, float *Wp)
{
    MapSet &m = *maps;
    gpuSolver->advectVelocity(U, V, W, Ui, Vi, Wi, m.BackwardX, m.BackwardY, m.BackwardZ, false);
    gpuSolver->compensateVelocity(U, V, W, Ui, Vi, Wi, m.ForwardX, m.ForwardY, m.ForwardZ,
                                  m.BackwardX, m.BackwardY, m.BackwardZ, false);
    const float blend = (TotalReinitCount != 0) ? BlendCoeff : 1.f;
    gpuSolver->advectVelocityDouble(U, V, W, Up, Vp, Wp, m.BackwardX, m.BackwardY, m.BackwardZ,
                                    m.BackwardXPrev, m.BackwardYPrev, m.BackwardZPrev, false, blend);
}

// Mapping.cpp:393-407
void MapperBaseGPU::advectField(float *f, float *fInit, float *fPrev)
{
    MapSet &m = *maps;
    gpuSolver->advectField(f, fInit, m.BackwardX, m.BackwardY, m.BackwardZ, false);
    gpuSolver->compensateField(f, fInit, m.ForwardX, m.ForwardY, m.ForwardZ,
                               m.BackwardX, m.BackwardY, m.BackwardZ, false);
    const float blend = (TotalReinitCount != 0) ? BlendCoeff : 1.f;
    gpuSolver->advectFieldDouble(f, fPrev, m.BackwardX, m.BackwardY, m.BackwardZ,
                                 m.BackwardXPrev, m.BackwardYPrev, m.BackwardZPrev, false, blend);
}

// Mapping.cpp:420-428 (note the Init/Change order swap towards the gpuMapper)
void MapperBaseGPU::accumulateVelocity(float *dUi, float *dVi, float *dWi, float *Uc, float *Vc, float *Wc, float coeff)
{
    MapSet &m = *maps;
    gpuSolver->accumulateVelocity(Uc, Vc, Wc, dUi, dVi, dWi, m.ForwardX, m.ForwardY, m.ForwardZ, false, coeff);
}

void MapperBaseGPU::accumulateField(float *dfInit, float *fChange)
{
    MapSet &m = *maps;
    gpuSolver->accumulateField(fChange, dfInit, m.ForwardX, m.ForwardY, m.ForwardZ, false, 1.0f);
}

// Mapping.cpp:430-447.  BackwardPrev <- Backward is a buffer swap (the old BackwardPrev content is
// dead and Backward is refilled right after); the two identity refills stay copies.
void MapperBaseGPU::reinitializeMapping()
{
    MapSet &m = *maps;
    TotalReinitCount++;
    m.BackwardXPrev.swap(m.BackwardX); m.BackwardYPrev.swap(m.BackwardY); m.BackwardZPrev.swap(m.BackwardZ);
    m.BackwardX.copy_from(m.InitX); m.BackwardY.copy_from(m.InitY); m.BackwardZ.copy_from(m.InitZ);
    m.ForwardX.copy_from(m.InitX); m.ForwardY.copy_from(m.InitY); m.ForwardZ.copy_from(m.InitZ);
}

} // namespace bqhost
