// mapping.hpp -- device-resident BiMocq map state and its update/advect/accumulate/reinit
// sequences.  Mirrors the reference's MapperBaseGPU (src/bimocq3D/Mapping.h:54-105,
// src/bimocq3D/Mapping.cpp:276-447).
#pragma once
#include <memory>
#include "gpu_mapper.hpp"

namespace bqhost {

// The twelve map fields of one MapperBaseGPU (Mapping.h:92-96).  Held behind a shared_ptr so the
// velocity and the scalar advector can share ONE set while their update/reinit schedules coincide
// (the reference re-initialises both every frame, SURVEY Q5: the two sets are then bit-identical
// and computing them twice is pure waste).
struct MapSet {
    DeviceField ForwardX, ForwardY, ForwardZ;
    DeviceField BackwardX, BackwardY, BackwardZ;
    DeviceField BackwardXPrev, BackwardYPrev, BackwardZPrev;
    DeviceField InitX, InitY, InitZ;
    bool alloc(const GridDims &g);
};

class MapperBaseGPU {
public:
    bool init(int ni, int nj, int nk, float h, float coeff, gpuMapper *mymapper);
    // use `owner`'s map set instead of an own one; update/reinit must then be driven through the
    // owner only, with noteSharedReinit() keeping this mapper's reinit counter in step
    void shareMapsOf(MapperBaseGPU &owner) { maps = owner.maps; shared_ = true; }
    bool sharesMaps() const { return shared_; }
    void noteSharedReinit() { TotalReinitCount++; }

    void updateForward(float *U, float *V, float *W, float cfldt, float dt);
    void updateBackward(float *U, float *V, float *W, float cfldt, float dt);
    void updateMapping(float *U, float *V, float *W, float cfldt, float dt);

    void advectVelocity(float *U, float *V, float *W, float *Ui, float *Vi, float *Wi,
                        float *Up, float *Vp, float *Wp);
    void advectField(float *f, float *fInit, float *fPrev);
    void accumulateVelocity(float *dUi, float *dVi, float *dWi, float *Uc, float *Vc, float *Wc, float coeff);
    void accumulateField(float *dfInit, float *fChange);
    void reinitializeMapping();

    GridDims g;
    float BlendCoeff = 1.f;
    unsigned TotalReinitCount = 0;
    std::shared_ptr<MapSet> maps;
    gpuMapper *gpuSolver = nullptr;

private:
    bool shared_ = false;
};

} // namespace bqhost
