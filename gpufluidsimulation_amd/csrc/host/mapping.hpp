// mapping.hpp -- device-resident BiMocq map state and its update/advect/accumulate/reinit
// sequences.  Mirrors the reference's MapperBaseGPU (src/bimocq3D/Mapping.h:54-105,
// src/bimocq3D/Mapping.cpp:276-447).
//
// z-slab ranks (multi-GPU): every method states how far its kernels reach along z
// (`require(fields, depth)`) before launching and records how many ghost planes of its outputs are
// still correct afterwards (`produced`), so ghost planes are exchanged exactly when an operator needs
// more of them than are valid.  On a single GPU both calls are no-ops.
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
    // upper bounds (in cells) of |map(x) - x| along any axis: how far a mapped position can sit from
    // the node it belongs to.  Grow by the step's CFL travel, reset by reinitializeMapping().
    int Dfwd = 0, Dback = 0, DbackPrev = 0;
    // Forward* still hold exactly what gpu_init_maps wrote (fresh or just re-initialised, not yet updated)
    bool fwdIdentity = true;
    // every value of the Backward* / Forward* arrays is 0 or a coordinate in [h/256, 1024 h] (gpu_maps_quarter_safe, checked
    // after each update; true for the identity maps): the operators may run the weight-1/4 lerps of the structured map
    // look-up in fp32 (FL_OPT_MAP_QUARTER_FP32) -- same values, fewer issue cycles
    bool backQ4 = true, fwdQ4 = true;
    bool alloc(const gpuMapper &m);
};

class MapperBaseGPU {
public:
    bool init(int ni, int nj, int nk, float h, float coeff, gpuMapper *mymapper);
    // use `owner`'s map set instead of an own one; update/reinit must then be driven through the
    // owner only, with noteSharedReinit() keeping this mapper's reinit counter in step
    void shareMapsOf(MapperBaseGPU &owner) { maps = owner.maps; shared_ = true; }
    bool sharesMaps() const { return shared_; }
    void noteSharedReinit() { TotalReinitCount++; }

    // dcells: upper bound of the distance (in cells) anything travels during this update
    void updateForward(DeviceField &U, DeviceField &V, DeviceField &W, float cfldt, float dt, int dcells);
    void updateBackward(DeviceField &U, DeviceField &V, DeviceField &W, float cfldt, float dt, int dcells);
    void updateMapping(DeviceField &U, DeviceField &V, DeviceField &W, float cfldt, float dt, int dcells);

    void advectVelocity(DeviceField &U, DeviceField &V, DeviceField &W,
                        DeviceField &Ui, DeviceField &Vi, DeviceField &Wi,
                        DeviceField &Up, DeviceField &Vp, DeviceField &Wp);
    void advectField(DeviceField &f, DeviceField &fInit, DeviceField &fPrev);
    // advectField(f1..) ; advectField(f2..) with the map look-ups shared between the two fields
    void advectFields2(DeviceField &f1, DeviceField &f1Init, DeviceField &f1Prev,
                       DeviceField &f2, DeviceField &f2Init, DeviceField &f2Prev);
    void accumulateVelocity(DeviceField &dUi, DeviceField &dVi, DeviceField &dWi,
                            DeviceField &Uc, DeviceField &Vc, DeviceField &Wc, float coeff);
    // accumulateVelocity(change 1, coeff1) ; accumulateVelocity(change 2, coeff2), one map look-up
    // first_uw_zero: the caller knows Uc1 and Wc1 to be identically zero (adding blend9 of a zero field changes
    // nothing): their look-ups are skipped and u, w take the second source only
    void accumulateVelocity2(DeviceField &dUi, DeviceField &dVi, DeviceField &dWi,
                             DeviceField &Uc1, DeviceField &Vc1, DeviceField &Wc1, float coeff1,
                             DeviceField &Uc2, DeviceField &Vc2, DeviceField &Wc2, float coeff2, bool first_uw_zero = false);
    void accumulateField(DeviceField &dfInit, DeviceField &fChange);
    void reinitializeMapping();
    // Mapping.cpp:495-519: sqrt of the largest round-trip error |x - fwd(back(x))|, |x - back(fwd(x))|
    float estimateDistortion();
    // stop sharing the owner's maps: this mapper gets its own identity set (only before the first update)
    bool unshareMaps();

    GridDims g;
    float BlendCoeff = 1.f;
    unsigned TotalReinitCount = 0;
    // The reference copies the DMC scratch buffer, whose border nodes (outside 2..n-3) are zero, back
    // into the backward map (GPU_Advection.h:464-468; the pre-copy that would have kept the border is
    // commented out at :335-337).  Wall-adjacent nodes of the compensation gather then sample positions
    // pulled 25 % towards the origin -- arbitrarily far along z.  Default false = replicate (single-GPU
    // parity with the reference); true = keep the incoming border (the evident intent), which is also the
    // only setting under which z-slab ranks reproduce a single GPU bit for bit.
    bool keepDmcBorder = false;
    // Measure how far along z the maps really carry their nodes after every update (gpu_map_travel_z) and use THAT as
    // Dback / Dfwd instead of the running sum of CFL travels: maps that live for many steps (BQ_OPT_REINIT_POLICY = 1)
    // then need ghost planes for what they do, not for the worst they could have done.  One more 8-byte read-back per
    // update.  Set by the solver together with BQ_OPT_REINIT_MAX_TRAVEL.
    bool measureTravel = false;
    float lastTravel[2] = { 0.f, 0.f };      // cells along z: backward, forward map (what the last measurement saw)
    std::shared_ptr<MapSet> maps;
    gpuMapper *gpuSolver = nullptr;

private:
    bool shared_ = false;
};

} // namespace bqhost
