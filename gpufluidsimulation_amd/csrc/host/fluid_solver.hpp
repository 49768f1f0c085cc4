// fluid_solver.hpp -- GPU-resident BiMocq smoke solver: the time-step state machine.
//
// Keeps the reference's BimocqGPUSolver surface (src/bimocq3D/BimocqGPUSolver.h:27-56):
//   BimocqGPUSolver(nx, ny, nz, L, vis_coeff, blend_coeff, scheme, gpuMapper*)
//   setSmoke(drop, raise, emitters) / advance(framenum, dt) / outputResult(frame, path)
// so the reference's driver loop (src/bimocq3D/main.cpp:151-159) runs against it unchanged.
// `FluidSolver` (the name BASELINE.json uses) is an alias with step()/dump() spellings.
#pragma once
#include <string>
#include <thread>
#include <vector>
#include "bimocq_solver.h"
#include "mapping.hpp"

namespace bqhost {

enum Scheme { BIMOCQ = 0, SEMILAG, MACCORMACK, MAC_REFLECTION };     // BimocqSolver.h:29

// The reference's Emitter carries an OpenVDB SDF that the GPU solver never samples
// (BimocqGPUSolver.cpp:376-392 uses hard-coded spheres); what it does use is kept.
struct Emitter {
    int emitFrame = 0;
    float emit_density = 0.f, emit_temperature = 0.f;
    float e_pos[3] = { 0.f, 0.f, 0.f };
    float radius = 0.f;
    float emiter = 0.f;                 // sign/scale of the x-velocity the source imposes
};

class BimocqGPUSolver {
public:
    BimocqGPUSolver(unsigned nx, unsigned ny, unsigned nz, float L, float vis_coeff, float blend_coeff,
                    Scheme inScheme, gpuMapper *mymapper);
    bool ok() const { return ok_; }

    void advance(int framenum, float dt);
    void advanceBimocq(int framenum, float dt);
    void advanceReflection(int framenum, float dt);
    float getCFL();
    void emitSmoke(int framenum, float dt);
    void addBuoyancy(float dt);
    void diffuseField(float *field, float *t0, float *t1, int ni, int nj, int nk, int iter, float nu, float dt);
    void diffuseFieldSlab(DeviceField &field, DeviceField &t0, DeviceField &t1, int bi, int bj, int bk, int iter, float nu, float dt);
    bool projection(bool with_delta = false);
    void velocityReinitialize();
    void scalarReinitialize();
    bool whole_grid_prev = true;            // BQ_OPT_WHOLE_GRID_PREV
    bool wholeGridPrev() const;             // blend != 1 on z-slab ranks with the zeroed map border: *Prev fields need whole-grid copies
    void setSmoke(float drop, float raise, const std::vector<Emitter> &emitters);
    long outputResult(unsigned frame, const std::string &filepath);
    // The same dump without stalling the simulation: the density is downloaded into pinned
    // memory on the copy stream, ordered after the step that produced it, and a worker thread writes the file
    // while the next advance() runs.  At most one dump is in flight: a second call first waits for the
    // previous one.  waitOutput() returns that dump's voxel count (or -1) and leaves nothing in flight.
    bool outputResultAsync(unsigned frame, const std::string &filepath);
    long waitOutput();
    ~BimocqGPUSolver();

    // FluidSolver spellings
    void step(int framenum, float dt) { advance(framenum, dt); }
    long dump(unsigned frame, const std::string &filepath) { return outputResult(frame, filepath); }

    // projection variant (compile-time `#if` in the reference, BimocqGPUSolver.cpp:408-466)
    bool  keep_full_state = false;      // BQ_OPT_FULL_STATE: also compute state nothing reads (see advanceBimocq)
    int   reinit_policy = 0;            // BQ_OPT_REINIT_POLICY
    bool  setReinitPolicy(int policy);
    // BQ_OPT_REINIT_MAX_TRAVEL (policy 1): a map set is also re-initialised when its measured z-travel (cells) + this step's
    // CFL travel + the sampling footprint would no longer fit `travel_limit` planes next step.  0 = no such rule (single GPU
    // default); z-slab ranks: the ghost depth G (default, and the largest value that makes sense there).
    int   travel_limit = 0;
    void  setTravelLimit(int cells);
    int   forced_reinits = 0;           // re-initialisations the travel rule (not the CPU solver's thresholds) caused
    int   vel_reinits = 0, scalar_reinits = 0;
    float last_vel_distortion = 0.f, last_scalar_distortion = 0.f;
    int   steps_taken = 0;
    DeviceField DensityTemp, TemperatureTemp, DensityExtern, TemperatureExtern;   // policy 1 only (:53-58)
    int   projection_kind = 0;          // BQ_PROJECTION_JACOBI / BQ_PROJECTION_MGCG
    int   mg_iters = 50;                // :444
    int   jacobi_iters = 100;           // :409
    float halfrdx = 0.5f;               // :410 (quarter-strength projection; 1.0 is the physical value)
    bool  verbose = false;              // print "[Bimocq GPU Time: ...]" like the reference (:126)
    // BQ_OPT_PROFILE_PHASES: event pairs on the compute stream around the phases of advanceBimocq -- where a step's time
    // goes on THIS rank, exposed communication included (the waits sit inside the phase that needs the data)
    enum Phase { PH_MAPS = 0, PH_ADVECT, PH_FORCES, PH_PROJECTION, PH_ACCUMULATE, PH_COUNT };
    bool  profile_phases = false;
    void  phaseMark(int phase);                     // closes the running phase and opens `phase` (PH_COUNT: just close)
    void  phaseTotals(double ms[PH_COUNT], long long *steps, bool reset);   // blocking

    float _alpha = 0.f, _beta = 0.f;    // smoke parameters (:529-534)
    Scheme myscheme;

    GridDims g;
    float CellSize, MaxVelocity = 0.f, Viscosity;
    float last_cfldt = 0.f, last_ms = 0.f;

    DeviceField VelocityU, VelocityV, VelocityW;
    DeviceField VelocityUInit, VelocityVInit, VelocityWInit;
    DeviceField VelocityUPrev, VelocityVPrev, VelocityWPrev;
    DeviceField VelocityUPrevAll, VelocityVPrevAll, VelocityWPrevAll, DensityPrevAll, TemperaturePrevAll;   // wholeGridPrev()
    DeviceField VelocityUTemp, VelocityVTemp, VelocityWTemp;
    DeviceField duProj, dvProj, dwProj, duExtern, dvExtern, dwExtern;
    DeviceField TempSrcU, TempSrcV, TempSrcW;
    DeviceField Density, DensityInit, DensityPrev;
    DeviceField Temperature, TemperatureInit, TemperaturePrev;
    DeviceField div, p, p_temp;         // the reference lends DensityTemp/TemperatureTemp/TempSrcV here (:410)
    DeviceField debugParam;             // 4096 floats, residual history (:412-417)
    // fp64 work arrays + level pyramid of the multigrid-CG projection (:60-90), allocated on first use
    struct Mgcg {
        DeviceBytes div, p, dir, residual, temp0, temp1, result;
        std::vector<DeviceBytes> b, x, r;
        std::vector<SCoarseLevelInfo> levels;
        bool ready = false;
        // z-slab ranks: the velocity of the WHOLE grid, assembled on every rank for the replicated solve (projectionMgcgSlabs)
        DeviceField gu, gv, gw;
    } mg;
    bool allocMgcg();
    bool projectionMgcgSlabs();
    bool projectionMgcgShared();                    // the levels shared between the slab ranks; false: not applicable here
    bool mgcg_shared = true;                        // BQ_OPT_MGCG_SHARED
    bool mgcg_shared_ran = false;                   // the last MGCG projection on slabs took the shared path
    std::vector<double> mgHistory() const;          // tempResult (4096 doubles), downloaded

    std::vector<float> host_density, host_u, host_v, host_w;    // outputResult staging (:538-541)

    gpuMapper *GpuSolver;
    MapperBaseGPU VelocityAdvector, ScalarAdvector;
    int vel_lastReinit = -11, scalar_lastReinit = -31;          // BimocqGPUSolver.h:109-110
    std::vector<Emitter> sim_emitter;

private:
    struct PhaseSpan { void *a, *b; int phase; };
    std::vector<PhaseSpan> phase_spans_;
    void *phase_open_ev_ = nullptr;
    int phase_open_ = -1;
    long long phase_steps_ = 0;
    bool ok_ = false;
    float *dump_host_ = nullptr;        // pinned staging buffer of the asynchronous dump
    DeviceField dump_dev_;              // device snapshot of the dumped frame (the next advance() rewrites Density)
    std::thread dump_thread_;
    long dump_result_ = 0;
};

using FluidSolver = BimocqGPUSolver;

// writeVDB's contract (utils/volumeMeshTools.h:33-60) in a dependency-free container; returns the
// number of voxels written or -1.  Defined in density_dump.cpp.
long write_density_dump(unsigned frame, const std::string &filepath, float voxel_size,
                        const float *density, int nx, int ny, int nz, int k_offset, int nz_global);
#ifdef HAVE_OPENVDB
long write_density_vdb(unsigned frame, const std::string &filepath, float voxel_size,
                       const float *density, int nx, int ny, int nz, int k_offset, int nz_global);
#endif

} // namespace bqhost
