// solver_capi.cpp -- C view of the host solver (include/bimocq_solver.h).
#include "fluid_solver.hpp"

#include <algorithm>
#include <memory>

using namespace bqhost;

struct bq_solver {
    std::unique_ptr<gpuMapper> mapper;
    std::unique_ptr<BimocqGPUSolver> solver;
    fl_context *ctx = nullptr;          // the context that was current when the solver was created (NULL: the default one)
};
// every entry point below acts on the solver's own context: several solvers -- on one device or on several -- can be driven
// from one thread in any order
#define BQ_ENTER(s) do { if (s) fl_context_make_current((s)->ctx); } while (0)

extern "C" {

bq_solver *bq_solver_create(int device, int nx, int ny, int nz, float L, float viscosity, float blend, int scheme)
{
    return bq_solver_create_slab(device, nx, ny, nz, L, viscosity, blend, scheme, 0, 1, 0);
}

bq_solver *bq_solver_create_slab(int device, int nx, int ny, int nz, float L, float viscosity, float blend, int scheme,
                                 int rank, int nranks, int ghost)
{
    if (nx < 8 || ny < 8 || nz < 8 || (scheme != BQ_SCHEME_BIMOCQ && scheme != BQ_SCHEME_MAC_REFLECTION)) return nullptr;
    SlabCtx sl;
    if (nranks > 1 || ghost > 0) {
        // even z-slabs: rank r owns planes [r*nz/nranks, (r+1)*nz/nranks); each must be deeper than the ghost zone
        if (nranks < 1 || rank < 0 || rank >= nranks || ghost < 2 || nz % nranks != 0 || nz / nranks < ghost + 1) {
            fl_report_error(FL_ERR_BAD_ARGUMENT, "bq_solver_create_slab: need nz % nranks == 0, ghost >= 2 and nz/nranks > ghost");
            return nullptr;
        }
        sl.on = true; sl.rank = rank; sl.nranks = nranks; sl.nkg = nz; sl.G = ghost;
        sl.own0 = rank * (nz / nranks); sl.own1 = sl.own0 + nz / nranks;
    }
    auto s = std::make_unique<bq_solver>();
    s->ctx = fl_context_current();
    s->mapper = std::make_unique<gpuMapper>(device, nx, ny, nz, L / nx, sl);   // main.cpp:151 / :37 (h = L/ni)
    if (!s->mapper->ok()) return nullptr;
    s->solver = std::make_unique<BimocqGPUSolver>(nx, ny, nz, L, viscosity, blend, scheme == BQ_SCHEME_MAC_REFLECTION ? MAC_REFLECTION : BIMOCQ, s->mapper.get());
    if (!s->solver->ok()) return nullptr;
    return s.release();
}

void bq_solver_destroy(bq_solver *s)
{
    BQ_ENTER(s);
    if (!s) return;
    fl_sync();
    s->solver.reset();
    s->mapper.reset();
    delete s;
}

void bq_solver_set_smoke(bq_solver *s, float drop, float rise, const bq_emitter *emitters, int n)
{
    if (!s) return;
    std::vector<Emitter> list;
    for (int a = 0; a < n; a++) {
        Emitter e;
        e.emitFrame = emitters[a].emit_frames;
        e.emit_density = emitters[a].density;
        e.emit_temperature = emitters[a].temperature;
        e.e_pos[0] = emitters[a].cx; e.e_pos[1] = emitters[a].cy; e.e_pos[2] = emitters[a].cz;
        e.radius = emitters[a].radius;
        e.emiter = emitters[a].emiter;
        list.push_back(e);
    }
    s->solver->setSmoke(drop, rise, list);
}

void bq_solver_set_projection(bq_solver *s, int kind, int iters, float halfrdx)
{
    if (!s) return;
    if (kind == BQ_PROJECTION_MGCG) {
        s->solver->projection_kind = kind;
        s->solver->mg_iters = iters;
    } else if (kind == BQ_PROJECTION_JACOBI) {
        s->solver->projection_kind = kind;
        s->solver->jacobi_iters = iters;
    } else {
        fl_report_error(FL_ERR_BAD_ARGUMENT, "bq_solver_set_projection: unknown projection kind");
        return;
    }
    s->solver->halfrdx = halfrdx;
}

void bq_solver_set_option(bq_solver *s, int option, int value)
{
    BQ_ENTER(s);
    if (!s) return;
    if (option == BQ_OPT_KEEP_DMC_BORDER) {
        s->solver->VelocityAdvector.keepDmcBorder = value != 0;
        s->solver->ScalarAdvector.keepDmcBorder = value != 0;
    } else if (option == BQ_OPT_FULL_STATE) {
        s->solver->keep_full_state = value != 0;
    } else if (option == BQ_OPT_FUSED_HOUSEKEEPING) {
        bqhost::gpuMapper &gm = *s->solver->GpuSolver;
        gm.fuse_housekeeping = value != 0;
        if (!value) {   // the unfused DMC update relies on scratch sets whose border nodes are zero; swaps may have left map data there
            gm.x_out.zero(); gm.y_out.zero(); gm.z_out.zero(); gm.x_out2.zero(); gm.y_out2.zero(); gm.z_out2.zero();
        }
    } else if (option == BQ_OPT_OVERLAP_EXCHANGES) {
        s->solver->GpuSolver->overlap_exchanges = value != 0;
    } else if (option == BQ_OPT_JACOBI_ENDS_FIRST) {
        s->solver->GpuSolver->jacobi_ends_first = value != 0;
    } else if (option == BQ_OPT_JACOBI_TRIPLES) {
        s->solver->GpuSolver->jacobi_triples = value != 0;
    } else if (option == BQ_OPT_MGCG_SHARED) {
        s->solver->mgcg_shared = value != 0;
    } else if (option == BQ_OPT_CONCURRENT_MAPS) {
        s->solver->GpuSolver->concurrent_maps = value != 0;
    } else if (option == BQ_OPT_WHOLE_GRID_PREV) {
        s->solver->whole_grid_prev = value != 0;
        if (!value) s->solver->GpuSolver->dropGlobalTwins();
    } else if (option == BQ_OPT_SHALLOW_BLOCKING_EXCHANGE) {
        s->solver->GpuSolver->shallow_blocking = value < 0 ? 0 : value;
    } else if (option == BQ_OPT_REINIT_MAX_TRAVEL) {
        s->solver->setTravelLimit(value);
    } else if (option == BQ_OPT_PROFILE_PHASES) {
        s->solver->profile_phases = value != 0;
    } else if (option == BQ_OPT_REINIT_POLICY) {
        s->solver->setReinitPolicy(value);
        s->solver->ScalarAdvector.keepDmcBorder = s->solver->VelocityAdvector.keepDmcBorder;
    }
}

int bq_solver_get_option(const bq_solver *s, int option)
{
    if (!s) return -1;
    switch (option) {
    case BQ_OPT_KEEP_DMC_BORDER:     return s->solver->VelocityAdvector.keepDmcBorder ? 1 : 0;
    case BQ_OPT_REINIT_POLICY:       return s->solver->reinit_policy;
    case BQ_OPT_FULL_STATE:          return s->solver->keep_full_state ? 1 : 0;
    case BQ_OPT_FUSED_HOUSEKEEPING:  return s->solver->GpuSolver->fuse_housekeeping ? 1 : 0;
    case BQ_OPT_OVERLAP_EXCHANGES:   return s->solver->GpuSolver->overlap_exchanges ? 1 : 0;
    case BQ_OPT_JACOBI_ENDS_FIRST:   return s->solver->GpuSolver->jacobi_ends_first ? 1 : 0;
    case BQ_OPT_JACOBI_TRIPLES:      return s->solver->GpuSolver->jacobi_triples ? 1 : 0;
    case BQ_OPT_CONCURRENT_MAPS:     return s->solver->GpuSolver->concurrent_maps ? 1 : 0;
    case BQ_OPT_WHOLE_GRID_PREV:     return s->solver->whole_grid_prev ? (s->solver->GpuSolver->globalTwin(&s->solver->VelocityUPrev) ? 2 : 1) : 0;
    case BQ_OPT_MGCG_SHARED:         return s->solver->mgcg_shared ? (s->solver->mgcg_shared_ran ? 2 : 1) : 0;    // 2: the last projection took it
    case BQ_OPT_SHALLOW_BLOCKING_EXCHANGE: return s->solver->GpuSolver->shallow_blocking;
    case BQ_OPT_PROFILE_PHASES:      return s->solver->profile_phases ? 1 : 0;
    case BQ_OPT_REINIT_MAX_TRAVEL:   return s->solver->travel_limit;
    default:                         return -1;
    }
}

int bq_solver_reinit_counts(const bq_solver *s, int which)
{
    if (!s) return 0;
    if (which == 2) return s->solver->forced_reinits;
    return which ? s->solver->scalar_reinits : s->solver->vel_reinits;
}

float bq_solver_last_distortion(const bq_solver *s, int which)
{
    if (!s) return 0.f;
    return which ? s->solver->last_scalar_distortion : s->solver->last_vel_distortion;
}

void bq_solver_advance(bq_solver *s, int framenum, float dt)
{
    BQ_ENTER(s);
    if (s) s->solver->advance(framenum, dt);
}

long bq_solver_output_result(bq_solver *s, unsigned frame, const char *path)
{
    BQ_ENTER(s);
    if (!s) return -1;
    return s->solver->outputResult(frame, path ? std::string(path) : std::string());
}

int bq_solver_output_result_async(bq_solver *s, unsigned frame, const char *path)
{
    BQ_ENTER(s);
    if (!s) return 0;
    return s->solver->outputResultAsync(frame, path ? std::string(path) : std::string()) ? 1 : 0;
}

long bq_solver_output_wait(bq_solver *s)
{
    BQ_ENTER(s);
    return s ? s->solver->waitOutput() : -1;
}

long bq_solver_download(bq_solver *s, int which, float *host, long capacity)
{
    BQ_ENTER(s);
    if (!s) return 0;
    BimocqGPUSolver &b = *s->solver;
    MapSet &m = *b.VelocityAdvector.maps;
    const DeviceField *f[BQ_F_COUNT] = {
        &b.Density, &b.Temperature, &b.VelocityU, &b.VelocityV, &b.VelocityW,
        &b.VelocityUInit, &b.VelocityVInit, &b.VelocityWInit, &b.DensityInit, &b.TemperatureInit,
        &m.ForwardX, &m.ForwardY, &m.ForwardZ, &m.BackwardX, &m.BackwardY, &m.BackwardZ, &b.p, &b.div };
    if (which < 0 || which >= BQ_F_COUNT) return 0;
    long count = (long)f[which]->count();
    if (host && capacity > 0)
        fl_memcpy_d2h(host, f[which]->get(), (size_t)std::min(count, capacity) * sizeof(float));
    return count;
}

/* slab geometry of this solver: {on, rank, nranks, nk_global, own0, own1, ghost, nk_local} */
void bq_solver_slab_info(const bq_solver *s, int out[8])
{
    if (!s || !out) return;
    const SlabCtx &sl = s->mapper->slab;
    out[0] = sl.on; out[1] = sl.rank; out[2] = sl.nranks; out[3] = sl.on ? sl.nkg : s->mapper->g.nk;
    out[4] = sl.on ? sl.own0 : 0; out[5] = sl.on ? sl.own1 : s->mapper->g.nk; out[6] = sl.G; out[7] = s->mapper->g.nk;
}

long bq_solver_mg_history(const bq_solver *s, double *host, long capacity)
{
    BQ_ENTER(s);
    if (!s || !s->solver->mg.ready) return 0;
    const std::vector<double> h = s->solver->mgHistory();
    if (host)
        for (long a = 0; a < capacity && a < (long)h.size(); a++) host[a] = h[(size_t)a];
    return (long)h.size();
}

long long bq_solver_phase_ms(bq_solver *s, double ms[BQ_PHASE_COUNT], int reset)
{
    BQ_ENTER(s);
    if (!s || !ms) return 0;
    long long steps = 0;
    s->solver->phaseTotals(ms, &steps, reset != 0);
    return steps;
}

float bq_solver_last_cfldt(const bq_solver *s) { return s ? s->solver->last_cfldt : 0.f; }
float bq_solver_last_ms(const bq_solver *s) { return s ? s->solver->last_ms : 0.f; }
int   bq_solver_reinit_count(const bq_solver *s) { return s ? (int)s->solver->VelocityAdvector.TotalReinitCount : 0; }

} // extern "C"
