// solver_capi.cpp -- C view of the host solver (include/bimocq_solver.h).
#include "fluid_solver.hpp"

#include <algorithm>
#include <memory>

using namespace bqhost;

struct bq_solver {
    std::unique_ptr<gpuMapper> mapper;
    std::unique_ptr<BimocqGPUSolver> solver;
};

extern "C" {

bq_solver *bq_solver_create(int device, int nx, int ny, int nz, float L, float viscosity, float blend, int scheme)
{
    if (nx < 8 || ny < 8 || nz < 8 || scheme != BQ_SCHEME_BIMOCQ) return nullptr;
    auto s = std::make_unique<bq_solver>();
    s->mapper = std::make_unique<gpuMapper>(device, nx, ny, nz, L / nx);       // main.cpp:151 / :37 (h = L/ni)
    if (!s->mapper->ok()) return nullptr;
    s->solver = std::make_unique<BimocqGPUSolver>(nx, ny, nz, L, viscosity, blend, BIMOCQ, s->mapper.get());
    if (!s->solver->ok()) return nullptr;
    return s.release();
}

void bq_solver_destroy(bq_solver *s)
{
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
    if (!s || kind != BQ_PROJECTION_JACOBI) return;
    s->solver->jacobi_iters = iters;
    s->solver->halfrdx = halfrdx;
}

void bq_solver_advance(bq_solver *s, int framenum, float dt)
{
    if (s) s->solver->advance(framenum, dt);
}

long bq_solver_output_result(bq_solver *s, unsigned frame, const char *path)
{
    if (!s) return -1;
    return s->solver->outputResult(frame, path ? std::string(path) : std::string());
}

long bq_solver_download(bq_solver *s, int which, float *host, long capacity)
{
    if (!s) return 0;
    BimocqGPUSolver &b = *s->solver;
    MapSet &m = *b.VelocityAdvector.maps;
    const DeviceField *f[BQ_F_COUNT] = {
        &b.Density, &b.Temperature, &b.VelocityU, &b.VelocityV, &b.VelocityW,
        &b.VelocityUInit, &b.VelocityVInit, &b.VelocityWInit, &b.DensityInit, &b.TemperatureInit,
        &m.ForwardX, &m.ForwardY, &m.ForwardZ, &m.BackwardX, &m.BackwardY, &m.BackwardZ, &b.p };
    if (which < 0 || which >= BQ_F_COUNT) return 0;
    long count = (long)f[which]->count();
    if (host && capacity > 0)
        fl_memcpy_d2h(host, f[which]->get(), (size_t)std::min(count, capacity) * sizeof(float));
    return count;
}

float bq_solver_last_cfldt(const bq_solver *s) { return s ? s->solver->last_cfldt : 0.f; }
float bq_solver_last_ms(const bq_solver *s) { return s ? s->solver->last_ms : 0.f; }
int   bq_solver_reinit_count(const bq_solver *s) { return s ? (int)s->solver->VelocityAdvector.TotalReinitCount : 0; }

} // extern "C"
