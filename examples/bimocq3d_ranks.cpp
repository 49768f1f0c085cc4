// bimocq3d_ranks.cpp -- the reference's driver loop (src/bimocq3D/main.cpp:137-159, GPU branch) as ONE RANK of an N-GPU
// run, in C++ only: no Python, no torch.distributed.  Every rank is this program, started once per GPU by whatever
// launcher is at hand (mpirun, srun, torchrun, a shell loop):
//
//   for r in 0 1 2 3 4 5 6 7; do RANK=$r WORLD_SIZE=8 build/bimocq3d_ranks 1024 1024 512 20 out 1 & done; wait
//
// RANK / WORLD_SIZE / LOCAL_RANK are read from the environment (LOCAL_RANK defaults to RANK: one node).  Rank 0 obtains the
// 128-byte ncclUniqueId from the library (fl_comm_unique_id) and publishes it as <outdir>/.nccl_id.<MASTER_PORT or BQ_JOB_ID> (written to a temporary
// name, then renamed: readers never see a partial file); the other ranks poll for it; all call fl_comm_init.  After that the
// library's own RCCL communicator moves ghost planes and wall sheets over xGMI (csrc/bq_halo.hip); nothing else is shared.
// Each rank dumps the planes it owns: <outdir>/density_render_%04d.k%05d.bqd (the slab files stitch to the global grid).
//
//   build/bimocq3d_ranks [NX=512] [NY=512] [NZ=512] [frames=20] [outdir=out] [scene=0 smoke|1 leapfrog] [ghost=8] [jacobi=200]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "bimocq_gpu.h"
#include "fluid_solver.hpp"

static int env_int(const char *name, int fallback)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : fallback;
}

// rank 0: write the id; everyone else: wait for it (up to `timeout_s`).  Returns false on failure.
static bool share_unique_id(const std::string &dir, int rank, unsigned char id[128], double timeout_s)
{
    // one file per job: launchers give every job its own MASTER_PORT (or set BQ_JOB_ID), so a rank that starts before rank 0
    // cannot pick up the id an earlier run left behind
    const char *job = std::getenv("BQ_JOB_ID") ? std::getenv("BQ_JOB_ID") : std::getenv("MASTER_PORT");
    const std::string path = dir + "/.nccl_id." + (job && *job ? job : "0"), tmp = path + ".tmp";
    if (rank == 0) {
        if (fl_comm_unique_id(id) != FL_OK) return false;
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f) return false;
        const bool ok = std::fwrite(id, 1, 128, f) == 128;
        std::fclose(f);
        return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < timeout_s) {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (f) {
            const size_t n = std::fread(id, 1, 128, f);
            std::fclose(f);
            if (n == 128) return true;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    return false;
}

int main(int argc, char **argv)
{
    using namespace bqhost;
    const int ni = argc > 1 ? std::atoi(argv[1]) : 512, nj = argc > 2 ? std::atoi(argv[2]) : 512, nk = argc > 3 ? std::atoi(argv[3]) : 512;
    const int total_frame = argc > 4 ? std::atoi(argv[4]) : 20;
    const std::string filepath = argc > 5 ? argv[5] : "out";
    const int scene = argc > 6 ? std::atoi(argv[6]) : 0;
    const int ghost = argc > 7 ? std::atoi(argv[7]) : 8;
    const int jacobi = argc > 8 ? std::atoi(argv[8]) : 200;
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
    if (ni < 8 || nj < 8 || nk < 8 || total_frame < 1 || world < 1 || rank < 0 || rank >= world || nk % world != 0) {
        std::fprintf(stderr, "usage: RANK=r WORLD_SIZE=n %s [NX] [NY] [NZ divisible by n] [frames] [outdir] [scene] [ghost] [jacobi]\n", argv[0]);
        return 2;
    }
    ::mkdir(filepath.c_str(), 0777);
    const float L = 1.f, h = L / (float)ni, dt = 2.f * h;
    if (fl_init(local) != FL_OK) { std::fprintf(stderr, "[rank %d] %s\n", rank, fl_last_error_string()); return 1; }
    if (world > 1) {
        unsigned char id[128];
        if (!share_unique_id(filepath, rank, id, 120.0)) { std::fprintf(stderr, "[rank %d] no ncclUniqueId (%s)\n", rank, fl_last_error_string()); return 1; }
        if (fl_comm_init(id, rank, world) != FL_OK) { std::fprintf(stderr, "[rank %d] fl_comm_init: %s\n", rank, fl_last_error_string()); return 1; }
    }

    SlabCtx slab;                                                    // even z-slabs, as bq_solver_create_slab cuts them
    if (world > 1) {
        slab.on = true; slab.rank = rank; slab.nranks = world; slab.nkg = nk; slab.G = ghost;
        slab.own0 = rank * (nk / world); slab.own1 = slab.own0 + nk / world;
    }
    auto *myGPUmapper = new gpuMapper(local, ni, nj, nk, h, slab);
    if (!myGPUmapper->ok()) { std::fprintf(stderr, "[rank %d] %s\n", rank, fl_last_error_string()); return 1; }
    BimocqGPUSolver mysolver(ni, nj, nk, L, 0.f, 1.f, BIMOCQ, myGPUmapper);
    if (!mysolver.ok()) { std::fprintf(stderr, "[rank %d] %s\n", rank, fl_last_error_string()); return 1; }

    float zc = 0.5f * (float)nk * h;
    if (scene == 1) {                                                // BASELINE config 5 (gpufluidsimulation_amd/scenes.py: leapfrog)
        Emitter a, b;
        a.emitFrame = b.emitFrame = 10; a.emit_density = b.emit_density = 1.f; a.emit_temperature = b.emit_temperature = 0.f;
        a.emiter = b.emiter = 1.f; a.radius = b.radius = 0.08f;
        // the ring axis passes BETWEEN nodes: a node on it would get 0/0 from the emitter's direction normalisation
        const float yc = 0.5f + 0.37f * h;
        zc += 0.29f * h;
        a.e_pos[0] = 0.15f; a.e_pos[1] = yc; a.e_pos[2] = zc;
        b.e_pos[0] = 0.35f; b.e_pos[1] = yc; b.e_pos[2] = zc;
        mysolver.setSmoke(0.f, 0.f, { a, b });
    } else {                                                         // SURVEY 8(d) rising smoke
        Emitter src;
        src.emitFrame = 1; src.emit_density = 1.f; src.emit_temperature = 1.f; src.emiter = 0.f;
        src.e_pos[0] = 0.5f; src.e_pos[1] = 0.2f; src.e_pos[2] = zc; src.radius = 0.1f;
        mysolver.setSmoke(0.f, 1.f, { src });
    }
    mysolver.jacobi_iters = jacobi;
    mysolver.verbose = rank == 0;

    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < total_frame; i++) {
        if (rank == 0) std::printf("Frame %d Starts !!!\n", i);
        mysolver.advance(i, dt);
        mysolver.outputResultAsync((unsigned)i, filepath);          // this rank's planes, written while frame i + 1 runs
        if (fl_last_error() != FL_OK) { std::fprintf(stderr, "[rank %d] %s\n", rank, fl_last_error_string()); return 1; }
    }
    const long last = mysolver.waitOutput();
    fl_sync();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long long st[4] = { 0, 0, 0, 0 };
    fl_comm_stats(st, 0);
    std::printf("[rank %d/%d] %d frames of %dx%dx%d (planes %d..%d) in %.3f s, %.1f Mvoxels/s of the global grid incl. dumps, "
                "%lld ghost exchanges, %.1f MB sent, last dump %s\n", rank, world, total_frame, ni, nj, nk,
                slab.on ? slab.own0 : 0, slab.on ? slab.own1 : nk, sec, (double)ni * nj * nk * total_frame / sec / 1e6,
                st[0], (double)(st[1] + st[3]) / 1e6, last >= 0 ? "ok" : "FAILED");
    if (world > 1) fl_comm_destroy();
    return last >= 0 ? 0 : 1;
}
