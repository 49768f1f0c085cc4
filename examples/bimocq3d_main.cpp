// bimocq3d_main.cpp -- the reference's driver loop (src/bimocq3D/main.cpp:137-159, GPU branch) on this library:
//   gpuMapper + BimocqGPUSolver(ni, nj, nk, L, viscosity, blend, scheme, mapper); setSmoke; advance(i, dt);
//   outputResult(i, path) every frame.
// The scene is the synthetic rising-smoke case of SURVEY 8(d) (one warm sphere near the floor) instead of the
// reference's OpenVDB-SDF emitters, which need OpenVDB.
//
// scene = 1 is BASELINE config 5's shape: an N x N x N/2 box (1024 x 1024 x 512 at N = 1024) with two coaxial vortex
// rings blown along x by the reference's emitter velocity formula (main.cpp:52-73, emiter = +1 for both: the rear
// ring catches up and threads the front one -- leapfrogging), no buoyancy, density dumped every frame.
//
//   make example && build/bimocq3d [N=128] [frames=20] [outdir=out] [scheme=0|3] [projection=0|1] [async=1] [scene=0|1]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "bimocq_gpu.h"
#include "fluid_solver.hpp"

int main(int argc, char **argv)
{
    using namespace bqhost;
    const int n = argc > 1 ? std::atoi(argv[1]) : 128;
    const int total_frame = argc > 2 ? std::atoi(argv[2]) : 20;
    const std::string filepath = argc > 3 ? argv[3] : "out";
    const int scheme = argc > 4 ? std::atoi(argv[4]) : 0;            // 0 BIMOCQ, 3 MAC_REFLECTION (main.cpp:51 ships 3)
    const int projection = argc > 5 ? std::atoi(argv[5]) : 0;        // 0 Jacobi, 1 multigrid-CG (what the binary ships)
    const bool async_dump = argc > 6 ? std::atoi(argv[6]) != 0 : true;
    const int scene = argc > 7 ? std::atoi(argv[7]) : 0;             // 0 rising smoke, 1 leapfrogging vortex rings
    if (n < 8 || total_frame < 1 || (scene == 1 && n % 2)) {
        std::fprintf(stderr, "usage: %s [N>=8] [frames] [outdir] [scheme] [projection] [async] [scene]\n", argv[0]); return 2;
    }

    const int ni = n, nj = n, nk = scene == 1 ? n / 2 : n;
    const float L = 1.f, h = L / (float)n, dt = 2.f * h;
    const float viscosity = 0.f, mapping_blend_coeff = 1.f;          // main.cpp:46-47
    const float smoke_rise = scene == 1 ? 0.f : 1.f, smoke_drop = 0.f;

    if (fl_init(0) != FL_OK) { std::fprintf(stderr, "%s\n", fl_last_error_string()); return 1; }
    auto *myGPUmapper = new gpuMapper(/*device*/0, ni, nj, nk, h);
    BimocqGPUSolver mysolver(ni, nj, nk, L, viscosity, mapping_blend_coeff, scheme == 3 ? MAC_REFLECTION : BIMOCQ, myGPUmapper);
    if (!myGPUmapper->ok() || !mysolver.ok()) { std::fprintf(stderr, "%s\n", fl_last_error_string()); return 1; }

    if (scene == 1) {
        Emitter a, b;                                                // main.cpp:75-78: 10 frames, density 1, +x velocity ring
        a.emitFrame = b.emitFrame = 10; a.emit_density = b.emit_density = 1.f; a.emit_temperature = b.emit_temperature = 0.f;
        a.emiter = b.emiter = 1.f; a.radius = b.radius = 0.08f;
        // the ring axis passes BETWEEN nodes: a node on it would get 0/0 from the emitter's direction normalisation
        const float yc = 0.5f + 0.37f * h, zc = 0.5f * (float)nk * h + 0.29f * h;
        a.e_pos[0] = 0.15f; a.e_pos[1] = yc; a.e_pos[2] = zc;
        b.e_pos[0] = 0.35f; b.e_pos[1] = yc; b.e_pos[2] = zc;
        mysolver.setSmoke(smoke_drop, smoke_rise, { a, b });
    } else {
        Emitter src;                                                 // one warm sphere, applied at frame 0 only
        src.emitFrame = 1; src.emit_density = 1.f; src.emit_temperature = 1.f; src.emiter = 0.f;
        src.e_pos[0] = 0.5f; src.e_pos[1] = 0.2f; src.e_pos[2] = 0.5f; src.radius = 0.1f;
        mysolver.setSmoke(smoke_drop, smoke_rise, { src });
    }
    if (projection == 1) { mysolver.projection_kind = BQ_PROJECTION_MGCG; mysolver.mg_iters = 50; }
    else                 { mysolver.jacobi_iters = 200; }
    mysolver.verbose = true;                                         // "[Bimocq GPU Time: ...ms ]" like the reference

    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < total_frame; i++) {
        std::printf("Frame %d Starts !!!\n", i);
        mysolver.advance(i, dt);
        if (async_dump) mysolver.outputResultAsync((unsigned)i, filepath);      // written while frame i + 1 runs
        else std::printf("[ Valid voxel: %ld ]\n", mysolver.outputResult((unsigned)i, filepath));
        if (fl_last_error() != FL_OK) { std::fprintf(stderr, "%s\n", fl_last_error_string()); return 1; }
    }
    const long last = async_dump ? mysolver.waitOutput() : 0;
    fl_sync();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%d frames of %dx%dx%d in %.3f s (%.1f Mvoxels/s incl. dumps)%s\n", total_frame, ni, nj, nk, sec,
                (double)ni * nj * nk * total_frame / sec / 1e6, async_dump ? (last >= 0 ? ", last dump ok" : ", last dump FAILED") : "");
    return 0;
}
