/*
 * bimocq_solver.h -- C view of the C++ host solver (gpufluidsimulation_amd/csrc/host/).
 *
 * The host solver keeps the reference's BimocqGPUSolver surface
 * (reference: src/bimocq3D/BimocqGPUSolver.h:27-56): construct with (nx, ny, nz, L, viscosity,
 * blend, scheme), setSmoke, advance(framenum, dt), outputResult(frame, path).  It is plain C++
 * that calls nothing but the C-ABI of include/bimocq_gpu.h; this header lets non-C++ hosts
 * (the Python tests and bench.py via ctypes) drive the same object.
 */
#ifndef BIMOCQ_SOLVER_H
#define BIMOCQ_SOLVER_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bq_solver bq_solver;

/* one spherical smoke source; the reference hard-codes two of these in
 * BimocqGPUSolver::emitSmoke (BimocqGPUSolver.cpp:387-390) */
typedef struct bq_emitter {
    float cx, cy, cz, radius, density, temperature, emiter;
    int   emit_frames;                  /* active while framenum < emit_frames */
} bq_emitter;

/* enum Scheme, BimocqSolver.h:29.  The reference's GPU solver implements BIMOCQ and MAC_REFLECTION
 * (BimocqGPUSolver.cpp:112-122); the reflection scheme runs with the corrected limiter (gpu_clamp_extrema). */
enum { BQ_SCHEME_BIMOCQ = 0, BQ_SCHEME_MAC_REFLECTION = 3 };
enum {
    BQ_PROJECTION_JACOBI = 0,           /* the `#if 0` branch of BimocqGPUSolver::projection (:408-417); iters = sweeps      */
    BQ_PROJECTION_MGCG = 1              /* the `#else` branch (:443-446): fp64 multigrid-CG; iters = outer iterations (50)  */
};

/* which-ids for bq_solver_download */
enum {
    BQ_F_RHO = 0, BQ_F_T, BQ_F_U, BQ_F_V, BQ_F_W, BQ_F_UINIT, BQ_F_VINIT, BQ_F_WINIT,
    BQ_F_RHOINIT, BQ_F_TINIT, BQ_F_FWDX, BQ_F_FWDY, BQ_F_FWDZ, BQ_F_BACKX, BQ_F_BACKY, BQ_F_BACKZ,
    BQ_F_P, BQ_F_DIV, BQ_F_COUNT
};

/* BimocqGPUSolver::BimocqGPUSolver (BimocqGPUSolver.cpp:3-106).  device: HIP device index.
 * Returns NULL on failure (see fl_last_error_string()). */
bq_solver *bq_solver_create(int device, int nx, int ny, int nz, float L,
                            float viscosity, float blend, int scheme);
/* z-slab rank of a multi-GPU run (one process per GPU): nz is the GLOBAL plane count, rank r of
 * nranks owns planes [r*nz/nranks, (r+1)*nz/nranks) and keeps `ghost` ghost planes per side
 * (>= CFL travel + 3; 8 covers CFL <= 5).  Requires fl_comm_init() first when nranks > 1. */
bq_solver *bq_solver_create_slab(int device, int nx, int ny, int nz, float L, float viscosity, float blend,
                                 int scheme, int rank, int nranks, int ghost);
void  bq_solver_slab_info(const bq_solver *s, int out[8]);
void  bq_solver_destroy(bq_solver *s);
/* setSmoke (BimocqGPUSolver.cpp:529-534): alpha = drop (rho coefficient), beta = rise (T) */
void  bq_solver_set_smoke(bq_solver *s, float drop, float rise, const bq_emitter *emitters, int n);
/* projection variant + parameters (compile-time `#if` in the reference, :408-466) */
void  bq_solver_set_projection(bq_solver *s, int kind, int iters, float halfrdx);
/* solver options: BQ_OPT_KEEP_DMC_BORDER (default 0 = reference behaviour: the backward map's border
 * nodes are zeroed by the DMC update; 1 = they keep their values, see csrc/host/mapping.hpp) */
enum {
    BQ_OPT_KEEP_DMC_BORDER = 1,
    /* 0 (default): both map sets are re-initialised every frame, as the reference's GPU solver does (`if (1)`,
     * BimocqGPUSolver.cpp:218-229).  1: distortion-driven re-initialisation with the CPU solver's rules
     * (BimocqSolver.cpp:165-229): velocity maps when their round-trip error exceeds 1 step-travel or after 10
     * frames, scalar maps above 5 or after 30 frames; the scalar snapshots are taken before the sources act
     * so that emission reaches DensityInit through the accumulation.  Set before the first advance().
     * z-slab ranks (round 3): maps that live for many steps must still fit the ghost zone, see BQ_OPT_REINIT_MAX_TRAVEL. */
    BQ_OPT_REINIT_POLICY = 2,
    /* 0 (default): state that nothing can read is not computed -- with blend == 1 and a re-initialisation every
     * frame the *Prev fields are never sampled, so the pre-reinit accumulation into *Init (which only survives as
     * *Prev) and the force delta feeding it are skipped.  1: execute the reference's full sequence.  Every field
     * reachable through this API, and every dump, is identical either way. */
    BQ_OPT_FULL_STATE = 3,
    /* 1 (default): the clears and copies the reference issues around the map operators (GPU_Advection.h:464-526,
     * GPU_kernel.cu:656-658) are done by the kernels themselves (FL_OPT_FUSED_HOUSEKEEPING of the operator ABI):
     * same values in every buffer, ~25 fewer memset/memcpy launches per step.  0: separate launches. */
    BQ_OPT_FUSED_HOUSEKEEPING = 4,
    /* z-slab ranks, 1 (default): the ghost-plane exchange in front of a map operator runs on the halo stream while the
     * operator works on the planes that cannot reach a ghost plane; the planes at both ends follow the exchange.
     * 0: exchange, then the whole operator.  Same values either way. */
    BQ_OPT_OVERLAP_EXCHANGES = 5,
    /* z-slab ranks, 0 (default): a BLOCKING ghost-plane refresh (the ones no operator hides: the projection's velocity
     * refresh, the limiter's) moves all G ghost planes, whatever depth was asked for -- the deeper validity spares later
     * operators their own exchange.  1: it moves only the planes asked for (3 instead of 24 MB per velocity refresh at
     * 512^2 planes); the operators that need more fetch it in their own, overlapped exchange.  Same values either way;
     * which is faster depends on the links (host-staged transport: 0).  2: the overlapped exchanges in front of the map
     * operators move only the planes the operator can reach as well (reach 4-5 of G = 8 planes at CFL 1-2). */
    BQ_OPT_SHALLOW_BLOCKING_EXCHANGE = 6,
    /* z-slab ranks with at least 2 G + 8 owned planes, 1 (default): the last two fused pairs of every pressure chunk that
     * another chunk follows run on the planes next to the slab ends first, the exchange for the next chunk starts there,
     * and their interiors (and the next chunk's first interior) run while it travels -- three launches hide the 8-plane
     * exchange instead of one, at two more short launches per chunk (+ 2 % compute on a 512 x 512 x 64 rank).  0: the
     * exchange starts when the chunk is complete.  Same values either way. */
    BQ_OPT_JACOBI_ENDS_FIRST = 7,
    /* 1: advanceBimocq brackets its phases with events on the compute stream (read with bq_solver_phase_ms): where the
     * step's time goes on this rank, the waits for ghost planes included in the phase that needs them.  Default 0. */
    BQ_OPT_PROFILE_PHASES = 8,
    /* policy 1 only.  T > 0: after every map update the z-travel of both maps of a set is measured (gpu_map_travel_z: max
     * |map_z - z| / h, all-reduced over the ranks) and replaces the running sum of CFL travels as the set's displacement
     * bound; a set whose bound + this step's CFL travel + the two cells of the sampling footprint exceeds T is
     * re-initialised at the end of the step even though the CPU solver's thresholds would let it live on.  z-slab ranks:
     * default and maximum T = the ghost depth G (what their ghost planes can serve); single GPU: default 0 (no such rule) --
     * give both the same T and they re-initialise on the same frames and produce the same fields
     * (tests/test_slab_multirank.py).  bq_solver_reinit_counts(s, 2) counts the re-initialisations this rule caused. */
    BQ_OPT_REINIT_MAX_TRAVEL = 9,
    /* z-slab ranks, G = 8, 1 (default): the six sweeps of a pressure chunk that follow its overlapped first pair run as TWO
     * fused triples (gpu_jacobi_sweep_triple_ranges: the LDS-exchanged three-sweep kernels on plane ranges) instead of three
     * pairs -- with BQ_OPT_JACOBI_ENDS_FIRST both triples do the planes next to the slab ends first, the exchange for the next
     * chunk starts there and the two interiors (and the next chunk's first interior) hide it.  Where the kernels do not apply
     * (rows wider than 512 floats, ...) and with 0 the pairs run.  Same values either way. */
    BQ_OPT_JACOBI_TRIPLES = 10,
    /* z-slab ranks with the multigrid-CG projection, 1 (default): the grid's fine levels are SHARED between the ranks
     * (gpu_multi_grid_conjugate_gradient_slab: owned + ghost planes per level, neighbour exchanges, all-gathered block dot
     * products; thin levels gathered and solved replicated) wherever gpu_mgcg_slab_supported says the decomposition allows it
     * -- no global fp64 array on any rank, the solve scales with the rank count.  0, and any decomposition it does not cover:
     * the replicated solve of round 3 (every rank assembles the whole velocity and solves the whole grid).  Same bits.
     * bq_solver_get_option returns 2 once a projection has taken the shared path. */
    BQ_OPT_MGCG_SHARED = 11,
    /* one GPU, 1: updateMapping launches the forward-map update on the library's auxiliary stream beside the backward map's DMC
     * sub-steps (fl_aux_*: disjoint arrays) instead of after them.  Same values.  Default 0: measured at 256^3 and 128^3, the
     * two kernels each fill the chip and gain nothing from running side by side (11.14-11.19 against 11.10-11.15 ms per step). */
    BQ_OPT_CONCURRENT_MAPS = 12,
    /* z-slab ranks, blend coefficient != 1, reference-faithful map border (BQ_OPT_KEEP_DMC_BORDER = 0), 1 (default): at every
     * re-initialisation each rank assembles whole-grid copies of the *Prev fields (the only time they change; one message per
     * peer and field) and the two-level advection samples those (gpu_advect_vel_double_global, include/bimocq_gpu.h): its second
     * look-up lands anywhere between the origin and the node once it meets the zeroed border cells of the previous backward
     * map, which no ghost zone or sheet bounds.  Bit-identical to one GPU.  0: local *Prev fields with ghost planes only --
     * exact as long as no node within a cell of the outermost window nodes is carried more than 3/4 of a cell towards a wall
     * between two re-initialisations (reads outside the slab return 0 otherwise).  get: 2 = copies are in use.  Costs five
     * whole-grid float arrays per rank. */
    BQ_OPT_WHOLE_GRID_PREV = 13
};
/* BQ_OPT_PROFILE_PHASES: milliseconds per phase summed over the steps since the last reset -- map update (DMC + RK3,
 * BimocqGPUSolver.cpp:136-139), advection with error compensation (:143-145), sources and forces (:157-177), projection
 * (:179-193), accumulation and re-initialisation (:195-229).  Returns the number of steps summed.  Blocking. */
enum { BQ_PHASE_MAPS = 0, BQ_PHASE_ADVECT, BQ_PHASE_FORCES, BQ_PHASE_PROJECTION, BQ_PHASE_ACCUMULATE, BQ_PHASE_COUNT };
long long bq_solver_phase_ms(bq_solver *s, double ms[BQ_PHASE_COUNT], int reset);
/* after a step: re-initialisation counts (which: 0 velocity maps, 1 scalar maps, 2 those forced by BQ_OPT_REINIT_MAX_TRAVEL) and the distortions the
 * last step measured (policy 1; 0 otherwise) */
int   bq_solver_reinit_counts(const bq_solver *s, int which);
float bq_solver_last_distortion(const bq_solver *s, int which);
void  bq_solver_set_option(bq_solver *s, int option, int value);
/* current value of an option above (-1: unknown option) */
int   bq_solver_get_option(const bq_solver *s, int option);
/* advance (BimocqGPUSolver.cpp:108-127) */
void  bq_solver_advance(bq_solver *s, int framenum, float dt);
/* outputResult (BimocqGPUSolver.cpp:536-543): D2H of rho,u,v,w and a sparse density dump
 * <path>/density_render_%04d.bqd for frame+1 (writeVDB's contract, utils/volumeMeshTools.h:33-60,
 * in a dependency-free container).  path == NULL: only refresh the host copies.  Returns the
 * number of voxels written (|rho| > 1e-4) or -1 on error. */
long  bq_solver_output_result(bq_solver *s, unsigned frame, const char *path);
/* copy one device field to host (blocking).  Returns its element count (0 on bad id); copies
 * min(count, capacity) elements when host != NULL. */
long  bq_solver_download(bq_solver *s, int which, float *host, long capacity);
/* tempResult of the last multigrid-CG projection (4096 doubles: CG sums at [0..2*iters+2], largest positive
 * residual per outer iteration at [2000..2000+iters] -- what the reference prints, BimocqGPUSolver.cpp:447-452).
 * Returns the count (0 before the first MGCG projection); copies min(count, capacity). */
long  bq_solver_mg_history(const bq_solver *s, double *host, long capacity);
/* The dump without stalling the simulation: asynchronous download on a third stream + a writer thread; the
 * file is the one bq_solver_output_result would write.  At most one dump in flight (a second call waits for
 * the first).  _wait returns the voxel count of the last asynchronous dump, or -1. */
int   bq_solver_output_result_async(bq_solver *s, unsigned frame, const char *path);
long  bq_solver_output_wait(bq_solver *s);
float bq_solver_last_cfldt(const bq_solver *s);
float bq_solver_last_ms(const bq_solver *s);          /* event time of the last advance()        */
int   bq_solver_reinit_count(const bq_solver *s);

#ifdef __cplusplus
}
#endif
#endif
