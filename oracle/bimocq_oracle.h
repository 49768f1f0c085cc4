/*
 * bimocq_oracle.h -- CPU restatement of the bimocq3D per-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and there only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (Hyberge/GPUFluidSimulation) holds no golden
 * vectors, tests or fixtures for this path and its kernels
 * cannot be built in this image without writing stand-ins for the CUDA runtime,
 * cuda-samples and TBB headers, which the build rules forbid.  The restatement
 * follows the reference source line by line (citations on every function,
 * relative to /root/reference/src/bimocq3D/) and is cross-checked by analytic
 * known-answer tests plus the run statistics that SURVEY.md section 8(c) recorded
 * from the reference's own kernels (tests/test_oracle_kat.py).
 *
 * Arithmetic contract (what "bit-exact" between this oracle and the HIP path means):
 *   - every expression is evaluated with the operand types and association of the
 *     reference source, fp32 unless the source promotes to double (lerp, RK3 stage
 *     points, buoyancy, the 9-point blend in cumulate/compensate);
 *   - no FMA contraction (-ffp-contract=off on both sides);
 *   - expf of the DMC integrator is orc_expf(): a fixed double-precision polynomial,
 *     identical on both sides, within 1 ulp of a correctly rounded expf;
 *   - loads that fall outside the sampled allocation return 0.0f (the reference
 *     leaves them undefined); in-allocation wrap-around reads are kept.
 */
#ifndef BIMOCQ_ORACLE_H
#define BIMOCQ_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* z-slab context (mirror of fl_set_slab, include/bimocq_gpu.h section 4): the buffers of every
 * following call hold the global planes [koff, koff + nk_local) of a grid with nk_global cell
 * planes, of which [own0, own1) are owned.  nk_global <= 0 switches back to a single domain. */
void orc_set_slab(int koff, int nk_global, int own0, int own1, int nk_local);

/* ---- math helpers exposed for tests ---- */
float orc_expf(float x);
void  orc_set_fast_lerp(int on);
float orc_lerp(float a, float b, float c);
float orc_sample(const float *b, int nx, int ny, int nz, float h,
                 float ox, float oy, float oz, float px, float py, float pz);

/* ---- operators: same argument order as the reference's extern "C" gpu_* ---- */
void orc_solve_forward(const float *u, const float *v, const float *w,
                       float *x_fwd, float *y_fwd, float *z_fwd,
                       float h, int ni, int nj, int nk, float cfldt, float dt);
void orc_solve_backwardDMC(const float *u, const float *v, const float *w,
                           const float *x_in, const float *y_in, const float *z_in,
                           float *x_out, float *y_out, float *z_out,
                           float h, int ni, int nj, int nk, float substep);
void orc_advect_velocity(float *u, float *v, float *w,
                         const float *u_init, const float *v_init, const float *w_init,
                         const float *bx, const float *by, const float *bz,
                         float h, int ni, int nj, int nk, int is_point);
void orc_advect_vel_double(float *u, float *v, float *w,
                           const float *utemp, const float *vtemp, const float *wtemp,
                           const float *bx, const float *by, const float *bz,
                           const float *bxp, const float *byp, const float *bzp,
                           float h, int ni, int nj, int nk, int is_point, float blend);
void orc_advect_field(float *field, const float *field_init,
                      const float *bx, const float *by, const float *bz,
                      float h, int ni, int nj, int nk, int is_point);
void orc_advect_field_double(float *field, const float *field_prev,
                             const float *bx, const float *by, const float *bz,
                             const float *bxp, const float *byp, const float *bzp,
                             float h, int ni, int nj, int nk, int is_point, float blend);
void orc_accumulate_velocity(const float *uc, const float *vc, const float *wc,
                             float *du_init, float *dv_init, float *dw_init,
                             const float *fx, const float *fy, const float *fz,
                             float h, int ni, int nj, int nk, int is_point, float coeff);
void orc_accumulate_component(const float *change, float *d_init, const float *fx, const float *fy, const float *fz,
                              float h, int ni, int nj, int nk, int axis, int is_point, float coeff);
void orc_accumulate_field(const float *change, float *dfield_init,
                          const float *fx, const float *fy, const float *fz,
                          float h, int ni, int nj, int nk, int is_point, float coeff);
void orc_estimate_distortion(float *dist,
                             const float *xb, const float *yb, const float *zb,
                             const float *xf, const float *yf, const float *zf,
                             float h, int ni, int nj, int nk);
void orc_add(float *f1, const float *f2, float coeff, int number);
void orc_compensate_velocity(float *u, float *v, float *w,
                             float *du, float *dv, float *dw,
                             float *u_src, float *v_src, float *w_src,
                             const float *fx, const float *fy, const float *fz,
                             const float *bx, const float *by, const float *bz,
                             float h, int ni, int nj, int nk, int is_point);
void orc_compensate_field(float *u, float *du, float *u_src,
                          const float *fx, const float *fy, const float *fz,
                          const float *bx, const float *by, const float *bz,
                          float h, int ni, int nj, int nk, int is_point);
void orc_compensate_error_velocity(const float *u, const float *v, const float *w,
                                   const float *du, const float *dv, const float *dw,
                                   float *u_src, float *v_src, float *w_src,
                                   const float *fx, const float *fy, const float *fz,
                                   float h, int ni, int nj, int nk, int is_point);
void orc_compensate_error_field(const float *u, const float *du, float *u_src,
                                const float *fx, const float *fy, const float *fz,
                                float h, int ni, int nj, int nk, int is_point);
void orc_clamp_extrema_box_w(const float *before, float *after, int ni, int nj, int nk_buffer);
void orc_clamp_extrema(const float *field, float *field_temp, const float *u, const float *v, const float *w,
                       int ni, int nj, int nk, int dimx, int dimy, int dimz, float ox, float oy, float oz,
                       float h, float dt);
void orc_semilag(float *field, const float *field_src,
                 const float *u, const float *v, const float *w,
                 int dim_x, int dim_y, int dim_z,
                 float h, int ni, int nj, int nk, float cfldt, float dt);
void orc_emit_smoke(float *u, float *v, float *w, float *rho, float *T,
                    float h, int ni, int nj, int nk,
                    float cx, float cy, float cz, float radius,
                    float density, float temperature, float emiter);
void orc_add_buoyancy(float *v, const float *rho, const float *T,
                      int ni, int nj, int nk, float alpha, float beta, float dt);
int orc_diffuse_sweeps(const float *field, float *in, float *out, int ni, int nj, int nk, int iter, float coef);
void orc_diffuse_field(float *field, float *tmp0, float *tmp1,
                       int ni, int nj, int nk, int iter, float coef);
void orc_add_field(float *out, const float *f1, const float *f2, float coeff, int number);
void orc_mad(float *out, const float *f1, const float *f2, float c1, float c2, int number);
void orc_clamp_extrema_box(const float *before, float *after, int ni, int nj, int nk);
void orc_divergence(const float *u, const float *v, const float *w, float *div,
                    int ni, int nj, int nk, float halfrdx);
/* restrict every operator to the local planes [k0, k1) (k0 < 0: off); nk_cells = local cell planes */
void orc_set_plane_window(int k0, int k1, int nk_cells);
/* z-slab ranks: cumulate_kernel's expression on the wall layers of its window, with the source read from an assembled copy
 * that holds the global planes [src_koff, src_koff + src_nk): dst = before + blend9(coeff * src(map(x))) on the window nodes
 * with i in xlist, j in ylist or GLOBAL plane in zlist (bimocq_oracle.c) */
/* orc_advect_vel_double / orc_advect_field_double on a z-slab rank (orc_set_slab) with the *_prev fields of the whole grid
 * (include/bimocq_gpu.h: gpu_advect_vel_double_global) */
void orc_advect_vel_double_global(float *u, float *v, float *w,
                                  const float *uprev_g, const float *vprev_g, const float *wprev_g,
                                  const float *bx, const float *by, const float *bz,
                                  const float *bxp, const float *byp, const float *bzp,
                                  float h, int ni, int nj, int nk, int is_point, float blend);
void orc_advect_field_double_global(float *field, const float *field_prev_g,
                                    const float *bx, const float *by, const float *bz,
                                    const float *bxp, const float *byp, const float *bzp,
                                    float h, int ni, int nj, int nk, int is_point, float blend);
void orc_accumulate_wall_fixup(const float *src, int src_koff, int src_nk, const float *before, float *dst,
                               const float *mx, const float *my, const float *mz,
                               float h, int ni, int nj, int nk, int axis, float coeff,
                               const int *xlist, int nxl, const int *ylist, int nyl, const int *zlist, int nzl);
void orc_jacobi_sweep(const float *p, const float *div, float *out,
                      int ni, int nj, int nk, float alpha, float beta);
/* ---- fp64 multigrid-CG projection (mgcg_oracle.c; GPU_kernel.cu:1420-1815) ---- */
typedef struct OrcCoarseLevel {          /* layout of SCoarseLevelInfo, GPU_Advection.h:15-24 */
    int ni, nj, nk;
    int number;
    double alpha;
    double beta;
    double *b;
    double *x;
    double *r;
} OrcCoarseLevel;
void orc_multi_grid_conjugate_gradient(float *u, float *v, float *w, double *div, double *p, double *dir,
                                       double *residual, double *temp0, double *temp1, double *tempResult,
                                       OrcCoarseLevel *levels, int levelNum, int iter, double halfrdx);
void orc_mg_divergence(const float *u, const float *v, const float *w, double *div, int ni, int nj, int nk, double halfrdx);
void orc_mg_poisson(const double *x, double *b, int ni, int nj, int nk);
void orc_mg_residual(double *r, const double *b, const double *x, int ni, int nj, int nk);
void orc_mg_dot_partials(const double *v0, const double *v1, double *output, long count);
void orc_mg_calc_sum(const double *v, double *output, long count, long per_thread, int iter_index);
void orc_mg_calc_max(const double *v, double *output, long count, int iter_index);
void orc_mg_smooth(double *x, const double *b, double *temp, double alpha, double beta, int ni, int nj, int nk, int iter);
void orc_mg_restrict(const double *residual, double *coarse, int ni, int nj, int nk, int ci, int cj, int ck);
void orc_mg_prolong(double *x, const double *coarse, int ni, int nj, int nk, int ci, int cj, int ck);
void orc_jacobi_sweep_range(const float *p, const float *div, float *out,
                            int ni, int nj, int nk, int k_begin, int k_end, float alpha, float beta);
void orc_gradient(float *field, const float *p, int nbi, int nbj, int nbk,
                  int dimx, int dimy, int dimz, float halfrdx);
/* residual r = div - (sum6 p - 6p) over interior cells: returns sum r^2 (double
 * accumulation, index order) and max |r| -- the re-specified A15 norms */
void orc_residual_norms(const float *div, const float *p, int ni, int nj, int nk,
                        double *sum_sq, float *max_abs);
/* debug != NULL: debug[it] = sum r^2, debug[2000+it] = max|r| of iterate `it`,
 * it = 0 .. iter-1 (the iterates the sweeps actually read)       */
void orc_projection_jacobi(float *u, float *v, float *w, float *div, float *p, float *p_temp,
                           float *debug, int ni, int nj, int nk, int iter,
                           float halfrdx, float alpha, float beta);
float orc_max_abs3(const float *u, const float *v, const float *w, int ni, int nj, int nk);

/* ---- whole-step state machine (BimocqGPUSolver::advanceBimocq restated) ---- */
typedef struct orc_emitter {
    float cx, cy, cz, radius, density, temperature, emiter;
    int   emit_frames;      /* source active while framenum < emit_frames */
} orc_emitter;

typedef struct orc_solver orc_solver;

orc_solver *orc_solver_create(int ni, int nj, int nk, float L, float viscosity, float blend);
void  orc_solver_destroy(orc_solver *s);
void  orc_solver_set_smoke(orc_solver *s, float drop_alpha, float rise_beta,
                           const orc_emitter *emitters, int n_emitters);
void  orc_solver_set_projection(orc_solver *s, int jacobi_iters, float halfrdx);
/* option 1: keep the backward map's border through the DMC update (0 = reference behaviour) */
/* projection kind: 0 = Jacobi (iters = sweeps), 1 = fp64 multigrid-CG (iters = outer iterations; 50 in
 * the reference, BimocqGPUSolver.cpp:444) */
void  orc_solver_set_projection_kind(orc_solver *s, int kind, int iters);
int   orc_solver_mg_levels(const orc_solver *s);
const double *orc_solver_mg_history(const orc_solver *s);
/* after a step: how often each map set was re-initialised so far (which: 0 velocity, 1 scalar) and the
 * distortions the last step measured (policy 1 only) */
int   orc_solver_reinit_counts(const orc_solver *s, int which);
float orc_solver_last_distortion(const orc_solver *s, int which);
void  orc_solver_set_option(orc_solver *s, int option, int value);
void  orc_solver_advance(orc_solver *s, int framenum, float dt);
/* which: 0 rho, 1 T, 2 u, 3 v, 4 w, 5 uinit, 6 vinit, 7 winit, 8 rhoinit, 9 Tinit,
 *        10..12 forward xyz (velocity mapper), 13..15 backward xyz, 16 p          */
const float *orc_solver_field(orc_solver *s, int which, long *count);
float orc_solver_last_cfldt(const orc_solver *s);

#ifdef __cplusplus
}
#endif
#endif
