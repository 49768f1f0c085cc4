/*
 * bimocq_gpu.h -- C-ABI of the MI355X (gfx950) bimocq3D hot path.
 *
 * Drop-in boundary: the 22 `extern "C" gpu_*` operators below carry exactly the
 * signatures of the reference's CUDA launchers (reference: src/bimocq3D/GPU_Advection.h:26-108,
 * implemented in src/bimocq3D/GPU_kernel.cu), so the reference's gpuMapper / MapperBaseGPU /
 * BimocqGPUSolver link against libbimocq_hip.so unchanged.  All pointers are DEVICE pointers
 * (hipMalloc / fl_malloc); fields are dense fp32, x fastest: index = i + nx*j + nx*ny*k.
 * Staggered sizes: u (ni+1,nj,nk), v (ni,nj+1,nk), w (ni,nj,nk+1); scalars and maps (ni,nj,nk).
 *
 * The fl_* group replaces the raw CUDA runtime calls gpuMapper makes
 * (GPU_Advection.h:214-326: findCudaDevice, cudaMalloc/Memset/Memcpy, cudaEvent*).
 * The last group is additive (no reference counterpart).
 *
 * Error convention: the reference's operators return void and check nothing
 * (SURVEY 8b).  Kept: every operator returns void, is asynchronous on the library's
 * compute stream, and latches the first failure (bad argument, HIP error, unsupported
 * operator) for fl_last_error().
 */
#ifndef BIMOCQ_GPU_H
#define BIMOCQ_GPU_H

#include <stddef.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* GPU_Advection.h:14-24 */
#define LEVEL_COUNT 6
struct SCoarseLevelInfo {
    int ni, nj, nk;
    int number;
    double alpha;
    double beta;
    double *b;
    double *x;
    double *r;
};
#ifndef __cplusplus
typedef struct SCoarseLevelInfo SCoarseLevelInfo;
#endif

/* ------------------------------------------------------------------------------------------
 * 1. Reference operators (same names, argument order and meaning)
 * ---------------------------------------------------------------------------------------- */

/* GPU_Advection.h:26-28 / GPU_kernel.cu:567-574.  In-place RK3 trace of the forward map,
 * sub-stepped by cfldt, interior nodes 2..n-3. */
void gpu_solve_forward(float *u, float *v, float *w,
                       float *x_fwd, float *y_fwd, float *z_fwd,
                       float h, int ni, int nj, int nk, float cfldt, float dt);

/* GPU_Advection.h:30-33 / GPU_kernel.cu:576-584.  One DMC backward sub-step in -> out.
 * out must not alias in; nodes outside 2..n-3 of out are left untouched. */
void gpu_solve_backwardDMC(float *u, float *v, float *w,
                           float *x_in, float *y_in, float *z_in,
                           float *x_out, float *y_out, float *z_out,
                           float h, int ni, int nj, int nk, float substep);

/* GPU_Advection.h:35-38 / GPU_kernel.cu:586-598.  Caller zeroes u,v,w first
 * (GPU_Advection.h:477-479): only the window 3+dim..nbuf-4 is written. */
void gpu_advect_velocity(float *u, float *v, float *w,
                         float *u_init, float *v_init, float *w_init,
                         float *backward_x, float *backward_y, float *backward_z,
                         float h, int ni, int nj, int nk, bool is_point);

/* GPU_Advection.h:40-44 / GPU_kernel.cu:600-618.  u = u*b + (1-b)*prev(psi_prev(psi_back)). */
void gpu_advect_vel_double(float *u, float *v, float *w,
                           float *utemp, float *vtemp, float *wtemp,
                           float *backward_x, float *backward_y, float *backward_z,
                           float *backward_xprev, float *backward_yprev, float *backward_zprev,
                           float h, int ni, int nj, int nk, bool is_point, float blend_coeff);

/* GPU_Advection.h:46-48 / GPU_kernel.cu:620-627 */
void gpu_advect_field(float *field, float *field_init,
                      float *backward_x, float *backward_y, float *backward_z,
                      float h, int ni, int nj, int nk, bool is_point);

/* GPU_Advection.h:50-53 / GPU_kernel.cu:629-638 */
void gpu_advect_field_double(float *field, float *field_init,
                             float *backward_x, float *backward_y, float *backward_z,
                             float *backward_xprev, float *backward_yprev, float *backward_zprev,
                             float h, int ni, int nj, int nk, bool is_point, float blend_coeff);

/* GPU_Advection.h:55-58 / GPU_kernel.cu:684-696.  d*_init += coeff * blend9(change(psi_fwd(x))). */
void gpu_accumulate_velocity(float *u_change, float *v_change, float *w_change,
                             float *du_init, float *dv_init, float *dw_init,
                             float *forward_x, float *forward_y, float *forward_z,
                             float h, int ni, int nj, int nk, bool is_point, float coeff);

/* GPU_Advection.h:60-62 / GPU_kernel.cu:698-705 */
void gpu_accumulate_field(float *field_change, float *dfield_init,
                          float *forward_x, float *forward_y, float *forward_z,
                          float h, int ni, int nj, int nk, bool is_point, float coeff);

/* GPU_Advection.h:64-67 / GPU_kernel.cu:707-716.  du (ni*nj*nk floats) receives the squared
 * round-trip distortion on nodes 2..n-3; caller zeroes it first (GPU_Advection.h:583). */
void gpu_estimate_distortion(float *du,
                             float *x_init, float *y_init, float *z_init,
                             float *x_fwd, float *y_fwd, float *z_fwd,
                             float h, int ni, int nj, int nk);

/* GPU_Advection.h:69 / GPU_kernel.cu:729-734.  field1 += coeff*field2 over exactly `number`
 * elements (the reference's grid overruns to the next multiple of 256; not replicated). */
void gpu_add(float *field1, float *field2, float coeff, int number);

/* GPU_Advection.h:71-76 / GPU_kernel.cu:640-666.  u,v,w: field in/out.  du,dv,dw: `init`, read,
 * then OVERWRITTEN with the uncompensated field.  u_src..: scratch, zeroed by the caller
 * (GPU_Advection.h:499-501). */
void gpu_compensate_velocity(float *u, float *v, float *w,
                             float *du, float *dv, float *dw,
                             float *u_src, float *v_src, float *w_src,
                             float *forward_x, float *forward_y, float *forward_z,
                             float *backward_x, float *backward_y, float *backward_z,
                             float h, int ni, int nj, int nk, bool is_point);

/* GPU_Advection.h:78-81 / GPU_kernel.cu:668-682.  All three buffers hold ni*nj*nk floats
 * (the reference's (ni+1)*nj*nk copy size is an overrun, not replicated). */
void gpu_compensate_field(float *u, float *du, float *u_src,
                          float *forward_x, float *forward_y, float *forward_z,
                          float *backward_x, float *backward_y, float *backward_z,
                          float h, int ni, int nj, int nk, bool is_point);

/* GPU_Advection.h:83-86 / GPU_kernel.cu:718-727 */
void gpu_semilag(float *field, float *field_src,
                 float *u, float *v, float *w,
                 int dim_x, int dim_y, int dim_z,
                 float h, int ni, int nj, int nk, float cfldt, float dt);

/* GPU_Advection.h:88-90 / GPU_kernel.cu:782-802.  ni,nj,nk are CELL dims. */
void gpu_emit_smoke(float *u, float *v, float *w, float *rho, float *T,
                    float h, int ni, int nj, int nk,
                    float centerX, float centerY, float centerZ, float radius,
                    float density, float temperature, float emiter);

/* GPU_Advection.h:92-93 / GPU_kernel.cu:825-832.  field = v component; ni,nj,nk CELL dims.
 * Implements the intended indexing (the reference mis-indexes rho/T for k>0):
 * v(i,j,k) += 0.5*dt*(beta*(T(j)+T(j-1)) - alpha*(rho(j)+rho(j-1))), 1 <= j <= nj-1. */
void gpu_add_buoyancy(float *field, float *density, float *temperature,
                      int ni, int nj, int nk, float alpha, float beta, float dt);

/* GPU_Advection.h:95 / GPU_kernel.cu:855-876.  ni,nj,nk are BUFFER dims.  fieldTemp0 receives a
 * copy of field and is the first sweep's input; on return field holds iterate iter-1. */
void gpu_diffuse_field(float *field, float *fieldTemp0, float *filedTemp1,
                       int ni, int nj, int nk, int iter, float coef);

/* GPU_Advection.h:97 / GPU_kernel.cu:885-890.  out = field1 + coeff*field2. */
void gpu_add_field(float *out, float *field1, float *field2, float coeff, int number);

/* GPU_Advection.h:99 / GPU_kernel.cu:1839-1895.  Caller zeroes div, p, p_temp
 * (GPU_Advection.h:604-606).  On return u,v,w are projected with iterate iter-1, which is
 * also what p holds (the reference's off-by-one, SURVEY Q1, kept).  debugParam may be NULL;
 * otherwise it needs >= 2000+iter floats and receives, for it = 0..iter-1,
 * debugParam[it] = sum r^2 and debugParam[2000+it] = max|r| of iterate `it`
 * (exact norms; the reference's buggy reductions are re-specified),
 * evaluated every fl_set_option(FL_OPT_RESIDUAL_STRIDE) iterates (0 = never).
 * A caller that does NOT clear p / p_temp (warm start) gets the reference's values too: sweeps ping-pong, odd
 * iterates carry p_temp's boundary shell, iter == 0 copies p_temp over p (:1876-1879).  Two sweeps share a launch only
 * where that cannot change a value: FL_OPT_JACOBI_FUSE = 1 (default) compares the two boundary shells first (one small
 * kernel and a 4-byte read-back per call), 2 skips the check on the caller's word, 0 never fuses.  p_temp is scratch:
 * its contents on return differ from the reference's (which performs one sweep more and discards it). */
void gpu_projection_jacobi(float *u, float *v, float *w, float *div, float *p, float *p_temp,
                           float *debugParam, int ni, int nj, int nk, int iter,
                           float halfrdx, float alpha, float beta);

/* GPU_Advection.h:101 / GPU_kernel.cu:892-950 -- MacCormack limiter of the reflection scheme, CORRECTED
 * (SURVEY 8f N3).  The reference kernel adds the stagger offset with the wrong sign, uses the departure
 * point's world coordinates as grid indices (:913-915) and tests/overwrites fieldTemp there from every
 * thread, so its output is undefined; this entry does what it evidently means:
 *   node x = (i - o) h, o = (ox, oy, oz) (0.5 along the staggered axis); x_d = x - dt u(x - dt/2 u(x)),
 *   clamped to [h, (n-1)h]; if fieldTemp at this node lies outside the min/max of the 8 values of `field`
 *   around x_d, it becomes the trilinear value of `field` at x_d.
 * ni, nj, nk: BUFFER dims (cells + dim).  field is read only; fieldTemp must not alias it. */
void gpu_clamp_extrema(float *field, float *fieldTemp, float *u, float *v, float *w,
                       int ni, int nj, int nk, int dimx, int dimy, int dimz,
                       float ox, float oy, float oz, float h, float dt);

/* GPU_Advection.h:103 / GPU_kernel.cu:959-964.  field = c1*field1 + c2*field2. */
void gpu_mad(float *field, float *field1, float *field2, float coeff1, float coeff2, int number);

/* GPU_Advection.h:105 -- OUT OF SCOPE (alternative solver, compiled out in the reference,
 * BimocqGPUSolver.cpp:419).  Latches FL_ERR_UNSUPPORTED. */
void gpu_conjugate_gradient(float *u, float *v, float *w, float *div, float *p,
                            float *residual, float *dir, float *dotR,
                            int ni, int nj, int nk, int iter, float halfrdx);

/* GPU_Advection.h:107-108 / GPU_kernel.cu:1764-1815 -- fp64 multigrid-corrected CG projection (what the
 * reference's shipped binary runs, BimocqGPUSolver.cpp:443-446: iter = 50, LEVEL_COUNT levels, halfrdx 0.5).
 * `levels` is a HOST array of levelNum entries holding DEVICE pointers (level l has dims (n-1)/2 of level
 * l-1, BimocqGPUSolver.cpp:68-90); div, p, dir, residual, temp0, temp1: levels[0].number doubles each;
 * tempResult: 4096 doubles -- [2i], [2i+1], [2i+2] the CG sums of outer iteration i, [2000+i] the largest
 * positive residual before iteration i.  p is cleared here; the caller owns and zero-allocates the rest
 * (boundary entries are never written).  Single GPU only: latches FL_ERR_UNSUPPORTED on a z-slab rank. */
void gpu_multi_grid_conjugate_gradient(float *u, float *v, float *w, double *div, double *p,
                                       double *dir, double *residual, double *temp0, double *temp1,
                                       double *tempResult, struct SCoarseLevelInfo *levels,
                                       int levelNum, int iter, double halfrdx);

/* ---- Limits of the operators above (part of the ABI contract; each violation latches an error, nothing is launched) ----
 *   * ONE field must stay below 2 GiB: 4 (ni+1)(nj+1)(nk+1) < 2^31 bytes.  The gather kernels (gpu_solve_*, gpu_advect_*,
 *     gpu_compensate_*, gpu_accumulate_*, gpu_semilag, gpu_clamp_extrema, gpu_estimate_distortion) address a field through
 *     a buffer resource descriptor with 32-bit byte offsets, and the offset 2 GiB is where a cell whose base index is
 *     negative is parked so that the descriptor's range check zeroes its corners (FL_ERR_BAD_ARGUMENT "field larger than
 *     2 GiB").  512^3 is 0.5 GiB; BASELINE config 5 (1024 x 1024 x 512 = 2.0 GiB per field) needs >= 2 z-slab ranks --
 *     bq_solver_create / gpuMapper refuse it on one rank with FL_ERR_UNSUPPORTED and name the rank count.
 *   * one plane must stay below 2^23 elements: (ni+1)(nj+1) < 8 388 608 (flat indices are formed with 24-bit multiplies).
 *   * nk + 1 <= 65 535: planes are launched along grid.z ("nk too large for grid.z").
 *   * the fused Jacobi kernels take rows of 32..1024 floats with ni % 4 == 0 and 16-byte aligned pointers; any other row
 *     runs the one-sweep kernels (same values).
 */

/* ------------------------------------------------------------------------------------------
 * 2. Runtime mini-ABI (replaces gpuMapper's direct CUDA runtime calls)
 * ---------------------------------------------------------------------------------------- */
enum {
    FL_OK = 0,
    FL_ERR_NO_DEVICE = 1,       /* no HIP device / wrong architecture                         */
    FL_ERR_HIP = 2,             /* a HIP runtime call failed (text in fl_last_error_string)   */
    FL_ERR_BAD_ARGUMENT = 3,
    FL_ERR_UNSUPPORTED = 4,
    FL_ERR_COMM = 5             /* RCCL failure                                               */
};

/* ---- contexts (round 3) ----
 * All state of this library -- device, the three streams, the option table, the error latch, the z-slab context and plane
 * window, the communicator, profiles, cached graphs -- belongs to a CONTEXT.  A program that never creates one uses the
 * default context and notices nothing.  fl_context_create(device) makes another (its own streams on `device`; NULL on
 * failure, the error latched on the caller's context), fl_context_make_current switches the calling THREAD to it (NULL: back
 * to the default context; thread-local, like hipSetDevice -- which it also calls), and every fl_* / gpu_* call of that thread
 * then acts on it.  One process can so drive several devices (SURVEY section 5's "single process, 8 devices": one context
 * and communicator per device, one thread each or one thread switching) or several solvers on one device side by side.
 * The host solver's C API (bimocq_solver.h) remembers the context a solver was created under and switches to it in every
 * call.  Buffers belong to the device, not to a context; a context must be current when its resources are used or freed. */
typedef struct fl_context fl_context;
fl_context *fl_context_create(int device);
void        fl_context_make_current(fl_context *ctx);
fl_context *fl_context_current(void);
void        fl_context_destroy(fl_context *ctx);       /* synchronises, destroys its communicator and streams */

/* Select device `device` (findCudaDevice, GPU_Advection.h:214-226), create the compute and
 * halo streams.  Returns FL_OK or an error code (the reference exit()s; we report). */
int   fl_init(int device);
void  fl_shutdown(void);
/* fl_shutdown for EVERY live context of the process (the default one and those of fl_context_create).  The first
 * successful fl_init registers it with atexit(), so a program that simply returns from main -- or a Python process that
 * ends -- releases the library's streams while the HIP runtime is still whole; callable directly, idempotent.  A
 * communicator that is still alive is abandoned, not destroyed (fl_comm_destroy is the orderly way). */
void  fl_shutdown_all(void);
/* cudaMalloc + cudaMemset(0) (allocGPUBuffer, GPU_Advection.h:322-326).  NULL on failure. */
void *fl_malloc(size_t bytes);
void  fl_free(void *p);
void  fl_memset(void *dst, int value, size_t bytes);                    /* async on the compute stream */
void  fl_memcpy_h2d(void *dst, const void *src, size_t bytes);          /* blocking */
void  fl_memcpy_d2h(void *dst, const void *src, size_t bytes);          /* blocking, after queued work */
void  fl_memcpy_d2d(void *dst, const void *src, size_t bytes);          /* async on the compute stream */
void  fl_sync(void);
/* An auxiliary compute stream for ONE operator that is independent of the ones that follow it (round 4): between
 * fl_aux_begin() and fl_aux_end() operators are launched on a second stream that first waits for everything queued on the
 * compute stream; after fl_aux_end() launches go to the compute stream again while the auxiliary work keeps running;
 * fl_aux_join() makes the compute stream wait for it.  The caller vouches for the independence.  The host solver runs the
 * forward-map update (gpu_solve_forward: reads the velocity, updates its own three arrays) beside the backward map's DMC
 * sub-steps this way on one GPU (BQ_OPT_CONCURRENT_MAPS). */
void  fl_aux_begin(void);
void  fl_aux_end(void);
void  fl_aux_join(void);
/* Asynchronous download for the dump path (replaces the blocking cudaMemcpy of copyDeviceToHost,
 * GPU_Advection.h:276-299): pinned host memory, a copy on the library's third stream ordered after the
 * compute work queued so far, and a ticket any host thread can wait on while the next step runs. */
void *fl_malloc_host(size_t bytes);
void  fl_free_host(void *p);
void *fl_download_begin(void *host_dst, const void *dev_src, size_t bytes);
int   fl_download_wait(void *ticket);
/* startEventRecord / endEventRecord (GPU_Advection.h:228-247) */
void *fl_event_create(void);
void  fl_event_record(void *ev);
float fl_event_elapsed_ms(void *start, void *stop);                     /* synchronises on `stop` */
void  fl_event_destroy(void *ev);
int         fl_last_error(void);
const char *fl_last_error_string(void);
void        fl_clear_error(void);
/* latch an error on behalf of host code that sits on top of this ABI (first error wins) */
void        fl_report_error(int code, const char *text);
/* the hipStream_t the operators launch on (for timing / interop with other runtimes) */
void *fl_compute_stream(void);

enum {
    FL_OPT_RESIDUAL_STRIDE = 1, /* evaluate Jacobi residual norms every k-th iterate (default 0)  */
    FL_OPT_SKIP_UNIT_BLEND = 2, /* gpu_advect_*_double with blend==1 (field*1 + 0*prev): 1 (default) = no launch, the
                                   values cannot change for finite prev; 2 = one-pass field+0 kernel (turns -0
                                   into +0 like the reference); 0 = the full two-level kernel               */
    FL_OPT_JACOBI_VARIANT  = 3, /* 0 = auto, 1 = generic scalar kernel, 2 = LDS-tiled kernel      */
    FL_OPT_PROFILE_JACOBI  = 4, /* record a hipEvent pair around each projection's sweep loop      */
    FL_OPT_JACOBI_KCHUNK   = 5, /* planes marched per block in the tiled kernel (0 = auto); for the fused kernels (which take
                                   their chunk length from FL_OPT_JACOBI_KCHUNK2): 1 / 2 = loads run one / two planes ahead */
    FL_OPT_JACOBI_ROWS     = 6, /* float4 rows per thread: tiled kernel 1, 2, 4; fused kernels 1, 2 (0 = auto); 4 = the three-sweep kernel
                                   that exchanges the intermediate levels' neighbour rows through LDS wherever it applies (auto: whole
                                   arrays with chunks of >= 24 planes), 5 = never that kernel (A/B timing) */
    FL_OPT_STRUCTURED_MAPS = 7, /* 9-point kernels: compile-time taps when h is a power of two (1)  */
    FL_OPT_JACOBI_FUSE     = 8, /* two or three sweeps per launch (4: at most two): 0 never, 1 in gpu_projection_jacobi after it has checked that p and
                                   p_temp carry the same boundary shell (default), 2 there without the check and also in
                                   gpu_jacobi_sweeps (caller vouches for equal boundary shells) */
    FL_OPT_JACOBI_KCHUNK2  = 9, /* planes marched per block in the fused kernel (0 = auto)           */
    FL_OPT_MGCG_GRAPH      = 10,/* 1 (default): the multigrid V-cycle is captured into a hipGraph once and
                                 * replayed in every outer iteration; 0: plain launches                */
    FL_OPT_FUSED_HOUSEKEEPING = 12, /* bit mask, default 0 (the reference's operator semantics).  Lets a caller drop the clears
                                 * and copies it issues around the map operators; what each bit makes the kernels do:
                                 *   1: gpu_advect_velocity/_field/_field2 and gpu_compensate_error_* write zeros outside their
                                 *      index window -- the outputs need not be cleared first (GPU_Advection.h:472-526);
                                 *   2: gpu_compensate_error_* store the uncompensated field into `init` (du/dv/dw) after
                                 *      reading it -- stage 2 of gpu_compensate_* (GPU_kernel.cu:656-658) needs no copy;
                                 *   4: gpu_solve_backwardDMC writes zeros to the border nodes of x/y/z_out (what the
                                 *      reference's cleared scratch holds there, GPU_Advection.h:464-468);
                                 *   8: gpu_solve_backwardDMC copies the border nodes of x/y/z_in to x/y/z_out instead.  */
    FL_OPT_FAST_LERP       = 11,/* 0 (default): the reference's double-evaluated lerp, results bit-identical to the
                                 * oracle.  1: every lerp of the gather kernels is one fp32 fma, fmaf(c, b-a, a) -- NOT the
                                 * reference arithmetic; deviation measured per grid size   */
    FL_OPT_MGCG_TILE       = 14,/* multigrid V-cycle, levels below 2^19 cells (63^3 and coarser; 2: below 2^21): 1 (default) smooths them with
                                 * the LDS tile kernel -- 4 (the 32-sweep calls) or 2 (the 4-sweep calls) sweeps per launch on a
                                 * 16^3 region per workgroup, and the clears of x / temp0 folded into the first two launches;
                                 * 0: one launch per sweep (mg_smooth_kernel).  Same values either way.          */
    FL_OPT_PROFILE_COMM    = 15,/* 1: every wait of the compute stream on the halo stream (fl_halo_exchange with wait != 0, fl_halo_wait,
                                 * fl_p2p_exchange) is bracketed by two timing events, and every in-stream all-reduce too; the
                                 * sums are read with fl_comm_profile().  Default 0.                                 */
    FL_OPT_RESERVE_CUS     = 16,/* k > 0: the compute stream is (re)created with a CU mask that leaves k compute units (k / 8 per
                                 * XCD) to the halo stream, so that RCCL's send/recv kernels start the moment an exchange is
                                 * issued instead of waiting for compute workgroups to drain; the Jacobi launchers size their
                                 * grids for the remaining CUs.  Setting it synchronises.  Default 0 (all CUs).       */
    FL_OPT_MGCG_BOTTOM     = 17,/* 1 (default): the two coarsest levels of the multigrid V-cycle -- when both fit one workgroup's
                                 * LDS (4096 and 512 cells: 15^3 and 7^3 of a 256^3 pyramid) -- run as ONE launch
                                 * (mg_vbottom_kernel: 32 sweeps, residual, restriction, 32 sweeps, prolongation, 4 sweeps)
                                 * instead of ~22, and the last launch of level 0's 32 sweeps also writes the residual that
                                 * follows.  0: one launch per operator.  Same values either way.                   */
    FL_OPT_FIELD_WINDOW    = 18,/* nine-point operators (gpu_advect_*, gpu_compensate_*, gpu_accumulate_*) on power-of-two
                                 * spacing: k > 0 runs them as z-marching blocks that read the sampled field out of a rolling
                                 * LDS window (bq_gather_march.hip.h) instead of gathering every corner from memory; a tap
                                 * outside the window takes the direct path, so values never change.  k = planes marched per
                                 * block (1 = chosen by the launcher).  0: the one-plane kernels.  Negative (default): on
                                 * with FL_OPT_FAST_LERP, whose kernels it speeds up, off in the exact arithmetic, which is
                                 * bound by its double-rounded lerps either way.  BQ_FIELD_WINDOW in the environment sets
                                 * the initial value.                                                                 */
    FL_OPT_COMM_CHECK      = 19,/* 1: every RCCL call of the z-slab path (ncclSend / ncclRecv of the exchanges, the scalar
                                 * all-reduces) is entered into a per-rank ledger that fl_comm_check() compares across the
                                 * ranks; the host solver calls it at the end of every step while the option is on.  A
                                 * debugging aid for the first runs on real links: a mismatch latches FL_ERR_COMM instead of
                                 * hanging or silently pairing the wrong messages.  Default 0.                        */
    FL_OPT_MGCG_FUSE       = 20,/* gpu_multi_grid_conjugate_gradient on rows of 256 or 512 cells: 1 runs the level-0 vector updates of
                                 * an outer iteration inside the stencil pass that follows each of them (update_x + residual;
                                 * add + residual + max + dot; update_dir + A dir + dot -- three launches for nine passes over
                                 * the arrays, bq_mgcg_fused.hip.inc) on grids of 2^20 cells and more, 2 on any grid of that row
                                 * length, 0 never (3: as 2 with the wave-per-row form of the kernels on rows of 256 -- the same
                                 * time, kept for A/B).  Same values.  The fused form keeps its intermediate vectors in temp1
                                 * and levels[0].b, i.e. it needs BOTH to be full-size arrays (ni nj nk doubles) -- as the
                                 * reference's caller allocates them (BimocqGPUSolver.cpp:65-66) but more than the reference's
                                 * kernels themselves touch of temp1; what temp0 / temp1 hold after the call differs (both are
                                 * scratch).  Hence the default is -1 = off until a caller says so: setting 1 is the caller's
                                 * word that its arrays have that size.  The host solver of this package (csrc/host), which
                                 * allocates them, sets 1 unless the option has been set before.                      */
    FL_OPT_MAP_QUARTER_FP32 = 13 /* 0 (default): every lerp of the structured map look-up follows the double-rounding
                                 * contract.  1: the caller vouches that every value of the map arrays it passes to the
                                 * 9-point operators is 0 or lies in [h/256, 1024 h] (gpu_maps_quarter_safe checks a map
                                 * set); the weight-1/4 lerps of the look-up then run as one fp32 fma, which is provably
                                 * the same value under that bound (bq_device.hip.h: lerp_q) -- same results, ~15 % fewer
                                 * issue cycles per launch.  The host solver checks its maps after every update.     */
};
void fl_set_option(int option, int value);
int  fl_get_option(int option);
/* FL_OPT_PROFILE_JACOBI: total milliseconds, sweep-kernel launches and Jacobi sweeps (a fused launch
 * performs two) of the sweep loops recorded since the previous call (blocking; resets the record) */
void fl_jacobi_profile(double *total_ms, long long *launches, long long *sweeps);
/* name of the two-sweep kernel the projection launched last ("" before the first; for reports) */
const char *fl_jacobi_kernel_name(void);
/* name of the fused kernel the last fp64 smoothing call (gpu_smoothing_jacobi, V_Cycle) launched first ("" if none; for
 * reports and tests) */
const char *fl_mg_smooth_kernel_name(void);
/* launches of the fused level-0 kernels (FL_OPT_MGCG_FUSE) since the previous call (resets the count; for reports and tests) */
long long fl_mg_fused_launches(void);

/* ------------------------------------------------------------------------------------------
 * 3. Additive entry points (no reference counterpart)
 * ---------------------------------------------------------------------------------------- */
/* precondition check of FL_OPT_MAP_QUARTER_FP32: 1 when every value of x, y, z is 0 or in [h/256, 1024 h]; blocking;
 * z-slab ranks get one common answer */
/* 1 when a gpu_max_abs3 since the last reset met a NaN or an Inf in the velocity (on any slab rank; reset != 0 clears).  The
 * CFL maximum skips NaNs like the reference's host scan (BimocqGPUSolver.cpp:348-373), so a run that has gone NaN -- e.g. a
 * source whose axis passes through grid nodes, SURVEY Q14 -- would otherwise look perfectly calm to its driver. */
int  fl_nonfinite_seen(int reset);
int  gpu_maps_quarter_safe(const float *x, const float *y, const float *z, float h, int ni, int nj, int nk);
/* the same check without a pass over the maps: while armed, gpu_solve_backwardDMC (word 0) and gpu_solve_forward (word 1)
 * flag every value they store that fails the test.  _reset(which) arms the guard and clears a word (which < 0: off);
 * _read fills ok[0], ok[1] (1 = nothing flagged since the reset); blocking, z-slab ranks get common answers */
void fl_map_guard_reset(int which);
void fl_map_guard_read(int ok[2]);
/* maps <- (i*h, j*h, k*h): the host loop + H2D of MapperBaseGPU::init (Mapping.cpp:306-328) */
void gpu_init_maps(float *x, float *y, float *z, float h, int ni, int nj, int nk);
/* device getCFL (BimocqGPUSolver.cpp:348-373): max(1e-4, max|u|,|v|,|w|); blocking */
float gpu_max_abs3(const float *u, const float *v, const float *w, int ni, int nj, int nk);
/* the three stages of gpu_projection_jacobi, separately launchable */
void gpu_divergence(const float *u, const float *v, const float *w, float *div,
                    int ni, int nj, int nk, float halfrdx);
/* `sweeps` Jacobi sweeps ping-ponging p <-> p_temp starting from p; returns 0 if the newest
 * iterate ends in p, 1 if it ends in p_temp.  Boundary cells are never written. */
int  gpu_jacobi_sweeps(float *p, const float *div, float *p_temp,
                       int ni, int nj, int nk, int sweeps, float alpha, float beta);
void gpu_gradient(float *u, float *v, float *w, const float *p,
                  int ni, int nj, int nk, float halfrdx);
/* gpu_gradient that also returns what it changed: (du, dv, dw) = new - old on the update window, 0 elsewhere --
 * the d*Proj = U - UTemp of BimocqGPUSolver.cpp:188-193 without the snapshot copies and the subtraction passes */
void gpu_gradient_delta(float *u, float *v, float *w, const float *p, float *du, float *dv, float *dw,
                        int ni, int nj, int nk, float halfrdx);
/* one Jacobi sweep in -> out over the local planes [k_begin, k_end) only: lets a z-slab host sweep the
 * planes that do not depend on ghost planes while those are still being exchanged */
void gpu_jacobi_sweep_range(const float *in, const float *div, float *out, int ni, int nj, int nk,
                            int k_begin, int k_end, float alpha, float beta);
/* TWO Jacobi sweeps in -> out in one launch, `out` written on the local planes [k0a, k1a) and [k0b, k1b) only (either
 * range may be empty): a z-slab host issues the first two sweeps after an exchange as the planes whose two-sweep stencil
 * stays inside the owned planes (while the ghost planes are in flight) and then the rest.  Both buffers must carry the
 * same boundary layer (as for FL_OPT_JACOBI_FUSE = 2).  Returns 1 when the fused kernel ran, 0 when it does not apply
 * to this grid -- nothing was launched and the caller falls back to gpu_jacobi_sweep_range. */
int gpu_jacobi_sweep_pair_ranges(const float *in, const float *div, float *out, int ni, int nj, int nk,
                                 int k0a, int k1a, int k0b, int k1b, float alpha, float beta);
/* THREE sweeps in -> out in one launch on the local planes [k0a, k1a) and [k0b, k1b) (either may be empty), through the
 * LDS-exchanged kernels (rows of 32 .. 512 floats): the pieces of a z-slab rank's pressure chunk.  `in` must be valid three
 * planes beyond each range; same precondition as above.  Returns 1 when it ran, 0 when it does not apply (nothing launched). */
int gpu_jacobi_sweep_triple_ranges(const float *in, const float *div, float *out, int ni, int nj, int nk,
                                   int k0a, int k1a, int k0b, int k1b, float alpha, float beta);
/* exact sum r^2 (double) and max|r| of r = div - (sum6 p - 6p) over interior cells; blocking */
void gpu_residual_norms(const float *div, const float *p, int ni, int nj, int nk,
                        double *sum_sq, float *max_abs);
/* stage 1 of gpu_compensate_velocity / _field on its own (compensate_kernel, GPU_kernel.cu:438-499,
 * launches :652-654 / :676): u_src = blend9(u(psi_fwd(x))) - du.  The full operator is then
 * [this] ; du <- u ; gpu_accumulate_*(u_src -> u, backward map, -0.5) ; gpu_clamp_extrema_box(du, u),
 * which lets a z-slab host refresh ghost planes between the stages. */
void gpu_compensate_error_velocity(float *u, float *v, float *w, float *du, float *dv, float *dw,
                                   float *u_src, float *v_src, float *w_src,
                                   float *forward_x, float *forward_y, float *forward_z,
                                   float h, int ni, int nj, int nk, bool is_point);
void gpu_compensate_error_field(float *u, float *du, float *u_src,
                                float *forward_x, float *forward_y, float *forward_z,
                                float h, int ni, int nj, int nk, bool is_point);
/* Batched forms: fields that live on the same nodes share ONE map look-up (the look-up is ~70 % of a
 * gather kernel's arithmetic).  Each is, result for result, the two single calls it names, in order.
 *   gpu_advect_field2            = gpu_advect_field(field1..) ; gpu_advect_field(field2..)
 *   gpu_compensate_error_field2  = gpu_compensate_error_field(u1..) ; (u2..)
 *   gpu_accumulate_field2        = gpu_accumulate_field(change1, dinit1, coeff1) ; (change2, dinit2, coeff2)
 *   gpu_accumulate_velocity2     = gpu_accumulate_velocity(change1 -> d*_init, coeff1) ; (change2 -> d*_init, coeff2)
 * (BimocqGPUSolver.cpp:144-145 advects rho and T back to back; :213-214 accumulates the force and the
 * projection change back to back.) */
void gpu_advect_field2(float *field1, float *field1_init, float *field2, float *field2_init,
                       float *backward_x, float *backward_y, float *backward_z,
                       float h, int ni, int nj, int nk, bool is_point);
void gpu_compensate_error_field2(float *u1, float *du1, float *u1_src, float *u2, float *du2, float *u2_src,
                                 float *forward_x, float *forward_y, float *forward_z,
                                 float h, int ni, int nj, int nk, bool is_point);
void gpu_accumulate_field2(float *change1, float *dinit1, float coeff1, float *change2, float *dinit2, float coeff2,
                           float *forward_x, float *forward_y, float *forward_z,
                           float h, int ni, int nj, int nk, bool is_point);
void gpu_accumulate_velocity2(float *u_change1, float *v_change1, float *w_change1, float coeff1,
                              float *u_change2, float *v_change2, float *w_change2, float coeff2,
                              float *du_init, float *dv_init, float *dw_init,
                              float *forward_x, float *forward_y, float *forward_z,
                              float h, int ni, int nj, int nk, bool is_point);
/* one component (axis 0/1/2 = u/v/w) of gpu_accumulate_velocity with one or two sources (change2 may be NULL):
 * d_init += blend9(coeff1*change1(psi(x))) [then += blend9(coeff2*change2(psi(x)))].  For hosts that know a
 * component's source to be identically zero (no force acts on u and w when only buoyancy is on). */
void gpu_accumulate_component(float *change1, float coeff1, float *change2, float coeff2, float *d_init,
                              float *forward_x, float *forward_y, float *forward_z,
                              float h, int ni, int nj, int nk, int axis, bool is_point);
/* gpu_accumulate_velocity when the caller KNOWS the forward map is the identity map gpu_init_maps wrote
 * (BimocqGPUSolver.cpp:222-223: accumulate right after reinitializeMapping): with power-of-two spacing
 * the mapped positions follow from the node indices alone, by the same lerps, and the map is not read.
 * Same results as gpu_accumulate_velocity on that map. */
void gpu_accumulate_velocity_identity(float *u_change, float *v_change, float *w_change,
                                      float *du_init, float *dv_init, float *dw_init,
                                      float *forward_x, float *forward_y, float *forward_z,
                                      float h, int ni, int nj, int nk, bool is_point, float coeff);
/* the sweeps of gpu_diffuse_field without its two copies: `sweeps` times out = (field + coef*sum6(in))/(1+6coef)
 * on interior cells, ping-ponging in <-> out; returns 0 if the newest iterate ends in `in`, 1 if in `out`.
 * Lets a z-slab host refresh ghost planes between chunks of sweeps. */
int gpu_diffuse_sweeps(const float *field, float *in, float *out, int ni, int nj, int nk, int sweeps, float coef);
/* smoothing_jacobi<double> (GPU_kernel.cu:1464-1483) on its own: `iter` (rounded up to even) sweeps of
 * x' = ((sum6 x) + alpha*b) * beta on interior cells, ping-ponging x <-> temp; the newest iterate ends in x.
 * x and temp must carry the same boundary layer (V_Cycle clears both): sweeps are fused pairwise. */
/* gpu_multi_grid_conjugate_gradient on a z-slab rank with the grid's fine levels SHARED between the ranks (round 4;
 * csrc/bq_mgcg_slab.hip.inc): every level the ranks share is stored as owned planes + ghost planes, the single-GPU launchers run
 * on those plane ranges with the ghost planes computed redundantly and refreshed by the neighbour exchange (fl_halo_exchange),
 * the float-narrowed block dot products are computed per rank and all-gathered for the reference's final sum, thin levels are
 * gathered and solved replicated.  Bit-identical to the single-domain solver.
 *   u, v, w      the rank's LOCAL velocity buffers, planes [own0 - ghost, own1 + ghost) (w one more), ghost planes correct
 *   tempResult   device, 4096 doubles: the reference's residual history (identical on every rank)
 * Collective (every rank of the communicator calls it with its own planes).  gpu_mgcg_slab_supported says whether a
 * decomposition can run this way: planes of a multiple of 256 cells, equal slabs that start at a multiple of 8 and hold at least
 * 17 planes, ghost >= 8; the host solver keeps the replicated solve otherwise.  On return the velocity is projected on the owned
 * planes and on 7 ghost planes of either side. */
int  gpu_mgcg_slab_supported(int ni, int nj, int nkg, int own0, int own1, int ghost, int rank, int nranks);
void gpu_multi_grid_conjugate_gradient_slab(float *u, float *v, float *w, double *tempResult,
                                            int ni, int nj, int nkg, int own0, int own1, int ghost, int iter, double halfrdx);
void gpu_smoothing_jacobi(double *x, double *b, double *temp, double alpha, double beta,
                          int ni, int nj, int nk, int iter);
/* max(0, max of field[0..count)) with NaNs skipped -- the host scan of MapperBaseGPU::estimateDistortion
 * (Mapping.cpp:500-516) over the buffer gpu_estimate_distortion filled, as a device reduction; blocking.
 * Intended for non-negative data (it reduces |x|).  Not slab-aware. */
float gpu_max_field(const float *field, size_t count);
/* the same over the planes this rank OWNS of a scalar-sized field of nk local planes, all-reduced over the slab ranks
 * (single GPU: the whole field): estimateDistortion on z-slab ranks */
float gpu_max_field_owned(const float *field, int ni, int nj, int nk);
/* out[0] / out[1]: how many cells along z the backward / forward map (their z components bz, fz) carries a node at most --
 * max |map_z - z| / h over the nodes the map updates write (2 <= i, j, k <= n - 3), owned planes, all-reduced over the slab
 * ranks; a NaN counts as infinity.  Tells a z-slab host whether maps that live for more than one step still fit its ghost
 * zone (csrc/host/fluid_solver.cpp: BQ_OPT_REINIT_MAX_TRAVEL).  Blocking. */
void  gpu_map_travel_z(const float *bz, const float *fz, float h, int ni, int nj, int nk, float out[2]);
/* clamp_extrema_box for a staggered buffer: dz = 1 for the w component (nk+1 planes) */
void gpu_clamp_extrema_box_w(const float *before, float *after, int ni, int nj, int nk_buffer);
/* clampExtrema_kernel (GPU_kernel.cu:146-167) on its own: after = clamp(after, min/max27(before)) */
void gpu_clamp_extrema_box(const float *before, float *after, int ni, int nj, int nk);

/* ------------------------------------------------------------------------------------------
 * 4. Multi-GPU: z-slab context and ghost-plane exchange (one process per GPU, RCCL over xGMI)
 * ---------------------------------------------------------------------------------------- */
/* Tell the operators that the buffers they are given hold the global cell planes
 * [koff, koff + nk_local) of a grid with nk_global planes, of which this rank owns [own0, own1).
 * Index windows, positions and clamps are then evaluated in global coordinates; reductions
 * (gpu_max_abs3, residual norms) count owned planes and are all-reduced.  nk_global <= 0 resets. */
void fl_set_slab(int koff, int nk_global, int own0, int own1, int nk_local);
/* Restrict the map operators (gpu_solve_forward, gpu_solve_backwardDMC, gpu_advect_velocity/_field/_field2,
 * gpu_compensate_error_*, gpu_accumulate_*) to the local cell planes [k0, k1): they produce exactly the nodes of those
 * planes (the w component's extra plane belongs to the window that reaches the last cell plane).  k0 < 0 switches it
 * off.  A z-slab host runs an operator on the planes that need no ghost data while the ghost planes are in flight
 * (fl_halo_exchange with wait = 0), then on the rest after fl_halo_wait.  Returns 1 if supported, 0 if not. */
int fl_set_plane_window(int k0, int k1);
/* rank 0 obtains a 128-byte ncclUniqueId, the host program distributes it, every rank calls init */
int  fl_comm_unique_id(void *id128);
int  fl_comm_init(const void *id128, int rank, int nranks);
void fl_comm_destroy(void);
/* runs every RCCL call of this library on a temporary one-rank communicator (single-GPU check of the
 * dlopen'ed binding); FL_OK or an error code with fl_last_error_string() set */
int  fl_comm_selftest(void);
/* How many RCCL communicators the current context holds: 2 = the ghost-plane exchanges (halo stream) and the in-stream
 * scalar all-reduces (compute stream) each have their own, so that neither waits for the other inside RCCL's per-communicator
 * launch order; 1 = one serves both (a RCCL without ncclCommSplit, or BQ_SINGLE_COMM=1 in the environment); 0 = none. */
int  fl_comm_count(void);
/* FL_OPT_COMM_CHECK: compare the ranks' ledgers of communicator calls since the last check (collective: every rank calls
 * it at the same point; two 8-to-24-byte all-reduces + one read-back).  FL_OK, or FL_ERR_COMM (latched) when a send has no
 * receive of the same size at the same position of its pair's sequence, or the all-reduce sequences differ.  perturb != 0
 * falsifies this rank's ledger first (tests). */
int  fl_comm_check(int perturb);
int  fl_comm_rank(void);
int  fl_comm_size(void);
/* refresh `depth` (<= G) ghost planes per side of n fields with the z-neighbours, one RCCL group on
 * the halo stream.  extra[f] = 1 for a w-type buffer (nk+1 planes).  wait != 0: the compute stream
 * waits for the exchange; else call fl_halo_wait() before touching the exchanged planes. */
void fl_halo_exchange(int n, float *const *fields, const size_t *plane_elems, const int *extra,
                      int nk_local, int G, int depth, int wait);
void fl_halo_wait(void);
/* traffic of this rank since the last reset: out[0] ghost-plane exchanges issued, out[1] bytes sent in them, out[2] point-to-point
 * message groups (wall sheets), out[3] bytes sent in them (counted also when the transport is the null or a host-side one) */
void fl_comm_stats(long long out[4], int reset);
/* FL_OPT_PROFILE_COMM: ms[0] / n[0] = milliseconds the compute stream spent blocked on the halo stream (communication that
 * no kernel hid; with a host-side transport: the wall time of its blocking calls) and the number of such waits,
 * ms[1] / n[1] = the in-stream scalar all-reduces (CFL maximum, map guard, norms), since the last reset.  Blocking. */
void fl_comm_profile(double ms[2], long long n[2], int reset);
/* ncclGetVersion() of the RCCL this process loaded (e.g. 22105), 0 when none is loaded */
int  fl_comm_rccl_version(void);
/* Host-side transport hook: replaces RCCL by two callbacks (blocking, called with the compute stream
 * idle).  `exchange` receives fl_halo_exchange's arguments and must move the planes itself
 * (fl_memcpy_d2h / its own wire / fl_memcpy_h2d; plane ranges as documented in bq_halo.hip);
 * `allreduce` reduces `count` host values in place.  Used to run slab ranks over gloo / shared
 * memory, e.g. several ranks on one GPU in the tests. */
typedef void (*fl_exchange_cb)(int n, float *const *fields, const size_t *plane_elems, const int *extra,
                               int nk_local, int G, int depth);
typedef void (*fl_allreduce_cb)(void *host_values, int count, int is_double, int is_max);
void fl_comm_set_custom(int rank, int nranks, fl_exchange_cb exchange, fl_allreduce_cb allreduce);
/* timing aid: play rank `rank` of `nranks` with a transport that moves nothing and never waits (the compute-side cost
 * of the z-slab path on one GPU; results near the slab boundary are meaningless) */
void fl_comm_set_null(int rank, int nranks);

/* ---- wall sheets: what makes z-slab ranks reproduce ONE GPU bit for bit in the reference-faithful mode -------------
 * The reference zeroes the border nodes of the backward map in every DMC update (GPU_Advection.h:464-468, the protecting
 * pre-copy is commented out at :335-337).  Stage 3 of gpu_compensate_* (cumulate_kernel with the backward map,
 * GPU_kernel.cu:659-661) interpolates towards those zeros on the first and last node layer of its index window: such a
 * tap lands at 1/4, 1/2 or 3/4 (or a product of those) of its position -- near the wall on one GPU, arbitrarily far
 * along z for a slab rank.  The cells those taps can touch form a few thin sheets; a slab rank collects them from the
 * ranks that own them into a copy of the sampled field that spans the needed global planes, and re-evaluates the wall
 * layers from that copy.  Pieces (host computes WHICH boxes; csrc/host/wall_sheets.*): */
typedef struct fl_box { int x0, x1, y0, y1, z0, z1; } fl_box;   /* half-open, GLOBAL indices of the field's buffer */
/* packed <- the boxes of `field`, one after the other, x fastest.  `field` holds the global planes [koff, koff + nk_field)
 * of a buffer with rows of nbi and planes of nbi*nbj floats; every box must lie inside them.  `boxes` is a HOST array. */
void fl_box_pack(const float *field, int nbi, int nbj, int nk_field, int koff, const fl_box *boxes, int nboxes, float *packed);
/* the inverse, into `field`; packed == NULL fills the boxes with NaN (a cell the plan missed must not pass for data) */
void fl_box_unpack(float *field, int nbi, int nbj, int nk_field, int koff, const fl_box *boxes, int nboxes, const float *packed);
/* the boxes straight from one field to another with the same rows that holds other global planes: dst(cell) <- src(cell) */
void fl_box_copy(const float *src, int nbi, int nbj, int nk_src, int koff_src, float *dst, int nk_dst, int koff_dst,
                 const fl_box *boxes, int nboxes);
/* n point-to-point messages in one RCCL group on the halo stream: send[m] (send_count[m] floats) goes to rank peers[m],
 * recv[m] (recv_count[m] floats) comes from it; a count may be 0.  Ordered after the compute work queued so far; the
 * compute stream waits for the transfers.  Peers need not be z-neighbours (xGMI is a full mesh: one link per pair). */
void fl_p2p_exchange(int n, const int *peers, float *const *send, const size_t *send_count,
                     float *const *recv, const size_t *recv_count);
/* the same without the compute stream waiting: the messages travel while the compute stream goes on; fl_halo_wait()
 * before the received data are read or the send buffers rewritten (a later fl_halo_exchange / fl_halo_wait pair covers it
 * too: the halo stream runs its exchanges in order).  A host-side transport completes inside the call. */
void fl_p2p_exchange_begin(int n, const int *peers, float *const *send, const size_t *send_count,
                           float *const *recv, const size_t *recv_count);
/* host-side transport for fl_p2p_exchange (see fl_comm_set_custom; called with the compute stream idle) */
typedef void (*fl_p2p_cb)(int n, const int *peers, float *const *send, const size_t *send_count,
                          float *const *recv, const size_t *recv_count);
void fl_comm_set_custom_p2p(fl_p2p_cb p2p);
/* cumulate_kernel's expression (GPU_kernel.cu:376-436) on the nodes of its index window with i in xlist, j in ylist or
 * GLOBAL plane in zlist (host arrays, at most 8 entries each), the source read from `src`, which holds the global
 * planes [src_koff, src_koff + src_nk) of the sampled field:  dst = before + blend9(coeff * src(map(x))), `before`
 * being dst's value ahead of the stage (gpu_compensate_*'s stage-2 copy).  axis: -1 scalar, 0/1/2 = u/v/w buffers.
 * The maps and before/dst are local slab buffers (slab context, plane window honoured). */
void gpu_accumulate_wall_fixup(const float *src, int src_koff, int src_nk, const float *before, float *dst,
                               const float *mx, const float *my, const float *mz,
                               float h, int ni, int nj, int nk, int axis, float coeff,
                               const int *xlist, int nxl, const int *ylist, int nyl, const int *zlist, int nzl);
/* gpu_advect_vel_double / gpu_advect_field_double (GPU_kernel.cu:236-310, 600-638) for a z-slab rank with blend_coeff != 1 in the
 * reference-faithful mode (zeroed map border).  The kernel's second look-up reads the PREVIOUS backward map at the
 * position the current one returns; where that lands in a cell with zeroed border nodes (more than 3/4 of a cell towards a wall
 * from the outermost nodes of the window) all three components come back scaled by the live nodes' weight s in [0, 1], and the
 * *_prev field is sampled at s * q -- anywhere on the segment from the origin to the node, on any rank's planes, with a
 * data-dependent s.  No sheet bounds that: these entry points take the *_prev fields of the WHOLE grid (nk_global planes, + 1
 * for w; plane 0 = global plane 0), which the host solver assembles once per re-initialisation -- the only time they change
 * (BimocqGPUSolver.cpp:503-527).  Everything else (fields, maps, windows, clamps) as in the local forms; global arrays below
 * 2 GiB.  Ref: Mapping.cpp:383-390. */
void gpu_advect_vel_double_global(float *u, float *v, float *w,
                                  float *uprev_global, float *vprev_global, float *wprev_global,
                                  float *backward_x, float *backward_y, float *backward_z,
                                  float *backward_xprev, float *backward_yprev, float *backward_zprev,
                                  float h, int ni, int nj, int nk, bool is_point, float blend_coeff);
void gpu_advect_field_double_global(float *field, float *field_prev_global,
                                    float *backward_x, float *backward_y, float *backward_z,
                                    float *backward_xprev, float *backward_yprev, float *backward_zprev,
                                    float h, int ni, int nj, int nk, bool is_point, float blend_coeff);

#ifdef __cplusplus
}
#endif
#endif /* BIMOCQ_GPU_H */
