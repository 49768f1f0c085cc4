/*
 * oracle_abi.c -- TEST-ONLY stand-in of the C-ABI (include/bimocq_gpu.h) on top of the CPU oracle.
 *
 * Lives under tests/ and is linked only into tests/_build/libbimocq_host_cpu.so, which the
 * `-m "not gpu"` tests use to exercise the C++ HOST code (csrc/host/: step state machine, buffer
 * swaps, shared map sets, dump writer) without a GPU.  "Device" memory is host memory here.
 * The product never links this file: libbimocq_host.so binds to libbimocq_hip.so only.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/bimocq_gpu.h"
#include "../../oracle/bimocq_oracle.h"
#include <math.h>

static int g_err = FL_OK;
static char g_text[256];
static int g_opt[8];

static void latch(int code, const char *what)
{
    if (g_err != FL_OK) return;
    g_err = code;
    snprintf(g_text, sizeof g_text, "%s", what);
}

int fl_init(int device) { (void)device; return FL_OK; }
void fl_shutdown(void) {}
void *fl_malloc(size_t bytes) { return calloc(bytes ? bytes : 4, 1); }
void fl_free(void *p) { free(p); }
void fl_memset(void *dst, int value, size_t bytes) { memset(dst, value, bytes); }
void fl_memcpy_h2d(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); }
void fl_memcpy_d2h(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); }
void fl_memcpy_d2d(void *dst, const void *src, size_t bytes) { if (dst != src) memmove(dst, src, bytes); }
void fl_sync(void) {}
void *fl_event_create(void) { return calloc(1, sizeof(struct timespec)); }
void fl_event_record(void *ev) { if (ev) clock_gettime(CLOCK_MONOTONIC, (struct timespec *)ev); }
float fl_event_elapsed_ms(void *a, void *b)
{
    struct timespec *s = (struct timespec *)a, *e = (struct timespec *)b;
    if (!s || !e) return -1.f;
    return (float)((e->tv_sec - s->tv_sec) * 1e3 + (e->tv_nsec - s->tv_nsec) * 1e-6);
}
void fl_event_destroy(void *ev) { free(ev); }
int fl_last_error(void) { return g_err; }
const char *fl_last_error_string(void) { return g_err == FL_OK ? "" : g_text; }
void fl_clear_error(void) { g_err = FL_OK; g_text[0] = 0; }
void *fl_compute_stream(void) { return NULL; }
/* options 0..7 are stored; FL_OPT_FUSED_HOUSEKEEPING (12) is implemented below; everything else is unknown (-1) */
static int g_fused = 0;
static int g_nk_local = 0;          /* local cell planes, from fl_set_slab */
void fl_set_option(int option, int value)
{
    if (option >= 0 && option < 8) g_opt[option] = value;
    if (option == FL_OPT_FUSED_HOUSEKEEPING) g_fused = value & 15;
}
int fl_get_option(int option)
{
    if (option == FL_OPT_FUSED_HOUSEKEEPING) return g_fused;
    return (option >= 0 && option < 8) ? g_opt[option] : -1;
}
/* plane window of the map operators (fl_set_plane_window): handed to the oracle's loops; the clears and copies of the
 * fused housekeeping below are restricted to the same planes */
static int g_win_on = 0, g_win_k0 = 0, g_win_k1 = 0, g_win_cells = 0;
static void win_planes(int nk_cells, int nbk, int *p0, int *p1)
{
    *p0 = 0; *p1 = nbk;
    if (g_win_on) { *p0 = g_win_k0 < nbk ? g_win_k0 : nbk; *p1 = g_win_k1 >= nk_cells ? nbk : (g_win_k1 < *p0 ? *p0 : g_win_k1); }
}
static void win_zero(float *f, size_t plane, int nk_cells, int nbk)
{ int a, b; win_planes(nk_cells, nbk, &a, &b); if (b > a) memset(f + plane * (size_t)a, 0, plane * (size_t)(b - a) * sizeof(float)); }
static void win_copy(float *dst, const float *src, size_t plane, int nk_cells, int nbk)
{ int a, b; win_planes(nk_cells, nbk, &a, &b); if (b > a) memcpy(dst + plane * (size_t)a, src + plane * (size_t)a, plane * (size_t)(b - a) * sizeof(float)); }
void fl_jacobi_profile(double *total_ms, long long *launches, long long *sweeps)
{ if (total_ms) *total_ms = 0; if (launches) *launches = 0; if (sweeps) *sweeps = 0; }

void gpu_solve_forward(float *u, float *v, float *w, float *x, float *y, float *z,
                       float h, int ni, int nj, int nk, float cfldt, float dt)
{ orc_solve_forward(u, v, w, x, y, z, h, ni, nj, nk, cfldt, dt); }
void gpu_solve_backwardDMC(float *u, float *v, float *w, float *xi, float *yi, float *zi,
                           float *xo, float *yo, float *zo, float h, int ni, int nj, int nk, float substep)
{
    const size_t pl = (size_t)ni * nj;
    if (g_fused & 8) { win_copy(xo, xi, pl, nk, nk); win_copy(yo, yi, pl, nk, nk); win_copy(zo, zi, pl, nk, nk); }
    else if (g_fused & 4) { win_zero(xo, pl, nk, nk); win_zero(yo, pl, nk, nk); win_zero(zo, pl, nk, nk); }
    orc_solve_backwardDMC(u, v, w, xi, yi, zi, xo, yo, zo, h, ni, nj, nk, substep);
}
void gpu_advect_velocity(float *u, float *v, float *w, float *ui, float *vi, float *wi,
                         float *bx, float *by, float *bz, float h, int ni, int nj, int nk, bool pt)
{
    if (g_fused & 1) { win_zero(u, (size_t)(ni + 1) * nj, nk, nk); win_zero(v, (size_t)ni * (nj + 1), nk, nk); win_zero(w, (size_t)ni * nj, nk, nk + 1); }
    orc_advect_velocity(u, v, w, ui, vi, wi, bx, by, bz, h, ni, nj, nk, pt);
}
void gpu_advect_vel_double(float *u, float *v, float *w, float *ut, float *vt, float *wt,
                           float *bx, float *by, float *bz, float *px, float *py, float *pz,
                           float h, int ni, int nj, int nk, bool pt, float blend)
{ orc_advect_vel_double(u, v, w, ut, vt, wt, bx, by, bz, px, py, pz, h, ni, nj, nk, pt, blend); }
void gpu_advect_field(float *f, float *fi, float *bx, float *by, float *bz, float h, int ni, int nj, int nk, bool pt)
{
    if (g_fused & 1) win_zero(f, (size_t)ni * nj, nk, nk);
    orc_advect_field(f, fi, bx, by, bz, h, ni, nj, nk, pt);
}
void gpu_advect_field_double(float *f, float *fp, float *bx, float *by, float *bz, float *px, float *py, float *pz,
                             float h, int ni, int nj, int nk, bool pt, float blend)
{ orc_advect_field_double(f, fp, bx, by, bz, px, py, pz, h, ni, nj, nk, pt, blend); }
void gpu_advect_vel_double_global(float *u, float *v, float *w, float *ug, float *vg, float *wg,
                                  float *bx, float *by, float *bz, float *px, float *py, float *pz,
                                  float h, int ni, int nj, int nk, bool pt, float blend)
{ orc_advect_vel_double_global(u, v, w, ug, vg, wg, bx, by, bz, px, py, pz, h, ni, nj, nk, pt, blend); }
void gpu_advect_field_double_global(float *f, float *fg, float *bx, float *by, float *bz, float *px, float *py, float *pz,
                                    float h, int ni, int nj, int nk, bool pt, float blend)
{ orc_advect_field_double_global(f, fg, bx, by, bz, px, py, pz, h, ni, nj, nk, pt, blend); }
void gpu_accumulate_velocity(float *uc, float *vc, float *wc, float *du, float *dv, float *dw,
                             float *fx, float *fy, float *fz, float h, int ni, int nj, int nk, bool pt, float coeff)
{ orc_accumulate_velocity(uc, vc, wc, du, dv, dw, fx, fy, fz, h, ni, nj, nk, pt, coeff); }
void gpu_accumulate_field(float *fc, float *df, float *fx, float *fy, float *fz,
                          float h, int ni, int nj, int nk, bool pt, float coeff)
{ orc_accumulate_field(fc, df, fx, fy, fz, h, ni, nj, nk, pt, coeff); }
void gpu_estimate_distortion(float *du, float *xb, float *yb, float *zb, float *xf, float *yf, float *zf,
                             float h, int ni, int nj, int nk)
{ orc_estimate_distortion(du, xb, yb, zb, xf, yf, zf, h, ni, nj, nk); }
void gpu_add(float *f1, float *f2, float coeff, int number) { orc_add(f1, f2, coeff, number); }
void gpu_compensate_velocity(float *u, float *v, float *w, float *du, float *dv, float *dw,
                             float *us, float *vs, float *ws, float *fx, float *fy, float *fz,
                             float *bx, float *by, float *bz, float h, int ni, int nj, int nk, bool pt)
{ orc_compensate_velocity(u, v, w, du, dv, dw, us, vs, ws, fx, fy, fz, bx, by, bz, h, ni, nj, nk, pt); }
void gpu_compensate_field(float *u, float *du, float *us, float *fx, float *fy, float *fz,
                          float *bx, float *by, float *bz, float h, int ni, int nj, int nk, bool pt)
{ orc_compensate_field(u, du, us, fx, fy, fz, bx, by, bz, h, ni, nj, nk, pt); }
void gpu_semilag(float *f, float *fs, float *u, float *v, float *w, int dx, int dy, int dz,
                 float h, int ni, int nj, int nk, float cfldt, float dt)
{ orc_semilag(f, fs, u, v, w, dx, dy, dz, h, ni, nj, nk, cfldt, dt); }
void gpu_emit_smoke(float *u, float *v, float *w, float *rho, float *T, float h, int ni, int nj, int nk,
                    float cx, float cy, float cz, float radius, float density, float temperature, float emiter)
{ orc_emit_smoke(u, v, w, rho, T, h, ni, nj, nk, cx, cy, cz, radius, density, temperature, emiter); }
void gpu_add_buoyancy(float *f, float *rho, float *T, int ni, int nj, int nk, float alpha, float beta, float dt)
{ orc_add_buoyancy(f, rho, T, ni, nj, nk, alpha, beta, dt); }
void gpu_diffuse_field(float *f, float *t0, float *t1, int ni, int nj, int nk, int iter, float coef)
{ orc_diffuse_field(f, t0, t1, ni, nj, nk, iter, coef); }
void gpu_add_field(float *out, float *f1, float *f2, float coeff, int number) { orc_add_field(out, f1, f2, coeff, number); }
void gpu_projection_jacobi(float *u, float *v, float *w, float *div, float *p, float *pt, float *dbg,
                           int ni, int nj, int nk, int iter, float halfrdx, float alpha, float beta)
{ orc_projection_jacobi(u, v, w, div, p, pt, g_opt[FL_OPT_RESIDUAL_STRIDE] > 0 ? dbg : NULL, ni, nj, nk, iter, halfrdx, alpha, beta); }
void gpu_mad(float *out, float *f1, float *f2, float c1, float c2, int number) { orc_mad(out, f1, f2, c1, c2, number); }
void gpu_conjugate_gradient(float *a, float *b, float *c, float *d, float *e, float *f, float *g, float *h,
                            int i, int j, int k, int l, float m)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; (void)j; (void)k; (void)l; (void)m;
  latch(FL_ERR_UNSUPPORTED, "gpu_conjugate_gradient"); }
void gpu_multi_grid_conjugate_gradient(float *a, float *b, float *c, double *d, double *e, double *f, double *g,
                                       double *h, double *i, double *j, struct SCoarseLevelInfo *k, int l, int m, double n)
{ /* SCoarseLevelInfo and OrcCoarseLevel share their layout (GPU_Advection.h:15-24) */
  orc_multi_grid_conjugate_gradient(a, b, c, d, e, f, g, h, i, j, (OrcCoarseLevel *)k, l, m, n); }

/* ---- slab context + communicator (mirrors csrc/bq_halo.hip with the custom transport only) ---- */
static int s_on, s_koff, s_nkg, s_own0, s_own1;
static int c_rank, c_nranks = 1;
static fl_exchange_cb c_exchange;
static fl_allreduce_cb c_allreduce;

void fl_report_error(int code, const char *text) { latch(code, text ? text : ""); }
/* no contexts in the stand-in: the host solver's C API only asks which one is current and switches back to it */
fl_context *fl_context_current(void) { return NULL; }
void fl_context_make_current(fl_context *c) { (void)c; }
int fl_set_plane_window(int k0, int k1)
{
    if (g_nk_local <= 0) return 0;  /* only meaningful on a slab rank */
    g_win_on = k0 >= 0; g_win_k0 = k0 < 0 ? 0 : k0; g_win_k1 = k1; g_win_cells = g_nk_local;
    orc_set_plane_window(k0, k1, g_nk_local);
    return 1;
}
void fl_set_slab(int koff, int nk_global, int own0, int own1, int nk_local)
{
    s_on = nk_global > 0; s_koff = koff; s_nkg = nk_global; s_own0 = own0; s_own1 = own1;
    g_nk_local = nk_global > 0 ? nk_local : 0;
    orc_set_slab(koff, nk_global, own0, own1, nk_local);
}
int fl_comm_unique_id(void *id128) { memset(id128, 0, 128); return FL_OK; }
int fl_comm_init(const void *id128, int rank, int nranks) { (void)id128; (void)rank; if (nranks > 1) latch(FL_ERR_COMM, "no RCCL in the CPU stand-in"); return g_err; }
void fl_comm_destroy(void) { c_rank = 0; c_nranks = 1; c_exchange = NULL; c_allreduce = NULL; }
int fl_comm_count(void) { return 0; }
int fl_comm_check(int perturb) { (void)perturb; return FL_OK; }
void fl_shutdown_all(void) {}
void fl_aux_begin(void) {}
void fl_aux_end(void) {}
void fl_aux_join(void) {}
int fl_comm_rank(void) { return c_rank; }
int fl_comm_size(void) { return c_nranks; }
void fl_comm_set_custom(int rank, int nranks, fl_exchange_cb ex, fl_allreduce_cb ar)
{ c_rank = rank; c_nranks = nranks; c_exchange = ex; c_allreduce = ar; }
void fl_halo_exchange(int n, float *const *fields, const size_t *plane_elems, const int *extra,
                      int nk_local, int G, int depth, int wait)
{
    (void)wait;
    if (c_nranks <= 1) return;
    if (!c_exchange) { latch(FL_ERR_COMM, "fl_halo_exchange: no transport"); return; }
    c_exchange(n, fields, plane_elems, extra, nk_local, G, depth);
}
void fl_halo_wait(void) {}
void fl_comm_stats(long long out[4], int reset) { (void)reset; if (out) for (int a = 0; a < 4; a++) out[a] = 0; }

/* ---- wall sheets (include/bimocq_gpu.h, section 4) on host memory ---- */
static fl_p2p_cb c_p2p;
void fl_comm_set_custom_p2p(fl_p2p_cb p2p) { c_p2p = p2p; }
void fl_p2p_exchange(int n, const int *peers, float *const *send, const size_t *send_count,
                     float *const *recv, const size_t *recv_count)
{
    if (c_nranks <= 1 || n <= 0) return;
    if (!c_p2p) { latch(FL_ERR_COMM, "fl_p2p_exchange: no transport"); return; }
    c_p2p(n, peers, send, send_count, recv, recv_count);
}
void fl_p2p_exchange_begin(int n, const int *peers, float *const *send, const size_t *send_count,
                           float *const *recv, const size_t *recv_count)
{
    fl_p2p_exchange(n, peers, send, send_count, recv, recv_count);
}
static void box_copy(float *field, int nbi, int nbj, int nk_field, int koff, const fl_box *boxes, int nboxes, float *packed, int mode)
{
    size_t t = 0;
    for (int b = 0; b < nboxes; b++) {
        const fl_box q = boxes[b];
        if (q.x0 < 0 || q.y0 < 0 || q.z0 < koff || q.x1 > nbi || q.y1 > nbj || q.z1 > koff + nk_field) { latch(FL_ERR_BAD_ARGUMENT, "fl_box_*: box outside the field"); return; }
        for (int z = q.z0; z < q.z1; z++)
            for (int y = q.y0; y < q.y1; y++)
                for (int x = q.x0; x < q.x1; x++, t++) {
                    size_t id = (size_t)x + (size_t)nbi * ((size_t)y + (size_t)nbj * (size_t)(z - koff));
                    if (mode == 0) packed[t] = field[id];
                    else if (mode == 1) field[id] = packed[t];
                    else field[id] = NAN;
                }
    }
}
void fl_box_pack(const float *field, int nbi, int nbj, int nk_field, int koff, const fl_box *boxes, int nboxes, float *packed)
{ box_copy((float *)field, nbi, nbj, nk_field, koff, boxes, nboxes, packed, 0); }
void fl_box_unpack(float *field, int nbi, int nbj, int nk_field, int koff, const fl_box *boxes, int nboxes, const float *packed)
{ box_copy(field, nbi, nbj, nk_field, koff, boxes, nboxes, (float *)packed, packed ? 1 : 2); }
void fl_box_copy(const float *src, int nbi, int nbj, int nk_src, int koff_src, float *dst, int nk_dst, int koff_dst,
                 const fl_box *boxes, int nboxes)
{
    for (int b = 0; b < nboxes; b++) {
        const fl_box q = boxes[b];
        if (q.x0 < 0 || q.y0 < 0 || q.z0 < koff_src || q.z0 < koff_dst || q.x1 > nbi || q.y1 > nbj || q.z1 > koff_src + nk_src || q.z1 > koff_dst + nk_dst) {
            latch(FL_ERR_BAD_ARGUMENT, "fl_box_copy: box outside a field"); return;
        }
        for (int z = q.z0; z < q.z1; z++)
            for (int y = q.y0; y < q.y1; y++)
                for (int x = q.x0; x < q.x1; x++)
                    dst[(size_t)x + (size_t)nbi * ((size_t)y + (size_t)nbj * (size_t)(z - koff_dst))] =
                        src[(size_t)x + (size_t)nbi * ((size_t)y + (size_t)nbj * (size_t)(z - koff_src))];
    }
}
void gpu_accumulate_wall_fixup(const float *src, int src_koff, int src_nk, const float *before, float *dst,
                               const float *mx, const float *my, const float *mz,
                               float h, int ni, int nj, int nk, int axis, float coeff,
                               const int *xlist, int nxl, const int *ylist, int nyl, const int *zlist, int nzl)
{ orc_accumulate_wall_fixup(src, src_koff, src_nk, before, dst, mx, my, mz, h, ni, nj, nk, axis, coeff, xlist, nxl, ylist, nyl, zlist, nzl); }

/* the stand-in has one arithmetic path: the precondition checks may simply say no */
void fl_map_guard_reset(int which) { (void)which; }
void fl_map_guard_read(int ok[2]) { if (ok) ok[0] = ok[1] = 0; }
int gpu_maps_quarter_safe(const float *x, const float *y, const float *z, float h, int ni, int nj, int nk)
{ (void)x; (void)y; (void)z; (void)h; (void)ni; (void)nj; (void)nk; return 0; }
void gpu_init_maps(float *x, float *y, float *z, float h, int ni, int nj, int nk)
{
    for (int k = 0; k < nk; k++)
        for (int j = 0; j < nj; j++)
            for (int i = 0; i < ni; i++) {
                size_t id = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * k);
                int kg = k + (s_on ? s_koff : 0);
                int in = !s_on || (kg >= 0 && kg < s_nkg);
                x[id] = in ? (float)i * h : 0.f; y[id] = in ? (float)j * h : 0.f; z[id] = in ? (float)kg * h : 0.f;
            }
}
float gpu_max_abs3(const float *u, const float *v, const float *w, int ni, int nj, int nk)
{
    float m = orc_max_abs3(u, v, w, ni, nj, nk);
    if (c_nranks > 1 && c_allreduce) c_allreduce(&m, 1, 0, 1);
    return m;
}
void gpu_compensate_error_velocity(float *u, float *v, float *w, float *du, float *dv, float *dw,
                                   float *us, float *vs, float *ws, float *fx, float *fy, float *fz,
                                   float h, int ni, int nj, int nk, bool pt)
{
    const size_t pu = (size_t)(ni + 1) * nj, pv = (size_t)ni * (nj + 1), pw = (size_t)ni * nj;
    if (g_fused & 1) { win_zero(us, pu, nk, nk); win_zero(vs, pv, nk, nk); win_zero(ws, pw, nk, nk + 1); }
    orc_compensate_error_velocity(u, v, w, du, dv, dw, us, vs, ws, fx, fy, fz, h, ni, nj, nk, pt);
    if (g_fused & 2) { win_copy(du, u, pu, nk, nk); win_copy(dv, v, pv, nk, nk); win_copy(dw, w, pw, nk, nk + 1); }
}
void gpu_compensate_error_field(float *u, float *du, float *us, float *fx, float *fy, float *fz,
                                float h, int ni, int nj, int nk, bool pt)
{
    if (g_fused & 1) win_zero(us, (size_t)ni * nj, nk, nk);
    orc_compensate_error_field(u, du, us, fx, fy, fz, h, ni, nj, nk, pt);
    if (g_fused & 2) win_copy(du, u, (size_t)ni * nj, nk, nk);
}
/* batched / shortcut forms: by definition the single operators in order */
void gpu_advect_field2(float *f1, float *f1i, float *f2, float *f2i, float *bx, float *by, float *bz,
                       float h, int ni, int nj, int nk, bool pt)
{ gpu_advect_field(f1, f1i, bx, by, bz, h, ni, nj, nk, pt); gpu_advect_field(f2, f2i, bx, by, bz, h, ni, nj, nk, pt); }
void gpu_compensate_error_field2(float *u1, float *du1, float *us1, float *u2, float *du2, float *us2,
                                 float *fx, float *fy, float *fz, float h, int ni, int nj, int nk, bool pt)
{ gpu_compensate_error_field(u1, du1, us1, fx, fy, fz, h, ni, nj, nk, pt); gpu_compensate_error_field(u2, du2, us2, fx, fy, fz, h, ni, nj, nk, pt); }
void gpu_accumulate_field2(float *c1, float *d1, float k1, float *c2, float *d2, float k2,
                           float *fx, float *fy, float *fz, float h, int ni, int nj, int nk, bool pt)
{ orc_accumulate_field(c1, d1, fx, fy, fz, h, ni, nj, nk, pt, k1); orc_accumulate_field(c2, d2, fx, fy, fz, h, ni, nj, nk, pt, k2); }
void gpu_accumulate_velocity2(float *u1, float *v1, float *w1, float k1, float *u2, float *v2, float *w2, float k2,
                              float *du, float *dv, float *dw, float *fx, float *fy, float *fz,
                              float h, int ni, int nj, int nk, bool pt)
{
    orc_accumulate_velocity(u1, v1, w1, du, dv, dw, fx, fy, fz, h, ni, nj, nk, pt, k1);
    orc_accumulate_velocity(u2, v2, w2, du, dv, dw, fx, fy, fz, h, ni, nj, nk, pt, k2);
}
void gpu_accumulate_component(float *c1, float k1, float *c2, float k2, float *d, float *fx, float *fy, float *fz,
                              float h, int ni, int nj, int nk, int axis, bool pt)
{
    orc_accumulate_component(c1, d, fx, fy, fz, h, ni, nj, nk, axis, pt, k1);
    if (c2) orc_accumulate_component(c2, d, fx, fy, fz, h, ni, nj, nk, axis, pt, k2);
}
void gpu_accumulate_velocity_identity(float *uc, float *vc, float *wc, float *du, float *dv, float *dw,
                                      float *fx, float *fy, float *fz, float h, int ni, int nj, int nk, bool pt, float coeff)
{ orc_accumulate_velocity(uc, vc, wc, du, dv, dw, fx, fy, fz, h, ni, nj, nk, pt, coeff); }
/* the slab-decomposed multigrid solver is HIP-only: the CPU stand-in answers "not supported" and the host solver keeps the replicated solve */
int gpu_mgcg_slab_supported(int ni, int nj, int nkg, int own0, int own1, int ghost, int rank, int nranks)
{ (void)ni; (void)nj; (void)nkg; (void)own0; (void)own1; (void)ghost; (void)rank; (void)nranks; return 0; }
void gpu_multi_grid_conjugate_gradient_slab(float *u, float *v, float *w, double *tempResult, int ni, int nj, int nkg, int own0, int own1,
                                            int ghost, int iter, double halfrdx)
{ (void)u; (void)v; (void)w; (void)tempResult; (void)ni; (void)nj; (void)nkg; (void)own0; (void)own1; (void)ghost; (void)iter; (void)halfrdx;
  latch(FL_ERR_UNSUPPORTED, "gpu_multi_grid_conjugate_gradient_slab: not in the CPU stand-in"); }
void gpu_smoothing_jacobi(double *x, double *b, double *temp, double alpha, double beta, int ni, int nj, int nk, int iter)
{ orc_mg_smooth(x, b, temp, alpha, beta, ni, nj, nk, iter); }
int gpu_diffuse_sweeps(const float *field, float *in, float *out, int ni, int nj, int nk, int sweeps, float coef)
{ return orc_diffuse_sweeps(field, in, out, ni, nj, nk, sweeps, coef); }
void *fl_malloc_host(size_t bytes) { return calloc(bytes ? bytes : 4, 1); }
void fl_free_host(void *p) { free(p); }
void *fl_download_begin(void *host_dst, const void *dev_src, size_t bytes) { memcpy(host_dst, dev_src, bytes); return (void *)1; }
int fl_download_wait(void *ticket) { return ticket ? FL_OK : FL_ERR_BAD_ARGUMENT; }
void gpu_clamp_extrema(float *field, float *ft, float *u, float *v, float *w, int ni, int nj, int nk,
                       int dx, int dy, int dz, float ox, float oy, float oz, float h, float dt)
{ orc_clamp_extrema(field, ft, u, v, w, ni, nj, nk, dx, dy, dz, ox, oy, oz, h, dt); }
float gpu_max_field(const float *field, size_t count)
{
    float m = 0.f;
    for (size_t q = 0; q < count; q++) if (fabsf(field[q]) > m) m = fabsf(field[q]);
    return m;
}
float gpu_max_field_owned(const float *field, int ni, int nj, int nk)
{
    const int p0 = s_on ? s_own0 - s_koff : 0, p1 = s_on ? s_own1 - s_koff : nk;
    const size_t plane = (size_t)ni * nj;
    float m = gpu_max_field(field + plane * p0, plane * (size_t)(p1 - p0));
    if (c_nranks > 1 && c_allreduce) c_allreduce(&m, 1, 0, 1);
    return m;
}
void gpu_map_travel_z(const float *bz, const float *fz, float h, int ni, int nj, int nk, float out[2])
{
    const int p0 = s_on ? s_own0 - s_koff : 0, p1 = s_on ? s_own1 - s_koff : nk, nkg = s_on ? s_nkg : nk;
    float m[2] = { 0.f, 0.f };
    for (int k = p0; k < p1; k++) {
        const int kg = k + (s_on ? s_koff : 0);
        if (!(kg > 1 && kg < nkg - 2)) continue;
        const float z = (float)kg * h;
        for (int j = 2; j < nj - 2; j++)
            for (int i = 2; i < ni - 2; i++) {
                const size_t id = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * k);
                float db = fabsf(bz[id] - z), df = fabsf(fz[id] - z);
                if (db != db) db = INFINITY;
                if (df != df) df = INFINITY;
                if (db > m[0]) m[0] = db;
                if (df > m[1]) m[1] = df;
            }
    }
    if (c_nranks > 1 && c_allreduce) c_allreduce(m, 2, 0, 1);
    out[0] = m[0] / h; out[1] = m[1] / h;
}
void gpu_clamp_extrema_box_w(const float *before, float *after, int ni, int nj, int nk)
{ orc_clamp_extrema_box_w(before, after, ni, nj, nk); }
void gpu_divergence(const float *u, const float *v, const float *w, float *div, int ni, int nj, int nk, float hr)
{ orc_divergence(u, v, w, div, ni, nj, nk, hr); }
int gpu_jacobi_sweeps(float *p, const float *div, float *pt, int ni, int nj, int nk, int sweeps, float alpha, float beta)
{
    float *in = p, *out = pt;
    for (int s = 0; s < sweeps; s++) { orc_jacobi_sweep(in, div, out, ni, nj, nk, alpha, beta); float *t = in; in = out; out = t; }
    return in == p ? 0 : 1;
}
void gpu_gradient_delta(float *u, float *v, float *w, const float *p, float *du, float *dv, float *dw,
                        int ni, int nj, int nk, float hr)
{
    size_t nu = (size_t)(ni + 1) * nj * nk, nv = (size_t)ni * (nj + 1) * nk, nw = (size_t)ni * nj * (nk + 1);
    memcpy(du, u, nu * sizeof(float)); memcpy(dv, v, nv * sizeof(float)); memcpy(dw, w, nw * sizeof(float));
    gpu_gradient(u, v, w, p, ni, nj, nk, hr);
    for (size_t q = 0; q < nu; q++) du[q] = u[q] - du[q];
    for (size_t q = 0; q < nv; q++) dv[q] = v[q] - dv[q];
    for (size_t q = 0; q < nw; q++) dw[q] = w[q] - dw[q];
}
void gpu_jacobi_sweep_range(const float *in, const float *div, float *out, int ni, int nj, int nk,
                            int k_begin, int k_end, float alpha, float beta)
{ orc_jacobi_sweep_range(in, div, out, ni, nj, nk, k_begin, k_end, alpha, beta); }
/* two sweeps, output on two plane ranges: L1 on the planes those outputs read (into a scratch copy of `in`, which
 * also gives it the same boundary layer), then L2 on the ranges themselves */
int gpu_jacobi_sweep_pair_ranges(const float *in, const float *div, float *out, int ni, int nj, int nk,
                                 int k0a, int k1a, int k0b, int k1b, float alpha, float beta)
{
    size_t n = (size_t)ni * nj * nk;
    float *l1 = (float *)malloc(n * sizeof(float));
    if (!l1) return 0;
    memcpy(l1, in, n * sizeof(float));
    const int r[2][2] = { { k0a, k1a }, { k0b, k1b } };
    for (int a = 0; a < 2; a++) {
        int k0 = r[a][0] < 0 ? 0 : r[a][0], k1 = r[a][1] > nk ? nk : r[a][1];
        if (k0 >= k1) continue;
        orc_jacobi_sweep_range(in, div, l1, ni, nj, nk, k0 - 1 < 0 ? 0 : k0 - 1, k1 + 1 > nk ? nk : k1 + 1, alpha, beta);
    }
    for (int a = 0; a < 2; a++) {
        int k0 = r[a][0] < 0 ? 0 : r[a][0], k1 = r[a][1] > nk ? nk : r[a][1];
        if (k0 >= k1) continue;
        orc_jacobi_sweep_range(l1, div, out, ni, nj, nk, k0, k1, alpha, beta);
    }
    free(l1);
    return 1;
}
/* three sweeps, output on two plane ranges: L1 on the planes two beyond, L2 one beyond, L3 on the ranges themselves */
int gpu_jacobi_sweep_triple_ranges(const float *in, const float *div, float *out, int ni, int nj, int nk,
                                   int k0a, int k1a, int k0b, int k1b, float alpha, float beta)
{
    size_t n = (size_t)ni * nj * nk;
    float *l1 = (float *)malloc(n * sizeof(float)), *l2 = (float *)malloc(n * sizeof(float));
    if (!l1 || !l2) { free(l1); free(l2); return 0; }
    memcpy(l1, in, n * sizeof(float));
    memcpy(l2, in, n * sizeof(float));
    const int r[2][2] = { { k0a, k1a }, { k0b, k1b } };
    for (int lev = 0; lev < 3; lev++)
        for (int a = 0; a < 2; a++) {
            int k0 = r[a][0] < 0 ? 0 : r[a][0], k1 = r[a][1] > nk ? nk : r[a][1];
            if (k0 >= k1) continue;
            const int ext = 2 - lev;
            const int e0 = k0 - ext < 0 ? 0 : k0 - ext, e1 = k1 + ext > nk ? nk : k1 + ext;
            if (lev == 0) orc_jacobi_sweep_range(in, div, l1, ni, nj, nk, e0, e1, alpha, beta);
            else if (lev == 1) orc_jacobi_sweep_range(l1, div, l2, ni, nj, nk, e0, e1, alpha, beta);
            else orc_jacobi_sweep_range(l2, div, out, ni, nj, nk, e0, e1, alpha, beta);
        }
    free(l1); free(l2);
    return 1;
}
void gpu_gradient(float *u, float *v, float *w, const float *p, int ni, int nj, int nk, float hr)
{
    orc_gradient(u, p, ni + 1, nj, nk, 1, 0, 0, hr);
    orc_gradient(v, p, ni, nj + 1, nk, 0, 1, 0, hr);
    orc_gradient(w, p, ni, nj, nk + 1, 0, 0, 1, hr);
}
void gpu_residual_norms(const float *div, const float *p, int ni, int nj, int nk, double *ss, float *mx)
{ double s; float m; orc_residual_norms(div, p, ni, nj, nk, &s, &m); if (ss) *ss = s; if (mx) *mx = m; }
void gpu_clamp_extrema_box(const float *before, float *after, int ni, int nj, int nk)
{ orc_clamp_extrema_box(before, after, ni, nj, nk); }
