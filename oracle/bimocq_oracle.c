/*
 * bimocq_oracle.c -- CPU restatement of the bimocq3D per-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see bimocq_oracle.h).  PARITY UNPINNED by reference
 * fixtures (the reference has none); every function cites the reference lines it
 * restates, relative to /root/reference/src/bimocq3D/.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared (oracle/Makefile).
 * OpenMP `parallel for` stands in for the reference's one-thread-per-voxel CUDA grid;
 * every kernel below is a pure per-voxel map (reads never alias another voxel's
 * writes inside one launch), so the loop order cannot change results.
 */
#include "bimocq_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } f3;

static inline f3 mk3(float x, float y, float z) { f3 r = { x, y, z }; return r; }

/* z-slab context (orc_set_slab), mirror of the library's fl_set_slab: the buffers hold the global
 * planes [koff, koff + nk_local) of a grid with nkg cell planes.  Windows, positions and clamps are
 * evaluated in global coordinates; off = single domain (koff = 0, nkg = nk). */
static int S_on = 0, S_koff = 0, S_nkg = 0, S_own0 = 0, S_own1 = 0, S_nkl = 0;
void orc_set_slab(int koff, int nk_global, int own0, int own1, int nk_local)
{
    S_nkl = nk_local;
    S_on = nk_global > 0; S_koff = koff; S_nkg = nk_global; S_own0 = own0; S_own1 = own1;
}
#define KOFF (S_on ? S_koff : 0)
#define NKG(nk) (S_on ? S_nkg : (nk))
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
/* local plane range [KLO, KHI) of a buffer with nkloc planes whose GLOBAL index lies in [glo, ghi) */
/* plane window (orc_set_plane_window; mirrors fl_set_plane_window of the operator ABI): while it is set, the operators
 * touch the local planes [W_k0, W_k1) only; a buffer with one plane more than cells gets its extra plane from the window
 * that reaches the last cell plane.  W_cells: local cell planes (needed for that rule), given with the window. */
static int W_on = 0, W_k0 = 0, W_k1 = 0, W_cells = 0;
void orc_set_plane_window(int k0, int k1, int nk_cells)
{
    W_on = k0 >= 0; W_k0 = k0 < 0 ? 0 : k0; W_k1 = k1 < W_k0 ? W_k0 : k1; W_cells = nk_cells;
}
#define WLO (W_on ? W_k0 : 0)
#define WHI(nkloc) (W_on ? (W_k1 >= W_cells ? (nkloc) : W_k1) : (nkloc))
#define KLO(glo) imax(imax(0, (glo) - KOFF), WLO)
#define KHI(ghi, nkloc) imin(imin((nkloc), (ghi) - KOFF), WHI(nkloc))

/* GPU_kernel.cu:9-12 */
static inline float clampf(float a, float lo, float hi) { return fminf(fmaxf(lo, a), hi); }

/* GPU_kernel.cu:14-20 */
static inline f3 clamp3(f3 p, f3 lo, f3 hi)
{
    return mk3(clampf(p.x, lo.x, hi.x), clampf(p.y, lo.y, hi.y), clampf(p.z, lo.z, hi.z));
}

/* GPU_kernel.cu:22-25: `(1.0-c)*a + c*b` -- (1.0-c) and the first product are double,
 * c*b is a float product, the sum is double, the return rounds to float. */
static int g_fast_lerp = 0;
/* the library's FL_OPT_FAST_LERP variant: every lerp is one fp32 fma, nothing else changes */
void orc_set_fast_lerp(int on) { g_fast_lerp = on != 0; }

float orc_lerp(float a, float b, float c)
{
    if (g_fast_lerp) return fmaf(c, b - a, a);
    float cb = c * b;
    return (float)((1.0 - (double)c) * (double)a + (double)cb);
}

/* GPU_kernel.cu:27-41 */
static inline float trilerp(float v000, float v001, float v010, float v011,
                            float v100, float v101, float v110, float v111,
                            float a, float b, float c)
{
    float l00 = orc_lerp(v000, v001, a);
    float l01 = orc_lerp(v010, v011, a);
    float l10 = orc_lerp(v100, v101, a);
    float l11 = orc_lerp(v110, v111, a);
    return orc_lerp(orc_lerp(l00, l01, b), orc_lerp(l10, l11, b), c);
}

static inline float ld(const float *b, long idx, long count)
{
    return (idx >= 0 && idx < count) ? b[idx] : 0.0f;
}

/* GPU_kernel.cu:43-62 sample_buffer: no index clamping, flat index arithmetic kept (a corner one past a
 * row/plane wraps into the next one); reads outside the allocation return 0 and a cell whose base corner
 * has a negative flat index -- memory before the array in the reference: undefined -- reads all zeros
 * (header, arithmetic contract). */
static inline float sample(const float *b, int nx, int ny, int nz, float h, f3 off, f3 pos)
{
    float sx = pos.x - off.x, sy = pos.y - off.y, sz = pos.z - off.z;
    float qx = sx / h, qy = sy / h, qz = sz / h;
    int i = (int)floorf(qx), j = (int)floorf(qy), k = (int)floorf(qz);
    float fx = qx - (float)i, fy = qy - (float)j, fz = qz - (float)k;
    long sj = nx, sk = (long)nx * ny, count = (long)nx * ny * nz;
    long base = (long)i + sj * j + sk * (k - KOFF);
    if (base < 0) base = count;                 /* every corner out of range */
    return trilerp(ld(b, base, count),           ld(b, base + 1, count),
                   ld(b, base + sj, count),      ld(b, base + sj + 1, count),
                   ld(b, base + sk, count),      ld(b, base + sk + 1, count),
                   ld(b, base + sk + sj, count), ld(b, base + sk + sj + 1, count),
                   fx, fy, fz);
}

/* the same for a buffer that holds the global planes [koff, koff + nz) whatever the slab context says (the wall-sheet
 * copy a z-slab rank assembles for the border nodes of the compensation, orc_accumulate_wall_fixup) */
static inline float sample_k(const float *b, int nx, int ny, int nz, int koff, float h, f3 off, f3 pos)
{
    float sx = pos.x - off.x, sy = pos.y - off.y, sz = pos.z - off.z;
    float qx = sx / h, qy = sy / h, qz = sz / h;
    int i = (int)floorf(qx), j = (int)floorf(qy), k = (int)floorf(qz);
    float fx = qx - (float)i, fy = qy - (float)j, fz = qz - (float)k;
    long sj = nx, sk = (long)nx * ny, count = (long)nx * ny * nz;
    long base = (long)i + sj * j + sk * (k - koff);
    if (base < 0) base = count;
    return trilerp(ld(b, base, count),           ld(b, base + 1, count),
                   ld(b, base + sj, count),      ld(b, base + sj + 1, count),
                   ld(b, base + sk, count),      ld(b, base + sk + 1, count),
                   ld(b, base + sk + sj, count), ld(b, base + sk + sj + 1, count),
                   fx, fy, fz);
}

float orc_sample(const float *b, int nx, int ny, int nz, float h,
                 float ox, float oy, float oz, float px, float py, float pz)
{
    return sample(b, nx, ny, nz, h, mk3(ox, oy, oz), mk3(px, py, pz));
}

/* GPU_kernel.cu:64-72 getVelocity: MAC-staggered components, origins (-h/2,0,0) etc.
 * (-0.5*h is a double product rounded to float: exact). */
static inline f3 get_velocity(const float *u, const float *v, const float *w,
                              float h, int nx, int ny, int nz, f3 pos)
{
    float mh = (float)(-0.5 * (double)h);
    float _u = sample(u, nx + 1, ny, nz, h, mk3(mh, 0.f, 0.f), pos);
    float _v = sample(v, nx, ny + 1, nz, h, mk3(0.f, mh, 0.f), pos);
    float _w = sample(w, nx, ny, nz + 1, h, mk3(0.f, 0.f, mh), pos);
    return mk3(_u, _v, _w);
}

/* GPU_kernel.cu:74-90 traceRK3 (Ralston RK3; stage points evaluated in double) */
static inline f3 trace_rk3(const float *u, const float *v, const float *w,
                           float h, int ni, int nj, int nk, float dt, f3 pos)
{
    float c1 = (float)(2.0 / 9.0 * (double)dt);
    float c2 = (float)(3.0 / 9.0 * (double)dt);
    float c3 = (float)(4.0 / 9.0 * (double)dt);
    f3 v1 = get_velocity(u, v, w, h, ni, nj, nk, pos);
    double hdt = 0.5 * (double)dt;
    f3 m1 = mk3((float)((double)pos.x + hdt * (double)v1.x),
                (float)((double)pos.y + hdt * (double)v1.y),
                (float)((double)pos.z + hdt * (double)v1.z));
    f3 v2 = get_velocity(u, v, w, h, ni, nj, nk, m1);
    double qdt = 0.75 * (double)dt;
    f3 m2 = mk3((float)((double)pos.x + qdt * (double)v2.x),
                (float)((double)pos.y + qdt * (double)v2.y),
                (float)((double)pos.z + qdt * (double)v2.z));
    f3 v3 = get_velocity(u, v, w, h, ni, nj, nk, m2);
    f3 out = mk3(pos.x + c1 * v1.x + c2 * v2.x + c3 * v3.x,
                 pos.y + c1 * v1.y + c2 * v2.y + c3 * v3.y,
                 pos.z + c1 * v1.z + c2 * v2.z + c3 * v3.z);
    return clamp3(out, mk3(h, h, h),
                  mk3((float)ni * h - h, (float)nj * h - h, (float)NKG(nk) * h - h));
}

/* GPU_kernel.cu:92-125 trace: sub-step by cfldt until |dt| is consumed */
static inline f3 trace(const float *u, const float *v, const float *w,
                       float h, int ni, int nj, int nk, float cfldt, float dt, f3 pos)
{
    float sgn = (dt > 0) ? 1.0f : -1.0f;
    float T = (dt > 0) ? dt : -dt;
    float t = 0.f, substep = cfldt;
    f3 p = pos;
    while (t < T) {
        if (t + substep > T) substep = T - t;
        p = trace_rk3(u, v, w, h, ni, nj, nk, (sgn > 0) ? substep : -substep, p);
        t += substep;
    }
    return p;
}

/* expf of the DMC integrator: fixed double polynomial, identical in the HIP path. */
float orc_expf(float xf)
{
    double x = (double)xf;
    if (!(x == x)) return xf;
    if (x > 90.0) x = 90.0;
    if (x < -110.0) x = -110.0;
    const double LOG2E = 1.4426950408889634074;
    const double LN2HI = 6.93147180369123816490e-01;
    const double LN2LO = 1.90821492927058770002e-10;
    double kd = floor(x * LOG2E + 0.5);
    double r = (x - kd * LN2HI) - kd * LN2LO;
    /* Taylor/Horner degree 13 on |r| <= 0.3466: truncation < 1e-17 */
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int64_t k = (int64_t)kd;
    uint64_t bits = (uint64_t)(k + 1023) << 52;
    double scale;
    memcpy(&scale, &bits, sizeof scale);
    return (float)(p * scale);
}

#define IDX3(i, j, k, nx, ny) ((long)(i) + (long)(nx) * (j) + (long)(nx) * (ny) * (k))

/* GPU_kernel.cu:127-144 forward_kernel + :567-574 launcher */
void orc_solve_forward(const float *u, const float *v, const float *w,
                       float *x_fwd, float *y_fwd, float *z_fwd,
                       float h, int ni, int nj, int nk, float cfldt, float dt)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2); k < KHI(NKG(nk) - 2, nk); k++)
        for (int j = 2; j < nj - 2; j++)
            for (int i = 2; i < ni - 2; i++) {
                long id = IDX3(i, j, k, ni, nj);
                f3 q = trace(u, v, w, h, ni, nj, nk, cfldt, dt, mk3(x_fwd[id], y_fwd[id], z_fwd[id]));
                x_fwd[id] = q.x; y_fwd[id] = q.y; z_fwd[id] = q.z;
            }
}

/* GPU_kernel.cu:169-204 DMC_backward_kernel + :576-584 launcher */
static inline float dmc_axis(float p, float vel, float a, float s)
{
    if ((double)fabsf(a) > 1e-4)
        return p - (1.0f - orc_expf(-a * s)) * vel / a;
    return p - vel * s;
}

void orc_solve_backwardDMC(const float *u, const float *v, const float *w,
                           const float *x_in, const float *y_in, const float *z_in,
                           float *x_out, float *y_out, float *z_out,
                           float h, int ni, int nj, int nk, float substep)
{
    const f3 zero = mk3(0.f, 0.f, 0.f);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2); k < KHI(NKG(nk) - 2, nk); k++)
        for (int j = 2; j < nj - 2; j++)
            for (int i = 2; i < ni - 2; i++) {
                long id = IDX3(i, j, k, ni, nj);
                f3 pt = mk3(h * (float)i, h * (float)j, h * (float)(k + KOFF));
                f3 vel = get_velocity(u, v, w, h, ni, nj, nk, pt);
                f3 tp = mk3((vel.x > 0) ? pt.x - h : pt.x + h,
                            (vel.y > 0) ? pt.y - h : pt.y + h,
                            (vel.z > 0) ? pt.z - h : pt.z + h);
                f3 tv = get_velocity(u, v, w, h, ni, nj, nk, tp);
                float ax = (vel.x - tv.x) / (pt.x - tp.x);
                float ay = (vel.y - tv.y) / (pt.y - tp.y);
                float az = (vel.z - tv.z) / (pt.z - tp.z);
                f3 pn = mk3(dmc_axis(pt.x, vel.x, ax, substep),
                            dmc_axis(pt.y, vel.y, ay, substep),
                            dmc_axis(pt.z, vel.z, az, substep));
                x_out[id] = sample(x_in, ni, nj, nk, h, zero, pn);
                y_out[id] = sample(y_in, ni, nj, nk, h, zero, pn);
                z_out[id] = sample(z_in, ni, nj, nk, h, zero, pn);
            }
}

/* ---- the 9-point (8 sub-voxel corners + centre) gather family ------------------
 * Offsets in the reference's order, GPU_kernel.cu:317-327 (identical in all five
 * kernels); is_point collapses to one evaluation at the centre with weight 1.     */
typedef struct { const float *x, *y, *z; } map3;

static inline f3 map_at(map3 m, int ni, int nj, int nk, float h, f3 pos)
{
    const f3 zero = mk3(0.f, 0.f, 0.f);
    return mk3(sample(m.x, ni, nj, nk, h, zero, pos),
               sample(m.y, ni, nj, nk, h, zero, pos),
               sample(m.z, ni, nj, nk, h, zero, pos));
}

typedef struct {
    f3 vol[8];
    int evals;
    float weight;
    f3 origin;
    int nbi, nbj, nbk;
} nine_t;

static nine_t nine_setup(float h, int ni, int nj, int nk, int dx, int dy, int dz, int is_point)
{
    nine_t n;
    float q = 0.25f * h, mq = -0.25f * h;
    n.vol[0] = mk3(q, q, q);   n.vol[1] = mk3(q, q, mq);
    n.vol[2] = mk3(q, mq, q);  n.vol[3] = mk3(q, mq, mq);
    n.vol[4] = mk3(mq, q, q);  n.vol[5] = mk3(mq, q, mq);
    n.vol[6] = mk3(mq, mq, q); n.vol[7] = mk3(mq, mq, mq);
    n.evals = 8;
    if (is_point) { n.vol[0] = mk3(0.f, 0.f, 0.f); n.evals = 1; }
    n.weight = (float)(1.0 / (double)(float)n.evals);
    n.origin = mk3(-(float)dx * 0.5f * h, -(float)dy * 0.5f * h, -(float)dz * 0.5f * h);
    n.nbi = ni + dx; n.nbj = nj + dy; n.nbk = nk + dz;
    return n;
}

static inline f3 nine_pos(const nine_t *n, float h, int i, int j, int k, int ii)
{
    f3 c = mk3((float)i * h + n->origin.x, (float)j * h + n->origin.y, (float)(k + KOFF) * h + n->origin.z);
    if (ii < 0) return c;
    return mk3(c.x + n->vol[ii].x, c.y + n->vol[ii].y, c.z + n->vol[ii].z);
}

/* GPU_kernel.cu:312-374 advect_kernel */
static void advect_comp(float *field, const float *field_init, map3 back,
                        float h, int ni, int nj, int nk, int dx, int dy, int dz, int is_point)
{
    nine_t n = nine_setup(h, ni, nj, nk, dx, dy, dz, is_point);
    f3 lo = mk3(h, h, h), hi = mk3(h * (float)ni - h, h * (float)nj - h, h * (float)NKG(nk) - h);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(3 + dz); k < KHI(NKG(nk) + dz - 3, n.nbk); k++)
        for (int j = 3 + dy; j < n.nbj - 3; j++)
            for (int i = 3 + dx; i < n.nbi - 3; i++) {
                float sum = 0.f;
                for (int ii = 0; ii < n.evals; ii++) {
                    f3 p0 = clamp3(map_at(back, ni, nj, nk, h, nine_pos(&n, h, i, j, k, ii)), lo, hi);
                    sum += n.weight * sample(field_init, n.nbi, n.nbj, n.nbk, h, n.origin, p0);
                }
                f3 pc = clamp3(map_at(back, ni, nj, nk, h, nine_pos(&n, h, i, j, k, -1)), lo, hi);
                float value = sample(field_init, n.nbi, n.nbj, n.nbk, h, n.origin, pc);
                field[IDX3(i, j, k, n.nbi, n.nbj)] = 0.5f * sum + 0.5f * value;
            }
}

/* GPU_kernel.cu:236-310 doubleAdvect_kernel */
/* prev_global: temp_field holds every plane of the grid (a z-slab rank's assembled copy, orc_advect_*_double_global) */
static void double_advect_comp(float *field, const float *temp_field, map3 back, map3 backprev,
                               float h, int ni, int nj, int nk, int dx, int dy, int dz,
                               int is_point, float blend, int prev_global)
{
    nine_t n = nine_setup(h, ni, nj, nk, dx, dy, dz, is_point);
    f3 lo = mk3(h, h, h), hi = mk3(h * (float)ni - h, h * (float)nj - h, h * (float)NKG(nk) - h);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(3 + dz); k < KHI(NKG(nk) + dz - 3, n.nbk); k++)
        for (int j = 3 + dy; j < n.nbj - 3; j++)
            for (int i = 3 + dx; i < n.nbi - 3; i++) {
                float sum = 0.f;
                for (int ii = 0; ii < n.evals; ii++) {
                    f3 mid = clamp3(map_at(back, ni, nj, nk, h, nine_pos(&n, h, i, j, k, ii)), lo, hi);
                    f3 fin = clamp3(map_at(backprev, ni, nj, nk, h, mid), lo, hi);
                    sum += n.weight * (prev_global ? sample_k(temp_field, n.nbi, n.nbj, NKG(nk) + dz, 0, h, n.origin, fin)
                                                   : sample(temp_field, n.nbi, n.nbj, n.nbk, h, n.origin, fin));
                }
                f3 mid = clamp3(map_at(back, ni, nj, nk, h, nine_pos(&n, h, i, j, k, -1)), lo, hi);
                f3 fin = clamp3(map_at(backprev, ni, nj, nk, h, mid), lo, hi);
                float value = prev_global ? sample_k(temp_field, n.nbi, n.nbj, NKG(nk) + dz, 0, h, n.origin, fin)
                                          : sample(temp_field, n.nbi, n.nbj, n.nbk, h, n.origin, fin);
                float prev_value = 0.5f * (sum + value);
                long id = IDX3(i, j, k, n.nbi, n.nbj);
                field[id] = field[id] * blend + (1.0f - blend) * prev_value;
            }
}

/* GPU_kernel.cu:376-436 cumulate_kernel: dst += blend9(coeff * src(map(x))) */
static void cumulate_comp(const float *src, float *dst, map3 m,
                          float h, int ni, int nj, int nk, int dx, int dy, int dz,
                          int is_point, float coeff)
{
    nine_t n = nine_setup(h, ni, nj, nk, dx, dy, dz, is_point);
    f3 lo = mk3(0.f, 0.f, 0.f), hi = mk3(h * (float)ni, h * (float)nj, h * (float)NKG(nk));
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2 + dz); k < KHI(NKG(nk) + dz - 2, n.nbk); k++)
        for (int j = 2 + dy; j < n.nbj - 2; j++)
            for (int i = 2 + dx; i < n.nbi - 2; i++) {
                float sum = 0.f;
                for (int ii = 0; ii < n.evals; ii++) {
                    f3 mp = clamp3(map_at(m, ni, nj, nk, h, nine_pos(&n, h, i, j, k, ii)), lo, hi);
                    sum += n.weight * coeff * sample(src, n.nbi, n.nbj, n.nbk, h, n.origin, mp);
                }
                f3 mp = clamp3(map_at(m, ni, nj, nk, h, nine_pos(&n, h, i, j, k, -1)), lo, hi);
                float value = coeff * sample(src, n.nbi, n.nbj, n.nbk, h, n.origin, mp);
                sum = (float)(0.5 * (double)sum + 0.5 * (double)value);
                dst[IDX3(i, j, k, n.nbi, n.nbj)] += sum;
            }
}

/* z-slab ranks, reference-faithful DMC border (GPU_Advection.h:464-468): the border nodes of the backward
 * map are zero, so the taps of the first and last node layer of cumulate_kernel's window that interpolate towards such
 * a node land at a fraction (1/4, 1/2, 3/4 or a product of those) of their position -- arbitrarily far along z.  A slab
 * rank assembles the cells those taps can touch in `src` (global planes [src_koff, src_koff + src_nk), same rows as the
 * local buffer) and re-evaluates exactly the nodes (i, j, kg) of the window with
 *     i in {xlist}  or  j in {ylist}  or  kg in {zlist}   (wall indices, lists of nx/ny/nz entries)
 * as dst = before + blend9(coeff * src(map(x))): the expression of cumulate_comp with `before` = dst's value ahead of it.
 * Planes: the slab context's, cut to the plane window. */
void orc_accumulate_wall_fixup(const float *src, int src_koff, int src_nk, const float *before, float *dst,
                               const float *mx, const float *my, const float *mz,
                               float h, int ni, int nj, int nk, int axis, float coeff,
                               const int *xlist, int nxl, const int *ylist, int nyl, const int *zlist, int nzl)
{
    map3 m = { mx, my, mz };
    const int dx = axis == 0, dy = axis == 1, dz = axis == 2;
    nine_t n = nine_setup(h, ni, nj, nk, dx, dy, dz, 0);
    f3 lo = mk3(0.f, 0.f, 0.f), hi = mk3(h * (float)ni, h * (float)nj, h * (float)NKG(nk));
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2 + dz); k < KHI(NKG(nk) + dz - 2, n.nbk); k++)
        for (int j = 2 + dy; j < n.nbj - 2; j++)
            for (int i = 2 + dx; i < n.nbi - 2; i++) {
                int hit = 0;
                for (int a = 0; a < nxl; a++) hit |= xlist[a] == i;
                for (int a = 0; a < nyl; a++) hit |= ylist[a] == j;
                for (int a = 0; a < nzl; a++) hit |= zlist[a] == k + KOFF;
                if (!hit) continue;
                float sum = 0.f;
                for (int ii = 0; ii < n.evals; ii++) {
                    f3 mp = clamp3(map_at(m, ni, nj, nk, h, nine_pos(&n, h, i, j, k, ii)), lo, hi);
                    sum += n.weight * coeff * sample_k(src, n.nbi, n.nbj, src_nk, src_koff, h, n.origin, mp);
                }
                f3 mp = clamp3(map_at(m, ni, nj, nk, h, nine_pos(&n, h, i, j, k, -1)), lo, hi);
                float value = coeff * sample_k(src, n.nbi, n.nbj, src_nk, src_koff, h, n.origin, mp);
                sum = (float)(0.5 * (double)sum + 0.5 * (double)value);
                long id = IDX3(i, j, k, n.nbi, n.nbj);
                dst[id] = before[id] + sum;
            }
}

/* GPU_kernel.cu:438-499 compensate_kernel: err = blend9(src(map(x))) - init(x) */
static void compensate_comp(const float *src, const float *init, float *err, map3 m,
                            float h, int ni, int nj, int nk, int dx, int dy, int dz, int is_point)
{
    nine_t n = nine_setup(h, ni, nj, nk, dx, dy, dz, is_point);
    f3 lo = mk3(0.f, 0.f, 0.f), hi = mk3(h * (float)ni, h * (float)nj, h * (float)NKG(nk));
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2 + dz); k < KHI(NKG(nk) + dz - 2, n.nbk); k++)
        for (int j = 2 + dy; j < n.nbj - 2; j++)
            for (int i = 2 + dx; i < n.nbi - 2; i++) {
                float sum = 0.f;
                for (int ii = 0; ii < n.evals; ii++) {
                    f3 mp = clamp3(map_at(m, ni, nj, nk, h, nine_pos(&n, h, i, j, k, ii)), lo, hi);
                    sum += n.weight * sample(src, n.nbi, n.nbj, n.nbk, h, n.origin, mp);
                }
                f3 mp = clamp3(map_at(m, ni, nj, nk, h, nine_pos(&n, h, i, j, k, -1)), lo, hi);
                float value = sample(src, n.nbi, n.nbj, n.nbk, h, n.origin, mp);
                sum = (float)(0.5 * (double)sum + 0.5 * (double)value);
                long id = IDX3(i, j, k, n.nbi, n.nbj);
                err[id] = sum - init[id];
            }
}

/* GPU_kernel.cu:146-167 clampExtrema_kernel (3x3x3 box limiter, interior only) */
/* nkgb: GLOBAL plane count of this buffer (nk on a single domain) */
static void clamp_box(const float *before, float *after, int ni, int nj, int nk, int nkgb)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = imax(1, KLO(1)); k < imin(nk - 1, KHI(nkgb - 1, nk)); k++)
        for (int j = 1; j < nj - 1; j++)
            for (int i = 1; i < ni - 1; i++) {
                long id = IDX3(i, j, k, ni, nj);
                float mx = before[id], mn = before[id];
                for (int kk = k - 1; kk <= k + 1; kk++)
                    for (int jj = j - 1; jj <= j + 1; jj++)
                        for (int ii = i - 1; ii <= i + 1; ii++) {
                            float b = before[IDX3(ii, jj, kk, ni, nj)];
                            if (b > mx) mx = b;
                            if (b < mn) mn = b;
                        }
                after[id] = fminf(fmaxf(mn, after[id]), mx);
            }
}

void orc_clamp_extrema_box(const float *before, float *after, int ni, int nj, int nk)
{
    clamp_box(before, after, ni, nj, nk, NKG(nk));
}

/* GPU_kernel.cu:586-598 */
void orc_advect_velocity(float *u, float *v, float *w,
                         const float *u_init, const float *v_init, const float *w_init,
                         const float *bx, const float *by, const float *bz,
                         float h, int ni, int nj, int nk, int is_point)
{
    map3 b = { bx, by, bz };
    advect_comp(u, u_init, b, h, ni, nj, nk, 1, 0, 0, is_point);
    advect_comp(v, v_init, b, h, ni, nj, nk, 0, 1, 0, is_point);
    advect_comp(w, w_init, b, h, ni, nj, nk, 0, 0, 1, is_point);
}

/* GPU_kernel.cu:600-618 */
void orc_advect_vel_double(float *u, float *v, float *w,
                           const float *utemp, const float *vtemp, const float *wtemp,
                           const float *bx, const float *by, const float *bz,
                           const float *bxp, const float *byp, const float *bzp,
                           float h, int ni, int nj, int nk, int is_point, float blend)
{
    map3 b = { bx, by, bz }, bp = { bxp, byp, bzp };
    double_advect_comp(u, utemp, b, bp, h, ni, nj, nk, 1, 0, 0, is_point, blend, 0);
    double_advect_comp(v, vtemp, b, bp, h, ni, nj, nk, 0, 1, 0, is_point, blend, 0);
    double_advect_comp(w, wtemp, b, bp, h, ni, nj, nk, 0, 0, 1, is_point, blend, 0);
}

/* the same on a z-slab rank (orc_set_slab) with the *_prev fields of the whole grid: include/bimocq_gpu.h,
 * gpu_advect_vel_double_global */
void orc_advect_vel_double_global(float *u, float *v, float *w,
                                  const float *uprev_g, const float *vprev_g, const float *wprev_g,
                                  const float *bx, const float *by, const float *bz,
                                  const float *bxp, const float *byp, const float *bzp,
                                  float h, int ni, int nj, int nk, int is_point, float blend)
{
    map3 b = { bx, by, bz }, bp = { bxp, byp, bzp };
    double_advect_comp(u, uprev_g, b, bp, h, ni, nj, nk, 1, 0, 0, is_point, blend, 1);
    double_advect_comp(v, vprev_g, b, bp, h, ni, nj, nk, 0, 1, 0, is_point, blend, 1);
    double_advect_comp(w, wprev_g, b, bp, h, ni, nj, nk, 0, 0, 1, is_point, blend, 1);
}
void orc_advect_field_double_global(float *field, const float *field_prev_g,
                                    const float *bx, const float *by, const float *bz,
                                    const float *bxp, const float *byp, const float *bzp,
                                    float h, int ni, int nj, int nk, int is_point, float blend)
{
    map3 b = { bx, by, bz }, bp = { bxp, byp, bzp };
    double_advect_comp(field, field_prev_g, b, bp, h, ni, nj, nk, 0, 0, 0, is_point, blend, 1);
}

/* GPU_kernel.cu:620-627 */
void orc_advect_field(float *field, const float *field_init,
                      const float *bx, const float *by, const float *bz,
                      float h, int ni, int nj, int nk, int is_point)
{
    map3 b = { bx, by, bz };
    advect_comp(field, field_init, b, h, ni, nj, nk, 0, 0, 0, is_point);
}

/* GPU_kernel.cu:629-638 */
void orc_advect_field_double(float *field, const float *field_prev,
                             const float *bx, const float *by, const float *bz,
                             const float *bxp, const float *byp, const float *bzp,
                             float h, int ni, int nj, int nk, int is_point, float blend)
{
    map3 b = { bx, by, bz }, bp = { bxp, byp, bzp };
    double_advect_comp(field, field_prev, b, bp, h, ni, nj, nk, 0, 0, 0, is_point, blend, 0);
}

/* GPU_kernel.cu:640-666.  du/dv/dw are read as `init` and then overwritten with the
 * uncompensated field; u_src.. must be zeroed by the caller
 * (GPU_Advection.h:499-501). */
void orc_compensate_velocity(float *u, float *v, float *w,
                             float *du, float *dv, float *dw,
                             float *u_src, float *v_src, float *w_src,
                             const float *fx, const float *fy, const float *fz,
                             const float *bx, const float *by, const float *bz,
                             float h, int ni, int nj, int nk, int is_point)
{
    map3 f = { fx, fy, fz }, b = { bx, by, bz };
    compensate_comp(u, du, u_src, f, h, ni, nj, nk, 1, 0, 0, is_point);
    compensate_comp(v, dv, v_src, f, h, ni, nj, nk, 0, 1, 0, is_point);
    compensate_comp(w, dw, w_src, f, h, ni, nj, nk, 0, 0, 1, is_point);
    memcpy(du, u, sizeof(float) * (size_t)(ni + 1) * nj * nk);
    memcpy(dv, v, sizeof(float) * (size_t)ni * (nj + 1) * nk);
    memcpy(dw, w, sizeof(float) * (size_t)ni * nj * (nk + 1));
    cumulate_comp(u_src, u, b, h, ni, nj, nk, 1, 0, 0, is_point, -0.5f);
    cumulate_comp(v_src, v, b, h, ni, nj, nk, 0, 1, 0, is_point, -0.5f);
    cumulate_comp(w_src, w, b, h, ni, nj, nk, 0, 0, 1, is_point, -0.5f);
    clamp_box(du, u, ni + 1, nj, nk, NKG(nk));
    clamp_box(dv, v, ni, nj + 1, nk, NKG(nk));
    clamp_box(dw, w, ni, nj, nk + 1, NKG(nk) + 1);
}

/* stage 1 of the two compensate operators on its own (GPU_kernel.cu:652-654 / :676), and the
 * limiter for the w buffer: used by the host solver's four-stage form of the operator */
void orc_compensate_error_velocity(const float *u, const float *v, const float *w,
                                   const float *du, const float *dv, const float *dw,
                                   float *u_src, float *v_src, float *w_src,
                                   const float *fx, const float *fy, const float *fz,
                                   float h, int ni, int nj, int nk, int is_point)
{
    map3 f = { fx, fy, fz };
    compensate_comp(u, du, u_src, f, h, ni, nj, nk, 1, 0, 0, is_point);
    compensate_comp(v, dv, v_src, f, h, ni, nj, nk, 0, 1, 0, is_point);
    compensate_comp(w, dw, w_src, f, h, ni, nj, nk, 0, 0, 1, is_point);
}
void orc_compensate_error_field(const float *u, const float *du, float *u_src,
                                const float *fx, const float *fy, const float *fz,
                                float h, int ni, int nj, int nk, int is_point)
{
    map3 f = { fx, fy, fz };
    compensate_comp(u, du, u_src, f, h, ni, nj, nk, 0, 0, 0, is_point);
}
void orc_clamp_extrema_box_w(const float *before, float *after, int ni, int nj, int nk_buffer)
{
    clamp_box(before, after, ni, nj, nk_buffer, NKG(nk_buffer - 1) + 1);
}

/* GPU_kernel.cu:668-682; the (ni+1)*nj*nk-sized copy of the reference (Q4) is not
 * replicated: the scalar buffers hold ni*nj*nk floats. */
void orc_compensate_field(float *u, float *du, float *u_src,
                          const float *fx, const float *fy, const float *fz,
                          const float *bx, const float *by, const float *bz,
                          float h, int ni, int nj, int nk, int is_point)
{
    map3 f = { fx, fy, fz }, b = { bx, by, bz };
    compensate_comp(u, du, u_src, f, h, ni, nj, nk, 0, 0, 0, is_point);
    memcpy(du, u, sizeof(float) * (size_t)ni * nj * nk);
    cumulate_comp(u_src, u, b, h, ni, nj, nk, 0, 0, 0, is_point, -0.5f);
    clamp_box(du, u, ni, nj, nk, NKG(nk));
}

/* GPU_kernel.cu:684-696 */
void orc_accumulate_velocity(const float *uc, const float *vc, const float *wc,
                             float *du_init, float *dv_init, float *dw_init,
                             const float *fx, const float *fy, const float *fz,
                             float h, int ni, int nj, int nk, int is_point, float coeff)
{
    map3 f = { fx, fy, fz };
    cumulate_comp(uc, du_init, f, h, ni, nj, nk, 1, 0, 0, is_point, coeff);
    cumulate_comp(vc, dv_init, f, h, ni, nj, nk, 0, 1, 0, is_point, coeff);
    cumulate_comp(wc, dw_init, f, h, ni, nj, nk, 0, 0, 1, is_point, coeff);
}

/* one component (axis 0/1/2) of the launch trio above */
void orc_accumulate_component(const float *change, float *d_init, const float *fx, const float *fy, const float *fz,
                              float h, int ni, int nj, int nk, int axis, int is_point, float coeff)
{
    map3 f = { fx, fy, fz };
    cumulate_comp(change, d_init, f, h, ni, nj, nk, axis == 0, axis == 1, axis == 2, is_point, coeff);
}

/* GPU_kernel.cu:698-705 */
void orc_accumulate_field(const float *change, float *dfield_init,
                          const float *fx, const float *fy, const float *fz,
                          float h, int ni, int nj, int nk, int is_point, float coeff)
{
    map3 f = { fx, fy, fz };
    cumulate_comp(change, dfield_init, f, h, ni, nj, nk, 0, 0, 0, is_point, coeff);
}

/* GPU_kernel.cu:501-537 estimate_kernel + :707-716 */
void orc_estimate_distortion(float *dist,
                             const float *xb, const float *yb, const float *zb,
                             const float *xf, const float *yf, const float *zf,
                             float h, int ni, int nj, int nk)
{
    map3 first = { xb, yb, zb }, second = { xf, yf, zf };
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2); k < KHI(NKG(nk) - 2, nk); k++)
        for (int j = 2; j < nj - 2; j++)
            for (int i = 2; i < ni - 2; i++) {
                f3 pt = mk3(h * (float)i, h * (float)j, h * (float)(k + KOFF));
                f3 back = map_at(first, ni, nj, nk, h, pt);
                f3 fwd = map_at(second, ni, nj, nk, h, back);
                float d_bf = (pt.x - fwd.x) * (pt.x - fwd.x) + (pt.y - fwd.y) * (pt.y - fwd.y)
                           + (pt.z - fwd.z) * (pt.z - fwd.z);
                f3 f2 = map_at(second, ni, nj, nk, h, pt);
                f3 b2 = map_at(first, ni, nj, nk, h, f2);
                float d_fb = (pt.x - b2.x) * (pt.x - b2.x) + (pt.y - b2.y) * (pt.y - b2.y)
                           + (pt.z - b2.z) * (pt.z - b2.z);
                dist[IDX3(i, j, k, ni, nj)] = fmaxf(d_bf, d_fb);
            }
}

/* GPU_kernel.cu:206-233 semilag_kernel + :718-727 */
void orc_semilag(float *field, const float *field_src,
                 const float *u, const float *v, const float *w,
                 int dx, int dy, int dz,
                 float h, int ni, int nj, int nk, float cfldt, float dt)
{
    f3 org = mk3(-(float)dx * 0.5f * h, -(float)dy * 0.5f * h, -(float)dz * 0.5f * h);
    int bi = ni + dx, bj = nj + dy, bk = nk + dz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2); k < KHI(NKG(nk) - 2, bk); k++)
        for (int j = 2; j < bj - 2 - dy; j++)
            for (int i = 2; i < bi - 2 - dx; i++) {
                f3 pt = mk3(h * (float)i + org.x, h * (float)j + org.y, h * (float)(k + KOFF) + org.z);
                f3 pn = trace(u, v, w, h, ni, nj, nk, cfldt, dt, pt);
                field[IDX3(i, j, k, bi, bj)] = sample(field_src, bi, bj, bk, h, org, pn);
            }
}

/* gpu_clamp_extrema / clamp_extrema_kernel (GPU_kernel.cu:892-950), the MacCormack limiter of the
 * reflection scheme, CORRECTED.  As written the reference kernel (a) adds the stagger offset
 * with the wrong sign, (b) uses the departure point's WORLD coordinates as grid indices (:913-915) and
 * (c) tests and overwrites fieldTemp at that bogus index from every thread at once -- its output is
 * undefined.  What the code evidently means, and what is built here:
 *   node x = (i - o) h; departure point by the kernel's own midpoint rule, x_d = x - dt u(x - dt/2 u(x)),
 *   clamped to [h, (n-1)h] like every other trace; the 8 values of `field` around x_d give min/max; if
 *   fieldTemp at THIS node lies outside, it is replaced by the trilinear value of `field` at x_d.
 * ni, nj, nk are BUFFER dims (cells + dim), o = (ox, oy, oz) = 0.5 along the staggered axis. */
void orc_clamp_extrema(const float *field, float *field_temp, const float *u, const float *v, const float *w,
                       int ni, int nj, int nk, int dimx, int dimy, int dimz, float ox, float oy, float oz,
                       float h, float dt)
{
    const int ci = ni - dimx, cj = nj - dimy, ck = nk - dimz;
    const f3 org = mk3(-ox * h, -oy * h, -oz * h);
    /* z-slab context: nk is the LOCAL buffer's plane count, local plane k is global plane k + KOFF */
    const f3 lo = mk3(h, h, h), hi = mk3((float)ci * h - h, (float)cj * h - h, (float)NKG(ck) * h - h);
    const float halfdt = 0.5f * dt;
    const long sj = ni, sk = (long)ni * nj, count = (long)ni * nj * nk;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nk; k++)
        for (int j = 0; j < nj; j++)
            for (int i = 0; i < ni; i++) {
                f3 pt = mk3(h * (float)i + org.x, h * (float)j + org.y, h * (float)(k + KOFF) + org.z);
                f3 vel = get_velocity(u, v, w, h, ci, cj, ck, pt);
                f3 px = mk3(pt.x - vel.x * halfdt, pt.y - vel.y * halfdt, pt.z - vel.z * halfdt);
                vel = get_velocity(u, v, w, h, ci, cj, ck, px);
                px = clamp3(mk3(pt.x - vel.x * dt, pt.y - vel.y * dt, pt.z - vel.z * dt), lo, hi);
                float qx = (px.x - org.x) / h, qy = (px.y - org.y) / h, qz = (px.z - org.z) / h;
                int gi = (int)floorf(qx), gj = (int)floorf(qy), gk = (int)floorf(qz);
                float cx = qx - (float)gi, cy = qy - (float)gj, cz = qz - (float)gk;
                long base = (long)gi + sj * gj + sk * (gk - KOFF);
                if (base < 0) base = count;
                float v0 = ld(field, base, count),           v1 = ld(field, base + 1, count);
                float v2 = ld(field, base + sj, count),      v3 = ld(field, base + sj + 1, count);
                float v4 = ld(field, base + sk, count),      v5 = ld(field, base + sk + 1, count);
                float v6 = ld(field, base + sk + sj, count), v7 = ld(field, base + sk + sj + 1, count);
                float mn = fminf(v0, fminf(v1, fminf(v2, fminf(v3, fminf(v4, fminf(v5, fminf(v6, v7)))))));
                float mx = fmaxf(v0, fmaxf(v1, fmaxf(v2, fmaxf(v3, fmaxf(v4, fmaxf(v5, fmaxf(v6, v7)))))));
                long id = (long)i + sj * j + sk * k;
                float t = field_temp[id];
                if (t < mn || t > mx)
                    field_temp[id] = trilerp(v0, v1, v2, v3, v4, v5, v6, v7, cx, cy, cz);
            }
}

/* GPU_kernel.cu:560-565 + :729-734 (exactly `number` elements; the reference's
 * rounded-up grid overrun is not replicated) */
void orc_add(float *f1, const float *f2, float coeff, int number)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < number; i++) f1[i] += coeff * f2[i];
}

/* GPU_kernel.cu:878-890 */
void orc_add_field(float *out, const float *f1, const float *f2, float coeff, int number)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < number; i++) out[i] = f1[i] + coeff * f2[i];
}

/* GPU_kernel.cu:952-964 */
void orc_mad(float *out, const float *f1, const float *f2, float c1, float c2, int number)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < number; i++) out[i] = c1 * f1[i] + c2 * f2[i];
}

/* GPU_kernel.cu:736-758 emit_smoke_velocity_kernel.  norm3df and hypotf are taken as the
 * correctly-rounded sqrt of the double sum of squares (identical in the HIP path; within 1 ulp
 * of any libm, and |y| <= hypot2(y,z) always holds so acosf never sees |ratio| > 1);
 * note the u-face offset (i-1/2)h is used for every component. */
static inline float norm3(float x, float y, float z)
{
    return (float)sqrt((double)x * (double)x + (double)y * (double)y + (double)z * (double)z);
}
static inline float hypot2(float y, float z)
{
    return (float)sqrt((double)y * (double)y + (double)z * (double)z);
}

/* acosf / cosf of the emitter's velocity ring (GPU_kernel.cu:750-752), restated with IEEE double +, -, *, /, sqrt and
 * floor only, so that every compiler and both sides of the parity tests produce the same bits (within 1 ulp of any
 * libm).  acos: asin's Taylor series on |x| <= 1/2 (22 terms, c_k = C(2k,k) / (4^k (2k+1)): truncation < 1e-15),
 * acos(r) = pi/2 - asin(r) for |r| <= 1/2, 2 asin(sqrt((1-|r|)/2)) beyond, reflected for r < 0.  cos: Cody-Waite
 * reduction by pi/2 (two-part constant, exact for the |x| < 2^19 this is specified on; the emitter passes 8 theta <=
 * 8 pi), Taylor series of sin / cos on |y| <= pi/4. */
static inline float orc_acosf(float rf)
{
    double r = (double)rf;
    if (!(r == r)) return rf;
    if (r > 1.0) r = 1.0;
    if (r < -1.0) r = -1.0;
    const double PI = 3.14159265358979311600e+00;
    const double PIO2 = 1.57079632679489655800e+00;
    double a = r < 0.0 ? -r : r;
    double x = a <= 0.5 ? a : sqrt((1.0 - a) * 0.5);
    double z = x * x;
    double p = 2104098963720.0 / 791648371998720.0;
    p = p * z + 538257874440.0 / 189115999977472.0;
    p = p * z + 137846528820.0 / 45079976738816.0;
    p = p * z + 35345263800.0 / 10720238370816.0;
    p = p * z + 9075135300.0 / 2542620639232.0;
    p = p * z + 2333606220.0 / 601295421440.0;
    p = p * z + 601080390.0 / 141733920768.0;
    p = p * z + 155117520.0 / 33285996544.0;
    p = p * z + 40116600.0 / 7784628224.0;
    p = p * z + 10400600.0 / 1811939328.0;
    p = p * z + 2704156.0 / 419430400.0;
    p = p * z + 705432.0 / 96468992.0;
    p = p * z + 184756.0 / 22020096.0;
    p = p * z + 48620.0 / 4980736.0;
    p = p * z + 12870.0 / 1114112.0;
    p = p * z + 3432.0 / 245760.0;
    p = p * z + 924.0 / 53248.0;
    p = p * z + 252.0 / 11264.0;
    p = p * z + 70.0 / 2304.0;
    p = p * z + 20.0 / 448.0;
    p = p * z + 6.0 / 80.0;
    p = p * z + 2.0 / 12.0;
    double s = x + x * (z * p);                 /* asin(x) */
    double res;
    if (a <= 0.5) res = r < 0.0 ? PIO2 + s : PIO2 - s;
    else res = r < 0.0 ? PI - 2.0 * s : 2.0 * s;
    return (float)res;
}

static inline float orc_cosf(float xf)
{
    double x = (double)xf;
    if (!(x == x)) return xf;
    if (x < 0.0) x = -x;
    if (!(x < 524288.0)) return (float)(x - x);  /* outside the specified range (Inf -> NaN, huge -> 0) */
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
    const double PIO2_LO = 6.07710050650619224932e-11;   /* pi/2 - PIO2_HI */
    double kd = floor(x * TWO_OVER_PI + 0.5);
    double y = (x - kd * PIO2_HI) - kd * PIO2_LO;
    double z = y * y;
    double c = -1.0 / 6402373705728000.0;        /* cos: sum (-1)^m z^m / (2m)!, m <= 9 */
    c = c * z + 1.0 / 20922789888000.0;
    c = c * z - 1.0 / 87178291200.0;
    c = c * z + 1.0 / 479001600.0;
    c = c * z - 1.0 / 3628800.0;
    c = c * z + 1.0 / 40320.0;
    c = c * z - 1.0 / 720.0;
    c = c * z + 1.0 / 24.0;
    c = c * z - 0.5;
    c = c * z + 1.0;
    double s = 1.0 / 355687428096000.0;          /* sin: y sum (-1)^m z^m / (2m+1)!, m <= 8 */
    s = s * z - 1.0 / 1307674368000.0;
    s = s * z + 1.0 / 6227020800.0;
    s = s * z - 1.0 / 39916800.0;
    s = s * z + 1.0 / 362880.0;
    s = s * z - 1.0 / 5040.0;
    s = s * z + 1.0 / 120.0;
    s = s * z - 1.0 / 6.0;
    s = y + y * (z * s);
    long long q = (long long)kd & 3;
    double res = q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
    return (float)res;
}

static void emit_velocity(float *field, float h, int ni, int nj, int nk, int nkgb,
                          float cx, float cy, float cz, float radius, float emiter)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2); k < KHI(nkgb - 2, nk); k++)
        for (int j = 2; j < nj - 2; j++)
            for (int i = 2; i < ni - 2; i++) {
                float dxp = (float)(((double)(float)i - 0.5) * (double)h - (double)cx);
                float dyp = (float)j * h - cy;
                float dzp = (float)(k + KOFF) * h - cz;
                if (norm3(dxp, dyp, dzp) < radius) {
                    float theta = orc_acosf(dyp / hypot2(dyp, dzp));
                    float c8 = orc_cosf((float)(8.0 * (double)theta));
                    field[IDX3(i, j, k, ni, nj)] =
                        (float)((double)emiter * 0.06 * (1.0 + 0.01 * (double)c8));
                }
            }
}

/* GPU_kernel.cu:760-780 emit_smoke_field_kernel */
static void emit_field(float *rho, float *T, float h, int ni, int nj, int nk,
                       float cx, float cy, float cz, float radius, float density, float temperature)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(2); k < KHI(NKG(nk) - 2, nk); k++)
        for (int j = 2; j < nj - 2; j++)
            for (int i = 2; i < ni - 2; i++) {
                float dxp = (float)i * h - cx, dyp = (float)j * h - cy, dzp = (float)(k + KOFF) * h - cz;
                if (norm3(dxp, dyp, dzp) < radius) {
                    long id = IDX3(i, j, k, ni, nj);
                    rho[id] = density;
                    T[id] = temperature;
                }
            }
}

/* GPU_kernel.cu:782-802 */
void orc_emit_smoke(float *u, float *v, float *w, float *rho, float *T,
                    float h, int ni, int nj, int nk,
                    float cx, float cy, float cz, float radius,
                    float density, float temperature, float emiter)
{
    emit_velocity(u, h, ni + 1, nj, nk, NKG(nk), cx, cy, cz, radius, emiter);
    emit_velocity(v, h, ni, nj + 1, nk, NKG(nk), cx, cy, cz, radius, 0.f);
    emit_velocity(w, h, ni, nj, nk + 1, NKG(nk) + 1, cx, cy, cz, radius, 0.f);
    emit_field(rho, T, h, ni, nj, nk, cx, cy, cz, radius, density, temperature);
}

/* GPU_kernel.cu:804-832 add_buoyancy with the INTENDED indexing (fixed not
 * replicated): the reference reads rho/T with the v-buffer flat index, which is only
 * right on the k=0 slab.  v(i,j,k) += 0.5*dt*(beta*(T(j)+T(j-1)) - alpha*(rho(j)+rho(j-1)))
 * for 1 <= j <= nj-1 (the faces that have a cell on both sides). */
void orc_add_buoyancy(float *v, const float *rho, const float *T,
                      int ni, int nj, int nk, float alpha, float beta, float dt)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(0); k < KHI(NKG(nk), nk); k++)
        for (int j = 1; j < nj; j++)
            for (int i = 0; i < ni; i++) {
                long c0 = IDX3(i, j, k, ni, nj), c1 = IDX3(i, j - 1, k, ni, nj);
                float d0 = rho[c0], T0 = T[c0], d1 = rho[c1], T1 = T[c1];
                float f = (float)(0.5 * (double)dt * (double)(beta * (T0 + T1) - alpha * (d0 + d1)));
                v[IDX3(i, j, k, ni, nj + 1)] += f;
            }
}

/* GPU_kernel.cu:834-876 gpu_diffuse_field: tmp0 <- field; iter ping-pong sweeps; field <-
 * the buffer that was the INPUT of the last sweep (iterate iter-1). */
void orc_diffuse_field(float *field, float *tmp0, float *tmp1,
                       int ni, int nj, int nk, int iter, float coef)
{
    size_t number = (size_t)ni * nj * nk;
    memcpy(tmp0, field, number * sizeof(float));
    int where = orc_diffuse_sweeps(field, tmp0, tmp1, ni, nj, nk, iter, coef);
    /* after the loop `out` is the buffer that was the INPUT of the last sweep */
    memcpy(field, where ? tmp0 : tmp1, number * sizeof(float));
}

/* the sweep loop of gpu_diffuse_field on its own; returns 0 if the newest iterate is in `in`, 1 if in `out` */
int orc_diffuse_sweeps(const float *field, float *in0, float *out0, int ni, int nj, int nk, int iter, float coef)
{
    float *in = in0, *out = out0;
    /* nk is a BUFFER dim (nk+1 for w): in slab mode the global plane count keeps that +1 */
    const int nkgb = S_on ? S_nkg + (nk - S_nkl) : nk;
    for (int it = 0; it < iter; it++) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int k = imax(1, KLO(1)); k < imin(nk - 1, KHI(nkgb - 1, nk)); k++)
            for (int j = 1; j < nj - 1; j++)
                for (int i = 1; i < ni - 1; i++) {
                    long id = IDX3(i, j, k, ni, nj);
                    float s = in[id - 1] + in[id + 1] + in[id - ni] + in[id + ni]
                            + in[id - (long)ni * nj] + in[id + (long)ni * nj];
                    out[id] = (field[id] + coef * s) / (1.0f + 6.0f * coef);
                }
        float *t = out; out = in; in = t;
    }
    return in == in0 ? 0 : 1;
}

/* GPU_kernel.cu:967-985 divergence_kernel (float) */
void orc_divergence(const float *u, const float *v, const float *w, float *div,
                    int ni, int nj, int nk, float halfrdx)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = KLO(0); k < KHI(NKG(nk), nk); k++)
        for (int j = 0; j < nj; j++)
            for (int i = 0; i < ni; i++) {
                float ul = u[IDX3(i, j, k, ni + 1, nj)], ur = u[IDX3(i + 1, j, k, ni + 1, nj)];
                float vf = v[IDX3(i, j, k, ni, nj + 1)], vb = v[IDX3(i, j + 1, k, ni, nj + 1)];
                float wd = w[IDX3(i, j, k, ni, nj)],     wu = w[IDX3(i, j, k + 1, ni, nj)];
                div[IDX3(i, j, k, ni, nj)] = halfrdx * ((ur - ul) + (vb - vf) + (wu - wd));
            }
}

/* GPU_kernel.cu:1819-1837 jacobi_kernel: interior only, fixed summation order */
void orc_jacobi_sweep(const float *p, const float *div, float *out,
                      int ni, int nj, int nk, float alpha, float beta)
{
    orc_jacobi_sweep_range(p, div, out, ni, nj, nk, 0, nk, alpha, beta);
}

/* the same sweep restricted to the local planes [k_begin, k_end) (slab hosts: interior first) */
void orc_jacobi_sweep_range(const float *p, const float *div, float *out,
                            int ni, int nj, int nk, int k_begin, int k_end, float alpha, float beta)
{
    long sj = ni, sk = (long)ni * nj;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = imax(imax(1, KLO(1)), k_begin); k < imin(imin(nk - 1, KHI(NKG(nk) - 1, nk)), k_end); k++)
        for (int j = 1; j < nj - 1; j++)
            for (int i = 1; i < ni - 1; i++) {
                long id = IDX3(i, j, k, ni, nj);
                out[id] = (p[id - 1] + p[id + 1] + p[id - sj] + p[id + sj] + p[id - sk] + p[id + sk]
                           + alpha * div[id]) * beta;
            }
}

/* GPU_kernel.cu:1024-1041 gradient_kernel (float): nb* are BUFFER dims, window 2..cell-1 */
void orc_gradient(float *field, const float *p, int nbi, int nbj, int nbk,
                  int dx, int dy, int dz, float halfrdx)
{
    int pi = nbi - dx, pj = nbj - dy, pk = nbk - dz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = imax(dz, KLO(2)); k < KHI(NKG(pk), pk); k++)
        for (int j = 2; j < pj; j++)
            for (int i = 2; i < pi; i++) {
                float p0 = p[IDX3(i, j, k, pi, pj)];
                float p1 = p[IDX3(i - dx, j - dy, k - dz, pi, pj)];
                field[IDX3(i, j, k, nbi, nbj)] -= halfrdx * (p0 - p1);
            }
}

/* A15 re-specified: r = b - (sum6 x - 6x) as update_residual_kernel
 * (GPU_kernel.cu:1239-1249, calc_poisson_value :1048-1060), exact sum r^2 and max |r|. */
void orc_residual_norms(const float *div, const float *p, int ni, int nj, int nk,
                        double *sum_sq, float *max_abs)
{
    long sj = ni, sk = (long)ni * nj;
    double ss = 0.0;
    float mx = 0.f;
    for (int k = imax(1, KLO(S_on ? imax(1, S_own0) : 1)); k < imin(nk - 1, KHI(S_on ? imin(S_nkg - 1, S_own1) : nk - 1, nk)); k++)
        for (int j = 1; j < nj - 1; j++)
            for (int i = 1; i < ni - 1; i++) {
                long id = IDX3(i, j, k, ni, nj);
                float ax = (p[id - 1] + p[id + 1] + p[id - sj] + p[id + sj] + p[id - sk] + p[id + sk])
                         - p[id] * 6;
                float r = div[id] - ax;
                ss += (double)r * (double)r;
                if (fabsf(r) > mx) mx = fabsf(r);
            }
    *sum_sq = ss;
    *max_abs = mx;
}

/* GPU_kernel.cu:1839-1895 gpu_projection_jacobi.  The reference runs `iter` sweeps and
 * then applies the OLDER buffer (iterate iter-1) in the gradient and leaves it in p
 * (SURVEY Q1); restated directly: iter-1 sweeps, result in p. */
void orc_projection_jacobi(float *u, float *v, float *w, float *div, float *p, float *p_temp,
                           float *debug, int ni, int nj, int nk, int iter,
                           float halfrdx, float alpha, float beta)
{
    size_t number = (size_t)ni * nj * nk;
    orc_divergence(u, v, w, div, ni, nj, nk, halfrdx);
    float *p_in = p, *p_out = p_temp;
    for (int it = 0; it + 1 < iter; it++) {
        if (debug) {
            double ss; float mx;
            orc_residual_norms(div, p_in, ni, nj, nk, &ss, &mx);
            debug[it] = (float)ss; debug[2000 + it] = mx;
        }
        orc_jacobi_sweep(p_in, div, p_out, ni, nj, nk, alpha, beta);
        float *t = p_in; p_in = p_out; p_out = t;
    }
    if (debug && iter > 0) {
        double ss; float mx;
        orc_residual_norms(div, p_in, ni, nj, nk, &ss, &mx);
        debug[iter - 1] = (float)ss; debug[2000 + iter - 1] = mx;
    }
    /* iter == 0: the swap loop does not run, p_out is still p_temp and is copied over p (GPU_kernel.cu:1876-1879) */
    if (iter == 0) memcpy(p, p_temp, number * sizeof(float));
    else if (p_in != p) memcpy(p, p_in, number * sizeof(float));
    orc_gradient(u, p, ni + 1, nj, nk, 1, 0, 0, halfrdx);
    orc_gradient(v, p, ni, nj + 1, nk, 0, 1, 0, halfrdx);
    orc_gradient(w, p, ni, nj, nk + 1, 0, 0, 1, halfrdx);
}

/* BimocqGPUSolver.cpp:348-373 getCFL: max(1e-4, max|u|,|v|,|w|) */
float orc_max_abs3(const float *u, const float *v, const float *w, int ni, int nj, int nk)
{
    float m = 1e-4f;
    /* a slab rank scans the planes it owns (the last rank also owns w's top plane); the caller all-reduces */
    const int p0 = S_on ? S_own0 - S_koff : 0, p1 = S_on ? S_own1 - S_koff : nk;
    const int wtop = (!S_on || S_own1 == S_nkg) ? 1 : 0;
    size_t pu = (size_t)(ni + 1) * nj, pv = (size_t)ni * (nj + 1), pw = (size_t)ni * nj;
    const float *u0 = u + pu * p0, *v0 = v + pv * p0, *w0 = w + pw * p0;
    size_t nu = pu * (p1 - p0), nv = pv * (p1 - p0), nw = pw * (p1 - p0 + wtop);
    for (size_t i = 0; i < nu; i++) if (fabsf(u0[i]) > m) m = fabsf(u0[i]);
    for (size_t i = 0; i < nv; i++) if (fabsf(v0[i]) > m) m = fabsf(v0[i]);
    for (size_t i = 0; i < nw; i++) if (fabsf(w0[i]) > m) m = fabsf(w0[i]);
    return m;
}

/* =====================  step state machine  ===================================== */

/* Mapping.h:92-96 MapperBaseGPU state */
typedef struct {
    float *fx, *fy, *fz, *bx, *by, *bz, *bpx, *bpy, *bpz, *ix, *iy, *iz;
    unsigned total_reinit;
} mapper_t;

struct orc_solver {
    int ni, nj, nk;
    size_t n, nu, nv, nw;
    float h, viscosity, blend, alpha, beta;
    int jacobi_iters; float halfrdx;
    /* projection(): 0 = Jacobi (`#if 0` branch), 1 = fp64 multigrid-CG (the `#else` branch, :443-446) */
    int projection_kind, mg_iters, mg_levels;
    /* 0: re-initialise both map sets every frame (the GPU solver's `if (1)`, BimocqGPUSolver.cpp:218-229);
     * 1: distortion-driven, the CPU solver's thresholds (BimocqSolver.cpp:165-185) */
    int reinit_policy;
    int travel_limit, forced_reinits;       /* option 4 (host: BQ_OPT_REINIT_MAX_TRAVEL), see solver_advance */
    int scheme;                 /* enum Scheme (BimocqSolver.h:29): 0 BIMOCQ, 3 MAC_REFLECTION */
    float max_v, last_vel_distortion, last_scalar_distortion;
    int vel_reinits, scalar_reinits;
    double *mg_div, *mg_p, *mg_dir, *mg_res, *mg_t0, *mg_t1, *mg_result;
    OrcCoarseLevel mg_level[6];
    /* BimocqGPUSolver.h:74-87 buffer roles */
    float *U, *V, *W, *Ui, *Vi, *Wi, *Up, *Vp, *Wp, *Ut, *Vt, *Wt;
    float *dUp, *dVp, *dWp, *dUe, *dVe, *dWe, *Su, *Sv, *Sw;
    float *rho, *rhoi, *rhop, *rhot, *rhoe, *T, *Ti, *Tp, *Tt, *Te;
    float *div, *p, *pt;
    /* gpuMapper scratch: GPU_Advection.h:122-136 */
    float *u_src, *v_src, *w_src, *xo, *yo, *zo;
    mapper_t vel, scal;
    int vel_last, scal_last;        /* BimocqGPUSolver.h:109-110 */
    int keep_dmc_border;            /* 0 = reference (border of the DMC scratch is zero), 1 = keep */
    orc_emitter *em; int n_em;
    float last_cfldt;
};

static float *zalloc(size_t n) { return (float *)calloc(n ? n : 1, sizeof(float)); }

/* Mapping.cpp:276-345 MapperBaseGPU::init: maps start as (i*h, j*h, k*h) */
static void mapper_init(mapper_t *m, int ni, int nj, int nk, float h)
{
    size_t n = (size_t)ni * nj * nk;
    float **all[] = { &m->fx, &m->fy, &m->fz, &m->bx, &m->by, &m->bz,
                      &m->bpx, &m->bpy, &m->bpz, &m->ix, &m->iy, &m->iz };
    for (int a = 0; a < 12; a++) *all[a] = zalloc(n);
    for (int k = 0; k < nk; k++)
        for (int j = 0; j < nj; j++)
            for (int i = 0; i < ni; i++) {
                long id = IDX3(i, j, k, ni, nj);
                m->ix[id] = (float)i * h; m->iy[id] = (float)j * h; m->iz[id] = (float)k * h;
            }
    float *dst[] = { m->fx, m->fy, m->fz, m->bx, m->by, m->bz, m->bpx, m->bpy, m->bpz };
    float *src[] = { m->ix, m->iy, m->iz };
    for (int a = 0; a < 9; a++) memcpy(dst[a], src[a % 3], n * sizeof(float));
    m->total_reinit = 0;
}

static void mapper_free(mapper_t *m)
{
    float *all[] = { m->fx, m->fy, m->fz, m->bx, m->by, m->bz, m->bpx, m->bpy, m->bpz, m->ix, m->iy, m->iz };
    for (int a = 0; a < 12; a++) free(all[a]);
}

orc_solver *orc_solver_create(int ni, int nj, int nk, float L, float viscosity, float blend)
{
    orc_solver *s = (orc_solver *)calloc(1, sizeof *s);
    s->ni = ni; s->nj = nj; s->nk = nk;
    s->h = L / (float)ni;                               /* BimocqGPUSolver.cpp:10 */
    s->viscosity = viscosity; s->blend = blend;
    s->n = (size_t)ni * nj * nk;
    s->nu = (size_t)(ni + 1) * nj * nk; s->nv = (size_t)ni * (nj + 1) * nk; s->nw = (size_t)ni * nj * (nk + 1);
    s->jacobi_iters = 100; s->halfrdx = 0.5f;           /* BimocqGPUSolver.cpp:409-410 */
    float **ub[] = { &s->U, &s->Ui, &s->Up, &s->Ut, &s->dUp, &s->dUe, &s->Su, &s->u_src };
    float **vb[] = { &s->V, &s->Vi, &s->Vp, &s->Vt, &s->dVp, &s->dVe, &s->Sv, &s->v_src };
    float **wb[] = { &s->W, &s->Wi, &s->Wp, &s->Wt, &s->dWp, &s->dWe, &s->Sw, &s->w_src };
    for (int a = 0; a < 8; a++) { *ub[a] = zalloc(s->nu); *vb[a] = zalloc(s->nv); *wb[a] = zalloc(s->nw); }
    float **sb[] = { &s->rho, &s->rhoi, &s->rhop, &s->rhot, &s->rhoe, &s->T, &s->Ti, &s->Tp, &s->Tt, &s->Te,
                     &s->div, &s->p, &s->pt, &s->xo, &s->yo, &s->zo };
    for (int a = 0; a < 16; a++) *sb[a] = zalloc(s->n);
    mapper_init(&s->vel, ni, nj, nk, s->h);
    mapper_init(&s->scal, ni, nj, nk, s->h);
    s->vel_last = -11; s->scal_last = -31;
    return s;
}

void orc_solver_destroy(orc_solver *s)
{
    if (!s) return;
    float *all[] = { s->U, s->V, s->W, s->Ui, s->Vi, s->Wi, s->Up, s->Vp, s->Wp, s->Ut, s->Vt, s->Wt,
                     s->dUp, s->dVp, s->dWp, s->dUe, s->dVe, s->dWe, s->Su, s->Sv, s->Sw,
                     s->rho, s->rhoi, s->rhop, s->rhot, s->rhoe, s->T, s->Ti, s->Tp, s->Tt, s->Te,
                     s->div, s->p, s->pt, s->u_src, s->v_src, s->w_src, s->xo, s->yo, s->zo };
    for (size_t a = 0; a < sizeof all / sizeof *all; a++) free(all[a]);
    mapper_free(&s->vel); mapper_free(&s->scal);
    free(s->mg_div); free(s->mg_p); free(s->mg_dir); free(s->mg_res); free(s->mg_t0); free(s->mg_t1); free(s->mg_result);
    for (int l = 0; l < s->mg_levels; l++) { free(s->mg_level[l].b); free(s->mg_level[l].x); free(s->mg_level[l].r); }
    free(s->em);
    free(s);
}

/* BimocqGPUSolver.cpp:529-534: _alpha = drop (rho coefficient), _beta = raise (T coefficient) */
void orc_solver_set_smoke(orc_solver *s, float drop_alpha, float rise_beta,
                          const orc_emitter *emitters, int n_emitters)
{
    s->alpha = drop_alpha; s->beta = rise_beta;
    free(s->em);
    s->em = (orc_emitter *)malloc(sizeof(orc_emitter) * (size_t)(n_emitters > 0 ? n_emitters : 1));
    if (n_emitters > 0) memcpy(s->em, emitters, sizeof(orc_emitter) * (size_t)n_emitters);
    s->n_em = n_emitters;
}

void orc_solver_set_option(orc_solver *s, int option, int value)
{
    if (option == 1) s->keep_dmc_border = value != 0;
    if (option == 2) s->reinit_policy = value;
    if (option == 3) s->scheme = value;
    if (option == 4) s->travel_limit = value < 0 ? 0 : value;
}

int orc_solver_reinit_counts(const orc_solver *s, int which) { return which == 2 ? s->forced_reinits : which ? s->scalar_reinits : s->vel_reinits; }
float orc_solver_last_distortion(const orc_solver *s, int which) { return which ? s->last_scalar_distortion : s->last_vel_distortion; }

/* BimocqGPUSolver.cpp:60-90: the fp64 work arrays and the level pyramid n -> (n - 1) / 2.  Levels that
 * would have no cell at all are left out (the reference launches empty grids for them). */
static void mg_alloc(orc_solver *s)
{
    if (s->mg_div) return;
    size_t n = s->n;
    s->mg_div = calloc(n, sizeof(double)); s->mg_p = calloc(n, sizeof(double)); s->mg_dir = calloc(n, sizeof(double));
    s->mg_res = calloc(n, sizeof(double)); s->mg_t0 = calloc(n, sizeof(double)); s->mg_t1 = calloc(n, sizeof(double));
    s->mg_result = calloc(4096, sizeof(double));
    int ni = s->ni, nj = s->nj, nk = s->nk;
    s->mg_levels = 0;
    for (int l = 0; l < 6; l++) {
        if (l) { ni = (ni - 1) / 2; nj = (nj - 1) / 2; nk = (nk - 1) / 2; }
        if (ni < 1 || nj < 1 || nk < 1) break;
        OrcCoarseLevel *L = &s->mg_level[l];
        L->ni = ni; L->nj = nj; L->nk = nk; L->number = ni * nj * nk;
        L->alpha = -1.0; L->beta = 1.0 / 6.0;
        L->b = calloc((size_t)L->number, sizeof(double)); L->x = calloc((size_t)L->number, sizeof(double));
        L->r = calloc((size_t)L->number, sizeof(double));
        s->mg_levels = l + 1;
    }
}

void orc_solver_set_projection_kind(orc_solver *s, int kind, int iters)
{
    s->projection_kind = kind;
    if (kind == 1) { s->mg_iters = iters; mg_alloc(s); }
    else s->jacobi_iters = iters;
}

int orc_solver_mg_levels(const orc_solver *s) { return s->mg_levels; }
const double *orc_solver_mg_history(const orc_solver *s) { return s->mg_result; }

void orc_solver_set_projection(orc_solver *s, int jacobi_iters, float halfrdx)
{
    s->jacobi_iters = jacobi_iters; s->halfrdx = halfrdx;
}

/* Mapping.cpp:347-373 updateMapping = updateBackward (host sub-step loop around the DMC
 * kernel + 3 copies out->in, GPU_Advection.h:460-470) then updateForward */
static void mapper_update(orc_solver *s, mapper_t *m, float cfldt, float dt)
{
    float T = 0.f, substep = cfldt;
    while (T < dt) {
        if (T + substep > dt) substep = dt - T;
        if (s->keep_dmc_border) {   /* the pre-copy the reference has commented out (GPU_Advection.h:335-337) */
            memcpy(s->xo, m->bx, s->n * sizeof(float));
            memcpy(s->yo, m->by, s->n * sizeof(float));
            memcpy(s->zo, m->bz, s->n * sizeof(float));
        }
        orc_solve_backwardDMC(s->U, s->V, s->W, m->bx, m->by, m->bz, s->xo, s->yo, s->zo,
                              s->h, s->ni, s->nj, s->nk, substep);
        memcpy(m->bx, s->xo, s->n * sizeof(float));
        memcpy(m->by, s->yo, s->n * sizeof(float));
        memcpy(m->bz, s->zo, s->n * sizeof(float));
        T += substep;
    }
    orc_solve_forward(s->U, s->V, s->W, m->fx, m->fy, m->fz, s->h, s->ni, s->nj, s->nk, cfldt, dt);
}

/* Mapping.cpp:430-447 */
static void mapper_reinit(orc_solver *s, mapper_t *m)
{
    size_t b = s->n * sizeof(float);
    m->total_reinit++;
    memcpy(m->bpx, m->bx, b); memcpy(m->bpy, m->by, b); memcpy(m->bpz, m->bz, b);
    memcpy(m->bx, m->ix, b);  memcpy(m->by, m->iy, b);  memcpy(m->bz, m->iz, b);
    memcpy(m->fx, m->ix, b);  memcpy(m->fy, m->iy, b);  memcpy(m->fz, m->iz, b);
}

/* MapperBaseGPU::estimateDistortion (Mapping.cpp:495-519): sqrt of the largest round-trip error of the
 * map pair, over the cells estimate_kernel writes (the scratch is cleared first; the reference scans a
 * scratch buffer that still holds older data outside that window) */
static float mapper_distortion(orc_solver *s, mapper_t *m)
{
    memset(s->u_src, 0, s->n * sizeof(float));
    orc_estimate_distortion(s->u_src, m->bx, m->by, m->bz, m->fx, m->fy, m->fz, s->h, s->ni, s->nj, s->nk);
    float mx = 0.f;
    for (size_t q = 0; q < s->n; q++) if (s->u_src[q] > mx) mx = s->u_src[q];
    return sqrtf(mx);
}

/* How many cells along z the maps of a set carry a node at most (the host's gpu_map_travel_z: max |map_z - z| / h over the
 * nodes the map updates write, a NaN counts as infinity), rounded up -- the displacement bound behind option 4. */
static int mapper_travel_cells(const orc_solver *s, const mapper_t *m)
{
    float mb = 0.f, mf = 0.f;
    for (int k = 2; k < s->nk - 2; k++) {
        const float z = (float)k * s->h;
        for (int j = 2; j < s->nj - 2; j++)
            for (int i = 2; i < s->ni - 2; i++) {
                const size_t id = (size_t)i + (size_t)s->ni * ((size_t)j + (size_t)s->nj * k);
                float db = fabsf(m->bz[id] - z), df = fabsf(m->fz[id] - z);
                if (db != db) db = INFINITY;
                if (df != df) df = INFINITY;
                if (db > mb) mb = db;
                if (df > mf) mf = df;
            }
    }
    const float tb = mb / s->h, tf = mf / s->h;
    const int cb = tb < 1.0e6f ? (int)ceil((double)tb) : 1000000, cf = tf < 1.0e6f ? (int)ceil((double)tf) : 1000000;
    return cb > cf ? cb : cf;
}

/* Mapping.cpp:393-407 + GPU_Advection.h:505-528 */
static void advect_scalar(orc_solver *s, float *f, float *finit, const float *fprev)
{
    mapper_t *m = &s->scal;
    memset(f, 0, s->n * sizeof(float));
    orc_advect_field(f, finit, m->bx, m->by, m->bz, s->h, s->ni, s->nj, s->nk, 0);
    memset(s->u_src, 0, s->n * sizeof(float));
    orc_compensate_field(f, finit, s->u_src, m->fx, m->fy, m->fz, m->bx, m->by, m->bz,
                         s->h, s->ni, s->nj, s->nk, 0);
    float b = (m->total_reinit != 0) ? s->blend : 1.f;
    orc_advect_field_double(f, fprev, m->bx, m->by, m->bz, m->bpx, m->bpy, m->bpz,
                            s->h, s->ni, s->nj, s->nk, 0, b);
}

/* BimocqGPUSolver::projection (:406-467) */
static void solver_project(orc_solver *s)
{
    const int ni = s->ni, nj = s->nj, nk = s->nk;
    const size_t bs = s->n * sizeof(float);
    if (s->projection_kind == 1) {
        /* :443-446 projectionMultiGrid(U, V, W, div, p, dir, residual, temp0, temp1, tempResult, levels, ...) */
        orc_multi_grid_conjugate_gradient(s->U, s->V, s->W, s->mg_div, s->mg_p, s->mg_dir, s->mg_res, s->mg_t0, s->mg_t1,
                                          s->mg_result, s->mg_level, s->mg_levels, s->mg_iters, (double)s->halfrdx);
    } else {
        /* GPU_Advection.h:602-608 zeroes div, p, p_temp first */
        memset(s->div, 0, bs); memset(s->p, 0, bs); memset(s->pt, 0, bs);
        orc_projection_jacobi(s->U, s->V, s->W, s->div, s->p, s->pt, NULL, ni, nj, nk,
                              s->jacobi_iters, s->halfrdx, -1.f, (float)(1.0 / 6.0));
    }
}

static void solver_sources(orc_solver *s, int framenum, float dt_buoyancy, float dt_diffuse, int emit)
{
    const int ni = s->ni, nj = s->nj, nk = s->nk;
    const float h = s->h;
    if (emit)
        for (int e = 0; e < s->n_em; e++)                                /* :376-392 */
            if (framenum < s->em[e].emit_frames)
                orc_emit_smoke(s->U, s->V, s->W, s->rho, s->T, h, ni, nj, nk,
                               s->em[e].cx, s->em[e].cy, s->em[e].cz, s->em[e].radius,
                               s->em[e].density, s->em[e].temperature, s->em[e].emiter);
    orc_add_buoyancy(s->V, s->rho, s->T, ni, nj, nk, s->alpha, s->beta, dt_buoyancy);  /* :394-397 */
    if (s->viscosity != 0.f) {                                          /* :167-172 / :286-291 incl. the aliasing of Q7 */
        float coef = s->viscosity * (dt_diffuse / (h * h));             /* :401 */
        orc_diffuse_field(s->U, s->Ut, s->Su, ni + 1, nj, nk, 20, coef);
        orc_diffuse_field(s->V, s->Vt, s->Sv, ni, nj + 1, nk, 20, coef);
        orc_diffuse_field(s->W, s->Wt, s->Sw, ni, nj, nk + 1, 20, coef);
    }
}

/* gpuMapper::semilagAdvectField / semilagAdvectVelocity (GPU_Advection.h:530-551): output cleared first */
static void semilag_scalar(orc_solver *s, float *dst, const float *src, float cfldt, float dt)
{
    memset(dst, 0, s->n * sizeof(float));
    orc_semilag(dst, src, s->U, s->V, s->W, 0, 0, 0, s->h, s->ni, s->nj, s->nk, cfldt, dt);
}
static void semilag_velocity(orc_solver *s, float *uo, float *vo, float *wo, const float *us, const float *vs, const float *ws,
                             float cfldt, float dt)
{
    memset(uo, 0, s->nu * sizeof(float)); memset(vo, 0, s->nv * sizeof(float)); memset(wo, 0, s->nw * sizeof(float));
    orc_semilag(uo, us, s->U, s->V, s->W, 1, 0, 0, s->h, s->ni, s->nj, s->nk, cfldt, dt);
    orc_semilag(vo, vs, s->U, s->V, s->W, 0, 1, 0, s->h, s->ni, s->nj, s->nk, cfldt, dt);
    orc_semilag(wo, ws, s->U, s->V, s->W, 0, 0, 1, s->h, s->ni, s->nj, s->nk, cfldt, dt);
}

/* BimocqGPUSolver::advanceReflection (:232-337) with the corrected limiter (orc_clamp_extrema).  Buffer
 * roles as in the reference: DensityTemp/TemperatureTemp = rhot/Tt, TempSrc* = Su/Sv/Sw, d*Proj = dUp.. */
static void advance_reflection(orc_solver *s, int framenum, float dt)
{
    const int ni = s->ni, nj = s->nj, nk = s->nk;
    const float h = s->h;
    const size_t bu = s->nu * sizeof(float), bv = s->nv * sizeof(float), bw = s->nw * sizeof(float), bs = s->n * sizeof(float);
    s->max_v = orc_max_abs3(s->U, s->V, s->W, ni, nj, nk);
    const float cfldt = h / s->max_v;
    s->last_cfldt = cfldt;

    float *scal[2] = { s->rho, s->T }, *tmp[2] = { s->rhot, s->Tt };
    for (int a = 0; a < 2; a++) {                                       /* :237-263 MacCormack on rho, T */
        semilag_scalar(s, tmp[a], scal[a], cfldt, -dt);
        semilag_scalar(s, s->Su, tmp[a], cfldt, dt);
        orc_add(tmp[a], s->Su, -0.5f, (int)s->n);
        orc_add(tmp[a], scal[a], 0.5f, (int)s->n);
        orc_clamp_extrema(scal[a], tmp[a], s->U, s->V, s->W, ni, nj, nk, 0, 0, 0, 0.f, 0.f, 0.f, h, dt);
        memcpy(scal[a], tmp[a], bs);
    }
    {                                                                   /* :267-287 velocity, half a step */
        semilag_velocity(s, s->Ut, s->Vt, s->Wt, s->U, s->V, s->W, cfldt, -0.5f * dt);
        semilag_velocity(s, s->Su, s->Sv, s->Sw, s->Ut, s->Vt, s->Wt, cfldt, 0.5f * dt);
        orc_add(s->Ut, s->Su, -0.5f, (int)s->nu); orc_add(s->Vt, s->Sv, -0.5f, (int)s->nv); orc_add(s->Wt, s->Sw, -0.5f, (int)s->nw);
        orc_add(s->Ut, s->U, 0.5f, (int)s->nu);   orc_add(s->Vt, s->V, 0.5f, (int)s->nv);   orc_add(s->Wt, s->W, 0.5f, (int)s->nw);
        orc_clamp_extrema(s->U, s->Ut, s->U, s->V, s->W, ni + 1, nj, nk, 1, 0, 0, 0.5f, 0.f, 0.f, h, 0.5f * dt);
        orc_clamp_extrema(s->V, s->Vt, s->U, s->V, s->W, ni, nj + 1, nk, 0, 1, 0, 0.f, 0.5f, 0.f, h, 0.5f * dt);
        orc_clamp_extrema(s->W, s->Wt, s->U, s->V, s->W, ni, nj, nk + 1, 0, 0, 1, 0.f, 0.f, 0.5f, h, 0.5f * dt);
        memcpy(s->U, s->Ut, bu); memcpy(s->V, s->Vt, bv); memcpy(s->W, s->Wt, bw);
    }
    solver_sources(s, framenum, 0.5f * dt, 0.5f * dt, 1);               /* :289-297 */
    memcpy(s->Ut, s->U, bu); memcpy(s->Vt, s->V, bv); memcpy(s->Wt, s->W, bw);     /* :299-303 */
    solver_project(s);                                                  /* :305 */
    orc_mad(s->dUp, s->U, s->Ut, 2.f, -1.f, (int)s->nu);                /* :307-309 reflection: 2 u_proj - u */
    orc_mad(s->dVp, s->V, s->Vt, 2.f, -1.f, (int)s->nv);
    orc_mad(s->dWp, s->W, s->Wt, 2.f, -1.f, (int)s->nw);
    semilag_velocity(s, s->Ut, s->Vt, s->Wt, s->dUp, s->dVp, s->dWp, cfldt, -0.5f * dt);      /* :311 */
    semilag_velocity(s, s->Su, s->Sv, s->Sw, s->Ut, s->Vt, s->Wt, cfldt, 0.5f * dt);          /* :313 */
    orc_add(s->Ut, s->Su, -0.5f, (int)s->nu); orc_add(s->Vt, s->Sv, -0.5f, (int)s->nv); orc_add(s->Wt, s->Sw, -0.5f, (int)s->nw);
    orc_add(s->Ut, s->dUp, 0.5f, (int)s->nu); orc_add(s->Vt, s->dVp, 0.5f, (int)s->nv); orc_add(s->Wt, s->dWp, 0.5f, (int)s->nw);
    /* :323-325: the limiter is given VelocityU/V/W as the source field (not d*Proj, which was advected) */
    orc_clamp_extrema(s->U, s->Ut, s->U, s->V, s->W, ni + 1, nj, nk, 1, 0, 0, 0.5f, 0.f, 0.f, h, 0.5f * dt);
    orc_clamp_extrema(s->V, s->Vt, s->U, s->V, s->W, ni, nj + 1, nk, 0, 1, 0, 0.f, 0.5f, 0.f, h, 0.5f * dt);
    orc_clamp_extrema(s->W, s->Wt, s->U, s->V, s->W, ni, nj, nk + 1, 0, 0, 1, 0.f, 0.f, 0.5f, h, 0.5f * dt);
    memcpy(s->U, s->Ut, bu); memcpy(s->V, s->Vt, bv); memcpy(s->W, s->Wt, bw);
    solver_sources(s, framenum, 0.5f * dt, 0.5f * dt, 0);               /* :330-337 (no emission the second time) */
    solver_project(s);
}

/* BimocqGPUSolver.cpp:129-230 advanceBimocq with the Jacobi projection branch
 * (:408-410; div/p/p_temp get dedicated buffers -- the reference lends it DensityTemp,
 * TemperatureTemp and TempSrcV, all dead at that point). */
void orc_solver_advance(orc_solver *s, int framenum, float dt)
{
    if (s->scheme == 3) { advance_reflection(s, framenum, dt); return; }        /* enum Scheme: MAC_REFLECTION */
    const int ni = s->ni, nj = s->nj, nk = s->nk;
    const float h = s->h;
    size_t bu = s->nu * sizeof(float), bv = s->nv * sizeof(float), bw = s->nw * sizeof(float), bs = s->n * sizeof(float);
    float proj_coeff = 2.f;

    /* getCFL(): BimocqGPUSolver.cpp:348-373 (evaluated on the current device fields: equal
     * to the reference's host copies when outputResult follows every advance) */
    s->max_v = orc_max_abs3(s->U, s->V, s->W, ni, nj, nk);
    float cfldt = h / s->max_v;
    s->last_cfldt = cfldt;
    const int policy = s->reinit_policy;

    mapper_update(s, &s->vel, cfldt, dt);                               /* :138 */
    mapper_update(s, &s->scal, cfldt, dt);                              /* :139 */

    /* :143 VelocityAdvector.advectVelocity -> Mapping.cpp:375-391 */
    {
        mapper_t *m = &s->vel;
        memset(s->U, 0, bu); memset(s->V, 0, bv); memset(s->W, 0, bw);  /* GPU_Advection.h:477-479 */
        orc_advect_velocity(s->U, s->V, s->W, s->Ui, s->Vi, s->Wi, m->bx, m->by, m->bz, h, ni, nj, nk, 0);
        memset(s->u_src, 0, bu); memset(s->v_src, 0, bv); memset(s->w_src, 0, bw);
        orc_compensate_velocity(s->U, s->V, s->W, s->Ui, s->Vi, s->Wi, s->u_src, s->v_src, s->w_src,
                                m->fx, m->fy, m->fz, m->bx, m->by, m->bz, h, ni, nj, nk, 0);
        float b = (m->total_reinit != 0) ? s->blend : 1.f;
        orc_advect_vel_double(s->U, s->V, s->W, s->Up, s->Vp, s->Wp, m->bx, m->by, m->bz,
                              m->bpx, m->bpy, m->bpz, h, ni, nj, nk, 0, b);
    }
    advect_scalar(s, s->rho, s->rhoi, s->rhop);                         /* :144 */
    advect_scalar(s, s->T, s->Ti, s->Tp);                               /* :145 */

    memcpy(s->Ut, s->U, bu); memcpy(s->Vt, s->V, bv); memcpy(s->Wt, s->W, bw);   /* :157-159 */
    /* policy 1 follows the CPU solver here (BimocqSolver.cpp:129-133): the scalar snapshots are taken BEFORE the
     * sources act, so that rho - rhoTemp is what emission added (the GPU solver takes them after) */
    if (policy == 1) { memcpy(s->rhot, s->rho, bs); memcpy(s->Tt, s->T, bs); }

    solver_sources(s, framenum, dt, dt, 1);                            /* :164-172 */

    orc_add_field(s->dUe, s->U, s->Ut, -1.f, (int)s->nu);               /* :175-177 */
    orc_add_field(s->dVe, s->V, s->Vt, -1.f, (int)s->nv);
    orc_add_field(s->dWe, s->W, s->Wt, -1.f, (int)s->nw);
    memcpy(s->Ut, s->U, bu); memcpy(s->Vt, s->V, bv); memcpy(s->Wt, s->W, bw);   /* :179-181 */

    solver_project(s);

    if (policy == 0) { memcpy(s->rhot, s->rho, bs); memcpy(s->Tt, s->T, bs); }   /* :185-186 */
    memcpy(s->dUp, s->U, bu); memcpy(s->dVp, s->V, bv); memcpy(s->dWp, s->W, bw);  /* :188-190 */
    orc_add(s->dUp, s->Ut, -1.f, (int)s->nu);                           /* :191-193 */
    orc_add(s->dVp, s->Vt, -1.f, (int)s->nv);
    orc_add(s->dWp, s->Wt, -1.f, (int)s->nw);
    memcpy(s->rhoe, s->rho, bs); memcpy(s->Te, s->T, bs);               /* :195-198: identically 0 (Q8) */
    orc_add(s->rhoe, s->rhot, -1.f, (int)s->n);
    orc_add(s->Te, s->Tt, -1.f, (int)s->n);

    int vel_reinit = 1, scalar_reinit = 1;
    if (policy == 1) {
        /* BimocqSolver.cpp:165-185: distortion of each map set in units of the step's travel */
        s->last_vel_distortion = mapper_distortion(s, &s->vel) / (s->max_v * dt);
        s->last_scalar_distortion = mapper_distortion(s, &s->scal) / (s->max_v * dt);
        vel_reinit = scalar_reinit = 0;
        if (s->last_vel_distortion > 1.f || framenum - s->vel_last > 10) { vel_reinit = 1; s->vel_last = framenum; proj_coeff = 1.f; }
        if (s->last_scalar_distortion > 5.f || framenum - s->scal_last > 30) { scalar_reinit = 1; s->scal_last = framenum; }
        if (s->travel_limit > 0) {
            /* option 4 (not in the reference: what lets z-slab ranks run this policy, csrc/host/fluid_solver.cpp): a map set
             * whose z-travel + this step's CFL travel + the sampling footprint would not fit `travel_limit` planes next step
             * is re-initialised now */
            const int dcells = (int)ceil((double)dt * (double)s->max_v / (double)h) + 1;
            if (!vel_reinit && mapper_travel_cells(s, &s->vel) + dcells + 2 > s->travel_limit) {
                vel_reinit = 1; s->vel_last = framenum; proj_coeff = 1.f; s->forced_reinits++;
            }
            if (!scalar_reinit && mapper_travel_cells(s, &s->scal) + dcells + 2 > s->travel_limit) {
                scalar_reinit = 1; s->scal_last = framenum; s->forced_reinits++;
            }
        }
    } else {
        if (framenum - s->vel_last > 10) { s->vel_last = framenum; proj_coeff = 1.f; }   /* :200-205 */
        if (framenum - s->scal_last > 30) { s->scal_last = framenum; }                    /* :207-211 */
    }

    {   /* :213-216 */
        mapper_t *m = &s->vel;
        orc_accumulate_velocity(s->dUe, s->dVe, s->dWe, s->Ui, s->Vi, s->Wi, m->fx, m->fy, m->fz, h, ni, nj, nk, 0, 1.f);
        orc_accumulate_velocity(s->dUp, s->dVp, s->dWp, s->Ui, s->Vi, s->Wi, m->fx, m->fy, m->fz, h, ni, nj, nk, 0, proj_coeff);
        mapper_t *q = &s->scal;
        orc_accumulate_field(s->rhoe, s->rhoi, q->fx, q->fy, q->fz, h, ni, nj, nk, 0, 1.f);
        orc_accumulate_field(s->Te, s->Ti, q->fx, q->fy, q->fz, h, ni, nj, nk, 0, 1.f);
    }

    if (vel_reinit) {   /* :218-223 `if (1)`: reinitialise every frame (Q5); policy 1: BimocqSolver.cpp:203-215 */
        mapper_t *m = &s->vel;
        s->vel_reinits++;
        mapper_reinit(s, m);
        memcpy(s->Up, s->Ui, bu); memcpy(s->Vp, s->Vi, bv); memcpy(s->Wp, s->Wi, bw);  /* :509-511 */
        memcpy(s->Ui, s->U, bu);  memcpy(s->Vi, s->V, bv);  memcpy(s->Wi, s->W, bw);   /* :513-515 */
        orc_accumulate_velocity(s->dUp, s->dVp, s->dWp, s->Ui, s->Vi, s->Wi, m->fx, m->fy, m->fz, h, ni, nj, nk, 0, 1.f);
    }
    if (scalar_reinit) {   /* :225-229; policy 1: BimocqSolver.cpp:216-227 */
        s->scalar_reinits++;
        mapper_reinit(s, &s->scal);
        memcpy(s->rhop, s->rhoi, bs); memcpy(s->Tp, s->Ti, bs);         /* :522-523 */
        memcpy(s->rhoi, s->rho, bs);  memcpy(s->Ti, s->T, bs);          /* :525-526 */
    }
}

const float *orc_solver_field(orc_solver *s, int which, long *count)
{
    const float *f[] = { s->rho, s->T, s->U, s->V, s->W, s->Ui, s->Vi, s->Wi, s->rhoi, s->Ti,
                         s->vel.fx, s->vel.fy, s->vel.fz, s->vel.bx, s->vel.by, s->vel.bz, s->p, s->div };
    size_t c[] = { s->n, s->n, s->nu, s->nv, s->nw, s->nu, s->nv, s->nw, s->n, s->n,
                   s->n, s->n, s->n, s->n, s->n, s->n, s->n, s->n };
    if (which < 0 || which > 17) { if (count) *count = 0; return NULL; }
    if (count) *count = (long)c[which];
    return f[which];
}

float orc_solver_last_cfldt(const orc_solver *s) { return s->last_cfldt; }
