// bq_advect.hip -- characteristic-map kernels and their
// extern "C" launchers.  One thread per buffer element, block = 64 (x) x 4 (y) so a wavefront
// owns 64 consecutive i of one row (coalesced stores, spatially coherent gathers).
//
// Every kernel restates one reference kernel with identical arithmetic (see bq_device.hip.h);
// the launch geometry, the buffer-descriptor loads and the fusion are ours.
#include "bq_device.hip.h"
#include "bq_host.h"

#include <algorithm>
#include <cstdint>
#include <type_traits>
#include <vector>

namespace bq {
inline namespace BQ_VARIANT {

#define BQ_IJK(nbi, nbj, nbk)                                   \
    const int i = blockIdx.x * 64 + threadIdx.x;                \
    const int j = blockIdx.y * 4 + threadIdx.y;                 \
    const int k = blockIdx.z + g.kw0;                           \
    if (i >= (nbi) || j >= (nbj) || k >= (nbk)) return;         \
    const int kg = k + g.koff;                                  \
    (void)kg;

static inline dim3 grid_for(int nbi, int nbj, int nbk) { return dim3((nbi + 63) / 64, (nbj + 3) / 4, nbk); }
static const dim3 kBlock(64, 4, 1);

// The 9-point kernels on the structured power-of-two path stage the map nodes of their block in LDS
// (bq_device.hip.h: stage_tiles), so every thread of a block has to reach the barrier: the index window
// ilo < i < ihi, jlo < j < jhi, klo < kg < khi is tested as a predicate, only whole blocks leave early.
// block_out: no thread of the block is inside the window (block-uniform, so leaving on it skips no barrier).
#define BQ_IJK_WINDOW(ilo, ihi, jlo, jhi, klo, khi)                                           \
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 4;                                      \
    const int i = i0 + threadIdx.x, j = j0 + threadIdx.y, k = blockIdx.z + g.kw0;             \
    const int kg = k + g.koff;                                                                \
    const bool block_out = !((klo) < kg && kg < (khi)) || i0 + 63 <= (ilo) || i0 >= (ihi) || j0 + 3 <= (jlo) || j0 >= (jhi); \
    const bool active = !block_out && (ilo) < i && i < (ihi) && (jlo) < j && j < (jhi);
// (SD >= 0 with P2: compile-time taps; SD >= 0 without P2: the tabled form for other spacings, bq_device.hip.h: MapTabs)
template <bool P2, bool PT, int SD> constexpr bool kStaged = !PT && SD >= 0;

// CELL dims of the LOCAL buffers plus the z-slab context: local plane k is global plane k + koff of a
// grid with nkg cell planes (single GPU: koff = 0, nkg = nk).  Index windows, positions and clamps
// are evaluated in GLOBAL coordinates so that a slab rank computes exactly what one GPU would.
struct Grid { int ni, nj, nk, koff, nkg, kw0; };      // kw0: first local plane of this launch (plane window, else 0)

// ---- 9-point stencil of sub-voxel sample positions (GPU_kernel.cu:317-348) ---------------
struct Nine {
    float q, mq;        // +-0.25f*h
    f3 org;             // buffer origin (-dim*0.5f*h)
    float h;
};
__device__ __forceinline__ Nine nine_setup(float h, int dx, int dy, int dz)
{
    Nine n;
    n.q = 0.25f * h; n.mq = -0.25f * h; n.h = h;
    n.org = mk3(-(float)dx * 0.5f * h, -(float)dy * 0.5f * h, -(float)dz * 0.5f * h);
    return n;
}
__device__ __forceinline__ f3 nine_centre(const Nine &n, int i, int j, int k)
{
    return mk3((float)i * n.h + n.org.x, (float)j * n.h + n.org.y, (float)k * n.h + n.org.z);
}
// corner ii in the reference's order: bit2 -> x sign, bit1 -> y sign, bit0 -> z sign (0 = +)
__device__ __forceinline__ f3 nine_corner(const Nine &n, f3 c, int ii)
{
    return mk3(c.x + ((ii & 4) ? n.mq : n.q), c.y + ((ii & 2) ? n.mq : n.q), c.z + ((ii & 1) ? n.mq : n.q));
}

// ---- A3: forward_kernel (GPU_kernel.cu:127-144) -------------------------------------------
// guard (may be null): word that gets a 1 when a stored map value fails tile_value_ok (fl_map_guard_*)
__device__ __forceinline__ void guard_map_values(int *guard, f3 r, float h)
{
    if (!guard) return;
    const float lo = h * 0.00390625f, hi = h * 1024.f;
    const bool bad = !(tile_value_ok(r.x, lo, hi) && tile_value_ok(r.y, lo, hi) && tile_value_ok(r.z, lo, hi));
    // at most ONE atomic per wave, and none once the word is set: a field full of bad values (a NaN that has spread, the
    // stale ghost planes of an emulated slab rank) used to issue one atomic per NODE to the same address -- 61 M of them
    // made dmc_kernel take 10.9 instead of 1.4 ms on an emulated 1024 x 1024 x 80 rank
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && __lane_id() == (unsigned)(__ffsll((long long)m) - 1) &&
        __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        atomicOr(guard, 1);
}

template <bool P2>
__global__ __launch_bounds__(256) void forward_kernel(const float *u, const float *v, const float *w,
                                                      float *xf, float *yf, float *zf,
                                                      Spacing sp, Grid g, float cfldt, float dt, int *guard)
{
    BQ_IJK(g.ni, g.nj, g.nk)
    if (!(i > 1 && i < g.ni - 2 && j > 1 && j < g.nj - 2 && kg > 1 && kg < g.nkg - 2)) return;
    Vel3 vel{make_field(u, g.ni + 1, g.nj, g.nk, g.koff), make_field(v, g.ni, g.nj + 1, g.nk, g.koff), make_field(w, g.ni, g.nj, g.nk + 1, g.koff)};
    f3 hi = mk3((float)g.ni * sp.h - sp.h, (float)g.nj * sp.h - sp.h, (float)g.nkg * sp.h - sp.h);
    size_t id = (size_t)i + (size_t)g.ni * j + (size_t)g.ni * g.nj * k;
    f3 q = trace<P2>(vel, sp, hi, cfldt, dt, mk3(xf[id], yf[id], zf[id]));
    xf[id] = q.x; yf[id] = q.y; zf[id] = q.z;
    guard_map_values(guard, q, sp.h);
}

// ---- A4: DMC_backward_kernel (GPU_kernel.cu:169-204) --------------------------------------
__device__ __forceinline__ float dmc_axis(float p, float vel, float a, float s)
{
    if ((double)fabsf(a) > 1e-4) return p - (1.0f - exp_portable(-a * s)) * vel / a;
    return p - vel * s;
}

template <bool P2>
__global__ __launch_bounds__(256) void dmc_kernel(const float *u, const float *v, const float *w,
                                                  const float *xi, const float *yi, const float *zi,
                                                  float *xo, float *yo, float *zo,
                                                  Spacing sp, Grid g, float substep, int border, int *guard)
{
    BQ_IJK(g.ni, g.nj, g.nk)
    if (!(i > 1 && i < g.ni - 2 && j > 1 && j < g.nj - 2 && kg > 1 && kg < g.nkg - 2)) {
        // border nodes are not updated (GPU_kernel.cu:175).  FL_OPT_FUSED_HOUSEKEEPING bits 4 / 8: the kernel itself
        // leaves there what the caller would otherwise have to prepare -- zeros (the reference's cleared scratch)
        // or the input map's value
        if (border) {
            const size_t b = (size_t)i + (size_t)g.ni * j + (size_t)g.ni * g.nj * k;
            const bool copy = border == 2;
            const f3 bv = mk3(copy ? xi[b] : 0.f, copy ? yi[b] : 0.f, copy ? zi[b] : 0.f);
            xo[b] = bv.x; yo[b] = bv.y; zo[b] = bv.z;
            guard_map_values(guard, bv, sp.h);
        }
        return;
    }
    const float h = sp.h;
    Vel3 vel{make_field(u, g.ni + 1, g.nj, g.nk, g.koff), make_field(v, g.ni, g.nj + 1, g.nk, g.koff), make_field(w, g.ni, g.nj, g.nk + 1, g.koff)};
    Map3 in{make_field(xi, g.ni, g.nj, g.nk, g.koff), make_field(yi, g.ni, g.nj, g.nk, g.koff), make_field(zi, g.ni, g.nj, g.nk, g.koff)};
    // the node (indices >= 2 inside the window) and its upwind neighbour are at least h away from the origin on every
    // axis: q >= 1 for both velocity look-ups (sample's GE1 form); the DMC departure point below is not bounded
    f3 pt = mk3(h * (float)i, h * (float)j, h * (float)kg);
    f3 vl = get_velocity<P2, true>(vel, sp, pt);
    f3 tp = mk3((vl.x > 0) ? pt.x - h : pt.x + h, (vl.y > 0) ? pt.y - h : pt.y + h, (vl.z > 0) ? pt.z - h : pt.z + h);
    f3 tv = get_velocity<P2, true>(vel, sp, tp);
    float ax = (vl.x - tv.x) / (pt.x - tp.x);
    float ay = (vl.y - tv.y) / (pt.y - tp.y);
    float az = (vl.z - tv.z) / (pt.z - tp.z);
    f3 pn = mk3(dmc_axis(pt.x, vl.x, ax, substep), dmc_axis(pt.y, vl.y, ay, substep), dmc_axis(pt.z, vl.z, az, substep));
    f3 r = map_at<P2>(in, sp, pn);
    size_t id = (size_t)i + (size_t)g.ni * j + (size_t)g.ni * g.nj * k;
    xo[id] = r.x; yo[id] = r.y; zo[id] = r.z;
    guard_map_values(guard, r, sp.h);
}

// The 9 mapped positions of a node: out[0..7] corners (reference order), out[8] centre.  SD >= 0 selects
// the structured power-of-two path (SD = staggered axis + 1, 0 = none); SD < 0 the generic one.
template <bool P2, bool PT, int SD>
__device__ __forceinline__ void mapped9(const Map3 &m, const Spacing &sp, const Nine &n, f3 c, int i, int j, int kl, f3 out[9])
{
    if constexpr (P2 && !PT && SD >= 0) {
        map9<SD == 1, SD == 2, SD == 3>(m, i, j, kl, out);
    } else {
        if (!PT) {
#pragma unroll
            for (int ii = 0; ii < 8; ii++) out[ii] = map_at<P2>(m, sp, nine_corner(n, c, ii));
        }
        out[8] = map_at<P2>(m, sp, c);
    }
}

// The 9-point blend of NF co-located fields at the (clamped) positions mp[0..8]: per field
//     sum = sum_ii w * field(mp[ii])  (ii = 0..7 in order; point mode: the centre alone),  value = field(mp[8])
// with the tap's cell and weights located once for all fields (they share dims and origin).
// Positions are inside the grid and origins are <= 0, so pos - org >= 0 (locate's NONNEG form).
// GE1: every position is >= h on every axis, hence q >= 1 and the lerps may take their one-fma form (lerp_w<true>).
template <bool P2, bool PT, int NF, bool GE1>
__device__ __forceinline__ void blend9_gather_w(const Field (&src)[NF], const Spacing &sp, f3 org, const f3 (&mp)[9],
                                                const float (&w)[NF], float (&sum)[NF], float (&value)[NF])
{
    if (!PT) {
#pragma unroll
        for (int ii = 0; ii < 8; ii++) {
            const Cell c = locate<P2, true>(src[0], sp, org, mp[ii]);
#pragma unroll
            for (int f = 0; f < NF; f++) sum[f] += w[f] * gather<GE1>(src[f], c);
        }
    }
    const Cell c = locate<P2, true>(src[0], sp, org, mp[8]);
#pragma unroll
    for (int f = 0; f < NF; f++) {
        value[f] = gather<GE1>(src[f], c);
        if (PT) sum[f] += w[f] * value[f];
    }
}
template <bool P2, bool PT, int NF, bool GE1>
__device__ __forceinline__ void blend9_gather(const Field (&src)[NF], const Spacing &sp, f3 org, const f3 (&mp)[9],
                                              float (&sum)[NF], float (&value)[NF])
{
    float w[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) w[f] = PT ? 1.0f : 0.125f;
    blend9_gather_w<P2, PT, NF, GE1>(src, sp, org, mp, w, sum, value);
}
// true when every lane of the wave has all nine mapped positions (point mode: the centre) >= h on every axis.  Tested
// before the clamp to [0, hi], which cannot lower a value that is >= h.  fminf skips a NaN: such a position is clamped
// to 0, where q is 0 or 1/2 and the weights are multiples of 2^-28 all the same.
template <bool PT>
__device__ __forceinline__ bool wave_all_ge(const f3 (&mp)[9], float h)
{
    float m = fminf(mp[8].x, fminf(mp[8].y, mp[8].z));
    if (!PT) {
#pragma unroll
        for (int a9 = 0; a9 < 8; a9++) m = fminf(m, fminf(mp[a9].x, fminf(mp[a9].y, mp[a9].z)));
    }
    return __all(m >= h);
}

// Batches: NF fields that live on the same nodes share one map look-up (density + temperature; the
// two velocity-change fields accumulated back to back).  Results per field are what NF single
// launches in the same order produce.
template <int NF> struct AdvectArgs { float *field[NF]; const float *init[NF]; };
template <int NF> struct CumulateArgs { const float *src[NF]; float *dst[NF]; float coeff[NF]; };
template <int NF> struct CompensateArgs { const float *src[NF]; float *init[NF]; float *err[NF]; };

// ---- A5: advect_kernel (GPU_kernel.cu:312-374) --------------------------------------------
// Q4: the caller vouches for the map's values (tile_value_ok): the quarter-weight map lerps run in fp32
template <bool P2, bool PT, int SD, int NF, bool Q4 = false>
__global__ __launch_bounds__(256, NF == 1 ? 7 : 5) void advect_kernel(AdvectArgs<NF> a,
                                                     const float *bx, const float *by, const float *bz,
                                                     Spacing sp, Grid g, int dx, int dy, int dz, int fused, MapTabs tabs)
{
    const int nbi = g.ni + dx, nbj = g.nj + dy, nbk = g.nk + dz;
    BQ_IJK_WINDOW(2 + dx, nbi - 3, 2 + dy, nbj - 3, 2 + dz, g.nkg + dz - 3)
    // FL_OPT_FUSED_HOUSEKEEPING bit 1: nodes outside the window get the zero the caller's clear would have left
    if ((fused & 1) && !active && i < nbi && j < nbj) {
#pragma unroll
        for (int f = 0; f < NF; f++) a.field[f][(size_t)i + (size_t)nbi * j + (size_t)nbi * nbj * k] = 0.f;
    }
    if (block_out) return;
    const float h = sp.h;
    Map3 back{make_field(bx, g.ni, g.nj, g.nk, g.koff), make_field(by, g.ni, g.nj, g.nk, g.koff), make_field(bz, g.ni, g.nj, g.nk, g.koff)};
    Nine n = nine_setup(h, dx, dy, dz);
    f3 lo = mk3(h, h, h), hi = mk3(h * (float)g.ni - h, h * (float)g.nj - h, h * (float)g.nkg - h);
    f3 c = nine_centre(n, i, j, kg);
    f3 mp[9];
    if constexpr (kStaged<P2, PT, SD>) {
        __shared__ float tile[3 * kTile];
        const Field mf[3] = {back.x, back.y, back.z};
        stage_tiles<3>(mf, i0, j0, k, tile);
        if (!active) return;
        if constexpr (P2) map9_lds<SD == 1, SD == 2, SD == 3, Q4>(tile, mp);
        else map9_lds_tab<SD == 1, SD == 2, SD == 3>(tile, tabs, i, j, kg, mp);
    } else {
        if (!active) return;
        mapped9<P2, PT, SD>(back, sp, n, c, i, j, k, mp);
    }
    // (an active thread implies ni, nj >= 7 and nkg >= 7, so lo = h <= hi)
#pragma unroll
    for (int a9 = 0; a9 < 9; a9++) mp[a9] = clamp3_ordered(mp[a9], lo, hi);
    // taps outermost: cell and weights of a tap are found once and serve every field of the batch
    Field src[NF];
    float sum[NF], value[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) { src[f] = make_field(a.init[f], nbi, nbj, nbk, g.koff); sum[f] = 0.f; }
    blend9_gather<P2, PT, NF, true>(src, sp, n.org, mp, sum, value);       // positions clamped to >= h
#pragma unroll
    for (int f = 0; f < NF; f++)
        a.field[f][(size_t)i + (size_t)nbi * j + (size_t)nbi * nbj * k] = 0.5f * sum[f] + 0.5f * value[f];
}

// ---- A7: doubleAdvect_kernel (GPU_kernel.cu:236-310) --------------------------------------
template <bool P2, bool PT, int SD>
__global__ __launch_bounds__(256) void double_advect_kernel(float *field, const float *prev,
                                                            const float *bx, const float *by, const float *bz,
                                                            const float *px, const float *py, const float *pz,
                                                            Spacing sp, Grid g, int dx, int dy, int dz, float blend, int prev_global)
{
    const int nbi = g.ni + dx, nbj = g.nj + dy, nbk = g.nk + dz;
    BQ_IJK(nbi, nbj, nbk)
    if (!(2 + dx < i && i < nbi - 3 && 2 + dy < j && j < nbj - 3 && 2 + dz < kg && kg < g.nkg + dz - 3)) return;
    const float h = sp.h;
    Map3 back{make_field(bx, g.ni, g.nj, g.nk, g.koff), make_field(by, g.ni, g.nj, g.nk, g.koff), make_field(bz, g.ni, g.nj, g.nk, g.koff)};
    Map3 bprev{make_field(px, g.ni, g.nj, g.nk, g.koff), make_field(py, g.ni, g.nj, g.nk, g.koff), make_field(pz, g.ni, g.nj, g.nk, g.koff)};
    // prev_global (gpu_advect_*_double_global): `prev` holds every plane of the grid, not this rank's -- the second look-up
    // lands anywhere between the origin and the node when it meets the zeroed border cells of the previous map
    Field src = prev_global ? make_field(prev, nbi, nbj, g.nkg + dz, 0) : make_field(prev, nbi, nbj, nbk, g.koff);
    Nine n = nine_setup(h, dx, dy, dz);
    f3 lo = mk3(h, h, h), hi = mk3(h * (float)g.ni - h, h * (float)g.nj - h, h * (float)g.nkg - h);
    f3 c = nine_centre(n, i, j, kg);
    f3 mp[9];
    mapped9<P2, PT, SD>(back, sp, n, c, i, j, k, mp);
    float sum = 0.f;
    if (PT) {
        f3 fin = clamp3(map_at<P2>(bprev, sp, clamp3(mp[8], lo, hi)), lo, hi);
        sum += 1.0f * sample<P2>(src, sp, n.org, fin);
    } else {
#pragma unroll
        for (int ii = 0; ii < 8; ii++) {
            f3 fin = clamp3(map_at<P2>(bprev, sp, clamp3(mp[ii], lo, hi)), lo, hi);
            sum += 0.125f * sample<P2>(src, sp, n.org, fin);
        }
    }
    f3 fin = clamp3(map_at<P2>(bprev, sp, clamp3(mp[8], lo, hi)), lo, hi);
    float value = sample<P2>(src, sp, n.org, fin);
    float prev_value = 0.5f * (sum + value);
    size_t id = (size_t)i + (size_t)nbi * j + (size_t)nbi * nbj * k;
    field[id] = field[id] * blend + (1.0f - blend) * prev_value;
}

// blend == 1 fast path: field*1 + 0*prev == field + (+-0) for finite prev; same window.
__global__ __launch_bounds__(256) void unit_blend_kernel(float *field, Grid g, int dx, int dy, int dz)
{
    const int nbi = g.ni + dx, nbj = g.nj + dy, nbk = g.nk + dz;
    BQ_IJK(nbi, nbj, nbk)
    if (!(2 + dx < i && i < nbi - 3 && 2 + dy < j && j < nbj - 3 && 2 + dz < kg && kg < g.nkg + dz - 3)) return;
    size_t id = (size_t)i + (size_t)nbi * j + (size_t)nbi * nbj * k;
    field[id] = field[id] + 0.0f;
}

// ---- A6/A8: cumulate_kernel (GPU_kernel.cu:376-436): dst += blend9(coeff*src(map(x))) ------
// ID: the map is the identity map of gpu_init_maps (mx/my/mz are not read).
template <bool P2, bool PT, int SD, int NF, bool ID, bool Q4 = false>
__global__ __launch_bounds__(256, NF == 1 ? 7 : 6) void cumulate_kernel(CumulateArgs<NF> a,
                                                       const float *mx, const float *my, const float *mz,
                                                       Spacing sp, Grid g, int dx, int dy, int dz, MapTabs tabs)
{
    const int nbi = g.ni + dx, nbj = g.nj + dy, nbk = g.nk + dz;
    BQ_IJK_WINDOW(1 + dx, nbi - 2, 1 + dy, nbj - 2, 1 + dz, g.nkg + dz - 2)
    if (block_out) return;
    const float h = sp.h;
    Map3 m{make_field(mx, g.ni, g.nj, g.nk, g.koff), make_field(my, g.ni, g.nj, g.nk, g.koff), make_field(mz, g.ni, g.nj, g.nk, g.koff)};
    Nine n = nine_setup(h, dx, dy, dz);
    f3 lo = mk3(0.f, 0.f, 0.f), hi = mk3(h * (float)g.ni, h * (float)g.nj, h * (float)g.nkg);
    f3 c = nine_centre(n, i, j, kg);
    const size_t id = (size_t)i + (size_t)nbi * j + (size_t)nbi * nbj * k;
    __shared__ float tile[kStaged<P2, PT, SD> && !ID ? 3 * kTile : 1];
    (void)tabs;
    if constexpr (ID) {
        // Identity map (node n holds n*h), power-of-two spacing.  A map component then varies along its
        // own axis only; map9's lerps along the other two axes combine equal values (lerp(a, a, c) == a
        // for the weights 0, 1/4, 1/2, 3/4) and the remaining one, between n*h and (n+1)*h, is exact in
        // fp32: the mapped positions ARE the node's 9 sample points, inside the clamp box.  In the
        // SOURCE field's own index space those are (i +- 1/4, j +- 1/4, k +- 1/4) and the centre: the
        // structured look-up with compile-time cells and weights (map9_component, unstaggered pattern)
        // applied to the field itself performs the very lerps locate() + gather() would.
        static_assert(P2 && !PT && SD >= 0, "identity shortcut: structured power-of-two path only");
        if (!active) return;                    // (this light kernel reads its nodes directly: staging them was slower)
#pragma unroll
        for (int f = 0; f < NF; f++) {
            Field src = make_field(a.src[f], nbi, nbj, nbk, g.koff);
            const float coeff = a.coeff[f];
            float s9[9];
            map9_component<0, 0, 0>(src, i, j, k, s9);
            float sum = 0.f;
#pragma unroll
            for (int ii = 0; ii < 8; ii++) sum += 0.125f * coeff * s9[ii];
            float value = coeff * s9[8];
            sum = (float)(0.5 * (double)sum + 0.5 * (double)value);
            a.dst[f][id] += sum;
        }
        return;
    }
    f3 mp[9];
    if constexpr (kStaged<P2, PT, SD>) {
        const Field mf[3] = {m.x, m.y, m.z};
        stage_tiles<3>(mf, i0, j0, k, tile);
        if (!active) return;
        if constexpr (P2) map9_lds<SD == 1, SD == 2, SD == 3, Q4>(tile, mp);
        else map9_lds_tab<SD == 1, SD == 2, SD == 3>(tile, tabs, i, j, kg, mp);
    } else {
        if (!active) return;
        mapped9<P2, PT, SD>(m, sp, n, c, i, j, k, mp);
    }
    const bool ge1 = NF == 1 && wave_all_ge<PT>(mp, h);   // (two fields: both code paths together need too many registers)
#pragma unroll
    for (int a9 = 0; a9 < 9; a9++) mp[a9] = clamp3_ordered(mp[a9], lo, hi);
    Field src[NF];
    float sum[NF], value[NF], w[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
        src[f] = make_field(a.src[f], nbi, nbj, nbk, g.koff);
        sum[f] = 0.f;
        w[f] = (PT ? 1.0f : 0.125f) * a.coeff[f];   // (0.125f * coeff) * sample: the reference's left-to-right product
    }
    if (NF == 1 && ge1) blend9_gather_w<P2, PT, NF, true>(src, sp, n.org, mp, w, sum, value);
    else                blend9_gather_w<P2, PT, NF, false>(src, sp, n.org, mp, w, sum, value);
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const float v = a.coeff[f] * value[f];
        a.dst[f][id] += (float)(0.5 * (double)sum[f] + 0.5 * (double)v);     // dst[0] == dst[1] is allowed: applied in order
    }
}

// ---- z-slab ranks: cumulate_kernel's expression on ONE wall layer of its index window, source in a wall-sheet copy --
// (include/bimocq_gpu.h: gpu_accumulate_wall_fixup.)  face_axis 0/1/2: the layer i == face_index, j == face_index or
// GLOBAL plane kg == face_index; threads cover the other two axes.  The map look-up is mapped9 (direct loads: the same
// lerps as the LDS-staged form), the gather is blend9_gather_w in its general form (bit-identical to the one-fma form
// wherever that applies) -- so a node comes out exactly as cumulate_kernel would produce it from the same values.
template <bool P2, int SD>
__global__ __launch_bounds__(256) void wall_fixup_kernel(const float *src, int src_nk, int src_koff,
                                                         const float *before, float *dst,
                                                         const float *mx, const float *my, const float *mz,
                                                         Spacing sp, Grid g, int dx, int dy, int dz, float coeff,
                                                         int face_axis, int face_index, int kw1)
{
    const int nbi = g.ni + dx, nbj = g.nj + dy, nbk = g.nk + dz;
    const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
    int i, j, k;
    if (face_axis == 0)      { i = face_index; j = a; k = b + g.kw0; }
    else if (face_axis == 1) { i = a; j = face_index; k = b + g.kw0; }
    else                     { i = a; j = b; k = face_index - g.koff; }
    if (i >= nbi || j >= nbj || k < g.kw0 || k >= kw1 || k >= nbk) return;
    const int kg = k + g.koff;
    if (!(1 + dx < i && i < nbi - 2 && 1 + dy < j && j < nbj - 2 && 1 + dz < kg && kg < g.nkg + dz - 2)) return;
    const float h = sp.h;
    Map3 m{make_field(mx, g.ni, g.nj, g.nk, g.koff), make_field(my, g.ni, g.nj, g.nk, g.koff), make_field(mz, g.ni, g.nj, g.nk, g.koff)};
    Nine n = nine_setup(h, dx, dy, dz);
    f3 lo = mk3(0.f, 0.f, 0.f), hi = mk3(h * (float)g.ni, h * (float)g.nj, h * (float)g.nkg);
    f3 c = nine_centre(n, i, j, kg);
    f3 mp[9];
    mapped9<P2, false, SD>(m, sp, n, c, i, j, k, mp);
#pragma unroll
    for (int a9 = 0; a9 < 9; a9++) mp[a9] = clamp3_ordered(mp[a9], lo, hi);
    const Field s1[1] = { make_field(src, nbi, nbj, src_nk, src_koff) };
    float sum[1] = { 0.f }, value[1], w[1] = { 0.125f * coeff };
    blend9_gather_w<P2, false, 1, false>(s1, sp, n.org, mp, w, sum, value);
    const float v = coeff * value[0];
    const size_t id = (size_t)i + (size_t)nbi * j + (size_t)nbi * nbj * k;
    dst[id] = before[id] + (float)(0.5 * (double)sum[0] + 0.5 * (double)v);
}

// ---- A6: compensate_kernel (GPU_kernel.cu:438-499): err = blend9(src(map(x))) - init(x) ----
template <bool P2, bool PT, int SD, int NF, bool Q4 = false>
__global__ __launch_bounds__(256, NF == 1 ? 7 : 6) void compensate_kernel(CompensateArgs<NF> a,
                                                         const float *mx, const float *my, const float *mz,
                                                         Spacing sp, Grid g, int dx, int dy, int dz, int fused, MapTabs tabs)
{
    const int nbi = g.ni + dx, nbj = g.nj + dy, nbk = g.nk + dz;
    BQ_IJK_WINDOW(1 + dx, nbi - 2, 1 + dy, nbj - 2, 1 + dz, g.nkg + dz - 2)
    const size_t id = (size_t)i + (size_t)nbi * j + (size_t)nbi * nbj * k;
    // FL_OPT_FUSED_HOUSEKEEPING.  Bit 2: init <- the uncompensated field, on EVERY node of the buffer (stage 2 of
    // gpu_compensate_*, GPU_kernel.cu:656-658).  A thread reads init only at its own node, before it stores there,
    // and no thread reads init anywhere else or writes src, so doing it here is race-free.  Bit 1: err = 0 outside
    // the window (the caller's clear).
    float init_own[NF];
    if (i < nbi && j < nbj) {
#pragma unroll
        for (int f = 0; f < NF; f++) {
            init_own[f] = a.init[f][id];
            if (fused & 2) a.init[f][id] = a.src[f][id];
            if ((fused & 1) && !active) a.err[f][id] = 0.f;
        }
    }
    if (block_out) return;
    const float h = sp.h;
    Map3 m{make_field(mx, g.ni, g.nj, g.nk, g.koff), make_field(my, g.ni, g.nj, g.nk, g.koff), make_field(mz, g.ni, g.nj, g.nk, g.koff)};
    Nine n = nine_setup(h, dx, dy, dz);
    f3 lo = mk3(0.f, 0.f, 0.f), hi = mk3(h * (float)g.ni, h * (float)g.nj, h * (float)g.nkg);
    f3 c = nine_centre(n, i, j, kg);
    f3 mp[9];
    if constexpr (kStaged<P2, PT, SD>) {
        __shared__ float tile[3 * kTile];
        const Field mf[3] = {m.x, m.y, m.z};
        stage_tiles<3>(mf, i0, j0, k, tile);
        if (!active) return;
        if constexpr (P2) map9_lds<SD == 1, SD == 2, SD == 3, Q4>(tile, mp);
        else map9_lds_tab<SD == 1, SD == 2, SD == 3>(tile, tabs, i, j, kg, mp);
    } else {
        if (!active) return;
        mapped9<P2, PT, SD>(m, sp, n, c, i, j, k, mp);
    }
    const bool ge1 = NF == 1 && wave_all_ge<PT>(mp, h);   // (two fields: both code paths together need too many registers)
#pragma unroll
    for (int a9 = 0; a9 < 9; a9++) mp[a9] = clamp3_ordered(mp[a9], lo, hi);
    Field src[NF];
    float sum[NF], value[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) { src[f] = make_field(a.src[f], nbi, nbj, nbk, g.koff); sum[f] = 0.f; }
    if (NF == 1 && ge1) blend9_gather<P2, PT, NF, true>(src, sp, n.org, mp, sum, value);
    else                blend9_gather<P2, PT, NF, false>(src, sp, n.org, mp, sum, value);
#pragma unroll
    for (int f = 0; f < NF; f++)
        a.err[f][id] = (float)(0.5 * (double)sum[f] + 0.5 * (double)value[f]) - init_own[f];
}

// ---- A6: clampExtrema_kernel (GPU_kernel.cu:146-167) --------------------------------------
__global__ __launch_bounds__(256) void clamp_box_kernel(const float *before, float *after, int ni, int nj, int nk,
                                                        int koff, int nkg)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= ni || j >= nj || k >= nk) return;
    const int kg = k + koff;                     // nkg: GLOBAL plane count of this buffer
    if (!(i > 0 && i < ni - 1 && j > 0 && j < nj - 1 && kg > 0 && kg < nkg - 1 && k > 0 && k < nk - 1)) return;
    const size_t sj = ni, sk = (size_t)ni * nj;
    const size_t id = (size_t)i + sj * j + sk * k;
    float mx = before[id], mn = mx;
#pragma unroll
    for (int kk = -1; kk <= 1; kk++)
#pragma unroll
        for (int jj = -1; jj <= 1; jj++)
#pragma unroll
            for (int ii = -1; ii <= 1; ii++) {
                float b = before[id + (ptrdiff_t)ii + (ptrdiff_t)jj * (ptrdiff_t)sj + (ptrdiff_t)kk * (ptrdiff_t)sk];
                if (b > mx) mx = b;
                if (b < mn) mn = b;
            }
    after[id] = fminf(fmaxf(mn, after[id]), mx);
}

// ---- N2: estimate_kernel (GPU_kernel.cu:501-537) ------------------------------------------
template <bool P2>
__global__ __launch_bounds__(256) void estimate_kernel(float *dist,
                                                       const float *xb, const float *yb, const float *zb,
                                                       const float *xf, const float *yf, const float *zf,
                                                       Spacing sp, Grid g)
{
    BQ_IJK(g.ni, g.nj, g.nk)
    if (!(i > 1 && i < g.ni - 2 && j > 1 && j < g.nj - 2 && kg > 1 && kg < g.nkg - 2)) return;
    const float h = sp.h;
    Map3 first{make_field(xb, g.ni, g.nj, g.nk, g.koff), make_field(yb, g.ni, g.nj, g.nk, g.koff), make_field(zb, g.ni, g.nj, g.nk, g.koff)};
    Map3 second{make_field(xf, g.ni, g.nj, g.nk, g.koff), make_field(yf, g.ni, g.nj, g.nk, g.koff), make_field(zf, g.ni, g.nj, g.nk, g.koff)};
    f3 pt = mk3(h * (float)i, h * (float)j, h * (float)kg);
    f3 back = map_at<P2>(first, sp, pt);
    f3 fwd = map_at<P2>(second, sp, back);
    float d_bf = (pt.x - fwd.x) * (pt.x - fwd.x) + (pt.y - fwd.y) * (pt.y - fwd.y) + (pt.z - fwd.z) * (pt.z - fwd.z);
    f3 f2 = map_at<P2>(second, sp, pt);
    f3 b2 = map_at<P2>(first, sp, f2);
    float d_fb = (pt.x - b2.x) * (pt.x - b2.x) + (pt.y - b2.y) * (pt.y - b2.y) + (pt.z - b2.z) * (pt.z - b2.z);
    dist[(size_t)i + (size_t)g.ni * j + (size_t)g.ni * g.nj * k] = fmaxf(d_bf, d_fb);
}

// ---- N3: semilag_kernel (GPU_kernel.cu:206-233) -------------------------------------------
template <bool P2>
__global__ __launch_bounds__(256) void semilag_kernel(float *field, const float *field_src,
                                                      const float *u, const float *v, const float *w,
                                                      Spacing sp, Grid g, int dx, int dy, int dz, float cfldt, float dt)
{
    const int bi = g.ni + dx, bj = g.nj + dy, bk = g.nk + dz;
    BQ_IJK(bi, bj, bk)
    if (!(i > 1 && i < bi - 2 - dx && j > 1 && j < bj - 2 - dy && kg > 1 && kg < g.nkg - 2)) return;
    const float h = sp.h;
    Vel3 vel{make_field(u, g.ni + 1, g.nj, g.nk, g.koff), make_field(v, g.ni, g.nj + 1, g.nk, g.koff), make_field(w, g.ni, g.nj, g.nk + 1, g.koff)};
    Field src = make_field(field_src, bi, bj, bk, g.koff);
    f3 org = mk3(-(float)dx * 0.5f * h, -(float)dy * 0.5f * h, -(float)dz * 0.5f * h);
    f3 hi = mk3((float)g.ni * h - h, (float)g.nj * h - h, (float)g.nkg * h - h);
    f3 pt = mk3(h * (float)i + org.x, h * (float)j + org.y, h * (float)kg + org.z);
    f3 pn = trace<P2>(vel, sp, hi, cfldt, dt, pt);
    field[(size_t)i + (size_t)bi * j + (size_t)bi * bj * k] = sample<P2>(src, sp, org, pn);
}

// ---- N3: clamp_extrema_kernel (GPU_kernel.cu:892-942), corrected -------------------------------
// MacCormack limiter of the reflection scheme.  The reference kernel adds the stagger offset with the wrong
// sign, uses the departure point's world coordinates as grid indices and tests/overwrites fieldTemp at that
// index from every thread: undefined output.  Built as it evidently means (oracle: orc_clamp_extrema):
// node x = (i - o) h, x_d = x - dt u(x - dt/2 u(x)) clamped to [h, (n-1)h]; min/max of the 8 values of
// `field` around x_d; fieldTemp at THIS node outside that range -> trilinear value of `field` at x_d.
template <bool P2>
__global__ __launch_bounds__(256) void clamp_extrema_kernel(const float *field, float *field_temp,
                                                            const float *u, const float *v, const float *w,
                                                            Spacing sp, int ni, int nj, int nk, int dx, int dy, int dz,
                                                            float ox, float oy, float oz, float dt, int koff, int nkg)
{
    // (ni, nj, nk: LOCAL buffer dims; z-slab ranks: local plane k is global plane k + koff of nkg cell planes)
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= ni || j >= nj || k >= nk) return;
    const float h = sp.h;
    const int ci = ni - dx, cj = nj - dy, ck = nk - dz;
    Vel3 vel{make_field(u, ci + 1, cj, ck, koff), make_field(v, ci, cj + 1, ck, koff), make_field(w, ci, cj, ck + 1, koff)};
    Field src = make_field(field, ni, nj, nk, koff);
    const f3 org = mk3(-ox * h, -oy * h, -oz * h);
    const f3 lo = mk3(h, h, h), hi = mk3((float)ci * h - h, (float)cj * h - h, (float)nkg * h - h);
    const float halfdt = 0.5f * dt;
    const f3 pt = mk3(h * (float)i + org.x, h * (float)j + org.y, h * (float)(k + koff) + org.z);
    f3 vl = get_velocity<P2>(vel, sp, pt);
    f3 px = mk3(pt.x - vl.x * halfdt, pt.y - vl.y * halfdt, pt.z - vl.z * halfdt);
    vl = get_velocity<P2>(vel, sp, px);
    px = clamp3(mk3(pt.x - vl.x * dt, pt.y - vl.y * dt, pt.z - vl.z * dt), lo, hi);
    const Cell c = locate<P2>(src, sp, org, px);
    float cv[8];
    corners(src, c, cv);
    const float mn = fminf(cv[0], fminf(cv[1], fminf(cv[2], fminf(cv[3], fminf(cv[4], fminf(cv[5], fminf(cv[6], cv[7])))))));
    const float mx = fmaxf(cv[0], fmaxf(cv[1], fmaxf(cv[2], fmaxf(cv[3], fmaxf(cv[4], fmaxf(cv[5], fmaxf(cv[6], cv[7])))))));
    const size_t id = (size_t)i + (size_t)ni * j + (size_t)ni * nj * k;
    const float t = field_temp[id];
    if (t < mn || t > mx) field_temp[id] = gather(src, c);
}

// ---- map-value scan behind FL_OPT_MAP_QUARTER_FP32 (bq_device.hip.h: tile_value_ok) ------------------------
__global__ __launch_bounds__(256) void maps_quarter_safe_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                const float *__restrict__ z, size_t n, float lo, float hi,
                                                                int *__restrict__ bad)
{
    bool ok = true;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (size_t)gridDim.x * 256)
        ok = ok && tile_value_ok(x[id], lo, hi) && tile_value_ok(y[id], lo, hi) && tile_value_ok(z[id], lo, hi);
    if (__any(!ok) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

} // inline namespace BQ_VARIANT
} // namespace bq

#include "bq_gather_march.hip.h"        // the same three operators as z-marching blocks with the field in an LDS window

namespace bq {
inline namespace BQ_VARIANT {

// ---- host-side dispatch helpers -----------------------------------------------------------
// local dims + the library's slab context (fl_set_slab); single GPU: koff = 0, nkg = nk
static inline Grid mk_grid(int ni, int nj, int nk)
{
    const Runtime &r = rt();
    if (r.slab_on) return Grid{ni, nj, nk, r.slab_koff, r.slab_nkg, 0};
    return Grid{ni, nj, nk, 0, nk, 0};
}
// the same with the plane window (fl_set_plane_window) applied: g.kw0 = first plane, *planes = how many planes of a
// buffer with nk + dz planes this launch covers (0: nothing to do)
static inline Grid mk_grid_win(int ni, int nj, int nk, int dz, int *planes)
{
    Grid g = mk_grid(ni, nj, nk);
    const Runtime &r = rt();
    int k0 = 0, k1 = nk + dz;
    if (r.win_on) {
        k0 = std::min(r.win_k0, nk + dz);
        k1 = r.win_k1 >= nk ? nk + dz : std::max(r.win_k1, k0);
    }
    g.kw0 = k0;
    *planes = k1 - k0;
    return g;
}

static bool dims_ok(int ni, int nj, int nk, const char *op)
{
    if (ni < 1 || nj < 1 || nk < 1) { latch(FL_ERR_BAD_ARGUMENT, op, "non-positive grid dims"); return false; }
    // byte offsets are 32-bit in the buffer descriptors and 2 GiB is the parking offset of out-of-range cells (bq_device.hip.h: corners()): (ni+1)*(nj+1)*(nk+1)*4 must stay below it
    double bytes = 4.0 * (double)(ni + 1) * (double)(nj + 1) * (double)(nk + 1);
    if (bytes >= 2147483648.0) { latch(FL_ERR_BAD_ARGUMENT, op, "field larger than 2 GiB"); return false; }
    if (nk + 1 > 65535) { latch(FL_ERR_BAD_ARGUMENT, op, "nk too large for grid.z"); return false; }
    // locate() forms the flat index with signed 24-bit multiplies: plane stride and indices must stay below 2^23
    if ((double)(ni + 1) * (double)(nj + 1) >= 8388608.0) { latch(FL_ERR_BAD_ARGUMENT, op, "plane larger than 2^23 elements"); return false; }
    return true;
}

// SDV: staggered axis + 1 (0 = none) -> structured power-of-two map look-up when the spacing allows it
// (FL_OPT_STRUCTURED_MAPS, default on); every other case takes the generic path (SD = -1).
#define BQ_DISPATCH2(KERNEL, P2V, PTV, SDV, GRID, ...)                                                      \
    do {                                                                                                    \
        hipStream_t st_ = rt().compute;                                                                     \
        if ((P2V) && !(PTV) && rt().opt_structured_maps) {                                                  \
            switch (SDV) {                                                                                  \
            case 0:  KERNEL<true, false, 0><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__); break;                  \
            case 1:  KERNEL<true, false, 1><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__); break;                  \
            case 2:  KERNEL<true, false, 2><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__); break;                  \
            default: KERNEL<true, false, 3><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__); break;                  \
            }                                                                                               \
        }                                                                                                   \
        else if (P2V) { if (PTV) KERNEL<true, true, -1><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__);             \
                        else     KERNEL<true, false, -1><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__); }          \
        else          { if (PTV) KERNEL<false, true, -1><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__);            \
                        else     KERNEL<false, false, -1><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__); }         \
        BQ_LAUNCH_CHECK(#KERNEL);                                                                           \
    } while (0)

#define BQ_DISPATCH1(KERNEL, P2V, GRID, ...)                                                                \
    do {                                                                                                    \
        hipStream_t st_ = rt().compute;                                                                     \
        if (P2V) KERNEL<true><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__);                                       \
        else     KERNEL<false><<<GRID, kBlock, 0, st_>>>(__VA_ARGS__);                                      \
        BQ_LAUNCH_CHECK(#KERNEL);                                                                           \
    } while (0)

// ---- tables of the structured look-up for spacings that are not a power of two (bq_device.hip.h: MapTabs) ----------------
// For every axis, stagger S and tap t the host evaluates, index by index, exactly what the kernels' generic path evaluates
// per tap: nine_centre ((float)idx * h + org, org = -S * 0.5f * h), nine_corner (+- 0.25f * h), map_at's locate with the
// map's origin 0 ((pos - 0.f) / h, floor, q - (float)floor) -- this translation unit is compiled -ffp-contract=off and the
// division is IEEE on both sides, so the numbers are the device's.  A pair (axis, S) conforms when the '+' / '-' taps land in
// the cell exact arithmetic predicts for every index >= 2 and the centre tap in that cell or the one below (unstaggered) --
// anything else (never seen) sends launches that need the pair down the generic path.
// Built once per (h, dims) and context; z is tabulated over the GLOBAL planes.
static MapTabs map_tabs(const Spacing &sp, const Grid &g, int *ok_mask)
{
    Runtime &r = rt();
    const int dims[3] = { g.ni, g.nj, g.nkg };
    if (r.map_tab_dev && r.map_tab_h == sp.h && r.map_tab_dims[0] == dims[0] && r.map_tab_dims[1] == dims[1] && r.map_tab_dims[2] == dims[2]) {
        *ok_mask = r.map_tab_ok;
        return MapTabs{ r.map_tab_dev, r.map_tab_dev + 18 * (size_t)r.map_tab_stride, r.map_tab_stride };
    }
    *ok_mask = 0;
    const int stride = std::max(dims[0], std::max(dims[1], dims[2])) + 2;
    std::vector<float> host((size_t)36 * stride, 0.f);              // 18 frac arrays, then 18 rel arrays
    const float h = sp.h, q = 0.25f * h, mq = -0.25f * h;
    int ok = 0;
    for (int a = 0; a < 3; a++)
        for (int S = 0; S < 2; S++) {
            bool conform = true;
            const float org = -(float)S * 0.5f * h;
            for (int t = 0; t < 3; t++) {
                float *frac = host.data() + (size_t)((a * 2 + S) * 3 + t) * stride;
                float *rel = frac + (size_t)18 * stride;
                for (int idx = 0; idx < dims[a] + S && idx < stride; idx++) {
                    const float c = (float)idx * h + org;
                    const float pos = t == 0 ? c + q : (t == 1 ? c + mq : c);
                    const float qx = (pos - 0.f) / h;
                    const float fl = floorf(qx);
                    const int cell = (int)fl;
                    frac[idx] = qx - (float)cell;
                    const int r0 = cell - (idx - 1);                // relative to the node block's first node
                    rel[idx] = (float)r0;
                    if (idx < 2) continue;                          // (never inside an operator's index window)
                    const int expect = S ? 0 : (t == 1 ? 0 : 1);    // tap_rel
                    if (t < 2 || S) { if (r0 != expect) conform = false; }
                    else if (r0 != 0 && r0 != 1) conform = false;
                }
            }
            if (conform) ok |= 1 << (a * 2 + S);
        }
    if (r.map_tab_dev) { (void)hipStreamSynchronize(r.compute); (void)hipFree(r.map_tab_dev); r.map_tab_dev = nullptr; }
    if (!BQ_HIP(hipMalloc((void **)&r.map_tab_dev, host.size() * sizeof(float)))) { r.map_tab_dev = nullptr; return MapTabs{ nullptr, nullptr, 0 }; }
    // (once per grid: on the compute stream, in front of the launches that read it, and waited for because `host` dies here)
    if (!BQ_HIP(hipMemcpyAsync(r.map_tab_dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, r.compute)) ||
        !BQ_HIP(hipStreamSynchronize(r.compute))) return MapTabs{ nullptr, nullptr, 0 };
    r.map_tab_h = sp.h; r.map_tab_dims[0] = dims[0]; r.map_tab_dims[1] = dims[1]; r.map_tab_dims[2] = dims[2];
    r.map_tab_stride = stride; r.map_tab_ok = ok;
    *ok_mask = ok;
    return MapTabs{ r.map_tab_dev, r.map_tab_dev + 18 * (size_t)stride, stride };
}
// the (axis, stagger) pairs a launch with staggered axis sd (0 none, 1 x, 2 y, 3 z) needs
static inline bool tabs_cover(int ok_mask, int sd)
{
    const int need = (1 << (0 * 2 + (sd == 1))) | (1 << (1 * 2 + (sd == 2))) | (1 << (2 * 2 + (sd == 3)));
    return (ok_mask & need) == need;
}

// Runtime (pow2 spacing, point sampling, staggered axis) -> compile-time <P2, PT, SD>; fn receives three
// integral_constant tags.  Returns whether a structured path was taken.  tabled: the spacing is not a power of two but the
// look-up tables cover this launch (map_tabs / tabs_cover) -- the structured kernels with P2 = false.
template <class Fn>
static bool dispatch_sd(bool p2, bool pt, int sd, Fn &&fn, bool tabled = false)
{
    using T = std::true_type; using F = std::false_type;
    bool structured = false;
    if (!p2 && !pt && tabled && rt().opt_structured_maps) {
        structured = true;
        switch (sd) {
        case 0:  fn(F{}, F{}, std::integral_constant<int, 0>{}); break;
        case 1:  fn(F{}, F{}, std::integral_constant<int, 1>{}); break;
        case 2:  fn(F{}, F{}, std::integral_constant<int, 2>{}); break;
        default: fn(F{}, F{}, std::integral_constant<int, 3>{}); break;
        }
        return structured;
    }
    if (p2 && !pt && rt().opt_structured_maps) {
        structured = true;
        switch (sd) {
        case 0:  fn(T{}, F{}, std::integral_constant<int, 0>{}); break;
        case 1:  fn(T{}, F{}, std::integral_constant<int, 1>{}); break;
        case 2:  fn(T{}, F{}, std::integral_constant<int, 2>{}); break;
        default: fn(T{}, F{}, std::integral_constant<int, 3>{}); break;
        }
    }
    else if (p2) { if (pt) fn(T{}, T{}, std::integral_constant<int, -1>{}); else fn(T{}, F{}, std::integral_constant<int, -1>{}); }
    else         { if (pt) fn(F{}, T{}, std::integral_constant<int, -1>{}); else fn(F{}, F{}, std::integral_constant<int, -1>{}); }
    return structured;
}
static inline int stag_axis(int dx, int dy, int dz) { return dx ? 1 : dy ? 2 : dz ? 3 : 0; }

// FL_OPT_FIELD_WINDOW: the structured power-of-two path of the three nine-point operators as gather_march_kernel.
// g carries the plane window (kw0); `planes` = how many planes from there.  Returns false when the option is off or the
// case is not the structured one (the caller then launches the one-plane kernel).
template <int KIND, int NF>
static bool march_launch(const MarchArgs<NF> &a, const float *mx, const float *my, const float *mz,
                         Spacing sp, Grid g, int planes, int dx, int dy, int dz, bool pt, bool q4, int fused)
{
    // -1 (default): on in the one-fma build, whose kernels the window moves from the texture-addresser path onto the VALU
    // floor (5-12 % per launch at 256^3); off in the exact build, which sits on its VALU floor already and only pays the
    // window's extra address arithmetic and lower occupancy (+8.6 % per step, profiles/r04_b_*)
#ifdef BQ_FAST_LERP
    const int opt = rt().opt_field_window < 0 ? 1 : rt().opt_field_window;
#else
    const int opt = rt().opt_field_window < 0 ? 0 : rt().opt_field_window;
#endif
    if (!opt || !sp.pow2 || pt || !rt().opt_structured_maps) return false;
    const int sd = stag_axis(dx, dy, dz);
    const int gx = (g.ni + dx + 63) / 64, gy = (g.nj + dy + 3) / 4;
    int kchunk = opt;
    if (opt == 1) {
        // enough blocks for every CU to hold its three a few times over, chunks long enough that the seven planes a block
        // stages before its first node stay a small share
        kchunk = 32;
        while (kchunk > 8 && (long)gx * gy * ((planes + kchunk - 1) / kchunk) < 3L * 4 * rt().num_cus) kchunk /= 2;
    }
    const dim3 grid(gx, gy, (planes + kchunk - 1) / kchunk);
    hipStream_t st = rt().compute;
    const int kw1 = g.kw0 + planes;
    // two fields per launch: the scalar pair (unstaggered) in all three operators, staggered components only in the
    // accumulation (gpu_accumulate_velocity2) -- nothing else is instantiated
    if (NF == 2 && KIND != kMarchCumulate && sd != 0) return false;
#define BQ_MARCH(SDV, Q4V) gather_march_kernel<KIND, SDV, NF, Q4V><<<grid, kBlock, 0, st>>>(a, mx, my, mz, sp, g, dx, dy, dz, fused, kchunk, kw1)
#define BQ_MARCH_SD(Q4V)                                                                     \
    if constexpr (NF == 2 && KIND != kMarchCumulate) { BQ_MARCH(0, Q4V); }                    \
    else switch (sd) { case 0: BQ_MARCH(0, Q4V); break; case 1: BQ_MARCH(1, Q4V); break; case 2: BQ_MARCH(2, Q4V); break; default: BQ_MARCH(3, Q4V); break; }
#ifdef BQ_FAST_LERP
    (void)q4;                                       // (the one-fma lerps do not know the quarter-weight distinction)
    BQ_MARCH_SD(false)
#else
    if (q4) { BQ_MARCH_SD(true) } else { BQ_MARCH_SD(false) }
#endif
#undef BQ_MARCH_SD
#undef BQ_MARCH
    BQ_LAUNCH_CHECK("gather_march_kernel");
    return true;
}

template <int NF>
static void advect_multi(AdvectArgs<NF> a, const float *bx, const float *by, const float *bz,
                         Spacing sp, Grid g, int dx, int dy, int dz, bool pt)
{
    int planes;
    g = mk_grid_win(g.ni, g.nj, g.nk, dz, &planes);      // plane window (fl_set_plane_window)
    if (planes <= 0) return;
    const dim3 grid = grid_for(g.ni + dx, g.nj + dy, planes);
    hipStream_t st = rt().compute;
    const bool q4 = rt().opt_map_quarter_fp32 != 0;
    {
        MarchArgs<NF> ma;
        for (int f = 0; f < NF; f++) { ma.src[f] = a.init[f]; ma.out[f] = a.field[f]; ma.aux[f] = nullptr; ma.coeff[f] = 1.f; }
        if (march_launch<kMarchAdvect, NF>(ma, bx, by, bz, sp, g, planes, dx, dy, dz, pt, q4, rt().opt_fused_housekeeping)) return;
    }
    int tab_ok = 0;
    const MapTabs tabs = (!sp.pow2 && !pt && rt().opt_structured_maps) ? map_tabs(sp, g, &tab_ok) : MapTabs{ nullptr, nullptr, 0 };
    dispatch_sd(sp.pow2, pt, stag_axis(dx, dy, dz), [&](auto P2, auto PT, auto SD) {
        constexpr bool p2 = decltype(P2)::value, ptc = decltype(PT)::value;
        constexpr int sd = decltype(SD)::value;
        if constexpr (p2 && kStaged<p2, ptc, sd>) {
            if (q4) { advect_kernel<p2, ptc, sd, NF, true><<<grid, kBlock, 0, st>>>(a, bx, by, bz, sp, g, dx, dy, dz, rt().opt_fused_housekeeping, tabs); return; }
        }
        advect_kernel<p2, ptc, sd, NF, false><<<grid, kBlock, 0, st>>>(a, bx, by, bz, sp, g, dx, dy, dz, rt().opt_fused_housekeeping, tabs);
    }, tabs.frac && tabs_cover(tab_ok, stag_axis(dx, dy, dz)));
    BQ_LAUNCH_CHECK("advect_kernel");
}
// identity: the caller vouches that mx/my/mz hold the identity map of gpu_init_maps; the shortcut is
// taken on the structured power-of-two path, otherwise the map is read like any other
template <int NF>
static void cumulate_multi(CumulateArgs<NF> a, const float *mx, const float *my, const float *mz,
                           Spacing sp, Grid g, int dx, int dy, int dz, bool pt, bool identity)
{
    int planes;
    g = mk_grid_win(g.ni, g.nj, g.nk, dz, &planes);
    if (planes <= 0) return;
    const dim3 grid = grid_for(g.ni + dx, g.nj + dy, planes);
    hipStream_t st = rt().compute;
    if (!identity) {
        MarchArgs<NF> ma;
        for (int f = 0; f < NF; f++) { ma.src[f] = a.src[f]; ma.out[f] = a.dst[f]; ma.aux[f] = nullptr; ma.coeff[f] = a.coeff[f]; }
        if (march_launch<kMarchCumulate, NF>(ma, mx, my, mz, sp, g, planes, dx, dy, dz, pt, rt().opt_map_quarter_fp32 != 0, 0)) return;
    }
    int tab_ok = 0;
    const MapTabs tabs = (!sp.pow2 && !pt && rt().opt_structured_maps) ? map_tabs(sp, g, &tab_ok) : MapTabs{ nullptr, nullptr, 0 };
    dispatch_sd(sp.pow2, pt, stag_axis(dx, dy, dz), [&](auto P2, auto PT, auto SD) {
        constexpr bool p2 = decltype(P2)::value, ptc = decltype(PT)::value;
        constexpr int sd = decltype(SD)::value;
        if constexpr (p2 && !ptc && sd >= 0) {
            if (identity) { cumulate_kernel<p2, ptc, sd, NF, true><<<grid, kBlock, 0, st>>>(a, mx, my, mz, sp, g, dx, dy, dz, tabs); return; }
            if (rt().opt_map_quarter_fp32) { cumulate_kernel<p2, ptc, sd, NF, false, true><<<grid, kBlock, 0, st>>>(a, mx, my, mz, sp, g, dx, dy, dz, tabs); return; }
        }
        cumulate_kernel<p2, ptc, sd, NF, false><<<grid, kBlock, 0, st>>>(a, mx, my, mz, sp, g, dx, dy, dz, tabs);
    }, tabs.frac && tabs_cover(tab_ok, stag_axis(dx, dy, dz)));
    BQ_LAUNCH_CHECK("cumulate_kernel");
}
template <int NF>
static void compensate_multi(CompensateArgs<NF> a, const float *mx, const float *my, const float *mz,
                             Spacing sp, Grid g, int dx, int dy, int dz, bool pt)
{
    int planes;
    g = mk_grid_win(g.ni, g.nj, g.nk, dz, &planes);
    if (planes <= 0) return;
    const dim3 grid = grid_for(g.ni + dx, g.nj + dy, planes);
    hipStream_t st = rt().compute;
    const bool q4 = rt().opt_map_quarter_fp32 != 0;
    {
        MarchArgs<NF> ma;
        for (int f = 0; f < NF; f++) { ma.src[f] = a.src[f]; ma.out[f] = a.err[f]; ma.aux[f] = a.init[f]; ma.coeff[f] = 1.f; }
        if (march_launch<kMarchCompensate, NF>(ma, mx, my, mz, sp, g, planes, dx, dy, dz, pt, q4, rt().opt_fused_housekeeping)) return;
    }
    int tab_ok = 0;
    const MapTabs tabs = (!sp.pow2 && !pt && rt().opt_structured_maps) ? map_tabs(sp, g, &tab_ok) : MapTabs{ nullptr, nullptr, 0 };
    dispatch_sd(sp.pow2, pt, stag_axis(dx, dy, dz), [&](auto P2, auto PT, auto SD) {
        constexpr bool p2 = decltype(P2)::value, ptc = decltype(PT)::value;
        constexpr int sd = decltype(SD)::value;
        if constexpr (p2 && kStaged<p2, ptc, sd>) {
            if (q4) { compensate_kernel<p2, ptc, sd, NF, true><<<grid, kBlock, 0, st>>>(a, mx, my, mz, sp, g, dx, dy, dz, rt().opt_fused_housekeeping, tabs); return; }
        }
        compensate_kernel<p2, ptc, sd, NF, false><<<grid, kBlock, 0, st>>>(a, mx, my, mz, sp, g, dx, dy, dz, rt().opt_fused_housekeeping, tabs);
    }, tabs.frac && tabs_cover(tab_ok, stag_axis(dx, dy, dz)));
    BQ_LAUNCH_CHECK("compensate_kernel");
}

static void advect_comp(float *f, const float *init, const float *bx, const float *by, const float *bz,
                        Spacing sp, Grid g, int dx, int dy, int dz, bool pt)
{
    advect_multi<1>(AdvectArgs<1>{{f}, {init}}, bx, by, bz, sp, g, dx, dy, dz, pt);
}
static void cumulate_comp(const float *src, float *dst, const float *mx, const float *my, const float *mz,
                          Spacing sp, Grid g, int dx, int dy, int dz, bool pt, float coeff, bool identity = false)
{
    cumulate_multi<1>(CumulateArgs<1>{{src}, {dst}, {coeff}}, mx, my, mz, sp, g, dx, dy, dz, pt, identity);
}
static void compensate_comp(const float *src, float *init, float *err, const float *mx, const float *my, const float *mz,
                            Spacing sp, Grid g, int dx, int dy, int dz, bool pt)
{
    compensate_multi<1>(CompensateArgs<1>{{src}, {init}, {err}}, mx, my, mz, sp, g, dx, dy, dz, pt);
}
static void double_comp(float *f, const float *prev, const float *bx, const float *by, const float *bz,
                        const float *px, const float *py, const float *pz,
                        Spacing sp, Grid g, int dx, int dy, int dz, bool pt, float blend, int prev_global = 0)
{
    // blend == 1: field*1 + 0*prev.  For finite prev that is field + (+-0): the value of every node is unchanged
    // (only a -0 would turn into +0, which no consumer can tell apart).  FL_OPT_SKIP_UNIT_BLEND = 1 (default): nothing
    // is launched; 2: the one-pass `field + 0` kernel (bit pattern of the reference); 0: the full kernel.
    if (blend == 1.0f && rt().opt_skip_unit_blend == 1) return;
    if (blend == 1.0f && rt().opt_skip_unit_blend == 2) {
        unit_blend_kernel<<<grid_for(g.ni + dx, g.nj + dy, g.nk + dz), kBlock, 0, rt().compute>>>(f, g, dx, dy, dz);
        BQ_LAUNCH_CHECK("unit_blend_kernel");
        return;
    }
    BQ_DISPATCH2(double_advect_kernel, sp.pow2, pt, (dx ? 1 : dy ? 2 : dz ? 3 : 0), grid_for(g.ni + dx, g.nj + dy, g.nk + dz), f, prev, bx, by, bz, px, py, pz, sp, g, dx, dy, dz, blend, prev_global);
}
// The same limiter, separable and k-marching: a thread owns one float4 column of row j, reduces min/max over the
// 3x3 (x, y) neighbourhood of each plane once (rows j-1, j, j+1 as float4 loads, x-neighbours from the neighbouring
// lanes) and combines three consecutive planes from registers -- 3 vector loads per plane instead of 27 scalar
// ones per node.  min/max are exact and order-free, so the result is the reference's for every finite field.
// Rows of at most 64 float4 (one wave); the launcher falls back to clamp_box_kernel otherwise.
struct MinMax4 { float4 lo, hi; };
__global__ __launch_bounds__(256) void clamp_box_march_kernel(const float *__restrict__ before, float *__restrict__ after,
                                                              int ni, int nj, int nk, int cw, int nby, int kchunk, int koff, int nkg)
{
    const int by = blockIdx.x % nby, bz = blockIdx.x / nby;
    const int rows = 256 / cw;
    const int c = threadIdx.x % cw, r = threadIdx.x / cw;
    const int xraw = 4 * c, j = by * rows + r;
    const int kA = max(1, 1 - koff), kB = min(nk - 1, nkg - 1 - koff);          // local planes the limiter updates
    const int kbeg = max(kA, bz * kchunk), kend = min(kB, bz * kchunk + kchunk);
    if (kbeg >= kend) return;
    const bool xok = xraw + 4 <= ni;        // whole float4 inside the row (a row of 4m + 1 floats: its last column is the tail)
    const bool active = xok && j >= 1 && j <= nj - 2;
    const int x = xok ? xraw : ni - 4;
    const size_t sj = ni, sk = (size_t)ni * nj;
    const size_t o_m = (size_t)x + sj * (size_t)min(max(j - 1, 0), nj - 1), o_0 = (size_t)x + sj * (size_t)min(max(j, 0), nj - 1),
                 o_p = (size_t)x + sj * (size_t)min(max(j + 1, 0), nj - 1);
    auto ld4 = [&](size_t off) -> float4 { return *reinterpret_cast<const float4 *>(before + off); };
    auto mn3 = [](float a, float b, float c2) { return fminf(fminf(a, b), c2); };
    auto mx3 = [](float a, float b, float c2) { return fmaxf(fmaxf(a, b), c2); };
    // min/max over the 3x3 (x, y) neighbourhood of every cell of this float4 on plane pl
    // rows of 4*cw + 1 floats (the u component: 257): the last lane of a row fetches the one extra column itself
    const bool tail = ni % 4 == 1 && xraw + 4 == ni - 1;
    // rows of several waves (cw > 64: 512 and 1024 wide grids): the lanes at a wave's ends fetch the column just outside
    // their wave themselves (three scalar loads per plane), what the lane exchange cannot reach
    const int lane = threadIdx.x & 63;
    const bool wedgeL = cw > 64 && lane == 0 && c > 0 && xok;
    const bool wedgeR = cw > 64 && lane == 63 && xraw + 4 < ni && !tail;
    auto plane_box = [&](int pl) -> MinMax4 {
        const size_t p0 = sk * (size_t)min(max(pl, 0), nk - 1);
        const float4 a = ld4(p0 + o_m), b = ld4(p0 + o_0), d = ld4(p0 + o_p);
        float4 lo = make_float4(mn3(a.x, b.x, d.x), mn3(a.y, b.y, d.y), mn3(a.z, b.z, d.z), mn3(a.w, b.w, d.w));
        float4 hi = make_float4(mx3(a.x, b.x, d.x), mx3(a.y, b.y, d.y), mx3(a.z, b.z, d.z), mx3(a.w, b.w, d.w));
        float llo = lane_up(lo.w), rlo = lane_down(lo.x), lhi = lane_up(hi.w), rhi = lane_down(hi.x);
        if (tail || wedgeR) {
            const float ta = before[p0 + o_m + 4], tb = before[p0 + o_0 + 4], td = before[p0 + o_p + 4];
            rlo = mn3(ta, tb, td); rhi = mx3(ta, tb, td);
        }
        if (wedgeL) {
            const float ta = before[p0 + o_m - 1], tb = before[p0 + o_0 - 1], td = before[p0 + o_p - 1];
            llo = mn3(ta, tb, td); lhi = mx3(ta, tb, td);
        }
        MinMax4 m;
        m.lo = make_float4(mn3(llo, lo.x, lo.y), mn3(lo.x, lo.y, lo.z), mn3(lo.y, lo.z, lo.w), mn3(lo.z, lo.w, rlo));
        m.hi = make_float4(mx3(lhi, hi.x, hi.y), mx3(hi.x, hi.y, hi.z), mx3(hi.y, hi.z, hi.w), mx3(hi.z, hi.w, rhi));
        return m;
    };
    MinMax4 Pm = plane_box(kbeg - 1), Pc = plane_box(kbeg);
    for (int k = kbeg; k < kend; k++) {
        const MinMax4 Pn = plane_box(k + 1);
        if (active) {
            float *dst = after + (size_t)x + sj * j + sk * k;
            const float4 v = *reinterpret_cast<const float4 *>(dst);
            float4 o;
            o.x = fminf(fmaxf(mn3(Pm.lo.x, Pc.lo.x, Pn.lo.x), v.x), mx3(Pm.hi.x, Pc.hi.x, Pn.hi.x));
            o.y = fminf(fmaxf(mn3(Pm.lo.y, Pc.lo.y, Pn.lo.y), v.y), mx3(Pm.hi.y, Pc.hi.y, Pn.hi.y));
            o.z = fminf(fmaxf(mn3(Pm.lo.z, Pc.lo.z, Pn.lo.z), v.z), mx3(Pm.hi.z, Pc.hi.z, Pn.hi.z));
            o.w = fminf(fmaxf(mn3(Pm.lo.w, Pc.lo.w, Pn.lo.w), v.w), mx3(Pm.hi.w, Pc.hi.w, Pn.hi.w));
            if (x >= 4 && x + 4 < ni) {
                *reinterpret_cast<float4 *>(dst) = o;
            } else {                            // the float4 touches the x boundary: interior cells only
                if (x >= 1) dst[0] = o.x;
                dst[1] = o.y;
                dst[2] = o.z;
                if (x + 3 < ni - 1) dst[3] = o.w;
            }
        }
        Pm = Pc; Pc = Pn;
    }
}

// nk: local buffer planes; dz: 1 for the w buffer (its global plane count is nkg + 1)
static void clamp_box(const float *before, float *after, int ni, int nj, int nk, int dz)
{
    Grid g = mk_grid(1, 1, nk - dz);
    // float4 columns; rows of 4m + 1 floats (u: 257) go through the same kernel with unaligned 16-byte accesses and one
    // extra column fetched by the last lane (102 -> 48 us at 257 x 256 x 256)
    const int nv = ni % 4 == 1 ? ni - 1 : ni;            // floats per row handled as float4
    const bool pow2row = nv >= 32 && nv <= 1024 && (nv & (nv - 1)) == 0;
    const bool vec_ok = nj >= 3 && nk >= 3 && rt().opt_jacobi_variant != 1 &&
                        ((ni % 4 == 0 && ni >= 32 && ni <= 1024 && (((uintptr_t)before | (uintptr_t)after) & 15u) == 0) ||
                         (ni % 4 == 1 && pow2row && (((uintptr_t)before | (uintptr_t)after) & 3u) == 0));
    if (vec_ok) {
        int cw = 16;
        while (cw * 4 < nv) cw *= 2;
        const int rows = 256 / cw, nby = (nj + rows - 1) / rows;
        int kchunk = 32;
        while (kchunk > 8 && (long)nby * ((nk + kchunk - 1) / kchunk) < 1024) kchunk /= 2;
        const int nbz = (nk + kchunk - 1) / kchunk;
        clamp_box_march_kernel<<<nby * nbz, 256, 0, rt().compute>>>(before, after, ni, nj, nk, cw, nby, kchunk, g.koff, g.nkg + dz);
        BQ_LAUNCH_CHECK("clamp_box_march_kernel");
        return;
    }
    clamp_box_kernel<<<grid_for(ni, nj, nk), kBlock, 0, rt().compute>>>(before, after, ni, nj, nk, g.koff, g.nkg + dz);
    BQ_LAUNCH_CHECK("clamp_box_kernel");
}

} // inline namespace BQ_VARIANT
} // namespace bq

using namespace bq;

// Entry points of this file exist twice in the library: this translation unit is compiled once as is (exact
// arithmetic, the C-ABI names) and once with -DBQ_FAST_LERP (names suffixed _fast).  The exact build's entry
// point forwards to its _fast twin when FL_OPT_FAST_LERP is set.
#ifdef BQ_FAST_LERP
#define BQ_ENTRY(name, params, args) void name##_fast params
#else
#define BQ_ENTRY(name, params, args)                                                     \
    void name##_fast params;                                                             \
    static void name##_impl params;                                                      \
    void name params { if (rt().opt_fast_lerp) { name##_fast args; return; } name##_impl args; } \
    static void name##_impl params
#endif

#define BQ_ENTER(op, ...)                                                  \
    if (!ensure_ready(op)) return;                                         \
    if (!dims_ok(ni, nj, nk, op)) return;                                  \
    {                                                                      \
        const void *ptrs_[] = { __VA_ARGS__ };                             \
        for (const void *p_ : ptrs_)                                       \
            if (!p_) { latch(FL_ERR_BAD_ARGUMENT, op, "null device pointer"); return; } \
    }

extern "C" {

BQ_ENTRY(gpu_solve_forward, (float *u, float *v, float *w, float *x_fwd, float *y_fwd, float *z_fwd,
                       float h, int ni, int nj, int nk, float cfldt, float dt), (u, v, w, x_fwd, y_fwd, z_fwd, h, ni, nj, nk, cfldt, dt))
{
    BQ_ENTER("gpu_solve_forward", u, v, w, x_fwd, y_fwd, z_fwd)
    BQ_REQUIRE(cfldt > 0.f || dt == 0.f, "gpu_solve_forward");     // cfldt <= 0 would never terminate
    int planes;
    Spacing sp = make_spacing(h); Grid g = mk_grid_win(ni, nj, nk, 0, &planes);
    if (planes <= 0) return;
    int *guard = rt().map_guard_on ? rt().map_guard + 1 : nullptr;
    BQ_DISPATCH1(forward_kernel, sp.pow2, grid_for(ni, nj, planes), u, v, w, x_fwd, y_fwd, z_fwd, sp, g, cfldt, dt, guard);
}

BQ_ENTRY(gpu_solve_backwardDMC, (float *u, float *v, float *w, float *x_in, float *y_in, float *z_in,
                           float *x_out, float *y_out, float *z_out,
                           float h, int ni, int nj, int nk, float substep), (u, v, w, x_in, y_in, z_in, x_out, y_out, z_out, h, ni, nj, nk, substep))
{
    BQ_ENTER("gpu_solve_backwardDMC", u, v, w, x_in, y_in, z_in, x_out, y_out, z_out)
    BQ_REQUIRE(x_in != x_out && y_in != y_out && z_in != z_out, "gpu_solve_backwardDMC");
    int planes;
    Spacing sp = make_spacing(h); Grid g = mk_grid_win(ni, nj, nk, 0, &planes);
    if (planes <= 0) return;
    const int border = (rt().opt_fused_housekeeping & 8) ? 2 : (rt().opt_fused_housekeeping & 4) ? 1 : 0;
    int *guard = rt().map_guard_on ? rt().map_guard : nullptr;
    BQ_DISPATCH1(dmc_kernel, sp.pow2, grid_for(ni, nj, planes), u, v, w, x_in, y_in, z_in, x_out, y_out, z_out, sp, g, substep, border, guard);
}

BQ_ENTRY(gpu_advect_velocity, (float *u, float *v, float *w, float *u_init, float *v_init, float *w_init,
                         float *backward_x, float *backward_y, float *backward_z,
                         float h, int ni, int nj, int nk, bool is_point), (u, v, w, u_init, v_init, w_init, backward_x, backward_y, backward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_advect_velocity", u, v, w, u_init, v_init, w_init, backward_x, backward_y, backward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    advect_comp(u, u_init, backward_x, backward_y, backward_z, sp, g, 1, 0, 0, is_point);
    advect_comp(v, v_init, backward_x, backward_y, backward_z, sp, g, 0, 1, 0, is_point);
    advect_comp(w, w_init, backward_x, backward_y, backward_z, sp, g, 0, 0, 1, is_point);
}

BQ_ENTRY(gpu_advect_vel_double, (float *u, float *v, float *w, float *utemp, float *vtemp, float *wtemp,
                           float *backward_x, float *backward_y, float *backward_z,
                           float *backward_xprev, float *backward_yprev, float *backward_zprev,
                           float h, int ni, int nj, int nk, bool is_point, float blend_coeff), (u, v, w, utemp, vtemp, wtemp, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, h, ni, nj, nk, is_point, blend_coeff))
{
    BQ_ENTER("gpu_advect_vel_double", u, v, w, utemp, vtemp, wtemp, backward_x, backward_y, backward_z,
             backward_xprev, backward_yprev, backward_zprev)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    double_comp(u, utemp, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, sp, g, 1, 0, 0, is_point, blend_coeff);
    double_comp(v, vtemp, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, sp, g, 0, 1, 0, is_point, blend_coeff);
    double_comp(w, wtemp, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, sp, g, 0, 0, 1, is_point, blend_coeff);
}

BQ_ENTRY(gpu_advect_field, (float *field, float *field_init, float *backward_x, float *backward_y, float *backward_z,
                      float h, int ni, int nj, int nk, bool is_point), (field, field_init, backward_x, backward_y, backward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_advect_field", field, field_init, backward_x, backward_y, backward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    advect_comp(field, field_init, backward_x, backward_y, backward_z, sp, g, 0, 0, 0, is_point);
}

BQ_ENTRY(gpu_advect_field_double, (float *field, float *field_prev, float *backward_x, float *backward_y, float *backward_z,
                             float *backward_xprev, float *backward_yprev, float *backward_zprev,
                             float h, int ni, int nj, int nk, bool is_point, float blend_coeff), (field, field_prev, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, h, ni, nj, nk, is_point, blend_coeff))
{
    BQ_ENTER("gpu_advect_field_double", field, field_prev, backward_x, backward_y, backward_z,
             backward_xprev, backward_yprev, backward_zprev)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    double_comp(field, field_prev, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, sp, g, 0, 0, 0, is_point, blend_coeff);
}

// the two-level advection on a z-slab rank with the *_prev fields of the WHOLE grid (include/bimocq_gpu.h)
BQ_ENTRY(gpu_advect_vel_double_global, (float *u, float *v, float *w, float *uprev_g, float *vprev_g, float *wprev_g,
                           float *backward_x, float *backward_y, float *backward_z,
                           float *backward_xprev, float *backward_yprev, float *backward_zprev,
                           float h, int ni, int nj, int nk, bool is_point, float blend_coeff), (u, v, w, uprev_g, vprev_g, wprev_g, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, h, ni, nj, nk, is_point, blend_coeff))
{
    BQ_ENTER("gpu_advect_vel_double_global", u, v, w, uprev_g, vprev_g, wprev_g, backward_x, backward_y, backward_z,
             backward_xprev, backward_yprev, backward_zprev)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    BQ_REQUIRE((double)(ni + 1) * (nj + 1) * (g.nkg + 1) * 4.0 < 2147483648.0, "gpu_advect_vel_double_global");
    double_comp(u, uprev_g, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, sp, g, 1, 0, 0, is_point, blend_coeff, 1);
    double_comp(v, vprev_g, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, sp, g, 0, 1, 0, is_point, blend_coeff, 1);
    double_comp(w, wprev_g, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, sp, g, 0, 0, 1, is_point, blend_coeff, 1);
}

BQ_ENTRY(gpu_advect_field_double_global, (float *field, float *field_prev_g, float *backward_x, float *backward_y, float *backward_z,
                             float *backward_xprev, float *backward_yprev, float *backward_zprev,
                             float h, int ni, int nj, int nk, bool is_point, float blend_coeff), (field, field_prev_g, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, h, ni, nj, nk, is_point, blend_coeff))
{
    BQ_ENTER("gpu_advect_field_double_global", field, field_prev_g, backward_x, backward_y, backward_z,
             backward_xprev, backward_yprev, backward_zprev)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    BQ_REQUIRE((double)ni * nj * g.nkg * 4.0 < 2147483648.0, "gpu_advect_field_double_global");
    double_comp(field, field_prev_g, backward_x, backward_y, backward_z, backward_xprev, backward_yprev, backward_zprev, sp, g, 0, 0, 0, is_point, blend_coeff, 1);
}

BQ_ENTRY(gpu_accumulate_velocity, (float *u_change, float *v_change, float *w_change,
                             float *du_init, float *dv_init, float *dw_init,
                             float *forward_x, float *forward_y, float *forward_z,
                             float h, int ni, int nj, int nk, bool is_point, float coeff), (u_change, v_change, w_change, du_init, dv_init, dw_init, forward_x, forward_y, forward_z, h, ni, nj, nk, is_point, coeff))
{
    BQ_ENTER("gpu_accumulate_velocity", u_change, v_change, w_change, du_init, dv_init, dw_init, forward_x, forward_y, forward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    cumulate_comp(u_change, du_init, forward_x, forward_y, forward_z, sp, g, 1, 0, 0, is_point, coeff);
    cumulate_comp(v_change, dv_init, forward_x, forward_y, forward_z, sp, g, 0, 1, 0, is_point, coeff);
    cumulate_comp(w_change, dw_init, forward_x, forward_y, forward_z, sp, g, 0, 0, 1, is_point, coeff);
}

BQ_ENTRY(gpu_accumulate_field, (float *field_change, float *dfield_init, float *forward_x, float *forward_y, float *forward_z,
                          float h, int ni, int nj, int nk, bool is_point, float coeff), (field_change, dfield_init, forward_x, forward_y, forward_z, h, ni, nj, nk, is_point, coeff))
{
    BQ_ENTER("gpu_accumulate_field", field_change, dfield_init, forward_x, forward_y, forward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    cumulate_comp(field_change, dfield_init, forward_x, forward_y, forward_z, sp, g, 0, 0, 0, is_point, coeff);
}

// ---- batched / shortcut forms of the entry points above (additive; same arithmetic) ----------------
BQ_ENTRY(gpu_advect_field2, (float *field1, float *field1_init, float *field2, float *field2_init,
                       float *backward_x, float *backward_y, float *backward_z,
                       float h, int ni, int nj, int nk, bool is_point), (field1, field1_init, field2, field2_init, backward_x, backward_y, backward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_advect_field2", field1, field1_init, field2, field2_init, backward_x, backward_y, backward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    advect_multi<2>(AdvectArgs<2>{{field1, field2}, {field1_init, field2_init}}, backward_x, backward_y, backward_z, sp, g, 0, 0, 0, is_point);
}

BQ_ENTRY(gpu_compensate_error_field2, (float *u1, float *du1, float *u1_src, float *u2, float *du2, float *u2_src,
                                 float *forward_x, float *forward_y, float *forward_z,
                                 float h, int ni, int nj, int nk, bool is_point), (u1, du1, u1_src, u2, du2, u2_src, forward_x, forward_y, forward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_compensate_error_field2", u1, du1, u1_src, u2, du2, u2_src, forward_x, forward_y, forward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    compensate_multi<2>(CompensateArgs<2>{{u1, u2}, {du1, du2}, {u1_src, u2_src}}, forward_x, forward_y, forward_z, sp, g, 0, 0, 0, is_point);
}

BQ_ENTRY(gpu_accumulate_field2, (float *change1, float *dinit1, float coeff1, float *change2, float *dinit2, float coeff2,
                           float *forward_x, float *forward_y, float *forward_z,
                           float h, int ni, int nj, int nk, bool is_point), (change1, dinit1, coeff1, change2, dinit2, coeff2, forward_x, forward_y, forward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_accumulate_field2", change1, dinit1, change2, dinit2, forward_x, forward_y, forward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    cumulate_multi<2>(CumulateArgs<2>{{change1, change2}, {dinit1, dinit2}, {coeff1, coeff2}}, forward_x, forward_y, forward_z, sp, g, 0, 0, 0, is_point, false);
}

// du_init += blend9(coeff1 * change1(psi(x))), then += blend9(coeff2 * change2(psi(x))): two
// gpu_accumulate_velocity calls with one map look-up
BQ_ENTRY(gpu_accumulate_velocity2, (float *u_change1, float *v_change1, float *w_change1, float coeff1,
                              float *u_change2, float *v_change2, float *w_change2, float coeff2,
                              float *du_init, float *dv_init, float *dw_init,
                              float *forward_x, float *forward_y, float *forward_z,
                              float h, int ni, int nj, int nk, bool is_point), (u_change1, v_change1, w_change1, coeff1, u_change2, v_change2, w_change2, coeff2, du_init, dv_init, dw_init, forward_x, forward_y, forward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_accumulate_velocity2", u_change1, v_change1, w_change1, u_change2, v_change2, w_change2,
             du_init, dv_init, dw_init, forward_x, forward_y, forward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    cumulate_multi<2>(CumulateArgs<2>{{u_change1, u_change2}, {du_init, du_init}, {coeff1, coeff2}}, forward_x, forward_y, forward_z, sp, g, 1, 0, 0, is_point, false);
    cumulate_multi<2>(CumulateArgs<2>{{v_change1, v_change2}, {dv_init, dv_init}, {coeff1, coeff2}}, forward_x, forward_y, forward_z, sp, g, 0, 1, 0, is_point, false);
    cumulate_multi<2>(CumulateArgs<2>{{w_change1, w_change2}, {dw_init, dw_init}, {coeff1, coeff2}}, forward_x, forward_y, forward_z, sp, g, 0, 0, 1, is_point, false);
}

// One velocity component of gpu_accumulate_velocity (axis 0/1/2 = u/v/w) with one or two sources:
//   d_init += blend9(coeff1 * change1(psi(x))) [ ; += blend9(coeff2 * change2(psi(x))) when change2 != NULL ]
// lets a host skip a component's source that is known to be identically zero.
BQ_ENTRY(gpu_accumulate_component, (float *change1, float coeff1, float *change2, float coeff2, float *d_init,
                              float *forward_x, float *forward_y, float *forward_z,
                              float h, int ni, int nj, int nk, int axis, bool is_point), (change1, coeff1, change2, coeff2, d_init, forward_x, forward_y, forward_z, h, ni, nj, nk, axis, is_point))
{
    BQ_ENTER("gpu_accumulate_component", change1, d_init, forward_x, forward_y, forward_z)
    BQ_REQUIRE(axis >= 0 && axis <= 2, "gpu_accumulate_component");
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    const int dx = axis == 0, dy = axis == 1, dz = axis == 2;
    if (change2)
        cumulate_multi<2>(CumulateArgs<2>{{change1, change2}, {d_init, d_init}, {coeff1, coeff2}}, forward_x, forward_y, forward_z,
                          sp, g, dx, dy, dz, is_point, false);
    else
        cumulate_comp(change1, d_init, forward_x, forward_y, forward_z, sp, g, dx, dy, dz, is_point, coeff1);
}

// gpu_accumulate_velocity for a forward map that IS the identity map of gpu_init_maps (right after a
// re-initialisation): the caller vouches for that; the map buffers are still passed and are read on every
// path without the shortcut (spacing not a power of two, FL_OPT_STRUCTURED_MAPS off).
BQ_ENTRY(gpu_accumulate_velocity_identity, (float *u_change, float *v_change, float *w_change,
                                      float *du_init, float *dv_init, float *dw_init,
                                      float *forward_x, float *forward_y, float *forward_z,
                                      float h, int ni, int nj, int nk, bool is_point, float coeff), (u_change, v_change, w_change, du_init, dv_init, dw_init, forward_x, forward_y, forward_z, h, ni, nj, nk, is_point, coeff))
{
    BQ_ENTER("gpu_accumulate_velocity_identity", u_change, v_change, w_change, du_init, dv_init, dw_init, forward_x, forward_y, forward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    cumulate_comp(u_change, du_init, forward_x, forward_y, forward_z, sp, g, 1, 0, 0, is_point, coeff, true);
    cumulate_comp(v_change, dv_init, forward_x, forward_y, forward_z, sp, g, 0, 1, 0, is_point, coeff, true);
    cumulate_comp(w_change, dw_init, forward_x, forward_y, forward_z, sp, g, 0, 0, 1, is_point, coeff, true);
}

BQ_ENTRY(gpu_estimate_distortion, (float *du, float *x_init, float *y_init, float *z_init,
                             float *x_fwd, float *y_fwd, float *z_fwd, float h, int ni, int nj, int nk), (du, x_init, y_init, z_init, x_fwd, y_fwd, z_fwd, h, ni, nj, nk))
{
    BQ_ENTER("gpu_estimate_distortion", du, x_init, y_init, z_init, x_fwd, y_fwd, z_fwd)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    BQ_DISPATCH1(estimate_kernel, sp.pow2, grid_for(ni, nj, nk), du, x_init, y_init, z_init, x_fwd, y_fwd, z_fwd, sp, g);
}

BQ_ENTRY(gpu_compensate_velocity, (float *u, float *v, float *w, float *du, float *dv, float *dw,
                             float *u_src, float *v_src, float *w_src,
                             float *forward_x, float *forward_y, float *forward_z,
                             float *backward_x, float *backward_y, float *backward_z,
                             float h, int ni, int nj, int nk, bool is_point), (u, v, w, du, dv, dw, u_src, v_src, w_src, forward_x, forward_y, forward_z, backward_x, backward_y, backward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_compensate_velocity", u, v, w, du, dv, dw, u_src, v_src, w_src,
             forward_x, forward_y, forward_z, backward_x, backward_y, backward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    const size_t nu = (size_t)(ni + 1) * nj * nk, nv = (size_t)ni * (nj + 1) * nk, nw = (size_t)ni * nj * (nk + 1);
    // error at time 0 into *_src (GPU_kernel.cu:652-654)
    compensate_comp(u, du, u_src, forward_x, forward_y, forward_z, sp, g, 1, 0, 0, is_point);
    compensate_comp(v, dv, v_src, forward_x, forward_y, forward_z, sp, g, 0, 1, 0, is_point);
    compensate_comp(w, dw, w_src, forward_x, forward_y, forward_z, sp, g, 0, 0, 1, is_point);
    // d* <- uncompensated field (:656-658; clobbers the caller's `init`)
    if (!(rt().opt_fused_housekeeping & 2)) {        // (with that bit the error kernels have stored it already)
        fl_memcpy_d2d(du, u, nu * sizeof(float));
        fl_memcpy_d2d(dv, v, nv * sizeof(float));
        fl_memcpy_d2d(dw, w, nw * sizeof(float));
    }
    // subtract half the back-mapped error (:659-661)
    cumulate_comp(u_src, u, backward_x, backward_y, backward_z, sp, g, 1, 0, 0, is_point, -0.5f);
    cumulate_comp(v_src, v, backward_x, backward_y, backward_z, sp, g, 0, 1, 0, is_point, -0.5f);
    cumulate_comp(w_src, w, backward_x, backward_y, backward_z, sp, g, 0, 0, 1, is_point, -0.5f);
    // limiter (:663-665)
    clamp_box(du, u, ni + 1, nj, nk, 0);
    clamp_box(dv, v, ni, nj + 1, nk, 0);
    clamp_box(dw, w, ni, nj, nk + 1, 1);
}

BQ_ENTRY(gpu_compensate_field, (float *u, float *du, float *u_src,
                          float *forward_x, float *forward_y, float *forward_z,
                          float *backward_x, float *backward_y, float *backward_z,
                          float h, int ni, int nj, int nk, bool is_point), (u, du, u_src, forward_x, forward_y, forward_z, backward_x, backward_y, backward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_compensate_field", u, du, u_src, forward_x, forward_y, forward_z, backward_x, backward_y, backward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    compensate_comp(u, du, u_src, forward_x, forward_y, forward_z, sp, g, 0, 0, 0, is_point);
    if (!(rt().opt_fused_housekeeping & 2)) fl_memcpy_d2d(du, u, (size_t)ni * nj * nk * sizeof(float));
    cumulate_comp(u_src, u, backward_x, backward_y, backward_z, sp, g, 0, 0, 0, is_point, -0.5f);
    clamp_box(du, u, ni, nj, nk, 0);
}

BQ_ENTRY(gpu_semilag, (float *field, float *field_src, float *u, float *v, float *w,
                 int dim_x, int dim_y, int dim_z, float h, int ni, int nj, int nk, float cfldt, float dt), (field, field_src, u, v, w, dim_x, dim_y, dim_z, h, ni, nj, nk, cfldt, dt))
{
    BQ_ENTER("gpu_semilag", field, field_src, u, v, w)
    BQ_REQUIRE(cfldt > 0.f || dt == 0.f, "gpu_semilag");
    BQ_REQUIRE((dim_x | dim_y | dim_z) == 0 || (dim_x + dim_y + dim_z) == 1, "gpu_semilag");
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    BQ_DISPATCH1(semilag_kernel, sp.pow2, grid_for(ni + dim_x, nj + dim_y, nk + dim_z), field, field_src, u, v, w, sp, g, dim_x, dim_y, dim_z, cfldt, dt);
}

BQ_ENTRY(gpu_clamp_extrema, (float *field, float *fieldTemp, float *u, float *v, float *w,
                       int ni, int nj, int nk, int dimx, int dimy, int dimz,
                       float ox, float oy, float oz, float h, float dt), (field, fieldTemp, u, v, w, ni, nj, nk, dimx, dimy, dimz, ox, oy, oz, h, dt))
{
    BQ_ENTER("gpu_clamp_extrema", field, fieldTemp, u, v, w)
    BQ_REQUIRE(field != fieldTemp && ((dimx | dimy | dimz) == 0 || (dimx + dimy + dimz) == 1) &&
               ni - dimx >= 1 && nj - dimy >= 1 && nk - dimz >= 1, "gpu_clamp_extrema");
    Spacing sp = make_spacing(h);
    const Grid g = mk_grid(ni - dimx, nj - dimy, nk - dimz);          // the slab context (single GPU: koff 0, nkg = cell planes)
    BQ_DISPATCH1(clamp_extrema_kernel, sp.pow2, grid_for(ni, nj, nk), field, fieldTemp, u, v, w, sp, ni, nj, nk,
                 dimx, dimy, dimz, ox, oy, oz, dt, g.koff, g.nkg);
}

BQ_ENTRY(gpu_clamp_extrema_box, (const float *before, float *after, int ni, int nj, int nk), (before, after, ni, nj, nk))
{
    BQ_ENTER("gpu_clamp_extrema_box", before, after)
    clamp_box(before, after, ni, nj, nk, 0);
}

BQ_ENTRY(gpu_clamp_extrema_box_w, (const float *before, float *after, int ni, int nj, int nk), (before, after, ni, nj, nk))
{
    BQ_ENTER("gpu_clamp_extrema_box_w", before, after)
    clamp_box(before, after, ni, nj, nk, 1);
}

BQ_ENTRY(gpu_compensate_error_velocity, (float *u, float *v, float *w, float *du, float *dv, float *dw,
                                   float *u_src, float *v_src, float *w_src,
                                   float *forward_x, float *forward_y, float *forward_z,
                                   float h, int ni, int nj, int nk, bool is_point), (u, v, w, du, dv, dw, u_src, v_src, w_src, forward_x, forward_y, forward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_compensate_error_velocity", u, v, w, du, dv, dw, u_src, v_src, w_src, forward_x, forward_y, forward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    compensate_comp(u, du, u_src, forward_x, forward_y, forward_z, sp, g, 1, 0, 0, is_point);
    compensate_comp(v, dv, v_src, forward_x, forward_y, forward_z, sp, g, 0, 1, 0, is_point);
    compensate_comp(w, dw, w_src, forward_x, forward_y, forward_z, sp, g, 0, 0, 1, is_point);
}

BQ_ENTRY(gpu_accumulate_wall_fixup, (const float *src, int src_koff, int src_nk, const float *before, float *dst,
                               const float *mx, const float *my, const float *mz,
                               float h, int ni, int nj, int nk, int axis, float coeff,
                               const int *xlist, int nxl, const int *ylist, int nyl, const int *zlist, int nzl),
         (src, src_koff, src_nk, before, dst, mx, my, mz, h, ni, nj, nk, axis, coeff, xlist, nxl, ylist, nyl, zlist, nzl))
{
    BQ_ENTER("gpu_accumulate_wall_fixup", src, before, dst, mx, my, mz)
    BQ_REQUIRE(axis >= -1 && axis <= 2 && src_nk >= 1 && nxl >= 0 && nyl >= 0 && nzl >= 0 && nxl <= 8 && nyl <= 8 && nzl <= 8 &&
               (nxl == 0 || xlist) && (nyl == 0 || ylist) && (nzl == 0 || zlist), "gpu_accumulate_wall_fixup");
    const int dx = axis == 0, dy = axis == 1, dz = axis == 2;
    if (4.0 * (double)(ni + dx) * (double)(nj + dy) * (double)src_nk >= 2147483648.0) {
        latch(FL_ERR_BAD_ARGUMENT, "gpu_accumulate_wall_fixup", "wall-sheet copy larger than 2 GiB"); return;
    }
    int planes;
    Spacing sp = make_spacing(h); Grid g = mk_grid_win(ni, nj, nk, dz, &planes);
    if (planes <= 0) return;
    const int kw1 = g.kw0 + planes;
    const int nbi = ni + dx, nbj = nj + dy;
    hipStream_t st = rt().compute;
    auto launch = [&](int face_axis, int face_index, int na, int nb) {
        const dim3 grid((na + 63) / 64, (nb + 3) / 4, 1);
        const bool structured = sp.pow2 && rt().opt_structured_maps;
        const int sd = stag_axis(dx, dy, dz);
#define BQ_WF(P2V, SDV) wall_fixup_kernel<P2V, SDV><<<grid, kBlock, 0, st>>>(src, src_nk, src_koff, before, dst, mx, my, mz, sp, g, dx, dy, dz, coeff, face_axis, face_index, kw1)
        if (structured) { switch (sd) { case 0: BQ_WF(true, 0); break; case 1: BQ_WF(true, 1); break; case 2: BQ_WF(true, 2); break; default: BQ_WF(true, 3); break; } }
        else if (sp.pow2) BQ_WF(true, -1);
        else BQ_WF(false, -1);
#undef BQ_WF
        BQ_LAUNCH_CHECK("wall_fixup_kernel");
    };
    for (int a = 0; a < nxl; a++) launch(0, xlist[a], nbj, planes);
    for (int a = 0; a < nyl; a++) launch(1, ylist[a], nbi, planes);
    for (int a = 0; a < nzl; a++) {
        const int kl = zlist[a] - g.koff;
        if (kl >= g.kw0 && kl < kw1) launch(2, zlist[a], nbi, nbj);
    }
}

#ifndef BQ_FAST_LERP
// ---- the same check fused into the map updates: gpu_solve_backwardDMC / gpu_solve_forward flag what they store -------
// fl_map_guard_reset(which): arm the guard and clear word `which` (0: backward, 1: forward; -1: switch the guard off);
// fl_map_guard_read(ok): ok[0], ok[1] = 1 when no value stored since the word's reset failed the test -- blocking, one
// 8-byte read-back for both; z-slab ranks agree on the answer.  A map update that also copies or keeps values it did not
// compute (border nodes of the forward map, ghost planes received from a neighbour) is covered as long as those values
// were themselves produced under the guard or by gpu_init_maps.
void fl_map_guard_reset(int which)
{
    if (!ensure_ready("fl_map_guard_reset")) return;
    Runtime &r = rt();
    if (which < 0) { r.map_guard_on = false; return; }
    if (which > 1) { latch(FL_ERR_BAD_ARGUMENT, "fl_map_guard_reset", "which must be 0, 1 or negative"); return; }
    if (!r.map_guard && !BQ_HIP(hipMalloc((void **)&r.map_guard, 16))) { r.map_guard = nullptr; return; }
    if (!BQ_HIP(hipMemsetAsync(r.map_guard + which, 0, 4, r.compute))) return;
    r.map_guard_on = true;
}

void fl_map_guard_read(int ok[2])
{
    if (!ok) return;
    ok[0] = ok[1] = 0;
    Runtime &r = rt();
    if (!r.ready || !r.map_guard) return;
    int *host = (int *)pinned(64);
    if (!host) return;
    hipStream_t st = r.compute;
    if (!BQ_HIP(hipMemcpyAsync(host, r.map_guard, 8, hipMemcpyDeviceToHost, st)) || !BQ_HIP(hipStreamSynchronize(st))) return;
    int bad[2] = { host[0], host[1] };
    if (comm_ranks() > 1) {                         // every rank reads its neighbours' values in its ghost planes
        float *dev = (float *)(r.map_guard + 2), *hf = (float *)host;
        hf[0] = bad[0] ? 1.f : 0.f; hf[1] = bad[1] ? 1.f : 0.f;
        if (!BQ_HIP(hipMemcpyAsync(dev, hf, 8, hipMemcpyHostToDevice, st)) || !comm_allreduce(dev, 2, false, true, st)) return;
        if (!BQ_HIP(hipMemcpyAsync(hf, dev, 8, hipMemcpyDeviceToHost, st)) || !BQ_HIP(hipStreamSynchronize(st))) return;
        bad[0] = hf[0] != 0.f; bad[1] = hf[1] != 0.f;
    }
    ok[0] = !bad[0]; ok[1] = !bad[1];
}

// 1 when every value of the three map arrays is 0 or lies in [h/256, 1024 h] (tile_value_ok): the precondition of
// FL_OPT_MAP_QUARTER_FP32.  Blocking (one 4-byte read-back); z-slab ranks agree on the answer (all-reduced).
int gpu_maps_quarter_safe(const float *x, const float *y, const float *z, float h, int ni, int nj, int nk)
{
    if (!ensure_ready("gpu_maps_quarter_safe") || !dims_ok(ni, nj, nk, "gpu_maps_quarter_safe")) return 0;
    if (!x || !y || !z || !(h > 0.f)) { latch(FL_ERR_BAD_ARGUMENT, "gpu_maps_quarter_safe", "null pointer or bad spacing"); return 0; }
    int *dflag = (int *)scratch(64);
    int *hflag = (int *)pinned(64);
    if (!dflag || !hflag) return 0;
    hipStream_t st = rt().compute;
    if (!BQ_HIP(hipMemsetAsync(dflag, 0, 4, st))) return 0;
    const size_t n = (size_t)ni * nj * nk;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    maps_quarter_safe_kernel<<<blocks, 256, 0, st>>>(x, y, z, n, h * 0.00390625f, h * 1024.f, dflag);
    if (!BQ_LAUNCH_CHECK("maps_quarter_safe_kernel")) return 0;
    float *fflag = (float *)(dflag + 4);
    if (comm_ranks() > 1) {                         // every rank must take the same view of the shared ghost planes
        // (int flag -> float so that the scalar all-reduce of the slab path can carry it)
        int hbad = 0;
        if (!BQ_HIP(hipMemcpyAsync(hflag, dflag, 4, hipMemcpyDeviceToHost, st)) || !BQ_HIP(hipStreamSynchronize(st))) return 0;
        hbad = *hflag;
        float fb = hbad ? 1.f : 0.f;
        if (!BQ_HIP(hipMemcpyAsync(fflag, &fb, 4, hipMemcpyHostToDevice, st))) return 0;
        if (!comm_allreduce(fflag, 1, false, true, st)) return 0;
        if (!BQ_HIP(hipMemcpyAsync(hflag, fflag, 4, hipMemcpyDeviceToHost, st)) || !BQ_HIP(hipStreamSynchronize(st))) return 0;
        return *(float *)hflag == 0.f ? 1 : 0;
    }
    if (!BQ_HIP(hipMemcpyAsync(hflag, dflag, 4, hipMemcpyDeviceToHost, st)) || !BQ_HIP(hipStreamSynchronize(st))) return 0;
    return *hflag == 0 ? 1 : 0;
}
#endif

BQ_ENTRY(gpu_compensate_error_field, (float *u, float *du, float *u_src,
                                float *forward_x, float *forward_y, float *forward_z,
                                float h, int ni, int nj, int nk, bool is_point), (u, du, u_src, forward_x, forward_y, forward_z, h, ni, nj, nk, is_point))
{
    BQ_ENTER("gpu_compensate_error_field", u, du, u_src, forward_x, forward_y, forward_z)
    Spacing sp = make_spacing(h); Grid g = mk_grid(ni, nj, nk);
    compensate_comp(u, du, u_src, forward_x, forward_y, forward_z, sp, g, 0, 0, 0, is_point);
}

} // extern "C"
