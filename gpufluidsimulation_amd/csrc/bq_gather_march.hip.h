// bq_gather_march.hip.h -- the nine-point gather kernels (advect / cumulate / compensate, GPU_kernel.cu:312-499) as
// z-MARCHING blocks that read the sampled field out of a rolling LDS window (round 4, FL_OPT_FIELD_WINDOW).
//
// Why.  The one-plane kernels of bq_advect.hip fetch the eight corners of each of a node's nine taps straight from
// memory: 36 two-dword gathers per sampled field and node-wave.  Round 3's counters put the texture-addresser path at
// 0.72-0.83 busy and the L1->L2 traffic at 3x the algorithmic bytes; the one-fma (`fast`) variant drops a third of the VALU
// work and gains nothing because that path is its bound.  Round 2 tried an LDS tile of the field per ONE-plane block and
// lost (six planes re-staged by every z-block).  Here a block of 64 x 4 nodes marches along z: per plane step it stages
// ONE new plane of the field window and ONE new plane of the map tile, so the vector-memory work per node-wave falls from
// 36 + 13.5 gathers to about 4 coalesced row loads, and every corner pair comes out of LDS.
//
// What is staged (WD = window displacement in cells, 2):
//   field window  x in [i0 - 4, i0 + 68)      72 floats per row (the block's 64 columns +- WD + 1, start 16-byte aligned)
//                 y in [j0 - WD - 1, j0 + 4 + WD + 1)                 4 + 2 WD + 2 rows
//                 z: a ring of 2 WD + 4 plane slots: the planes k - WD - 1 .. k + WD + 1 a tap of plane k can touch + the
//                    one being staged; slot of plane p = p & 7, and a copy of slot 0 behind slot 7 so that "plane p + 1" is
//                    always the next slot (no second wrap computation per tap)
//   map tile      the 66 x 6 nodes of bq_device.hip.h (stage_tiles) as a ring of 4 plane slots (k - 1, k, k + 1 + staging)
//
// Exactness.  A tap whose cell and its +1 corners lie inside the window AND inside the array (no flat-index wrap, no
// out-of-allocation corner) reads the very floats the direct path would load; every other tap of every lane takes the
// direct path (corners(): buffer loads with the descriptor's range check) -- per tap, under the lane's own predicate.  So
// the window is a cache, never a semantics change: wild maps (zeroed DMC borders, NaN, positions far away) produce the
// same bits as the one-plane kernels, only slower.  The arithmetic (locate, lerps, blend) is the shared code of
// bq_device.hip.h in both builds (exact / BQ_FAST_LERP).
#pragma once
#include "bq_device.hip.h"

namespace bq {
inline namespace BQ_VARIANT {

constexpr int kWD = 2;
constexpr int kWX = 72, kWXoff = 4;
constexpr int kWY = 4 + 2 * kWD + 2, kWYoff = kWD + 1;
constexpr int kWR = 8;                          // ring slots (power of two >= 2 WD + 4)
constexpr int kWPS = kWX * kWY;                 // floats per plane slot
constexpr int kWSlots = kWR + 1;                // + the copy of slot 0
constexpr int kWField = kWSlots * kWPS;         // floats per field window
constexpr int kMR = 4;                          // map ring slots
constexpr int kMPS = 3 * kTileX * kTileY;       // floats per map plane slot (three components)
static_assert(2 * kWD + 4 <= kWR, "ring too short for the window");

enum { kMarchAdvect = 0, kMarchCumulate = 1, kMarchCompensate = 2 };

// src: sampled field; out: advect -> field, cumulate -> dst, compensate -> err; aux: compensate -> init
template <int NF> struct MarchArgs { const float *src[NF]; float *out[NF]; float *aux[NF]; float coeff[NF]; };

struct CellW { int i, j, kl; float fx, fy, fz; };

// locate() of bq_device.hip.h in its P2 + NONNEG form, keeping the three indices (kl: LOCAL plane)
__device__ __forceinline__ CellW locate_ijk(const Field &f, const Spacing &sp, f3 off, f3 pos)
{
    const float qx = __builtin_fmaf(pos.x, sp.inv_h, -off.x * sp.inv_h);
    const float qy = __builtin_fmaf(pos.y, sp.inv_h, -off.y * sp.inv_h);
    const float qz = __builtin_fmaf(pos.z, sp.inv_h, -off.z * sp.inv_h);
    CellW c;
    c.i = floor_to_int(qx); c.j = floor_to_int(qy); c.kl = floor_to_int(qz) - f.koff;
    c.fx = __builtin_amdgcn_fractf(qx); c.fy = __builtin_amdgcn_fractf(qy); c.fz = __builtin_amdgcn_fractf(qz);
    return c;
}

// the seven lerps of gather() on eight corner values that are already in registers
template <bool FMA>
__device__ __forceinline__ float blend_corners(const float (&v)[8], const CellW &c)
{
    double ox, oy, oz;
    if (FMA) { ox = (double)(1.0f - c.fx); oy = (double)(1.0f - c.fy); oz = (double)(1.0f - c.fz); }
    else { ox = 1.0 - (double)c.fx; oy = 1.0 - (double)c.fy; oz = 1.0 - (double)c.fz; }
    const float l00 = lerp_w<FMA>(v[0], v[1], c.fx, ox);
    const float l01 = lerp_w<FMA>(v[2], v[3], c.fx, ox);
    const float l10 = lerp_w<FMA>(v[4], v[5], c.fx, ox);
    const float l11 = lerp_w<FMA>(v[6], v[7], c.fx, ox);
    const float m0 = lerp_w<FMA>(l00, l01, c.fy, oy);
    const float m1 = lerp_w<FMA>(l10, l11, c.fy, oy);
    return lerp_w<FMA>(m0, m1, c.fz, oz);
}

// What a block knows about its window at one plane step (all wave-uniform).
struct WindowView {
    int xlo, xspan, ylo, yspan, zlo, zspan;     // cells [lo, lo + span] may be read from the window (corner + 1 included)
    int addr0;                                  // LDS byte address of window element (x = 0, y = 0) of slot 0, field 0
    int base;                                   // LDS byte address of the window's first element (always valid)
};

// The eight corners of tap cell c for NF co-located fields: from the window where the lane's cell is inside, else from memory.
template <int NF>
__device__ __forceinline__ void window_corners(const Field (&src)[NF], const WindowView &w, const CellW &c, float (&v)[NF][8])
{
    const bool in = (unsigned)(c.i - w.xlo) <= (unsigned)w.xspan && (unsigned)(c.j - w.ylo) <= (unsigned)w.yspan &&
                    (unsigned)(c.kl - w.zlo) <= (unsigned)w.zspan;
    if (in) {
        const int slot = c.kl & (kWR - 1);
        const unsigned a = (unsigned)(w.addr0 + c.i * 4) + (unsigned)__mul24(c.j, kWX * 4) + (unsigned)__mul24(slot, kWPS * 4);
        const float *p = (const float *)(__attribute__((address_space(3))) const float *)(size_t)a;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const float *q = p + f * kWField;
            v[f][0] = q[0];              v[f][1] = q[1];
            v[f][2] = q[kWX];            v[f][3] = q[kWX + 1];
            v[f][4] = q[kWPS];           v[f][5] = q[kWPS + 1];
            v[f][6] = q[kWPS + kWX];     v[f][7] = q[kWPS + kWX + 1];
        }
    } else {
        Cell g;
        const int idx = c.i + __mul24(src[0].nx, c.j) + __mul24(src[0].nx * src[0].ny, c.kl);
        g.base = idx < 0 ? 0x80000000u : (unsigned)idx * 4u;
        g.fx = c.fx; g.fy = c.fy; g.fz = c.fz;
#pragma unroll
        for (int f = 0; f < NF; f++) corners(src[f], g, v[f]);
    }
}

// blend9_gather_w of bq_advect.hip with the corners read through the window, tap by tap: each tap branches on its own
// predicate (the form the kernels ship with: see the batched form below for what was measured against it)
template <int NF, bool GE1>
__device__ __forceinline__ void blend9_window_pertap(const Field (&src)[NF], const WindowView &wv, const Spacing &sp, f3 org, const f3 (&mp)[9],
                                              const float (&w)[NF], float (&sum)[NF], float (&value)[NF])
{
#pragma unroll
    for (int ii = 0; ii < 8; ii++) {
        const CellW c = locate_ijk(src[0], sp, org, mp[ii]);
        float v[NF][8];
        window_corners<NF>(src, wv, c, v);
#pragma unroll
        for (int f = 0; f < NF; f++) sum[f] += w[f] * blend_corners<GE1>(v[f], c);
    }
    const CellW c = locate_ijk(src[0], sp, org, mp[8]);
    float v[NF][8];
    window_corners<NF>(src, wv, c, v);
#pragma unroll
    for (int f = 0; f < NF; f++) value[f] = blend_corners<GE1>(v[f], c);
}

// One tap, located: the LDS byte address of its cell's corner 000 in the window (field 0) when the lane's cell is inside,
// the window's first byte otherwise (any valid address: the values read there are replaced, see blend9_window).
struct TapW { unsigned addr; float fx, fy, fz; };

__device__ __forceinline__ bool tap_locate(const Field &f, const WindowView &w, const Spacing &sp, f3 org, f3 pos, TapW &t)
{
    const CellW c = locate_ijk(f, sp, org, pos);
    const bool in = (unsigned)(c.i - w.xlo) <= (unsigned)w.xspan && (unsigned)(c.j - w.ylo) <= (unsigned)w.yspan &&
                    (unsigned)(c.kl - w.zlo) <= (unsigned)w.zspan;
    const int slot = c.kl & (kWR - 1);
    const unsigned a = (unsigned)(w.addr0 + c.i * 4) + (unsigned)__mul24(c.j, kWX * 4) + (unsigned)__mul24(slot, kWPS * 4);
    t.addr = in ? a : (unsigned)w.base;
    t.fx = c.fx; t.fy = c.fy; t.fz = c.fz;
    return in;
}

template <int NF>
__device__ __forceinline__ void tap_read(const TapW &t, float (&v)[NF][8])
{
    const float *p = (const float *)(__attribute__((address_space(3))) const float *)(size_t)t.addr;
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const float *q = p + f * kWField;
        v[f][0] = q[0];              v[f][1] = q[1];
        v[f][2] = q[kWX];            v[f][3] = q[kWX + 1];
        v[f][4] = q[kWPS];           v[f][5] = q[kWPS + 1];
        v[f][6] = q[kWPS + kWX];     v[f][7] = q[kWPS + kWX + 1];
    }
}

template <bool FMA>
__device__ __forceinline__ float blend_tap(const float (&v)[8], const TapW &t)
{
    CellW c; c.i = c.j = c.kl = 0; c.fx = t.fx; c.fy = t.fy; c.fz = t.fz;
    return blend_corners<FMA>(v, c);
}

// blend9_gather_w of bq_advect.hip with the corners read through the window, in batches of TB taps:
//   1. locate the batch's taps (cell, weights, in-window predicate, LDS address)
//   2. read all their corners from LDS -- unconditionally, no branch: TB x NF x 4 two-dword reads in flight together
//   3. only if some lane of the wave has a tap outside the window (wave-uniform test): those lanes fetch those taps' corners
//      from memory, exactly as the one-plane kernels do (corners(): descriptor range check, flat-index wrap)
//   4. the lerps
// The common case is straight-line code; the direct path costs nothing but one scalar branch per batch when nobody needs it.
// Measured at 256^3 (profiles/r04_f_*): no faster than the tap-by-tap form where every tap is served by the window (the
// kernels are bound by VALU issue, not by the latency of the reads: 330 against 332 us), and 5-15 % SLOWER in the accumulations
// through the backward map, whose zeroed border sends some lane of half of all waves to the direct path: those
// waves then pay the reads AND a second locate.  Kept selectable (TB > 0) for maps without that border.
template <int NF, bool GE1, int TB>
__device__ __forceinline__ void blend9_window(const Field (&src)[NF], const WindowView &wv, const Spacing &sp, f3 org, const f3 (&mp)[9],
                                              const float (&w)[NF], float (&sum)[NF], float (&value)[NF])
{
#pragma unroll
    for (int b0 = 0; b0 < 9; b0 += TB) {
        TapW tap[TB];
        bool in[TB];
        bool all_in = true;
#pragma unroll
        for (int t = 0; t < TB; t++)
            if (b0 + t < 9) { in[t] = tap_locate(src[0], wv, sp, org, mp[b0 + t], tap[t]); all_in = all_in && in[t]; }
        float v[TB][NF][8];
#pragma unroll
        for (int t = 0; t < TB; t++)
            if (b0 + t < 9) tap_read<NF>(tap[t], v[t]);
        if (__any(!all_in)) {
#pragma unroll
            for (int t = 0; t < TB; t++)
                if (b0 + t < 9) {
                    if (!in[t]) {
                        const CellW c = locate_ijk(src[0], sp, org, mp[b0 + t]);
                        Cell g;
                        const int idx = c.i + __mul24(src[0].nx, c.j) + __mul24(src[0].nx * src[0].ny, c.kl);
                        g.base = idx < 0 ? 0x80000000u : (unsigned)idx * 4u;
                        g.fx = c.fx; g.fy = c.fy; g.fz = c.fz;
#pragma unroll
                        for (int f = 0; f < NF; f++) corners(src[f], g, v[t][f]);
                    }
                }
        }
#pragma unroll
        for (int t = 0; t < TB; t++)
            if (b0 + t < 9) {
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    const float s = blend_tap<GE1>(v[t][f], tap[t]);
                    if (b0 + t < 8) sum[f] += w[f] * s; else value[f] = s;
                }
            }
        __builtin_amdgcn_sched_barrier(0);          // (keeps the next batch's look-ups from being hoisted over this one: registers)
    }
}

// ---- the structured map look-up (bq_device.hip.h: map9_nodes) split along z ---------------------------------------------------
// map9_nodes interpolates a node's 3 x 3 x 3 block of map nodes along x, then y, then z.  Its first two stages work on ONE
// plane of nodes at a time and do not depend on the node's own plane: what plane p contributes to node k is what it
// contributes to k - 1 and k + 1.  A marching thread therefore evaluates them once per plane it enters -- the five (x tap, y tap)
// combinations the nine taps use: four corners and the centre -- keeps the last NZ planes' results in registers and runs only
// the z stage per node: 22 lerps per component and node instead of 48, the very same operations on the very same operands.
// combination index: tx * 2 + ty for the corner taps (tx, ty in {0 '+', 1 '-'}), 4 for the centre
template <int SX, int SY, bool Q4>
__device__ __forceinline__ void map_plane_stage(const float *t, int o, float (&ly)[5])
{
    constexpr int NX = SX ? 2 : 3, NY = SY ? 2 : 3;
    float N[NY][NX];
#pragma unroll
    for (int y = 0; y < NY; y++)
#pragma unroll
        for (int x = 0; x < NX; x++) N[y][x] = t[o + y * kTileX + x];
    float LX[3][NY];
#pragma unroll
    for (int tx = 0; tx < 3; tx++) {
        const int r = tap_rel(SX, tx);
        const float c = tap_frac(SX, tx);
#pragma unroll
        for (int y = 0; y < NY; y++) LX[tx][y] = lerp_q<Q4>(N[y][r], N[y][r + 1], c);
    }
#pragma unroll
    for (int tx = 0; tx < 2; tx++)
#pragma unroll
        for (int ty = 0; ty < 2; ty++) {
            const int r = tap_rel(SY, ty);
            ly[tx * 2 + ty] = lerp_q<Q4>(LX[tx][r], LX[tx][r + 1], tap_frac(SY, ty));
        }
    {
        const int r = tap_rel(SY, 2);
        ly[4] = lerp_q<Q4>(LX[2][r], LX[2][r + 1], tap_frac(SY, 2));
    }
}

// the z stage: P[z] = the plane stage of map plane kl - 1 + z (NZ = 3, or 2 when the component is staggered along z)
template <int SZ, bool Q4, int NZ>
__device__ __forceinline__ void map_z_stage(const float (&P)[NZ][5], float (&out)[9])
{
#pragma unroll
    for (int ii = 0; ii < 8; ii++) {
        const int tx = (ii >> 2) & 1, ty = (ii >> 1) & 1, tz = ii & 1;
        const int r = tap_rel(SZ, tz);
        out[ii] = lerp_q<Q4>(P[r][tx * 2 + ty], P[r + 1][tx * 2 + ty], tap_frac(SZ, tz));
    }
    const int r = tap_rel(SZ, 2);
    out[8] = lerp_q<Q4>(P[r][4], P[r + 1][4], tap_frac(SZ, 2));
}

// One thread's share of a plane of the field window / of the map tile, fixed for the whole march.
struct StageSlots {
    int foff[3];        // field window: element offset inside a plane (row-major, 72 per row), -1 = nothing
    int fsrc[3];        // ... its offset inside a source plane (x + nbi * y), -1 = outside the array
    int moff[2];        // map tile: element offset inside a component's plane (66 per row), -1 = nothing
    int msrc[2];        // ... flat offset x + nx * y inside a map plane (may be negative or beyond a row: same flat-index rule as stage_tiles)
};

template <int KIND, int SD, int NF, bool Q4>
__global__ __launch_bounds__(256, NF == 1 ? 3 : 2) void gather_march_kernel(MarchArgs<NF> a, const float *mx, const float *my, const float *mz,
                                                              Spacing sp, Grid g, int dx, int dy, int dz, int fused, int kchunk, int kw1)
{
    __shared__ float fwin[NF * kWField];
    __shared__ float mring[kMR * kMPS];
    const int nbi = g.ni + dx, nbj = g.nj + dy, nbk = g.nk + dz;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 4;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int i = i0 + threadIdx.x, j = j0 + threadIdx.y;
    const int kb = g.kw0 + blockIdx.z * kchunk, ke = min(kb + kchunk, kw1);
    // index window of the operator (GPU_kernel.cu:353, :414, :476)
    const int m = KIND == kMarchAdvect ? 2 : 1;
    const int ilo = m + dx, ihi = nbi - m - 1, jlo = m + dy, jhi = nbj - m - 1, klo = m + dz, khi = g.nkg + dz - m - 1;
    const bool ij_in = ilo < i && i < ihi && jlo < j && j < jhi;
    const bool block_ij_out = i0 + 63 <= ilo || i0 >= ihi || j0 + 3 <= jlo || j0 >= jhi;
    const bool node = i < nbi && j < nbj;
    const size_t plane = (size_t)nbi * nbj;
    const size_t id0 = (size_t)i + (size_t)nbi * j;

    // ---- housekeeping the one-plane kernels do on the side (FL_OPT_FUSED_HOUSEKEEPING), per node and plane --------------------
    auto housekeeping = [&](int k, bool active, float (&init_own)[NF]) {
        if (!node) return;
        const size_t id = id0 + plane * k;
        if (KIND == kMarchAdvect) {
            if ((fused & 1) && !active) {
#pragma unroll
                for (int f = 0; f < NF; f++) a.out[f][id] = 0.f;
            }
        } else if (KIND == kMarchCompensate) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                init_own[f] = a.aux[f][id];
                if (fused & 2) a.aux[f][id] = a.src[f][id];
                if ((fused & 1) && !active) a.out[f][id] = 0.f;
            }
        }
    };
    if (block_ij_out) {                                     // no node of this block is ever inside the window: no staging
        for (int k = kb; k < ke; k++) { float io[NF]; housekeeping(k, false, io); }
        return;
    }

    // ---- this thread's staging slots -----------------------------------------------------------------------------------------
    StageSlots ss;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int e = tid + 256 * q;
        ss.foff[q] = -1; ss.fsrc[q] = -1;
        if (e < kWPS) {
            const int r = e / kWX, c = e - r * kWX;
            const int x = i0 - kWXoff + c, y = j0 - kWYoff + r;
            ss.foff[q] = e;
            if (x >= 0 && x < nbi && y >= 0 && y < nbj) ss.fsrc[q] = x + nbi * y;
        }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int e = tid + 256 * q;
        ss.moff[q] = -1; ss.msrc[q] = 0;
        if (e < kTileX * kTileY) {
            const int r = e / kTileX, c = e - r * kTileX;
            ss.moff[q] = e;
            ss.msrc[q] = (i0 - 1 + c) + g.ni * (j0 - 1 + r);
        }
    }
    Field src[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) src[f] = make_field(a.src[f], nbi, nbj, nbk, g.koff);
    const Field mf[3] = { make_field(mx, g.ni, g.nj, g.nk, g.koff), make_field(my, g.ni, g.nj, g.nk, g.koff), make_field(mz, g.ni, g.nj, g.nk, g.koff) };
    const int msk = g.ni * g.nj;

    // loads of one plane into registers / registers into the ring (two halves so that a step can overlap them with its compute)
    auto load_field_plane = [&](int p, float (&r)[NF][3]) {
        const bool pin = p >= 0 && p < nbk;
#pragma unroll
        for (int f = 0; f < NF; f++)
#pragma unroll
            for (int q = 0; q < 3; q++)
                r[f][q] = (pin && ss.fsrc[q] >= 0) ? ldf(src[f], (unsigned)(ss.fsrc[q] + (int)plane * p) * 4u) : 0.f;
    };
    auto store_field_plane = [&](int p, const float (&r)[NF][3]) {
        const int slot = p & (kWR - 1);
#pragma unroll
        for (int f = 0; f < NF; f++)
#pragma unroll
            for (int q = 0; q < 3; q++)
                if (ss.foff[q] >= 0) {
                    fwin[f * kWField + slot * kWPS + ss.foff[q]] = r[f][q];
                    if (slot == 0) fwin[f * kWField + kWR * kWPS + ss.foff[q]] = r[f][q];
                }
    };
    auto load_map_plane = [&](int p, float (&r)[3][2]) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                r[c][q] = ss.moff[q] >= 0 ? ldf(mf[c], (unsigned)(ss.msrc[q] + msk * p) * 4u) : 0.f;
    };
    auto store_map_plane = [&](int p, const float (&r)[3][2]) {
        const int slot = p & (kMR - 1);
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                if (ss.moff[q] >= 0) mring[slot * kMPS + c * (kTileX * kTileY) + ss.moff[q]] = r[c][q];
    };

    // ---- prologue: the planes step kb needs ------------------------------------------------------------------------------------
    for (int p = kb - kWD - 1; p <= kb + kWD + 1; p++) { float r[NF][3]; load_field_plane(p, r); store_field_plane(p, r); }
    for (int p = kb - 1; p <= kb + 1; p++) { float r[3][2]; load_map_plane(p, r); store_map_plane(p, r); }
    __syncthreads();

    const float h = sp.h;
    const Nine n = nine_setup(h, dx, dy, dz);
    const f3 lo = KIND == kMarchAdvect ? mk3(h, h, h) : mk3(0.f, 0.f, 0.f);
    const f3 hi = KIND == kMarchAdvect ? mk3(h * (float)g.ni - h, h * (float)g.nj - h, h * (float)g.nkg - h)
                                       : mk3(h * (float)g.ni, h * (float)g.nj, h * (float)g.nkg);
    // plane stages of the map look-up carried from node to node (map_plane_stage): P[c][z] belongs to map plane k - 1 + z
    constexpr int SXc = SD == 1, SYc = SD == 2, SZc = SD == 3;
    constexpr int NZ = SZc ? 2 : 3;
    // ZR: carry them (45 more live registers).  Measured at 256^3 (profiles/r04_e_*): the two-field kernels, which LDS holds at
    // two waves per SIMD anyway, run 4-10 % faster with it; the single-field kernels need their third wave more than the 12 %
    // fewer instructions (332 -> 395 us at two waves) and re-evaluate all NZ planes per node instead, as map9_nodes does.
    constexpr bool ZR = NF == 2 && !(KIND == kMarchCompensate && SD == 0);
    float P[3][NZ][5];
    const float *mt = mring + threadIdx.y * kTileX + threadIdx.x;
    if (ZR && ij_in) {
#pragma unroll
        for (int z = 0; z < NZ - 1; z++) {
            const int o = ((kb - 1 + z) & (kMR - 1)) * kMPS;
#pragma unroll
            for (int c = 0; c < 3; c++) map_plane_stage<SXc, SYc, Q4>(mt + c * (kTileX * kTileY), o, P[c][z]);
        }
    }
    WindowView wv;
    wv.xlo = max(i0 - kWXoff, 0);           wv.xspan = min(i0 - kWXoff + kWX - 1, nbi - 1) - 1 - wv.xlo;
    wv.ylo = max(j0 - kWYoff, 0);           wv.yspan = min(j0 - kWYoff + kWY - 1, nbj - 1) - 1 - wv.ylo;
    wv.base = (int)(unsigned)(size_t)fwin;
    wv.addr0 = wv.base - ((i0 - kWXoff) * 4 + (j0 - kWYoff) * (kWX * 4));

    for (int k = kb; k < ke; k++) {
        const int kg = k + g.koff;
        const bool plane_in = klo < kg && kg < khi;
        const bool active = plane_in && ij_in;
        // next planes: issued now, landed in the ring after this plane's arithmetic
        float rf[NF][3], rm[3][2];
        const bool more = k + 1 < ke;
        if (more) { load_field_plane(k + kWD + 2, rf); load_map_plane(k + 2, rm); }
        float init_own[NF];
        housekeeping(k, active, init_own);
        if (ZR && ij_in) {                                  // the newest map plane this node touches: k + 1 (k when staggered along z)
            const int o = ((k + NZ - 2) & (kMR - 1)) * kMPS;
#pragma unroll
            for (int c = 0; c < 3; c++) map_plane_stage<SXc, SYc, Q4>(mt + c * (kTileX * kTileY), o, P[c][NZ - 1]);
        }
        if (active) {
            if (!ZR) {
#pragma unroll
                for (int z = 0; z < NZ; z++) {
                    const int o = ((k - 1 + z) & (kMR - 1)) * kMPS;
#pragma unroll
                    for (int c = 0; c < 3; c++) map_plane_stage<SXc, SYc, Q4>(mt + c * (kTileX * kTileY), o, P[c][z]);
                }
            }
            wv.zlo = max(k - kWD - 1, 0);   wv.zspan = min(k + kWD + 1, nbk - 1) - 1 - wv.zlo;
            f3 mp[9];
            {
                float x9[9], y9[9], z9[9];
                map_z_stage<SZc, Q4, NZ>(P[0], x9);
                map_z_stage<SZc, Q4, NZ>(P[1], y9);
                map_z_stage<SZc, Q4, NZ>(P[2], z9);
#pragma unroll
                for (int t = 0; t < 9; t++) mp[t] = mk3(x9[t], y9[t], z9[t]);
            }
            // exact build, single field: the one-fma lerps where every position of the wave is >= h (bq_advect.hip: wave_all_ge);
            // advect clamps to >= h.  The fast build's lerps do not depend on it.
#ifdef BQ_FAST_LERP
            constexpr bool kTestGe = false;
#else
            constexpr bool kTestGe = KIND != kMarchAdvect && NF == 1;
#endif
            bool ge1 = KIND == kMarchAdvect;
            if (kTestGe) {
                float mn = fminf(mp[8].x, fminf(mp[8].y, mp[8].z));
#pragma unroll
                for (int t = 0; t < 8; t++) mn = fminf(mn, fminf(mp[t].x, fminf(mp[t].y, mp[t].z)));
                ge1 = __all(mn >= h);
            }
#pragma unroll
            for (int t = 0; t < 9; t++) mp[t] = clamp3_ordered(mp[t], lo, hi);
            float sum[NF], value[NF], w[NF];
#pragma unroll
            for (int f = 0; f < NF; f++) { sum[f] = 0.f; w[f] = KIND == kMarchCumulate ? 0.125f * a.coeff[f] : 0.125f; }
            // 0: tap by tap; 3 / 5: batches (see blend9_window).  The two-field advection (unstaggered, forward-free map
            // border: every tap in the window) is the one launch the batched form serves better: 487 against 531 / 619 us
            constexpr int TB = (KIND == kMarchAdvect && NF == 2) ? 3 : 0;
            if constexpr (TB == 0) {
                if (KIND == kMarchAdvect) blend9_window_pertap<NF, true>(src, wv, sp, n.org, mp, w, sum, value);
                else if (kTestGe && ge1)  blend9_window_pertap<NF, true>(src, wv, sp, n.org, mp, w, sum, value);
                else                      blend9_window_pertap<NF, false>(src, wv, sp, n.org, mp, w, sum, value);
            } else {
                constexpr int TBB = TB > 0 ? TB : 3;
                if (KIND == kMarchAdvect) blend9_window<NF, true, TBB>(src, wv, sp, n.org, mp, w, sum, value);
                else if (kTestGe && ge1)  blend9_window<NF, true, TBB>(src, wv, sp, n.org, mp, w, sum, value);
                else                      blend9_window<NF, false, TBB>(src, wv, sp, n.org, mp, w, sum, value);
            }
            const size_t id = id0 + plane * k;
#pragma unroll
            for (int f = 0; f < NF; f++) {
                if (KIND == kMarchAdvect) a.out[f][id] = 0.5f * sum[f] + 0.5f * value[f];
                else if (KIND == kMarchCumulate) {
                    const float v = a.coeff[f] * value[f];
                    a.out[f][id] += (float)(0.5 * (double)sum[f] + 0.5 * (double)v);
                } else a.out[f][id] = (float)(0.5 * (double)sum[f] + 0.5 * (double)value[f]) - init_own[f];
            }
        }
        if (ZR && ij_in) {                                  // the planes move down by one for the next node
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int z = 0; z < NZ - 1; z++)
#pragma unroll
                    for (int q = 0; q < 5; q++) P[c][z][q] = P[c][z + 1][q];
        }
        if (more) { store_field_plane(k + kWD + 2, rf); store_map_plane(k + 2, rm); }
        __syncthreads();
    }
}

} // inline namespace BQ_VARIANT
} // namespace bq
