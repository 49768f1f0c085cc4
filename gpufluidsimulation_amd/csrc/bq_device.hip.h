// bq_device.hip.h -- device-side building blocks of the gather ("semi-Lagrangian") family.
//
// Arithmetic contract (DESIGN.md "Numerics"): every expression keeps the operand types and
// association of the reference source (src/bimocq3D/GPU_kernel.cu:9-125) so that results are
// bit-identical to oracle/bimocq_oracle.c.  The file is compiled with -ffp-contract=off; the
// double-typed pieces of the reference (lerp, RK3 stage points) stay double here.
//
// gfx950 has no image/sampler hardware (__HIP_NO_IMAGE_SUPPORT), so "trilinear sampling" is
// eight raw buffer loads + seven lerps.  Loads go through a buffer resource descriptor whose
// num_records is the field's byte size: the hardware range check returns 0 for any corner that
// falls outside the allocation (the oracle's `ld`), per dword, with no compare in the shader.
//
// Two arithmetic variants of everything in this file exist side by side in the library (inline namespaces
// bq::exact / bq::fast, selected per translation unit by BQ_FAST_LERP): `exact` is the contract above; `fast`
// evaluates a lerp as ONE fp32 fma, fmaf(c, b - a, a), and changes nothing else (oracle: orc_set_fast_lerp).
// The single-field gather kernels are bound by VALU instruction issue (~3.8 cycles per wave64 instruction whatever its
// type), so `fast` -- the variant SURVEY 8(d) calls for next to the exact one -- wins what its shorter lerps save in
// instruction count (~10 % of such a kernel); it stays 3 orders of magnitude inside the 1e-5 RMS tolerance over
// 200 steps (tests/test_gpu_solver.py).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

#ifdef BQ_FAST_LERP
#define BQ_VARIANT fast
#else
#define BQ_VARIANT exact
#endif

namespace bq {
inline namespace BQ_VARIANT {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }

// A dense x-fastest fp32 field as the kernels see it.  On a z-slab rank the buffer holds the
// global planes [koff, koff + nz): sample positions are global, the plane index is made local here.
struct Field {
    __amdgpu_buffer_rsrc_t rsrc;
    int nx, ny;            // row pitch and slab pitch factor (elements)
    int koff;              // global k of local plane 0 (0 on a single GPU)
};

__device__ __forceinline__ Field make_field(const float *p, int nx, int ny, int nz, int koff = 0)
{
    Field f;
    unsigned bytes = (unsigned)nx * (unsigned)ny * (unsigned)nz * 4u;
    f.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000);
    f.nx = nx; f.ny = ny; f.koff = koff;
    return f;
}

__device__ __forceinline__ float ldf(const Field &f, unsigned byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(f.rsrc, byte_off, 0, 0));
}

// Neighbour-lane exchange across the whole wave64 as DPP moves (wave_shr:1 / wave_shl:1, GFX9 encodings
// 0x138 / 0x130): a VALU instruction, where __shfl_up/__shfl_down compile to ds_bpermute_b32 on the LDS pipe.
// lane_up(x): lane l receives lane l-1's x (lane 0 keeps its own); lane_down(x): lane l receives lane l+1's.
__device__ __forceinline__ float lane_up(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_down(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_up(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = (int)b, hi = (int)(b >> 32);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)h2 << 32) | l2));
}
__device__ __forceinline__ double lane_down(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = (int)b, hi = (int)(b >> 32);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)h2 << 32) | l2));
}

// Grid spacing.  When h is a power of two (every BASELINE config: L=1, N=2^k) x/h == x*(1/h)
// bit for bit, which removes three IEEE divisions (~11 VALU each) from every sample.
struct Spacing {
    float h, inv_h;
    int pow2;
};

inline Spacing make_spacing(float h)
{
    int e = 0;
    float m = frexpf(h, &e);
    Spacing sp;
    sp.h = h;
    sp.inv_h = 1.0f / h;
    sp.pow2 = (m == 0.5f) && (e > -100) && (e < 100);
    return sp;
}

// (Round 4 measured, and did not keep, division by the constant h as fma(x, zh, x * zl) with (zh, zl) = 1 / h as a float pair:
// correctly rounded for every mantissa of x for h = 0.002f, 1/24, 0.01f, 1/300 ... -- checked exhaustively -- but only while
// no product leaves the normal range, and the per-wave guard for that (three votes and a second code path per locate) made the
// reference grid's step 8.6 -> 20.5 ms: EXPERIMENTS.md section 9.)
template <bool P2>
__device__ __forceinline__ float div_h(float s, const Spacing &sp)
{
    if (P2) return s * sp.inv_h;
    return s / sp.h;
}

// GPU_kernel.cu:9-12
__device__ __forceinline__ float clampf(float a, float lo, float hi) { return fminf(fmaxf(lo, a), hi); }
__device__ __forceinline__ f3 clamp3(f3 p, f3 lo, f3 hi)
{
    return mk3(clampf(p.x, lo.x, hi.x), clampf(p.y, lo.y, hi.y), clampf(p.z, lo.z, hi.z));
}
// The same clamp as one v_med3_f32 per component, for lo <= hi: the median of (a, lo, hi) is the clamped value, and
// for a NaN the instruction returns min3 = lo, which is what fminf(fmaxf(lo, NaN), hi) gives.
__device__ __forceinline__ f3 clamp3_ordered(f3 p, f3 lo, f3 hi)
{
    return mk3(__builtin_amdgcn_fmed3f(p.x, lo.x, hi.x), __builtin_amdgcn_fmed3f(p.y, lo.y, hi.y), __builtin_amdgcn_fmed3f(p.z, lo.z, hi.z));
}

// GPU_kernel.cu:22-25 with the (1.0 - c) factor hoisted (it is exact to hoist: same value)
// FMA: the caller vouches that c = q - floor(q) with q >= 1, i.e. c is a multiple of 2^-23.  Then omc = 1 - c has at
// most 24 significant bits, omc*a at most 48: the product is exact in double and fma(omc, a, cb) rounds exactly like
// the contract's separate multiply and add -- one f64 instruction fewer per lerp.
template <bool FMA = false>
__device__ __forceinline__ float lerp_w(float a, float b, float c, double omc)
{
#ifdef BQ_FAST_LERP
    (void)omc;
    return __builtin_fmaf(c, b - a, a);
#else
    float cb = c * b;
    if (FMA) return (float)__builtin_fma(omc, (double)a, (double)cb);
    return (float)(omc * (double)a + (double)cb);
#endif
}

// lerp with a compile-time weight c in {0, 1/4, 1/2, 3/4}: omc*a has at most 26 significant bits, so it
// is exact in double and fma(omc, a, cb) rounds exactly like the reference's separate multiply and add
// -- one f64 instruction instead of two.
__device__ __forceinline__ float lerp_const(float a, float b, float c, double omc)
{
#ifdef BQ_FAST_LERP
    (void)omc;
    return __builtin_fmaf(c, b - a, a);
#else
    float cb = c * b;
    return (float)__builtin_fma(omc, (double)a, (double)cb);
#endif
}

// The same compile-time-weight lerp, cheaper where that is provably bit-identical.  On gfx950 a wave64 VALU
// instruction costs one 4-cycle issue slot whatever its type (only packed fp32 is faster), so what counts is the
// instruction count: lerp_const is 5 (mul, 2 cvt, fma_f64, cvt), fmaf(1 - c, a, c*b) is 2.  c*b is the very float
// product of the contract and (1 - c)*a is exact inside the fma, so the fp32 form differs from lerp_const by one
// rounding (to float) instead of two (to double, then to float).  The two can only disagree when the double sum lands
// exactly on a float midpoint that the exact sum misses:
//   c = 0, 1/2, 3/4: (1 - c)*a is itself a float and the sum of two floats cannot do that -- always identical;
//   c = 1/4: 3/4*a can sit on a float midpoint (results differ when 0 < |c*b| <= 2^-53 |3/4 a|) -- stays lerp_const.
// tools/lerp_q_check.c compares both forms on 1.2e9 random and adversarial operand pairs per weight.
// Q4: the caller vouches that the c = 1/4 lerps may take the fp32 form too (a tile whose values passed tile_value_ok):
// with 3/4*a + 1/4*b exactly representable in double the contract's two roundings collapse into RN32 of the exact sum,
// which is what fmaf(3/4, a, b/4) returns (b/4 is exact).  Exactness in double needs |a| and |b| within 2^25 of each
// other or one of them zero: the 26 bits of 3a and the 24 of b then fit 53.
template <bool Q4 = false>
__device__ __forceinline__ float lerp_q(float a, float b, float c)
{
#ifdef BQ_FAST_LERP
    return __builtin_fmaf(c, b - a, a);
#else
    if (c == 0.25f) return Q4 ? __builtin_fmaf(0.75f, a, 0.25f * b) : lerp_const(a, b, c, 0.75);
    if (c == 0.0f) return __builtin_fmaf(0.0f, b, a);     // 1*a + 0*b: a for a finite b, NaN otherwise -- one instruction
    return __builtin_fmaf(1.0f - c, a, c * b);
#endif
}

__device__ __forceinline__ int floor_to_int(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Cell + weights of one sample position (GPU_kernel.cu:45-51)
struct Cell {
    unsigned base;          // byte offset of corner 000; 2 GiB (out of range) when its flat index is negative
    float fx, fy, fz;
};

// NONNEG: the caller vouches that pos - off >= 0 on every axis (positions clamped into the grid, origins <= 0).
// Then q - floor(q) is exact and below 1, which is what v_fract_f32 returns: one instruction instead of two.
template <bool P2, bool NONNEG = false>
__device__ __forceinline__ Cell locate(const Field &f, const Spacing &sp, f3 off, f3 pos)
{
    float qx, qy, qz;
    if (P2 && NONNEG) {
        // (pos - off) * (1/h) with 1/h a power of two: scaling commutes with the rounding of the difference, so this
        // is fl(pos/h - off/h) -- one fma (the product is exact); -off/h is 0 or 1/2
        qx = __builtin_fmaf(pos.x, sp.inv_h, -off.x * sp.inv_h);
        qy = __builtin_fmaf(pos.y, sp.inv_h, -off.y * sp.inv_h);
        qz = __builtin_fmaf(pos.z, sp.inv_h, -off.z * sp.inv_h);
    } else {
        qx = div_h<P2>(pos.x - off.x, sp);
        qy = div_h<P2>(pos.y - off.y, sp);
        qz = div_h<P2>(pos.z - off.z, sp);
    }
    // int(floorf(q)) (GPU_kernel.cu:47-49) in one instruction: v_cvt_flr_i32_f32 converts with round-toward-minus-
    // infinity, which is floor followed by the (exact) conversion for every |q| < 2^31
    const int i = floor_to_int(qx), j = floor_to_int(qy), k = floor_to_int(qz);
    Cell c;
    if (NONNEG) { c.fx = __builtin_amdgcn_fractf(qx); c.fy = __builtin_amdgcn_fractf(qy); c.fz = __builtin_amdgcn_fractf(qz); }
    else { c.fx = qx - (float)i; c.fy = qy - (float)j; c.fz = qz - (float)k; }
    // row and plane strides times signed 24-bit indices (fields stay below 2^29 elements, dims_ok): two full-rate
    // v_mad_i32_i24 instead of the 64-bit multiply-adds a plain `int` product turns into
    int idx = i + __mul24(f.nx, j) + __mul24(f.nx * f.ny, k - f.koff);
    c.base = idx < 0 ? 0x80000000u : (unsigned)idx * 4u;    // negative base: every corner out of range (see corners())
    return c;
}

// The 8 corner values of a cell.  Contract (oracle: sample()): a cell whose base corner has a NEGATIVE flat
// index reads all zeros; otherwise a corner whose flat index lies in [0, count) is read and anything else is
// 0 (a corner one past a row/plane wraps into the next one, as the reference's flat indexing does).
// locate() parks a negative base at 2 GiB, beyond every field (dims_ok caps them at 2 GiB), so the range
// check of the buffer descriptor delivers exactly that -- no compare or branch per corner.  (Without the
// parking a base of index -1 would wrap through 2^32: the address unit adds the instruction's immediate
// +4 beyond 32 bits and reports out-of-range where the wrapped offset 0 is valid.)
__device__ __forceinline__ void corners(const Field &f, const Cell &c, float (&v)[8])
{
    const unsigned sj = (unsigned)f.nx * 4u, sk = (unsigned)f.nx * (unsigned)f.ny * 4u;
    v[0] = ldf(f, c.base);           v[1] = ldf(f, c.base + 4u);
    v[2] = ldf(f, c.base + sj);      v[3] = ldf(f, c.base + sj + 4u);
    v[4] = ldf(f, c.base + sk);      v[5] = ldf(f, c.base + sk + 4u);
    v[6] = ldf(f, c.base + sk + sj); v[7] = ldf(f, c.base + sk + sj + 4u);
}

// GPU_kernel.cu:27-41 + :53-61
template <bool FMA = false>
__device__ __forceinline__ float gather(const Field &f, const Cell &c)
{
    float v[8];
    corners(f, c, v);
    double ox, oy, oz;
    if (FMA) {      // weights are multiples of 2^-23 here (q >= 1), so 1 - c is exact in fp32 as well
        ox = (double)(1.0f - c.fx); oy = (double)(1.0f - c.fy); oz = (double)(1.0f - c.fz);
    } else {
        ox = 1.0 - (double)c.fx; oy = 1.0 - (double)c.fy; oz = 1.0 - (double)c.fz;
    }
    float l00 = lerp_w<FMA>(v[0], v[1], c.fx, ox);
    float l01 = lerp_w<FMA>(v[2], v[3], c.fx, ox);
    float l10 = lerp_w<FMA>(v[4], v[5], c.fx, ox);
    float l11 = lerp_w<FMA>(v[6], v[7], c.fx, ox);
    float m0 = lerp_w<FMA>(l00, l01, c.fy, oy);
    float m1 = lerp_w<FMA>(l10, l11, c.fy, oy);
    return lerp_w<FMA>(m0, m1, c.fz, oz);
}

// GPU_kernel.cu:43-62 sample_buffer
// GE1: the caller vouches that pos - off >= h on every axis (q >= 1): locate's NONNEG form and the one-fma lerps apply
template <bool P2, bool GE1 = false>
__device__ __forceinline__ float sample(const Field &f, const Spacing &sp, f3 off, f3 pos)
{
    return gather<GE1>(f, locate<P2, GE1>(f, sp, off, pos));
}

// three co-located fields (the x/y/z maps share cell and weights: GPU_kernel.cu:350-352 etc.)
struct Map3 { Field x, y, z; };

template <bool P2>
__device__ __forceinline__ f3 map_at(const Map3 &m, const Spacing &sp, f3 pos)
{
    Cell c = locate<P2>(m.x, sp, mk3(0.f, 0.f, 0.f), pos);
    return mk3(gather(m.x, c), gather(m.y, c), gather(m.z, c));
}

// ---- structured map look-up for the 9-point kernels, power-of-two spacing ----------------------
// The 9 sample points of a node (8 sub-voxel corners at +-h/4 and the centre) are looked up in a
// map that lives on the nodes.  With h = 2^-m the positions (i - S/2 +- 1/4) h are exact in fp32,
// so cell and weights of every tap are known at compile time (S = 1 when the sampled component is
// staggered along that axis):
//     S = 0:  '+' -> cell n,   w 0.25     '-' -> cell n-1, w 0.75     centre -> cell n,   w 0
//     S = 1:  '+' -> cell n-1, w 0.75     '-' -> cell n-1, w 0.25     centre -> cell n-1, w 0.5
// All taps read the same 3x3x3 (2 along a staggered axis) block of map nodes: it is loaded once
// per component and the lerps that several taps share (same x-weight and node row, ...) are
// evaluated once: 45 lerps instead of 63 and 27 loads instead of 72 per component -- the very
// same operations as locate()+gather() would perform, hence bit-identical results.
__device__ constexpr int tap_rel(int S, int t) { return S ? 0 : (t == 1 ? 0 : 1); }
__device__ constexpr float tap_frac(int S, int t)
{
    return S ? (t == 0 ? 0.75f : (t == 1 ? 0.25f : 0.5f)) : (t == 0 ? 0.25f : (t == 1 ? 0.75f : 0.0f));
}

// out[0..7]: corners in the reference's order (bit2 x, bit1 y, bit0 z; 0 = '+'), out[8]: centre.
// (i, j, kl): node indices in the map's LOCAL index space.
// node(x, y, z): value of map node (i - 1 + x, j - 1 + y, kl - 1 + z) by flat index (see NodesGlobal / NodesLds)
template <int SX, int SY, int SZ, class Nodes, bool Q4 = false>
__device__ __forceinline__ void map9_nodes(const Nodes &node, float out[9])
{
    constexpr int NX = SX ? 2 : 3, NY = SY ? 2 : 3, NZ = SZ ? 2 : 3;
    float N[NZ][NY][NX];
#pragma unroll
    for (int z = 0; z < NZ; z++)
#pragma unroll
        for (int y = 0; y < NY; y++)
#pragma unroll
            for (int x = 0; x < NX; x++)
                N[z][y][x] = node(x, y, z);
    // level 1: along x, for taps '+' (0), '-' (1) on every node row, centre (2) on the rows it needs
    float LX[3][NZ][NY];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int r = tap_rel(SX, t);
        const float c = tap_frac(SX, t);
#pragma unroll
        for (int z = 0; z < NZ; z++)
#pragma unroll
            for (int y = 0; y < NY; y++)
                LX[t][z][y] = lerp_q<Q4>(N[z][y][r], N[z][y][r + 1], c);
    }
    // level 2: along y
    float LY[3][3][NZ];     // [tx][ty][z]; centre only pairs with centre
#pragma unroll
    for (int tx = 0; tx < 3; tx++)
#pragma unroll
        for (int ty = 0; ty < 3; ty++) {
            if ((tx == 2) != (ty == 2)) continue;
            const int r = tap_rel(SY, ty);
            const float c = tap_frac(SY, ty);
#pragma unroll
            for (int z = 0; z < NZ; z++)
                LY[tx][ty][z] = lerp_q<Q4>(LX[tx][z][r], LX[tx][z][r + 1], c);
        }
    // level 3: along z
#pragma unroll
    for (int ii = 0; ii < 8; ii++) {
        const int tx = (ii >> 2) & 1, ty = (ii >> 1) & 1, tz = ii & 1;
        const int r = tap_rel(SZ, tz);
        out[ii] = lerp_q<Q4>(LY[tx][ty][r], LY[tx][ty][r + 1], tap_frac(SZ, tz));
    }
    {
        const int r = tap_rel(SZ, 2);
        out[8] = lerp_q<Q4>(LY[2][2][r], LY[2][2][r + 1], tap_frac(SZ, 2));
    }
}

// the 3x3x3 block straight from memory (one buffer load per node)
struct NodesGlobal {
    const Field &f; int base;
    __device__ __forceinline__ float operator()(int x, int y, int z) const
    {
        return ldf(f, (unsigned)(base + x + f.nx * y + f.nx * f.ny * z) * 4u);
    }
};
template <int SX, int SY, int SZ>
__device__ __forceinline__ void map9_component(const Field &f, int i, int j, int kl, float out[9])
{
    map9_nodes<SX, SY, SZ>(NodesGlobal{f, (i - 1) + f.nx * (j - 1) + f.nx * f.ny * (kl - 1)}, out);
}

// ---- the same look-up from a tile staged in LDS ------------------------------------------------
// A 64 x 4 block at (i0, j0) on plane kl needs the nodes x in [i0-1, i0+64], y in [j0-1, j0+4], z in [kl-1, kl+1]:
// 66 x 6 x 3 per component.  Neighbouring threads share 26 of their 27 nodes, so reading them through L1 costs
// 27 wave-loads per component and wave; staged, a wave issues 4.5 aligned row loads and reads the rest from LDS.
// Each tile element is loaded by the same flat index the direct path uses (a column left of 0 / right of nx - 1
// is the neighbouring row's end, outside the allocation reads 0), so both paths see identical values.
constexpr int kTileX = 66, kTileY = 6, kTileZ = 3, kTile = kTileX * kTileY * kTileZ;

// A map value that lets the quarter-weight lerps of map9 run in fp32 (lerp_q<true>): zero, or a coordinate in
// [h/256, 1024 h].  Every value map9 then combines is a convex combination (weights 0, 1/4, 1/2, 3/4, 1, three levels)
// of such values: zero, or within [h/256/64, 1024 h] -- any two of them less than 2^24 apart, inside lerp_q's 2^25.
// Map components are coordinates in [0, n h], n <= 1024, zero on the border the DMC update clears: in practice every
// map passes (gpu_maps_quarter_safe scans a map set); a NaN, an Inf, a negative or a denormal-small value anywhere
// keeps every launch on that map on the double-rounding path.
__device__ __forceinline__ bool tile_value_ok(float v, float lo, float hi) { return v == 0.f || (v >= lo && v <= hi); }

// all 256 threads of the block call this (no early exit before it); ends with a barrier
template <int NC>
__device__ __forceinline__ void stage_tiles(const Field (&f)[NC], int i0, int j0, int kl, float *tile)
{
    const int lane = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int nx = f[0].nx, sk = f[0].nx * f[0].ny;
    for (int r = wv; r < kTileY * kTileZ; r += 4) {
        const int z = r / kTileY, y = r - z * kTileY;
        const unsigned off = (unsigned)((i0 + lane) + nx * (j0 - 1 + y) + sk * (kl - 1 + z)) * 4u;
#pragma unroll
        for (int c = 0; c < NC; c++) tile[c * kTile + r * kTileX + 1 + lane] = ldf(f[c], off);
    }
    if (wv == 0 && lane < 2 * kTileY * kTileZ) {            // the two edge columns
        const int r = lane >> 1, X = (lane & 1) ? kTileX - 1 : 0;
        const int z = r / kTileY, y = r - z * kTileY;
        const unsigned off = (unsigned)((i0 - 1 + X) + nx * (j0 - 1 + y) + sk * (kl - 1 + z)) * 4u;
#pragma unroll
        for (int c = 0; c < NC; c++) tile[c * kTile + r * kTileX + X] = ldf(f[c], off);
    }
    __syncthreads();
}

struct NodesLds {
    const float *t;         // tile + (threadIdx.y * kTileX + threadIdx.x): this thread's node (i - 1, j - 1, kl - 1)
    __device__ __forceinline__ float operator()(int x, int y, int z) const { return t[(z * kTileY + y) * kTileX + x]; }
};
template <int SX, int SY, int SZ, bool Q4 = false>
__device__ __forceinline__ void map9_lds(const float *tile, f3 out[9])
{
    const float *t = tile + threadIdx.y * kTileX + threadIdx.x;
    float x[9], y[9], z[9];
    map9_nodes<SX, SY, SZ, NodesLds, Q4>(NodesLds{t}, x);
    map9_nodes<SX, SY, SZ, NodesLds, Q4>(NodesLds{t + kTile}, y);
    map9_nodes<SX, SY, SZ, NodesLds, Q4>(NodesLds{t + 2 * kTile}, z);
#pragma unroll
    for (int a = 0; a < 9; a++) out[a] = mk3(x[a], y[a], z[a]);
}

template <int SX, int SY, int SZ>
__device__ __forceinline__ void map9(const Map3 &m, int i, int j, int kl, f3 out[9])
{
    float x[9], y[9], z[9];
    map9_component<SX, SY, SZ>(m.x, i, j, kl, x);
    map9_component<SX, SY, SZ>(m.y, i, j, kl, y);
    map9_component<SX, SY, SZ>(m.z, i, j, kl, z);
#pragma unroll
    for (int a = 0; a < 9; a++) out[a] = mk3(x[a], y[a], z[a]);
}

// ---- the structured look-up on spacings that are NOT a power of two (round 4) -----------------------------------------------
// With h = 2^-m the nine sample points of a node hit the map at compile-time cells and weights (above).  For any other h --
// the reference's own scene has h = 0.002 -- the generic path evaluates (pos - off) / h per tap: 27 IEEE divisions and 72
// loads per map component and node, which is what made those grids 3x slower per voxel.  But a tap's cell and weight along
// an axis depend on the node's index along THAT axis, the tap (+, -, centre) and the stagger only, so the host evaluates the
// very expressions of nine_centre / nine_corner / locate once per index (bq_advect.hip: map_tabs) and the kernels read
//     frac[axis][S][tap][index]      the weight q - floor(q)
//     rel [axis][S][centre][index]   0 / 1: the centre tap's cell relative to the node block's first node
// The '+' / '-' taps land in the cell exact arithmetic predicts for every index of every spacing tried (the table builder
// verifies it, else the launch takes the generic path); the centre tap of an unstaggered axis, (i h) / h, rounds to just below
// i for a few indices (63, 125, 126 at h = 0.002f) and then reads cell i - 1 with weight 1 - 2^-24: hence `rel`.  The lerps
// are the contract's general form (weights are no longer multiples of 1/4), shared between taps exactly as in map9_nodes:
// same operations on the same operands as locate() + gather(), hence the same bits.
struct MapTabs { const float *frac; const float *rel; int stride; };
__device__ __forceinline__ int tab_at(int axis, int S, int tap, int stride) { return ((axis * 2 + S) * 3 + tap) * stride; }

// out[0..7]: corners in the reference's order, out[8]: centre.  f?[t]: weight of tap t along the axis; r?: the centre
// tap's relative cell (0 or 1 on an unstaggered axis, 0 on the staggered one)
template <int SX, int SY, int SZ, class Nodes>
__device__ __forceinline__ void map9_nodes_tab(const Nodes &node, const float (&fx)[3], const float (&fy)[3], const float (&fz)[3],
                                               int rcx, int rcy, int rcz, float out[9])
{
    constexpr int NX = SX ? 2 : 3, NY = SY ? 2 : 3, NZ = SZ ? 2 : 3;
    float N[NZ][NY][NX];
#pragma unroll
    for (int z = 0; z < NZ; z++)
#pragma unroll
        for (int y = 0; y < NY; y++)
#pragma unroll
            for (int x = 0; x < NX; x++) N[z][y][x] = node(x, y, z);
    auto lerp = [](float a, float b, float c) { return lerp_w<false>(a, b, c, 1.0 - (double)c); };
    float LX[3][NZ][NY];
#pragma unroll
    for (int t = 0; t < 3; t++) {
#pragma unroll
        for (int z = 0; z < NZ; z++)
#pragma unroll
            for (int y = 0; y < NY; y++) {
                float a, b;
                if (t < 2 || SX) { const int r = tap_rel(SX, t); a = N[z][y][r]; b = N[z][y][r + 1]; }
                else { a = rcx ? N[z][y][1] : N[z][y][0]; b = rcx ? N[z][y][NX - 1] : N[z][y][1]; }
                LX[t][z][y] = lerp(a, b, fx[t]);
            }
    }
    float LY[3][3][NZ];
#pragma unroll
    for (int tx = 0; tx < 3; tx++)
#pragma unroll
        for (int ty = 0; ty < 3; ty++) {
            if ((tx == 2) != (ty == 2)) continue;
#pragma unroll
            for (int z = 0; z < NZ; z++) {
                float a, b;
                if (ty < 2 || SY) { const int r = tap_rel(SY, ty); a = LX[tx][z][r]; b = LX[tx][z][r + 1]; }
                else { a = rcy ? LX[tx][z][1] : LX[tx][z][0]; b = rcy ? LX[tx][z][NY - 1] : LX[tx][z][1]; }
                LY[tx][ty][z] = lerp(a, b, fy[ty]);
            }
        }
#pragma unroll
    for (int ii = 0; ii < 8; ii++) {
        const int tx = (ii >> 2) & 1, ty = (ii >> 1) & 1, tz = ii & 1;
        const int r = tap_rel(SZ, tz);
        out[ii] = lerp(LY[tx][ty][r], LY[tx][ty][r + 1], fz[tz]);
    }
    {
        float a, b;
        if (SZ) { a = LY[2][2][0]; b = LY[2][2][1]; }
        else { a = rcz ? LY[2][2][1] : LY[2][2][0]; b = rcz ? LY[2][2][NZ - 1] : LY[2][2][1]; }
        out[8] = lerp(a, b, fz[2]);
    }
}

// (i, j, kg): the node's indices in the map's GLOBAL index space (the tables are global along z)
template <int SX, int SY, int SZ>
__device__ __forceinline__ void map9_lds_tab(const float *tile, const MapTabs &tb, int i, int j, int kg, f3 out[9])
{
    float fx[3], fy[3], fz[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        fx[t] = tb.frac[tab_at(0, SX, t, tb.stride) + i];
        fy[t] = tb.frac[tab_at(1, SY, t, tb.stride) + j];
        fz[t] = tb.frac[tab_at(2, SZ, t, tb.stride) + kg];
    }
    const int rcx = SX ? 0 : (int)tb.rel[tab_at(0, 0, 2, tb.stride) + i];
    const int rcy = SY ? 0 : (int)tb.rel[tab_at(1, 0, 2, tb.stride) + j];
    const int rcz = SZ ? 0 : (int)tb.rel[tab_at(2, 0, 2, tb.stride) + kg];
    const float *t = tile + threadIdx.y * kTileX + threadIdx.x;
    float x[9], y[9], z[9];
    map9_nodes_tab<SX, SY, SZ, NodesLds>(NodesLds{t}, fx, fy, fz, rcx, rcy, rcz, x);
    map9_nodes_tab<SX, SY, SZ, NodesLds>(NodesLds{t + kTile}, fx, fy, fz, rcx, rcy, rcz, y);
    map9_nodes_tab<SX, SY, SZ, NodesLds>(NodesLds{t + 2 * kTile}, fx, fy, fz, rcx, rcy, rcz, z);
#pragma unroll
    for (int a = 0; a < 9; a++) out[a] = mk3(x[a], y[a], z[a]);
}

// MAC velocity (GPU_kernel.cu:64-72)
struct Vel3 { Field u, v, w; };

template <bool P2, bool GE1 = false>
__device__ __forceinline__ f3 get_velocity(const Vel3 &vel, const Spacing &sp, f3 pos)
{
    float mh = (float)(-0.5 * (double)sp.h);
    return mk3(sample<P2, GE1>(vel.u, sp, mk3(mh, 0.f, 0.f), pos),
               sample<P2, GE1>(vel.v, sp, mk3(0.f, mh, 0.f), pos),
               sample<P2, GE1>(vel.w, sp, mk3(0.f, 0.f, mh), pos));
}

// The same look-up at a position that is not known to be inside the grid: one test per wave decides between the short
// form (every lane at least h from the origin on every axis: q >= 1) and the general one.  A NaN fails the test.
template <bool P2>
__device__ __forceinline__ f3 get_velocity_auto(const Vel3 &vel, const Spacing &sp, f3 pos)
{
    if (__all(fminf(pos.x, fminf(pos.y, pos.z)) >= sp.h && pos.x == pos.x && pos.y == pos.y && pos.z == pos.z))
        return get_velocity<P2, true>(vel, sp, pos);
    return get_velocity<P2, false>(vel, sp, pos);
}

// GPU_kernel.cu:74-90 traceRK3
template <bool P2>
__device__ __forceinline__ f3 trace_rk3(const Vel3 &vel, const Spacing &sp, f3 hi, float dt, f3 pos)
{
    float c1 = (float)(2.0 / 9.0 * (double)dt);
    float c2 = (float)(3.0 / 9.0 * (double)dt);
    float c3 = (float)(4.0 / 9.0 * (double)dt);
    f3 v1 = get_velocity_auto<P2>(vel, sp, pos);
    double hdt = 0.5 * (double)dt;
    f3 m1 = mk3((float)((double)pos.x + hdt * (double)v1.x),
                (float)((double)pos.y + hdt * (double)v1.y),
                (float)((double)pos.z + hdt * (double)v1.z));
    f3 v2 = get_velocity_auto<P2>(vel, sp, m1);
    double qdt = 0.75 * (double)dt;
    f3 m2 = mk3((float)((double)pos.x + qdt * (double)v2.x),
                (float)((double)pos.y + qdt * (double)v2.y),
                (float)((double)pos.z + qdt * (double)v2.z));
    f3 v3 = get_velocity_auto<P2>(vel, sp, m2);
    f3 out = mk3(pos.x + c1 * v1.x + c2 * v2.x + c3 * v3.x,
                 pos.y + c1 * v1.y + c2 * v2.y + c3 * v3.y,
                 pos.z + c1 * v1.z + c2 * v2.z + c3 * v3.z);
    return clamp3(out, mk3(sp.h, sp.h, sp.h), hi);
}

// GPU_kernel.cu:92-125 trace
template <bool P2>
__device__ __forceinline__ f3 trace(const Vel3 &vel, const Spacing &sp, f3 hi, float cfldt, float dt, f3 pos)
{
    const bool fwd = dt > 0;
    float T = fwd ? dt : -dt;
    float t = 0.f, substep = cfldt;
    f3 p = pos;
    while (t < T) {
        if (t + substep > T) substep = T - t;
        p = trace_rk3<P2>(vel, sp, hi, fwd ? substep : -substep, p);
        t += substep;
    }
    return p;
}

// expf of the DMC integrator: the oracle's orc_expf, operation for operation.
__device__ __forceinline__ float exp_portable(float xf)
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
    long long k = (long long)kd;
    unsigned long long bits = (unsigned long long)(k + 1023) << 52;
    double scale = __builtin_bit_cast(double, bits);
    return (float)(p * scale);
}

// wave64 reductions (DPP/bpermute via __shfl_xor; 64 lanes)
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

} // inline namespace BQ_VARIANT
} // namespace bq
