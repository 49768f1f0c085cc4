// bq_project.hip -- pressure projection: divergence, Jacobi
// sweeps, gradient subtraction, residual norms, viscous diffusion sweeps.
//
// The Jacobi sweep is THE roofline kernel of this path: 12 algorithmic bytes per voxel per
// sweep (read p, read div, write p').  jacobi_tile_kernel stages each p plane once through
// LDS (tile + one-deep halo), marches along k with the k-1 / k / k+1 centre values in
// registers, and moves everything as 16-byte vectors; see DESIGN.md "Jacobi kernel".
// Summation order is the reference's (GPU_kernel.cu:1834), so results are bit-identical.
#include "bq_device.hip.h"
#include "bq_buffer.hip.h"
#include <type_traits>
#include "bq_host.h"
#include <algorithm>
#include <vector>

namespace bq {

// z-slab context of a launch: local plane k is global plane k + koff; nkg = global cell planes;
// [klo, khi): optional restriction of a stencil launch to a range of LOCAL planes (used to sweep the
// slab interior while the ghost planes are still in flight).
struct Slab { int koff, nkg, klo, khi; };
// Output planes of a fused two-sweep launch: chunks bz < nchA march [k0a, k1a), the others [k0b, k1b) (used to
// split a launch into the part that needs no ghost planes and the two parts next to them); L1 is evaluated
// wherever those outputs need it.  Whole array: {0, nz, 0, 0, nbz}.
struct PairRanges { int k0a, k1a, k0b, k1b, nchA; };

// ---- divergence_kernel (GPU_kernel.cu:967-985) --------------------------------------------
__global__ __launch_bounds__(256) void divergence_kernel(const float *__restrict__ u, const float *__restrict__ v,
                                                         const float *__restrict__ w, float *__restrict__ div,
                                                         int ni, int nj, int nk, float halfrdx, Slab sl)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    const int kg = k + sl.koff;
    if (i >= ni || j >= nj || kg < 0 || kg >= sl.nkg) return;
    const size_t iu = (size_t)i + (size_t)(ni + 1) * ((size_t)j + (size_t)nj * k);
    const size_t iv = (size_t)i + (size_t)ni * ((size_t)j + (size_t)(nj + 1) * k);
    const size_t ic = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * k);
    float ul = u[iu], ur = u[iu + 1];
    float vf = v[iv], vb = v[iv + ni];
    float wd = w[ic], wu = w[ic + (size_t)ni * nj];
    div[ic] = halfrdx * ((ur - ul) + (vb - vf) + (wu - wd));
}

// ---- gradient_kernel x3 fused (GPU_kernel.cu:1024-1041, launches :1881-1891) -------------
// All three components update the same cell window 2..n-1, so one pass reads p once.
__global__ __launch_bounds__(256) void gradient_kernel(float *__restrict__ u, float *__restrict__ v, float *__restrict__ w,
                                                       const float *__restrict__ p, int ni, int nj, int nk, float halfrdx, Slab sl)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    const int kg = k + sl.koff;
    if (i < 2 || i >= ni || j < 2 || j >= nj || kg < 2 || kg >= sl.nkg || k < 1) return;
    const size_t ic = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * k);
    const float p0 = p[ic];
    const size_t iu = (size_t)i + (size_t)(ni + 1) * ((size_t)j + (size_t)nj * k);
    const size_t iv = (size_t)i + (size_t)ni * ((size_t)j + (size_t)(nj + 1) * k);
    u[iu] -= halfrdx * (p0 - p[ic - 1]);
    v[iv] -= halfrdx * (p0 - p[ic - ni]);
    w[ic] -= halfrdx * (p0 - p[ic - (size_t)ni * nj]);
}

// The same update that also hands out what it changed: d = u_new - u_old on the window, 0 elsewhere -- exactly
// the d*Proj = U - UTemp that BimocqGPUSolver.cpp:188-193 forms from a snapshot taken before the projection
// (u_new and u_old are the very floats that subtraction sees), without the three snapshot copies and the three
// subtraction passes.  One thread per node of the (ni+1, nj+1, nk+1) super-grid.
__global__ __launch_bounds__(256) void gradient_delta_kernel(float *__restrict__ u, float *__restrict__ v, float *__restrict__ w,
                                                             const float *__restrict__ p,
                                                             float *__restrict__ du, float *__restrict__ dv, float *__restrict__ dw,
                                                             int ni, int nj, int nk, float halfrdx, Slab sl)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i > ni || j > nj || k > nk) return;
    const int kg = k + sl.koff;
    const bool win = !(i < 2 || i >= ni || j < 2 || j >= nj || kg < 2 || kg >= sl.nkg || k < 1 || k >= nk);
    const size_t iu = (size_t)i + (size_t)(ni + 1) * ((size_t)j + (size_t)nj * k);
    const size_t iv = (size_t)i + (size_t)ni * ((size_t)j + (size_t)(nj + 1) * k);
    const size_t ic = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * k);
    if (win) {
        const float p0 = p[ic];
        const float uo = u[iu], vo = v[iv], wo = w[ic];
        const float un = uo - halfrdx * (p0 - p[ic - 1]);
        const float vn = vo - halfrdx * (p0 - p[ic - ni]);
        const float wn = wo - halfrdx * (p0 - p[ic - (size_t)ni * nj]);
        u[iu] = un; v[iv] = vn; w[ic] = wn;
        du[iu] = un - uo; dv[iv] = vn - vo; dw[ic] = wn - wo;
    } else {
        if (j < nj && k < nk) du[iu] = 0.f;                  // the u buffer has ni+1 columns
        if (i < ni && k < nk) dv[iv] = 0.f;                  // the v buffer has nj+1 rows
        if (i < ni && j < nj) dw[ic] = 0.f;                  // the w buffer has nk+1 planes
    }
}

// ---- generic Jacobi sweep: any dims, one thread per cell (GPU_kernel.cu:1819-1837) --------
__global__ __launch_bounds__(256) void jacobi_generic_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                             float *__restrict__ out, int ni, int nj, int nk,
                                                             float alpha, float beta, Slab sl)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    const int kg = k + sl.koff;
    if (!(i > 0 && i < ni - 1 && j > 0 && j < nj - 1 && k > 0 && k < nk - 1 && kg > 0 && kg < sl.nkg - 1 && k >= sl.klo && k < sl.khi)) return;
    const size_t sj = ni, sk = (size_t)ni * nj;
    const size_t id = (size_t)i + sj * j + sk * k;
    out[id] = (p[id - 1] + p[id + 1] + p[id - sj] + p[id + sj] + p[id - sk] + p[id + sk] + alpha * div[id]) * beta;
}

// ---- LDS-tiled, k-marching Jacobi sweep -----------------------------------------------------
// Block = 256 threads.  Tile = TX (= 4*TXV) floats in x by TY (= R*256/TXV) rows in y; the block
// marches over `kchunk` planes.  Per plane each thread owns R float4 of the tile: it loads them
// once from HBM (as plane k+1), keeps them in registers while they serve as k+1, k and k-1, and
// publishes them to LDS when they are plane k so that neighbours can read x+-1 / y+-1.
// LDS rows carry a 4-float pad on each side: the x halo column sits at pad[3] / pad'[0] and the
// 16-byte alignment of the interior is kept.
template <int TXV, int R>
struct JTile {
    static constexpr int RP = 256 / TXV;        // rows per pass
    static constexpr int TY = RP * R;
    static constexpr int TX = TXV * 4;
    static constexpr int LW = TX + 8;           // LDS row pitch (floats)
    static constexpr int PLANE = (TY + 2) * LW; // floats per LDS plane
};

template <int TXV, int R>
__global__ __launch_bounds__(256) void jacobi_tile_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                          float *__restrict__ out, int nx, int ny, int nz,
                                                          int kchunk, float alpha, float beta, Slab sl)
{
    using T = JTile<TXV, R>;
    __shared__ __attribute__((aligned(16))) float lds[2][T::PLANE];

    const int tid = threadIdx.x;
    const int lx = tid % TXV, rp = tid / TXV;
    const int x0 = blockIdx.x * T::TX, j0 = blockIdx.y * T::TY;
    // local planes 1..nz-2 that are interior planes of the GLOBAL grid
    const int kbeg = max(max(max(1, 1 - sl.koff), sl.klo), (int)blockIdx.z * kchunk);
    const int kend = min(min(min(nz - 1, sl.nkg - 1 - sl.koff), sl.khi), (int)blockIdx.z * kchunk + kchunk);
    if (kbeg >= kend) return;

    const int x = x0 + 4 * lx;                  // first column of this thread's float4 (nx % 4 == 0)
    const bool xin = x < nx;
    const size_t sj = nx, sk = (size_t)nx * ny;

    // halo assignments: two extra rows (j0-1, j0+TY) as float4, two extra columns as scalars
    const bool hrow_thread = tid < 2 * TXV;
    const int hr_row = (tid / TXV) ? T::TY : -1;
    const int hr_j = j0 + hr_row;
    const int hr_x = x0 + 4 * (tid % TXV);
    const bool hr_ok = hrow_thread && hr_j >= 0 && hr_j < ny && hr_x < nx;
    const int hc_id = tid - (256 - 2 * (T::TY + 2));           // >= 0 for the column-halo threads
    const bool hcol_thread = hc_id >= 0;
    const int hc_side = hcol_thread ? hc_id / (T::TY + 2) : 0;  // 0 = left, 1 = right
    const int hc_row = hcol_thread ? hc_id % (T::TY + 2) - 1 : 0;
    const int hc_j = j0 + hc_row;
    const int hc_x = hc_side ? x0 + T::TX : x0 - 1;
    const bool hc_ok = hcol_thread && hc_j >= 0 && hc_j < ny && hc_x >= 0 && hc_x < nx;

    float4 pm[R], pc[R], pn[R];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // prologue: plane kbeg-1 -> pm (own only), plane kbeg -> pc + LDS[0] (own + halos)
#pragma unroll
    for (int m = 0; m < R; m++) {
        const int j = j0 + rp + m * T::RP;
        const bool ok = xin && j < ny;
        const size_t g = (size_t)x + sj * j;
        pm[m] = ok ? *reinterpret_cast<const float4 *>(p + g + sk * (kbeg - 1)) : zero4;
        pc[m] = ok ? *reinterpret_cast<const float4 *>(p + g + sk * kbeg) : zero4;
    }
    {
        float4 hr = hr_ok ? *reinterpret_cast<const float4 *>(p + (size_t)hr_x + sj * hr_j + sk * kbeg) : zero4;
        float hc = hc_ok ? p[(size_t)hc_x + sj * hc_j + sk * kbeg] : 0.f;
        float *L = lds[0];
#pragma unroll
        for (int m = 0; m < R; m++)
            *reinterpret_cast<float4 *>(L + (rp + m * T::RP + 1) * T::LW + 4 + 4 * lx) = pc[m];
        if (hrow_thread) *reinterpret_cast<float4 *>(L + (hr_row + 1) * T::LW + 4 + 4 * (tid % TXV)) = hr;
        if (hcol_thread) L[(hc_row + 1) * T::LW + (hc_side ? 4 + T::TX : 3)] = hc;
    }

    int cur = 0;
    for (int k = kbeg; k < kend; k++) {
        // (1) issue the loads of plane k+1 (own + halos) and div(k)
        float4 dv[R];
        const bool last = (k + 1 >= nz);       // never true (kend <= nz-1) but keeps loads in range
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int j = j0 + rp + m * T::RP;
            const bool ok = xin && j < ny;
            const size_t g = (size_t)x + sj * j;
            pn[m] = (ok && !last) ? *reinterpret_cast<const float4 *>(p + g + sk * (k + 1)) : zero4;
            dv[m] = ok ? *reinterpret_cast<const float4 *>(div + g + sk * k) : zero4;
        }
        float4 hr = (hr_ok && !last) ? *reinterpret_cast<const float4 *>(p + (size_t)hr_x + sj * hr_j + sk * (k + 1)) : zero4;
        float hc = (hc_ok && !last) ? p[(size_t)hc_x + sj * hc_j + sk * (k + 1)] : 0.f;

        // (2) plane k is complete in LDS[cur]
        __syncthreads();
        const float *L = lds[cur];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int row = rp + m * T::RP;
            const int j = j0 + row;
            const float *c = L + (row + 1) * T::LW + 4 + 4 * lx;
            const float left = c[-1], right = c[4];
            const float4 fr = *reinterpret_cast<const float4 *>(c - T::LW);   // j-1
            const float4 bk = *reinterpret_cast<const float4 *>(c + T::LW);   // j+1
            const float4 q = pc[m];
            float4 o;
            // ((((((l + r) + f) + b) + d) + u) + alpha*div) * beta   -- GPU_kernel.cu:1834
            o.x = (left + q.y + fr.x + bk.x + pm[m].x + pn[m].x + alpha * dv[m].x) * beta;
            o.y = (q.x + q.z + fr.y + bk.y + pm[m].y + pn[m].y + alpha * dv[m].y) * beta;
            o.z = (q.y + q.w + fr.z + bk.z + pm[m].z + pn[m].z + alpha * dv[m].z) * beta;
            o.w = (q.z + right + fr.w + bk.w + pm[m].w + pn[m].w + alpha * dv[m].w) * beta;
            if (xin && j >= 1 && j < ny - 1) {
                float *dst = out + (size_t)x + sj * j + sk * k;
                if (x >= 4 && x + 4 < nx) {
                    *reinterpret_cast<float4 *>(dst) = o;
                } else {                        // the float4 touches the x boundary: interior lanes only
                    if (x >= 1) dst[0] = o.x;
                    dst[1] = o.y;
                    dst[2] = o.z;
                    if (x + 3 < nx - 1) dst[3] = o.w;
                }
            }
        }

        // (3) publish plane k+1 to the other LDS buffer, rotate registers
        float *Ln = lds[cur ^ 1];
#pragma unroll
        for (int m = 0; m < R; m++) {
            *reinterpret_cast<float4 *>(Ln + (rp + m * T::RP + 1) * T::LW + 4 + 4 * lx) = pn[m];
            pm[m] = pc[m];
            pc[m] = pn[m];
        }
        if (hrow_thread) *reinterpret_cast<float4 *>(Ln + (hr_row + 1) * T::LW + 4 + 4 * (tid % TXV)) = hr;
        if (hcol_thread) Ln[(hc_row + 1) * T::LW + (hc_side ? 4 + T::TX : 3)] = hc;
        cur ^= 1;
    }
}

// ---- register-marching Jacobi sweep (no LDS, no barrier) -----------------------------------------
// At 256^3 the three arrays (201 MB) live in the 256 MiB Infinity Cache, where a plain triad streams
// the same 12 B/voxel in ~25 us (tools/membw.hip): the sweep is then bound by how many independent
// 16-byte loads a CU keeps in flight, and a barrier per plane (the LDS kernel above) throttles that.
// Here a thread owns one float4 column of one row and marches along k with the k-1/k/k+1 centre
// values in registers; the j-1/j+1 rows are re-read as float4 (L1/L2 hits: the neighbouring waves of
// the block load them as their own centres), x-1/x+4 come from the neighbouring lanes by shuffle.
// The loads of plane k+1 are issued before plane k is computed, so each wave always has 4 vector
// loads in flight and never waits on another wave.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void jacobi_march_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                                  float *__restrict__ out, int nx, int ny, int nz,
                                                                  int cw, int nbx, int nby, int kchunk, float alpha, float beta, Slab sl)
{
    // XCD-aware block order: blocks b and b+8 share an XCD (its L2); give each XCD a contiguous run
    // of (k-chunk, row-block) pairs so halo rows/planes are re-read from the same L2.
    const int nblk = gridDim.x;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);
    const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);

    const int rows = (WAVES * 64) / cw;             // rows of the tile (cw float4 columns each)
    const int c = threadIdx.x % cw, r = threadIdx.x / cw;
    const int xraw = (bx * cw + c) * 4, j = by * rows + r;
    const int kbeg = max(max(max(1, 1 - sl.koff), sl.klo), bz * kchunk);
    const int kend = min(min(min(nz - 1, sl.nkg - 1 - sl.koff), sl.khi), bz * kchunk + kchunk);
    if (kbeg >= kend) return;
    const bool active = xraw < nx && j >= 1 && j <= ny - 2;
    const int x = xraw < nx ? xraw : nx - 4;                    // lanes past the row read its last float4, never store
    const size_t sj = nx, sk = (size_t)nx * ny;
    const size_t g = (size_t)x + sj * (active ? j : 1);         // inactive lanes read a valid row, never store
    const int lane = threadIdx.x & 63;
    // a lane at a wave edge whose x-neighbour lives in another wave (rows longer than one wave)
    const bool fix_l = lane == 0 && c > 0, fix_r = lane == 63 && c < cw - 1;
    const bool tile_l = c == 0 && x > 0, tile_r = c == cw - 1 && x + 4 < nx;   // neighbour in another block

    // no branch or select per load: out-of-range lanes and the plane past the chunk's end are clamped into the
    // array, what they load is never used for a stored cell
#define BQ_LD4(ptr, off) (*reinterpret_cast<const float4 *>((ptr) + (off)))
    float4 pm = BQ_LD4(p, g + sk * (kbeg - 1));
    float4 pc = BQ_LD4(p, g + sk * kbeg);
    float4 pn = BQ_LD4(p, g + sk * (kbeg + 1));
    float4 fr = BQ_LD4(p, g - sj + sk * kbeg);
    float4 bk = BQ_LD4(p, g + sj + sk * kbeg);
    float4 dv = BQ_LD4(div, g + sk * kbeg);

    for (int k = kbeg; k < kend; k++) {
        // prefetch everything plane k+1 needs (its k+2 centre, its j+-1 rows, its div)
        const size_t k2 = (size_t)min(k + 2, nz - 1);       // k + 1 <= kend <= nz - 1 needs no clamp
        const float4 pn2 = BQ_LD4(p, g + sk * k2);
        const float4 fr2 = BQ_LD4(p, g - sj + sk * (k + 1));
        const float4 bk2 = BQ_LD4(p, g + sj + sk * (k + 1));
        const float4 dv2 = BQ_LD4(div, g + sk * (k + 1));
        float left = lane_up(pc.w), right = lane_down(pc.x);
        if (fix_l || tile_l) left = p[g - 1 + sk * k];
        if (fix_r || tile_r) right = p[g + 4 + sk * k];
        float4 o;
        // ((((((l + r) + f) + b) + d) + u) + alpha*div) * beta   -- GPU_kernel.cu:1834
        o.x = (left + pc.y + fr.x + bk.x + pm.x + pn.x + alpha * dv.x) * beta;
        o.y = (pc.x + pc.z + fr.y + bk.y + pm.y + pn.y + alpha * dv.y) * beta;
        o.z = (pc.y + pc.w + fr.z + bk.z + pm.z + pn.z + alpha * dv.z) * beta;
        o.w = (pc.z + right + fr.w + bk.w + pm.w + pn.w + alpha * dv.w) * beta;
        if (active) {
            float *dst = out + g + sk * k;
            if (x >= 4 && x + 4 < nx) {
                *reinterpret_cast<float4 *>(dst) = o;
            } else {                            // the float4 touches the x boundary: interior cells only
                if (x >= 1) dst[0] = o.x;
                dst[1] = o.y;
                dst[2] = o.z;
                if (x + 3 < nx - 1) dst[3] = o.w;
            }
        }
        pm = pc; pc = pn; pn = pn2; fr = fr2; bk = bk2; dv = dv2;
    }
#undef BQ_LD4
}

// ---- two Jacobi sweeps per launch (temporal fusion, register marching) ---------------------------
// The single-sweep kernel moves 12 B/voxel/sweep through the fabric.  Two consecutive sweeps
//     L1 = J(L0),  L2 = J(L1)        (J = jacobi_kernel, GPU_kernel.cu:1819-1837)
// are fused so that L1 never leaves the registers: a thread marches along k for one float4 column of
// row j, keeps L0 of the rows j-1, j, j+1 for three planes in registers, evaluates L1 on those three
// rows (the two outer ones redundantly: they are what the neighbouring waves compute as their centre)
// and from them L2 on row j, one plane behind.  Per plane: 8 float4 loads (5 of them L1/L2 hits),
// one store, for TWO sweeps -- the fabric traffic per sweep halves.  Each value is produced by exactly
// the reference's expression, so the result is bit-identical to two single sweeps.
// Preconditions (checked by the launcher): nx <= 256 (a row fits one wave; x-neighbours by shuffle)
// and both ping-pong buffers carry the same boundary layer (gpu_projection_jacobi's contract: the
// caller zeroes both), because boundary cells are never written and L1's boundary is taken from L0.
// WIDE: a row spans several waves (nx > 256 floats).  x-neighbours across a wave boundary cannot be shuffled:
// lane 0 / lane 63 fetch the L0 values of the cell just outside the wave from memory and evaluate the single L1
// value there themselves (a few one-lane loads per plane; no LDS, no barrier) -- the same scheme as the fp64
// smoother mg_smooth2_kernel (bq_mgcg.hip).
template <int WAVES, bool WIDE>
__global__ __launch_bounds__(WAVES * 64) void jacobi_march2_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                                   float *__restrict__ out, int nx, int ny, int nz,
                                                                   int cw, int nby, int kchunk, float alpha, float beta, Slab sl, PairRanges rg)
{
    const int nblk = gridDim.x;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);      // XCD-contiguous block order
    const int by = b % nby, bz = b / nby;
    const int rows = (WAVES * 64) / cw;
    const int c = threadIdx.x % cw, r = threadIdx.x / cw;
    const int xraw = 4 * c, j = by * rows + r;
    // local planes a sweep may update: inside the array AND inside the global domain (z-slab ranks);
    // everything else counts as boundary and keeps its input value in L1
    const int kA = max(1, 1 - sl.koff), kB = min(nz - 1, sl.nkg - 1 - sl.koff);
    const int r0 = bz < rg.nchA ? rg.k0a + bz * kchunk : rg.k0b + (bz - rg.nchA) * kchunk, r1 = bz < rg.nchA ? rg.k1a : rg.k1b;
    const int kbeg = max(kA, r0), kend = min(min(kB, r1), r0 + kchunk);
    if (kbeg >= kend) return;
    const bool xok = xraw < nx;
    const bool active = xok && j >= 1 && j <= ny - 2;
    // Out-of-range rows, planes and lanes are CLAMPED into the array instead of being zero-filled: whatever
    // they load only ever feeds cells that are boundary (kept from L0) or not stored at all.  No branch per load.
    const int x = xok ? xraw : nx - 4;
    const size_t sj = nx, sk = (size_t)nx * ny;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool xlo = x == 0, xhi = x + 3 == nx - 1;
    auto rowoff = [&](int row) -> size_t { return (size_t)x + sj * (size_t)min(max(row, 0), ny - 1); };
    const size_t o_m2 = rowoff(j - 2), o_m1 = rowoff(j - 1), o_0 = rowoff(j), o_p1 = rowoff(j + 1), o_p2 = rowoff(j + 2);
    auto plane = [&](int pl) -> size_t { return sk * (size_t)min(max(pl, 0), nz - 1); };
    auto ld4 = [&](const float *ptr, size_t off) -> float4 { return *reinterpret_cast<const float4 *>(ptr + off); };
    const bool rowb_m1 = j - 1 <= 0 || j - 1 >= ny - 1, rowb_0 = j <= 0 || j >= ny - 1, rowb_p1 = j + 1 <= 0 || j + 1 >= ny - 1;
    // edge lanes (WIDE): lane 0 looks after the cell just left of the wave, lane 63 after the cell just right of it
    const int lane = threadIdx.x & 63;
    const bool edgeL = WIDE && lane == 0 && xok && xraw > 0, edgeR = WIDE && lane == 63 && xraw + 4 < nx;
    const bool edge = edgeL || edgeR;
    const int xe = edgeL ? xraw - 1 : xraw + 4;                        // the outside cell
    const int xo = edgeL ? xe - 1 : xe + 1;                            // its own outer x-neighbour
    const size_t e_m1 = (size_t)min(max(xe, 0), nx - 1) + sj * (size_t)min(max(j - 1, 0), ny - 1),
                 e_0 = (size_t)min(max(xe, 0), nx - 1) + sj * (size_t)min(max(j, 0), ny - 1),
                 e_p1 = (size_t)min(max(xe, 0), nx - 1) + sj * (size_t)min(max(j + 1, 0), ny - 1),
                 e_o = (size_t)min(max(xo, 0), nx - 1) + sj * (size_t)min(max(j, 0), ny - 1);
    const bool xe_boundary = xe <= 0 || xe >= nx - 1;

    // one Jacobi evaluation on a float4; ce = centre row (x-neighbours from the neighbouring lanes, or `outside`
    // at a wave edge), fr/bk = rows -+1, dn/up = planes -+1.  Boundary cells keep the input value.
    auto jac = [&](float4 ce, float4 fr, float4 bk, float4 dn, float4 up, float4 dv, float outside, bool boundary) -> float4 {
        float left = lane_up(ce.w), right = lane_down(ce.x);
        if (WIDE) {
            if (lane == 0) left = outside;
            if (lane == 63) right = outside;
        }
        float4 o;
        o.x = (left + ce.y + fr.x + bk.x + dn.x + up.x + alpha * dv.x) * beta;
        o.y = (ce.x + ce.z + fr.y + bk.y + dn.y + up.y + alpha * dv.y) * beta;
        o.z = (ce.y + ce.w + fr.z + bk.z + dn.z + up.z + alpha * dv.z) * beta;
        o.w = (ce.z + right + fr.w + bk.w + dn.w + up.w + alpha * dv.w) * beta;
        if (boundary) return ce;
        if (xlo) o.x = ce.x;
        if (xhi) o.w = ce.w;
        return o;
    };

    // L0 of rows j-1, j, j+1 on three consecutive planes; q = plane whose L1 is being built.  The loads of plane
    // q+2 are issued before plane q is computed.  The three plane sets rotate roles (below, centre, above) from
    // one plane to the next: the loop body is written once as `phase` and instantiated three times with the
    // sets permuted, so that no register moves are needed to rotate them.
    float4 A[3], B[3], C[3], Dv[3], Hf, Hb;
    int q = kbeg - 1;
    {
        const size_t pm = plane(q - 1), pc = plane(q), pn = plane(q + 1);
        A[0] = ld4(p, pm + o_m1); A[1] = ld4(p, pm + o_0); A[2] = ld4(p, pm + o_p1);
        B[0] = ld4(p, pc + o_m1); B[1] = ld4(p, pc + o_0); B[2] = ld4(p, pc + o_p1);
        C[0] = ld4(p, pn + o_m1); C[1] = ld4(p, pn + o_0); C[2] = ld4(p, pn + o_p1);
        Dv[0] = ld4(div, pc + o_m1); Dv[1] = ld4(div, pc + o_0); Dv[2] = ld4(div, pc + o_p1);
        Hf = ld4(p, pc + o_m2); Hb = ld4(p, pc + o_p2);
    }
    float4 Mc[3] = { zero4, zero4, zero4 };     // L1 on plane q-1
    float4 Mm = zero4;                          // L1 of row j on plane q-2
    float4 Dprev = zero4;                       // div of row j on plane q-1
    // the outside cell (edge lanes, WIDE only): L0 on rows j-1..j+1 at plane q (Ec), on row j at planes q-1 (Em) and
    // q+1 (En), its outer x-neighbour at plane q (Eo), its div (Eb); Xp: its L1 on row j, plane q-1
    float Ec[3] = { 0.f, 0.f, 0.f }, Em = 0.f, En = 0.f, Eo = 0.f, Eb = 0.f, Xp = 0.f;
    if (edge) {
        const size_t pm = plane(q - 1), pc = plane(q), pn = plane(q + 1);
        Ec[0] = p[pc + e_m1]; Ec[1] = p[pc + e_0]; Ec[2] = p[pc + e_p1];
        Em = p[pm + e_0]; En = p[pn + e_0]; Eo = p[pc + e_o]; Eb = div[pc + e_0];
    }

    // one plane: Lm/Lc/Ln = L0 on planes q-1/q/q+1; on return Lm holds plane q+2 (it becomes the next "above")
    auto phase = [&](float4 (&Lm)[3], float4 (&Lc)[3], float4 (&Ln)[3]) {
        const size_t pa = plane(q + 2), pb = plane(q + 1);
        float4 La[3], Da[3];
        La[0] = ld4(p, pa + o_m1); La[1] = ld4(p, pa + o_0); La[2] = ld4(p, pa + o_p1);
        Da[0] = ld4(div, pb + o_m1); Da[1] = ld4(div, pb + o_0); Da[2] = ld4(div, pb + o_p1);
        const float4 Hfa = ld4(p, pb + o_m2), Hba = ld4(p, pb + o_p2);
        float E2[3] = { 0.f, En, 0.f }, En2 = 0.f, Eo2 = 0.f, Eb2 = 0.f;
        if (edge) {
            E2[0] = p[pb + e_m1]; E2[2] = p[pb + e_p1];
            En2 = p[pa + e_0]; Eo2 = p[pb + e_o]; Eb2 = div[pb + e_0];
        }
        // L1 on plane q for rows j-1, j, j+1
        const bool qb = q < kA || q >= kB;
        float4 M[3];
        M[0] = jac(Lc[0], Hf, Lc[1], Lm[0], Ln[0], Dv[0], Ec[0], qb || rowb_m1);
        M[1] = jac(Lc[1], Lc[0], Lc[2], Lm[1], Ln[1], Dv[1], Ec[1], qb || rowb_0);
        M[2] = jac(Lc[2], Lc[1], Hb, Lm[2], Ln[2], Dv[2], Ec[2], qb || rowb_p1);
        // L1 of the outside cell on row j, plane q (edge lanes; a boundary cell keeps L0)
        float X = Ec[1];
        if (edge && !(qb || rowb_0 || xe_boundary)) {
            const float l = edgeL ? Eo : Lc[1].w, rr = edgeL ? Lc[1].x : Eo;
            X = (l + rr + Ec[0] + Ec[2] + Em + En + alpha * Eb) * beta;
        }
        // L2 on plane q-1 for row j
        const int k = q - 1;
        const float4 o = jac(Mc[1], Mc[0], Mc[2], Mm, M[1], Dprev, Xp, false);
        if (active && k >= kbeg && k < kend) {
            float *dst = out + (size_t)x + sj * j + sk * k;
            if (x >= 4 && x + 4 < nx) {
                *reinterpret_cast<float4 *>(dst) = o;
            } else {
                if (x >= 1) dst[0] = o.x;
                dst[1] = o.y;
                dst[2] = o.z;
                if (x + 3 < nx - 1) dst[3] = o.w;
            }
        }
        Mm = Mc[1];
        Dprev = Dv[1];
        Xp = X;
        Em = Ec[1]; En = En2; Eo = Eo2; Eb = Eb2;
#pragma unroll
        for (int a = 0; a < 3; a++) { Mc[a] = M[a]; Lm[a] = La[a]; Dv[a] = Da[a]; Ec[a] = E2[a]; }
        Hf = Hfa; Hb = Hba;
        q++;
    };
    while (q <= kend) {
        phase(A, B, C);
        if (q > kend) break;
        phase(B, C, A);
        if (q > kend) break;
        phase(C, A, B);
    }
}

// ---- the same fusion with TWO rows per thread ----------------------------------------------------
// jacobi_march2_kernel evaluates L1 on three rows to produce L2 on one: the two outer rows are what the
// neighbouring waves compute as their own centre.  Here a thread owns the float4 columns of rows j and j + 1: L0
// of the rows j-1 .. j+2 on three planes in registers, L1 on those four rows, L2 on the two inner ones -- four L1
// evaluations for two outputs instead of three for one, and 10 float4 loads (rows j-2 .. j+3 of p, j-1 .. j+2 of
// div) for two outputs instead of 8 for one.  Same expression per value, so bit-identical to the other sweep
// kernels.  One wave per row (nx <= 256).  Preconditions as for jacobi_march2_kernel.
template <int WAVES, bool WIDE>
__global__ __launch_bounds__(WAVES * 64) void jacobi_march2r_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                                    float *__restrict__ out, int nx, int ny, int nz,
                                                                    int cw, int nby, int kchunk, float alpha, float beta, Slab sl, PairRanges rg)
{
    const int nblk = gridDim.x;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);      // XCD-contiguous block order
    const int by = b % nby, bz = b / nby;
    const int rows = (WAVES * 64) / cw;
    const int c = threadIdx.x % cw, r = threadIdx.x / cw;
    const int xraw = 4 * c, j = 2 * (by * rows + r);
    const int kA = max(1, 1 - sl.koff), kB = min(nz - 1, sl.nkg - 1 - sl.koff);
    const int r0 = bz < rg.nchA ? rg.k0a + bz * kchunk : rg.k0b + (bz - rg.nchA) * kchunk, r1 = bz < rg.nchA ? rg.k1a : rg.k1b;
    const int kbeg = max(kA, r0), kend = min(min(kB, r1), r0 + kchunk);
    if (kbeg >= kend) return;
    const bool xok = xraw < nx;
    const bool active0 = xok && j >= 1 && j <= ny - 2, active1 = xok && j + 1 >= 1 && j + 1 <= ny - 2;
    // out-of-range rows, planes and lanes are clamped into the array (see jacobi_march2_kernel)
    const int x = xok ? xraw : nx - 4;
    const size_t sj = nx, sk = (size_t)nx * ny;
    const bool xlo = x == 0, xhi = x + 3 == nx - 1;
    auto rowat = [&](int col, int row) -> size_t { return (size_t)min(max(col, 0), nx - 1) + sj * (size_t)min(max(row, 0), ny - 1); };
    const size_t o_m2 = rowat(x, j - 2), o_p3 = rowat(x, j + 3);
    size_t o[4];
    bool rowb[4];
#pragma unroll
    for (int a = 0; a < 4; a++) { o[a] = rowat(x, j - 1 + a); rowb[a] = j - 1 + a <= 0 || j - 1 + a >= ny - 1; }
    // WIDE (a row spans several waves): the edge lanes look after the cell just outside their wave, on both rows --
    // the scheme of jacobi_march2_kernel, twice
    const int lane = threadIdx.x & 63;
    const bool edgeL = WIDE && lane == 0 && xok && xraw > 0, edgeR = WIDE && lane == 63 && xraw + 4 < nx;
    const bool edge = edgeL || edgeR;
    const int xe = edgeL ? xraw - 1 : xraw + 4, xo = edgeL ? xe - 1 : xe + 1;
    size_t e[4], eo[2];
#pragma unroll
    for (int a = 0; a < 4; a++) e[a] = rowat(xe, j - 1 + a);
    eo[0] = rowat(xo, j); eo[1] = rowat(xo, j + 1);
    const bool xe_boundary = xe <= 0 || xe >= nx - 1;
    auto plane = [&](int pl) -> size_t { return sk * (size_t)min(max(pl, 0), nz - 1); };
    auto ld4 = [&](const float *ptr, size_t off) -> float4 { return *reinterpret_cast<const float4 *>(ptr + off); };
    auto jac = [&](float4 ce, float4 fr, float4 bk, float4 dn, float4 up, float4 dv, float outside, bool boundary) -> float4 {
        float left = lane_up(ce.w), right = lane_down(ce.x);
        if (WIDE) {
            if (lane == 0) left = outside;
            if (lane == 63) right = outside;
        }
        float4 v;
        v.x = (left + ce.y + fr.x + bk.x + dn.x + up.x + alpha * dv.x) * beta;
        v.y = (ce.x + ce.z + fr.y + bk.y + dn.y + up.y + alpha * dv.y) * beta;
        v.z = (ce.y + ce.w + fr.z + bk.z + dn.z + up.z + alpha * dv.z) * beta;
        v.w = (ce.z + right + fr.w + bk.w + dn.w + up.w + alpha * dv.w) * beta;
        if (boundary) return ce;
        if (xlo) v.x = ce.x;
        if (xhi) v.w = ce.w;
        return v;
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 A[4], B[4], C[4], Dv[4], Hf, Hb;
    int q = kbeg - 1;
    // the outside cell: L0 on rows j-1 .. j+2 at plane q (Ec), on rows j, j+1 at planes q-1 / q+1 (Em / En), its outer
    // x-neighbour (Eo) and its div (Eb) on rows j, j+1 at plane q; Xp: its L1 on rows j, j+1 at plane q-1
    float Ec[4] = { 0.f, 0.f, 0.f, 0.f }, Em[2] = { 0.f, 0.f }, En[2] = { 0.f, 0.f }, Eo[2] = { 0.f, 0.f }, Eb[2] = { 0.f, 0.f }, Xp[2] = { 0.f, 0.f };
    {
        const size_t pm = plane(q - 1), pc = plane(q), pn = plane(q + 1);
#pragma unroll
        for (int a = 0; a < 4; a++) { A[a] = ld4(p, pm + o[a]); B[a] = ld4(p, pc + o[a]); C[a] = ld4(p, pn + o[a]); Dv[a] = ld4(div, pc + o[a]); }
        Hf = ld4(p, pc + o_m2); Hb = ld4(p, pc + o_p3);
        if (edge) {
#pragma unroll
            for (int a = 0; a < 4; a++) Ec[a] = p[pc + e[a]];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) { Em[rr] = p[pm + e[rr + 1]]; En[rr] = p[pn + e[rr + 1]]; Eo[rr] = p[pc + eo[rr]]; Eb[rr] = div[pc + e[rr + 1]]; }
        }
    }
    float4 Mc[4] = { zero4, zero4, zero4, zero4 };      // L1 on plane q-1, rows j-1 .. j+2
    float4 Mm[2] = { zero4, zero4 };                    // L1 of rows j, j+1 on plane q-2
    float4 Dprev[2] = { zero4, zero4 };                 // div of rows j, j+1 on plane q-1
    auto phase = [&](float4 (&Lm)[4], float4 (&Lc)[4], float4 (&Ln)[4]) {
        const size_t pa = plane(q + 2), pb = plane(q + 1);
        float4 La[4], Da[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { La[a] = ld4(p, pa + o[a]); Da[a] = ld4(div, pb + o[a]); }
        const float4 Hfa = ld4(p, pb + o_m2), Hba = ld4(p, pb + o_p3);
        float E2[4] = { 0.f, En[0], En[1], 0.f }, En2[2] = { 0.f, 0.f }, Eo2[2] = { 0.f, 0.f }, Eb2[2] = { 0.f, 0.f };
        if (edge) {
            E2[0] = p[pb + e[0]]; E2[3] = p[pb + e[3]];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) { En2[rr] = p[pa + e[rr + 1]]; Eo2[rr] = p[pb + eo[rr]]; Eb2[rr] = div[pb + e[rr + 1]]; }
        }
        const bool qb = q < kA || q >= kB;
        float4 M[4];
        M[0] = jac(Lc[0], Hf, Lc[1], Lm[0], Ln[0], Dv[0], Ec[0], qb || rowb[0]);
        M[1] = jac(Lc[1], Lc[0], Lc[2], Lm[1], Ln[1], Dv[1], Ec[1], qb || rowb[1]);
        M[2] = jac(Lc[2], Lc[1], Lc[3], Lm[2], Ln[2], Dv[2], Ec[2], qb || rowb[2]);
        M[3] = jac(Lc[3], Lc[2], Hb, Lm[3], Ln[3], Dv[3], Ec[3], qb || rowb[3]);
        // L1 of the outside cell on rows j, j+1 at plane q (edge lanes; a boundary cell keeps L0)
        float X[2] = { Ec[1], Ec[2] };
        if (edge) {
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                if (qb || rowb[rr + 1] || xe_boundary) continue;
                const float l = edgeL ? Eo[rr] : Lc[rr + 1].w, rg2 = edgeL ? Lc[rr + 1].x : Eo[rr];
                X[rr] = (l + rg2 + Ec[rr] + Ec[rr + 2] + Em[rr] + En[rr] + alpha * Eb[rr]) * beta;
            }
        }
        const int k = q - 1;
        const float4 o0 = jac(Mc[1], Mc[0], Mc[2], Mm[0], M[1], Dprev[0], Xp[0], false);
        const float4 o1 = jac(Mc[2], Mc[1], Mc[3], Mm[1], M[2], Dprev[1], Xp[1], false);
        if (k >= kbeg && k < kend) {
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                if (!(rr ? active1 : active0)) continue;
                const float4 v = rr ? o1 : o0;
                float *dst = out + (size_t)x + sj * (j + rr) + sk * k;
                if (x >= 4 && x + 4 < nx) {
                    *reinterpret_cast<float4 *>(dst) = v;
                } else {
                    if (x >= 1) dst[0] = v.x;
                    dst[1] = v.y;
                    dst[2] = v.z;
                    if (x + 3 < nx - 1) dst[3] = v.w;
                }
            }
        }
        Mm[0] = Mc[1]; Mm[1] = Mc[2];
        Dprev[0] = Dv[1]; Dprev[1] = Dv[2];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) { Xp[rr] = X[rr]; Em[rr] = Ec[rr + 1]; En[rr] = En2[rr]; Eo[rr] = Eo2[rr]; Eb[rr] = Eb2[rr]; }
#pragma unroll
        for (int a = 0; a < 4; a++) { Mc[a] = M[a]; Lm[a] = La[a]; Dv[a] = Da[a]; Ec[a] = E2[a]; }
        Hf = Hfa; Hb = Hba;
        q++;
    };
    while (q <= kend) {
        phase(A, B, C);
        if (q > kend) break;
        phase(B, C, A);
        if (q > kend) break;
        phase(C, A, B);
    }
}

// ---- the two-row fused kernel again, written for the instruction count --------------------------------------------
// jacobi_march2r_kernel is bound by instruction ISSUE, not by memory: one wave per SIMD issues one VALU instruction
// per ~4 cycles, and the compiler's rendering of that kernel spends ~680 instructions per plane (64-bit address
// arithmetic for every load, ~120 register moves to rotate planes and to pair operands, selects for boundaries that a
// wave almost never has).  Same algorithm, same expression per value, different plumbing:
//   * loads and stores through buffer descriptors: per-thread row offsets computed once (VGPR), the plane offset in an
//     SGPR -- no address arithmetic in the loop;
//   * every plane set lives in a ring of four buffers indexed by (plane mod 4); the loop is unrolled four times so that
//     all ring indices are compile-time: nothing is ever moved to rotate planes;
//   * float4 columns as two float2 halves (clang ext vectors -> v_pk_add_f32 / v_pk_mul_f32);
//   * boundary rows are a property of the BLOCK (template flag EDGE: only the first and last row blocks pay the
//     selects), boundary planes one wave-uniform branch;
//   * x-boundary columns are stored too: they hold L0's value (a sweep never changes them), which is what `out`
//     already holds there (the fused kernels' precondition: both buffers carry the same boundary layer).
// Rows of one wave (nx <= 256); wider rows keep jacobi_march2r_kernel<.., true>.
struct R4 { v2f a, b; };                                    // (x, y), (z, w) of one float4 column

__device__ __forceinline__ R4 ld_r4(v4i rs, unsigned voff, unsigned soff)
{
    const v4f v = bq_buffer_load_x4(rs, (int)voff, (int)soff, 0);
    return R4{v2f{v.x, v.y}, v2f{v.z, v.w}};
}
__device__ __forceinline__ float ld_f(v4i rs, unsigned voff, unsigned soff) { return bq_buffer_load_x1(rs, (int)voff, (int)soff, 0); }
// AUX: cache policy bits of the store (2 = nt: a streaming store, for arrays that will not be read again from the caches)
template <int AUX = 0>
__device__ __forceinline__ void st_r4(R4 v, v4i rs, unsigned voff, unsigned soff)
{
    bq_buffer_store_x4(v4f{v.a.x, v.a.y, v.b.x, v.b.y}, rs, (int)voff, (int)soff, AUX);
}

// jacobi_kernel's expression (GPU_kernel.cu:1833) on a float4 column: ((((((l + r) + f) + b) + d) + u) + alpha div) beta
// WIDE (rows of several waves): the x-neighbour of a wave's first / last lane lives in another wave; `outside` is its
// value, supplied by that lane itself (edgeL / edgeR mark the two lanes)
// PRE: dv already holds alpha * div (the product is formed once per loaded div value and reused by every level that needs
// it -- the same float either way)
// own + (value of `from` in lane - 1) / (lane + 1), the neighbour taken through the add's own DPP operand: one instruction
// where a DPP move, a move to pair the operands and a packed add stood.  A lane without a source lane (0 / 63) reads 0
// (bound_ctrl): those are x-boundary or out-of-range columns, whose result is replaced or never stored.  The s_nop covers the
// two wait states a DPP read needs after a VALU write of its source, which the compiler cannot see inside the asm.
// `volatile` (round 4, ADVICE): the result depends on EXEC and on the neighbouring lanes, which the compiler cannot see either --
// a pure two-operand asm could be sunk into a divergent region or merged across EXEC changes; volatile statements stay where
// the source puts them (every use is in wave-uniform code of the plane loop).  Same ISA at today's -O3, same timings.
__device__ __forceinline__ float add_from_left_lane(float from, float own)
{
    float r;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(from), "v"(own));
    return r;
}
__device__ __forceinline__ float add_from_right_lane(float from, float own)
{
    float r;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(from), "v"(own));
    return r;
}

// DPPADD (rows of one wave only): the l + r stage as four scalar adds, the two outer ones with a DPP operand
template <bool WIDE = false, bool PRE = false, bool DPPADD = false>
__device__ __forceinline__ R4 jac_r4(R4 ce, R4 fr, R4 bk, R4 dn, R4 up, R4 dv, float alpha, float beta, bool xlo, bool xhi,
                                     float outside = 0.f, bool edgeL = false, bool edgeR = false)
{
    if (DPPADD && !WIDE) {
        // (l + r) per cell: cell 0 = left + x1, 1 = x0 + x2, 2 = x1 + x3, 3 = x2 + right -- the same sums as below
        v2f s0 = v2f{add_from_left_lane(ce.b.y, ce.a.y), ce.a.x + ce.b.x};
        v2f s1 = v2f{ce.a.y + ce.b.y, add_from_right_lane(ce.a.x, ce.b.x)};
        s0 = s0 + fr.a; s1 = s1 + fr.b;
        s0 = s0 + bk.a; s1 = s1 + bk.b;
        s0 = s0 + dn.a; s1 = s1 + dn.b;
        s0 = s0 + up.a; s1 = s1 + up.b;
        if (PRE) { s0 = s0 + dv.a; s1 = s1 + dv.b; }
        else     { s0 = s0 + alpha * dv.a; s1 = s1 + alpha * dv.b; }
        s0 = s0 * beta; s1 = s1 * beta;
        if (xlo) s0.x = ce.a.x;
        if (xhi) s1.y = ce.b.y;
        return R4{s0, s1};
    }
    float left = lane_up(ce.b.y), right = lane_down(ce.a.x);
    if (WIDE) {
        if (edgeL) left = outside;
        if (edgeR) right = outside;
    }
    // (the compiler forms two v_pk_add_f32 here and assembles their operand pairs {left, x0}, {x1, x2}, {x3, right} with
    // six moves; four scalar adds with the neighbour lanes taken through the adds' own DPP operand (inline asm) are 4
    // instructions instead of 10, but measured no faster: 13.25 vs 13.23 us per sweep in the three-sweep kernel)
    v2f s0 = v2f{left + ce.a.y, ce.a.x + ce.b.x};
    v2f s1 = v2f{ce.a.y + ce.b.y, ce.b.x + right};
    s0 = s0 + fr.a; s1 = s1 + fr.b;
    s0 = s0 + bk.a; s1 = s1 + bk.b;
    s0 = s0 + dn.a; s1 = s1 + dn.b;
    s0 = s0 + up.a; s1 = s1 + up.b;
    if (PRE) { s0 = s0 + dv.a; s1 = s1 + dv.b; }
    else     { s0 = s0 + alpha * dv.a; s1 = s1 + alpha * dv.b; }
    s0 = s0 * beta; s1 = s1 * beta;
    if (xlo) s0.x = ce.a.x;
    if (xhi) s1.y = ce.b.y;
    return R4{s0, s1};
}

// PF: how many planes ahead the loads run (1 or 2; 3 and 4 measured no better anywhere, tools/batches/r02_m.sh).  The rings hold 3 + PF planes and the loop is unrolled 3 + PF times.
template <bool WIDE, int PF>
__global__ __launch_bounds__(256) void jacobi_lean2r_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                            float *__restrict__ out, int nx, int ny, int nz,
                                                            int cw, int nby, int kchunk, float alpha, float beta, Slab sl, PairRanges rg)
{
    constexpr int P = 3 + PF;                                       // ring period
    // PF = 2 is the form for arrays that come from HBM (larger than the Infinity Cache): loads two planes ahead and
    // streaming stores (512^3: 174.6 -> 162.8 us per sweep; in-cache sizes lose with them: 256^3 15.9 -> 20.3).
    constexpr bool HBM = PF >= 2;
    constexpr int ST = HBM ? 2 : 0;
    const int nblk = gridDim.x;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);      // XCD-contiguous block order
    const int by = b % nby, bz = b / nby;
    const int rows = 256 / cw;
    // a wave holds one row pair when rows are at least one wave long: the row index is then wave-uniform
    const int c = threadIdx.x % cw;
    const int r = cw >= 64 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / cw) : (int)threadIdx.x / cw;
    const int xraw = 4 * c, j = 2 * (by * rows + r);
    const int kA = max(1, 1 - sl.koff), kB = min(nz - 1, sl.nkg - 1 - sl.koff);
    const int r0 = bz < rg.nchA ? rg.k0a + bz * kchunk : rg.k0b + (bz - rg.nchA) * kchunk, r1 = bz < rg.nchA ? rg.k1a : rg.k1b;
    const int kbeg = max(kA, r0), kend = min(min(kB, r1), r0 + kchunk);
    if (kbeg >= kend) return;
    const bool xok = xraw < nx;
    const bool active0 = xok && j >= 1 && j <= ny - 2, active1 = xok && j + 1 >= 1 && j + 1 <= ny - 2;
    const int x = xok ? xraw : nx - 4;                              // out-of-range lanes, rows, planes: clamped into the array
    const bool xlo = x == 0, xhi = x + 3 == nx - 1;
    const unsigned bytes = (unsigned)nx * (unsigned)ny * (unsigned)nz * 4u;
    const v4i rp = make_rsrc4(p, bytes), rd = make_rsrc4(div, bytes), ro = make_rsrc4(out, bytes);
    unsigned vo[6];                                                 // byte offsets of this thread's column in rows j-2 .. j+3
    bool rowb[4];                                                   // rows j-1 .. j+2 are boundary rows (keep L0)
#pragma unroll
    for (int a = 0; a < 6; a++) vo[a] = ((unsigned)x + (unsigned)nx * (unsigned)min(max(j - 2 + a, 0), ny - 1)) * 4u;
#pragma unroll
    for (int a = 0; a < 4; a++) rowb[a] = j - 1 + a <= 0 || j - 1 + a >= ny - 1;
    // only the first and the last row block can hold a boundary row: everybody else runs the loop without the selects
    const bool edge_block = 2 * by * rows - 1 <= 0 || 2 * (by * rows + rows - 1) + 2 >= ny - 1;
    const unsigned pstride = (unsigned)nx * (unsigned)ny * 4u;
    auto po = [&](int pl) -> unsigned { return pstride * (unsigned)min(max(pl, 0), nz - 1); };
    // WIDE: the first / last lane of a wave looks after the column just outside its wave (xe) -- it needs that column's
    // L0 on rows j-1 .. j+2 (x-neighbour of our own first sweep) and its L1 on rows j, j+1 (x-neighbour of our second
    // sweep), which it evaluates itself from that column's own neighbours (outer x-neighbour xo, rows, planes, div).
    // The other 62 lanes run the same eight scalar loads per plane, but with an offset beyond the descriptor's range: the
    // range check answers 0 and nothing goes to the caches (512^3: 200 -> 175 us per sweep against loading their own
    // column; an exec-masked branch around the loads does the same at 512^3 but costs 7 % in-cache).
    const int lane = threadIdx.x & 63;
    const bool edgeL = WIDE && lane == 0 && xok && xraw > 0, edgeR = WIDE && lane == 63 && xraw + 4 < nx;
    const bool edge = edgeL || edgeR;
    const int xe = edgeL ? xraw - 1 : (edgeR ? xraw + 4 : x), xo = edgeL ? xe - 1 : (edgeR ? xe + 1 : x);
    const bool xe_boundary = xe <= 0 || xe >= nx - 1;
    unsigned ve[4], vx[2];
#pragma unroll
    for (int a = 0; a < 4; a++) ve[a] = ((unsigned)min(max(xe, 0), nx - 1) + (unsigned)nx * (unsigned)min(max(j - 1 + a, 0), ny - 1)) * 4u;
#pragma unroll
    for (int a = 0; a < 2; a++) vx[a] = ((unsigned)min(max(xo, 0), nx - 1) + (unsigned)nx * (unsigned)min(max(j + a, 0), ny - 1)) * 4u;
    if (WIDE && !edge) {
        // an offset beyond the descriptor's range: the load returns 0 without touching the caches (2 GiB + any plane
        // offset of an array below 2 GiB neither wraps nor lands inside it)
#pragma unroll
        for (int a = 0; a < 4; a++) ve[a] = 0x80000000u;
        vx[0] = vx[1] = 0x80000000u;
    }

    // Rings indexed by the plane's slot (t + d) mod P, t = q - (kbeg - 1) the iteration number, d the plane's distance
    // from q -- all compile-time inside the unrolled loop, so no value is ever moved to rotate planes:
    //   L0[.][0..3]  p on rows j-1 .. j+2            (live: planes q-1 .. q+PF; q+1+PF arriving)
    //   H[.][0..1]   p on rows j-2, j+3              (live: plane q .. q+PF-1; q+PF arriving)
    //   L1[.][0..3]  first sweep on rows j-1 .. j+2  (q being made; q-1; q-2 (rows j, j+1))
    //   D[.][0..3]   div on rows j-1 .. j+2          (q-1 (rows j, j+1) .. q+PF-1; q+PF arriving)
    //   WIDE: E[.][0..3] p(xe) on rows j-1 .. j+2 like L0; Eo / Eb p(xo) / div(xe) on rows j, j+1 like H; X the outside
    //   column's L1 on rows j, j+1 (q being made, q-1 live)
    auto run = [&](auto EDGE_T) {
    constexpr bool EDGE = decltype(EDGE_T)::value;
    R4 L0[P][4], H[P][2], L1[P][4], D[P][4];
    float E[P][4], Eo[P][2], Eb[P][2], X[P][2];
    const R4 zero = R4{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
#pragma unroll
    for (int a = 0; a < P; a++) {
#pragma unroll
        for (int bb = 0; bb < 4; bb++) L1[a][bb] = zero;
        X[a][0] = 0.f; X[a][1] = 0.f;
    }
    int q = kbeg - 1;
#pragma unroll
    for (int d = -1; d <= PF; d++) {                                // prologue: planes q-1 .. q+PF
        constexpr int dummy = 0; (void)dummy;
        const int sl_ = (d + P) % P;
        const unsigned pp = po(q + d);
#pragma unroll
        for (int a = 0; a < 4; a++) L0[sl_][a] = ld_r4(rp, vo[a + 1], pp);
        if (WIDE) {
#pragma unroll
            for (int a = 0; a < 4; a++) E[sl_][a] = ld_f(rp, ve[a], pp);
        }
        if (d >= 0 && d < PF) {
#pragma unroll
            for (int a = 0; a < 4; a++) D[sl_][a] = ld_r4(rd, vo[a + 1], pp);
            H[sl_][0] = ld_r4(rp, vo[0], pp); H[sl_][1] = ld_r4(rp, vo[5], pp);
            if (WIDE) {
#pragma unroll
                for (int a = 0; a < 2; a++) { Eo[sl_][a] = ld_f(rp, vx[a], pp); Eb[sl_][a] = ld_f(rd, ve[a + 1], pp); }
            }
        }
    }
#define BQ_SL(T, d) (((T) + (d) + P) % P)
#define BQ_LEAN_PHASE(T)                                                                                            \
    {                                                                                                               \
        constexpr int im = BQ_SL(T, -1), ic = BQ_SL(T, 0), in_ = BQ_SL(T, 1), ia = BQ_SL(T, 1 + PF), ha = BQ_SL(T, PF); \
        constexpr int mp = BQ_SL(T, -1), mpp = BQ_SL(T, -2);                                                        \
        const unsigned pa = po(q + 1 + PF), pb = po(q + PF);                                                        \
        _Pragma("unroll") for (int a = 0; a < 4; a++) { L0[ia][a] = ld_r4(rp, vo[a + 1], pa); D[ha][a] = ld_r4(rd, vo[a + 1], pb); } \
        H[ha][0] = ld_r4(rp, vo[0], pb); H[ha][1] = ld_r4(rp, vo[5], pb);                                           \
        if (WIDE) {                                                                                                 \
            _Pragma("unroll") for (int a = 0; a < 4; a++) E[ia][a] = ld_f(rp, ve[a], pa);                              \
            _Pragma("unroll") for (int a = 0; a < 2; a++) { Eo[ha][a] = ld_f(rp, vx[a], pb); Eb[ha][a] = ld_f(rd, ve[a + 1], pb); } \
        }                                                                                                           \
        const bool qb = q < kA || q >= kB;                                                                          \
        if (qb) {                                               /* a boundary plane keeps L0 */                      \
            _Pragma("unroll") for (int a = 0; a < 4; a++) L1[ic][a] = L0[ic][a];                                      \
        } else {                                                                                                    \
            /* alpha * div once per value: rows j, j+1 reuse the product in the second sweep */                      \
            _Pragma("unroll") for (int a = 0; a < 4; a++) { D[ic][a].a = alpha * D[ic][a].a; D[ic][a].b = alpha * D[ic][a].b; } \
            L1[ic][0] = jac_r4<WIDE, true>(L0[ic][0], H[ic][0], L0[ic][1], L0[im][0], L0[in_][0], D[ic][0], alpha, beta, xlo, xhi, E[ic][0], edgeL, edgeR);   \
            L1[ic][1] = jac_r4<WIDE, true>(L0[ic][1], L0[ic][0], L0[ic][2], L0[im][1], L0[in_][1], D[ic][1], alpha, beta, xlo, xhi, E[ic][1], edgeL, edgeR);  \
            L1[ic][2] = jac_r4<WIDE, true>(L0[ic][2], L0[ic][1], L0[ic][3], L0[im][2], L0[in_][2], D[ic][2], alpha, beta, xlo, xhi, E[ic][2], edgeL, edgeR);  \
            L1[ic][3] = jac_r4<WIDE, true>(L0[ic][3], L0[ic][2], H[ic][1], L0[im][3], L0[in_][3], D[ic][3], alpha, beta, xlo, xhi, E[ic][3], edgeL, edgeR);   \
            if (EDGE) {                                                                                             \
                _Pragma("unroll") for (int a = 0; a < 4; a++)                                                        \
                    if (rowb[a]) L1[ic][a] = L0[ic][a];                                                             \
            }                                                                                                       \
        }                                                                                                           \
        if (WIDE) {                                             /* the outside column's own first sweep, rows j, j+1 */ \
            _Pragma("unroll") for (int rr = 0; rr < 2; rr++) {                                                       \
                const float own = edgeL ? L0[ic][rr + 1].a.x : L0[ic][rr + 1].b.y;                                   \
                const float l = edgeL ? Eo[ic][rr] : own, rg2 = edgeL ? own : Eo[ic][rr];                           \
                const float v = (l + rg2 + E[ic][rr] + E[ic][rr + 2] + E[im][rr + 1] + E[in_][rr + 1] + alpha * Eb[ic][rr]) * beta; \
                const bool keep = qb || xe_boundary || (j + rr <= 0 || j + rr >= ny - 1);                           \
                X[ic][rr] = keep ? E[ic][rr + 1] : v;                                                               \
            }                                                                                                       \
        }                                                                                                           \
        const int k = q - 1;                                                                                        \
        if (k >= kbeg && k < kend) {                                                                                \
            const R4 o0 = jac_r4<WIDE, true>(L1[mp][1], L1[mp][0], L1[mp][2], L1[mpp][1], L1[ic][1], D[mp][1], alpha, beta, xlo, xhi, X[mp][0], edgeL, edgeR); \
            const R4 o1 = jac_r4<WIDE, true>(L1[mp][2], L1[mp][1], L1[mp][3], L1[mpp][2], L1[ic][2], D[mp][2], alpha, beta, xlo, xhi, X[mp][1], edgeL, edgeR); \
            const unsigned pk = pstride * (unsigned)k;                                                              \
            if (active0) st_r4<ST>(o0, ro, vo[2], pk);                                                              \
            if (active1) st_r4<ST>(o1, ro, vo[3], pk);                                                              \
        }                                                                                                           \
        q++;                                                                                                        \
    }
    while (true) {
        BQ_LEAN_PHASE(0)
        if (q > kend) break;
        BQ_LEAN_PHASE(1)
        if (q > kend) break;
        BQ_LEAN_PHASE(2)
        if (q > kend) break;
        BQ_LEAN_PHASE(3)
        if (q > kend) break;
        if constexpr (P > 4) {
            BQ_LEAN_PHASE(4)
            if (q > kend) break;
        }
        if constexpr (P > 5) {
            BQ_LEAN_PHASE(5)
            if (q > kend) break;
        }
        if constexpr (P > 6) {
            BQ_LEAN_PHASE(6)
            if (q > kend) break;
        }
    }
    };
    if (edge_block) run(std::true_type{}); else run(std::false_type{});
#undef BQ_LEAN_PHASE
}

// ---- THREE sweeps per launch (rows of one wave, whole array) ---------------------------------------------------------
// With the lean plumbing the two-sweep kernel is bound by the fabric again (6.9 TB/s of real traffic at 256^3); what is
// left is to move fewer bytes per sweep.  Same scheme, one level deeper: a thread owns the float4 columns of rows j, j+1
// and keeps, in rings of four planes with compile-time indices,
//     L0 on rows j-2 .. j+3 (+ rows j-3, j+4 of the centre plane),  L1 on rows j-2 .. j+3,  L2 on rows j-1 .. j+2,
//     div on rows j-2 .. j+3;
// per plane q it makes L1(q) on six rows, L2(q-1) on four, L3(q-2) on its two -- 12 evaluations and 14 float4 loads for
// two outputs of three sweeps (two sweeps: 6 and 10 for two).  ~400 registers: one wave per SIMD.  Every value is
// jacobi_kernel's expression, planes and rows outside the interior keep their input through all three levels, so the
// result is bit-identical to three single sweeps.  Preconditions as for the two-sweep kernels.
template <int PF>
__global__ __launch_bounds__(256) void jacobi_lean3r_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                            float *__restrict__ out, int nx, int ny, int nz,
                                                            int cw, int nby, int kchunk, float alpha, float beta, Slab sl)
{
    constexpr int P = 3 + PF;                                       // ring period; loads run PF planes ahead
    const int nblk = gridDim.x;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);      // XCD-contiguous block order
    const int by = b % nby, bz = b / nby;
    const int rows = 256 / cw;
    const int c = threadIdx.x % cw;
    const int r = cw >= 64 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / cw) : (int)threadIdx.x / cw;   // (see the two-sweep kernel)
    const int xraw = 4 * c, j = 2 * (by * rows + r);
    const int kA = max(1, 1 - sl.koff), kB = min(nz - 1, sl.nkg - 1 - sl.koff);
    const int kbeg = max(kA, bz * kchunk), kend = min(kB, bz * kchunk + kchunk);
    if (kbeg >= kend) return;
    const bool xok = xraw < nx;
    const bool active0 = xok && j >= 1 && j <= ny - 2, active1 = xok && j + 1 >= 1 && j + 1 <= ny - 2;
    const int x = xok ? xraw : nx - 4;
    const bool xlo = x == 0, xhi = x + 3 == nx - 1;
    const unsigned bytes = (unsigned)nx * (unsigned)ny * (unsigned)nz * 4u;
    const v4i rp = make_rsrc4(p, bytes), rd = make_rsrc4(div, bytes), ro = make_rsrc4(out, bytes);
    unsigned vo[8];                                                 // rows j-3 .. j+4
    bool rowb[6];                                                   // rows j-2 .. j+3 are boundary rows
#pragma unroll
    for (int a = 0; a < 8; a++) vo[a] = ((unsigned)x + (unsigned)nx * (unsigned)min(max(j - 3 + a, 0), ny - 1)) * 4u;
#pragma unroll
    for (int a = 0; a < 6; a++) rowb[a] = j - 2 + a <= 0 || j - 2 + a >= ny - 1;
    const bool edge_block = 2 * by * rows - 2 <= 0 || 2 * (by * rows + rows - 1) + 3 >= ny - 1;
    const unsigned pstride = (unsigned)nx * (unsigned)ny * 4u;
    auto po = [&](int pl) -> unsigned { return pstride * (unsigned)min(max(pl, 0), nz - 1); };
    auto run = [&](auto EDGE_T) {
    constexpr bool EDGE = decltype(EDGE_T)::value;
    // rings indexed by (iteration + distance) mod P, all compile-time inside the unrolled loop:
    //   L0: planes q-1, q, q+1 .. q+PF live, q+1+PF arriving;  H (rows j-3, j+4) and D: q .. q+PF-1 (D also q-1, q-2), q+PF arriving;
    //   L1: q (being made), q-1, q-2;  L2: q-1 (being made), q-2, q-3
    R4 L0[P][6], H[P][2], L1[P][6], L2[P][4], D[P][6];
    const R4 zero = R4{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
#pragma unroll
    for (int a = 0; a < P; a++) {
#pragma unroll
        for (int bb = 0; bb < 6; bb++) { L1[a][bb] = zero; D[a][bb] = zero; }
#pragma unroll
        for (int bb = 0; bb < 4; bb++) L2[a][bb] = zero;
    }
    int q = kbeg - 2;
#define BQ_SL3(T, d) ((((T) + (d)) % P + P) % P)
#pragma unroll
    for (int d = -1; d <= PF; d++) {                                // prologue: planes q-1 .. q+PF
        const int sl_ = BQ_SL3(0, d);
        const unsigned pp = po(q + d);
#pragma unroll
        for (int a = 0; a < 6; a++) L0[sl_][a] = ld_r4(rp, vo[a + 1], pp);
        if (d >= 0 && d < PF) {
#pragma unroll
            for (int a = 0; a < 6; a++) D[sl_][a] = ld_r4(rd, vo[a + 1], pp);
            H[sl_][0] = ld_r4(rp, vo[0], pp); H[sl_][1] = ld_r4(rp, vo[7], pp);
        }
    }
#define BQ_LEAN3_PHASE(T)                                                                                           \
    {                                                                                                               \
        constexpr int im = BQ_SL3(T, -1), ic = BQ_SL3(T, 0), in_ = BQ_SL3(T, 1), ia = BQ_SL3(T, 1 + PF);  /* L0 */    \
        constexpr int hc = BQ_SL3(T, 0), ha = BQ_SL3(T, PF);                                                        \
        constexpr int a0 = BQ_SL3(T, 0), a1 = BQ_SL3(T, -1), a2 = BQ_SL3(T, -2);             /* L1: q, q-1, q-2 */    \
        constexpr int b1 = BQ_SL3(T, -1), b2 = BQ_SL3(T, -2), b3 = BQ_SL3(T, -3);            /* L2: q-1, q-2, q-3 */  \
        constexpr int d0 = BQ_SL3(T, 0), d1 = BQ_SL3(T, -1), d2 = BQ_SL3(T, -2);                                    \
        const unsigned pa = po(q + 1 + PF), pb = po(q + PF);                                                        \
        _Pragma("unroll") for (int a = 0; a < 6; a++) { L0[ia][a] = ld_r4(rp, vo[a + 1], pa); D[ha][a] = ld_r4(rd, vo[a + 1], pb); } \
        H[ha][0] = ld_r4(rp, vo[0], pb); H[ha][1] = ld_r4(rp, vo[7], pb);                                           \
        if (q < kA || q >= kB) {                                /* first sweep on plane q, rows j-2 .. j+3 */        \
            _Pragma("unroll") for (int a = 0; a < 6; a++) L1[a0][a] = L0[ic][a];                                      \
        } else {                                                                                                    \
            /* alpha * div once per value: this plane's first sweep and the later sweeps on it reuse the product */   \
            _Pragma("unroll") for (int a = 0; a < 6; a++) { D[d0][a].a = alpha * D[d0][a].a; D[d0][a].b = alpha * D[d0][a].b; } \
            _Pragma("unroll") for (int a = 0; a < 6; a++) {                                                          \
                const R4 fr = a == 0 ? H[hc][0] : L0[ic][a == 0 ? 0 : a - 1], bk = a == 5 ? H[hc][1] : L0[ic][a == 5 ? 5 : a + 1]; \
                L1[a0][a] = jac_r4<false, true>(L0[ic][a], fr, bk, L0[im][a], L0[in_][a], D[d0][a], alpha, beta, xlo, xhi); \
                if (EDGE && rowb[a]) L1[a0][a] = L0[ic][a];                                                         \
            }                                                                                                       \
        }                                                                                                           \
        if (q - 1 < kA || q - 1 >= kB) {                        /* second sweep on plane q-1, rows j-1 .. j+2 */     \
            _Pragma("unroll") for (int a = 0; a < 4; a++) L2[b1][a] = L1[a1][a + 1];                                  \
        } else {                                                                                                    \
            _Pragma("unroll") for (int a = 0; a < 4; a++) {                                                          \
                L2[b1][a] = jac_r4<false, true>(L1[a1][a + 1], L1[a1][a], L1[a1][a + 2], L1[a2][a + 1], L1[a0][a + 1], D[d1][a + 1], alpha, beta, xlo, xhi); \
                if (EDGE && rowb[a + 1]) L2[b1][a] = L1[a1][a + 1];                                                 \
            }                                                                                                       \
        }                                                                                                           \
        const int k = q - 2;                                                                                        \
        if (k >= kbeg && k < kend) {                            /* third sweep on plane q-2, rows j, j+1: stored */  \
            const R4 o0 = jac_r4<false, true>(L2[b2][1], L2[b2][0], L2[b2][2], L2[b3][1], L2[b1][1], D[d2][2], alpha, beta, xlo, xhi); \
            const R4 o1 = jac_r4<false, true>(L2[b2][2], L2[b2][1], L2[b2][3], L2[b3][2], L2[b1][2], D[d2][3], alpha, beta, xlo, xhi); \
            const unsigned pk = pstride * (unsigned)k;                                                              \
            if (active0) st_r4(o0, ro, vo[3], pk);                                                                  \
            if (active1) st_r4(o1, ro, vo[4], pk);                                                                  \
        }                                                                                                           \
        q++;                                                                                                        \
    }
    while (true) {
        BQ_LEAN3_PHASE(0)
        if (q > kend + 1) break;
        BQ_LEAN3_PHASE(1)
        if (q > kend + 1) break;
        BQ_LEAN3_PHASE(2)
        if (q > kend + 1) break;
        BQ_LEAN3_PHASE(3)
        if (q > kend + 1) break;
        if constexpr (P > 4) {
            BQ_LEAN3_PHASE(4)
            if (q > kend + 1) break;
        }
    }
    };
    if (edge_block) run(std::true_type{}); else run(std::false_type{});
#undef BQ_LEAN3_PHASE
#undef BQ_SL3
}

// ---- THREE sweeps per launch, the intermediate levels' neighbour rows exchanged through LDS (round 3) ------------------
// jacobi_lean3r_kernel is bound by instruction ISSUE: a thread that owns rows j, j+1 evaluates the first sweep on six rows and
// the second on four to produce the third on two -- 12 evaluations for 6 useful ones -- and its ~400 registers leave one wave
// per SIMD, where an instruction costs ~5 cycles instead of ~2.5.  Here every wave evaluates each level on ITS OWN two rows
// only and fetches the two neighbouring rows of the level below from LDS, where the waves that own them have put them:
//   block = W + 2 waves: W row pairs that produce output + one HALO wave at either end, which evaluates the first two levels
//           on the two rows outside the block (the block's edge pairs need them) and stores nothing;
//   per plane step q, per wave: L1(q) on its rows -> LDS A[q & 1]; L2(q-1) on its rows from L1(q-1) (own rows: registers,
//           rows j-1, j+2: A[(q-1) & 1]) -> LDS B[(q-1) & 1]; L3(q-2) from L2(q-2) (own: registers, neighbours: B[q & 1]) ->
//           global; ONE barrier.  Each buffer written in step q was last read in step q-1, before that step's barrier.
// 6 evaluations and 6 float4 loads per plane and wave instead of 12 and 14, ~190 registers: two waves per SIMD.  Every value
// is jacobi_kernel's expression on the same operands, boundary rows / planes / columns keep their input through all three
// levels: bit-identical to three single sweeps (and to the other fused kernels).  Rows of one wave, whole arrays; same
// precondition as the other fused kernels (both buffers carry the same boundary layer).
// R: rows per wave (2: a wave owns a row pair and has one neighbour row of each level in its own registers; 1: one row per
// wave, both neighbour rows come from LDS -- twice the waves for the same work).  S: sweeps per launch (3 or 4): level s is
// needed S - s rows outside the block, so a block carries H = ceil((S - 1) / R) halo waves at either end; a chunk marches
// 2 (S - 1) warm-up planes.  All rings have period 4 (planes q-1, q, q+1 of the input live, q+2 arriving; three planes of every
// intermediate level; div of the S planes the levels are working on -- with S = 4 the slot of the oldest one is refilled at
// the END of the step, after the last level has used it).
// A halo wave whose nearest row lies dn rows outside the block owes the block levels 1 .. S - dn only: the march is instantiated
// once per level count and a block's waves run different code between the same barriers (S = 4 in row pairs: 11.4 -> 10.0 us
// per sweep at 256^3; S = 3 unchanged, 10.7 -- its waves wait for each other, not for the VALU).
template <int W, int R, int S>
__global__ __launch_bounds__((W + 2 * ((S - 1 + R - 1) / R)) * 64) void jacobi_lds_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                                  float *__restrict__ out, int nx, int ny, int nz,
                                                                  int nby, int nblk, int kchunk, float alpha, float beta, Slab sl, PairRanges rg)
{
    static_assert((R == 1 || R == 2) && (S == 3 || S == 4), "one or two rows per wave, three or four sweeps per launch");
    constexpr int H = (S - 1 + R - 1) / R, NW = W + 2 * H, NS = NW * R, P = 4, A = P - 2;   // A: the plane loaded in step q is q + A
    __shared__ v4f lds[S - 1][2][NS][64];                           // [level - 1][plane parity][row slot][lane]
    // XCD-contiguous block order: the grid is padded to a multiple of 8 blocks (nblk real ones), XCD x = blockIdx % 8 works
    // through its own run of consecutive (row block, chunk) pairs -- neighbouring row blocks share their halo rows in one L2
    const int per = (int)gridDim.x >> 3;
    const int b = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (b >= nblk) return;                                          // padding (block-uniform, before any barrier)
    const int by = b % nby, bz = b / nby;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int jb = by * (W * R);                                    // first output row of the block
    const int j = jb + R * (wv - H);                                // this wave's rows: j .. j + R - 1 (halo waves: outside the block)
    const bool halo = wv < H || wv >= NW - H;
    // a halo wave whose nearest row lies dn rows outside the block owes the block levels 1 .. S - dn only (wave-uniform)
    const int dn = wv < H ? R * (H - wv) - (R - 1) : (wv >= NW - H ? R * (wv - (NW - H)) + 1 : 0);
    const int smax = S - dn;
    const int kA = max(1, 1 - sl.koff), kB = min(nz - 1, sl.nkg - 1 - sl.koff);
    // output planes: chunk bz of plane range A or B (whole arrays: A = [0, nz), B empty)
    const int c0 = bz < rg.nchA ? rg.k0a + bz * kchunk : rg.k0b + (bz - rg.nchA) * kchunk, c1 = bz < rg.nchA ? rg.k1a : rg.k1b;
    const int kbeg = max(kA, c0), kend = min(min(kB, c1), c0 + kchunk);
    if (kbeg >= kend) return;                                       // (block-uniform: no barrier is skipped by part of a block)
    const int xraw = 4 * lane;
    const bool xok = xraw < nx;
    const int x = xok ? xraw : nx - 4;                              // out-of-range lanes, rows, planes: clamped into the array
    const bool xlo = x == 0, xhi = x + 3 == nx - 1;
    bool active[R], rowb[R];                                        // rowb: boundary (or outside) rows keep L0
#pragma unroll
    for (int a = 0; a < R; a++) {
        active[a] = xok && !halo && j + a >= 1 && j + a <= ny - 2;
        rowb[a] = j + a <= 0 || j + a >= ny - 1;
    }
    const unsigned bytes = (unsigned)nx * (unsigned)ny * (unsigned)nz * 4u;
    const v4i rp = make_rsrc4(p, bytes), rd = make_rsrc4(div, bytes), ro = make_rsrc4(out, bytes);
    unsigned vo[R + 2];                                             // byte offsets of this thread's column in rows j-1 .. j+R
#pragma unroll
    for (int a = 0; a < R + 2; a++) vo[a] = ((unsigned)x + (unsigned)nx * (unsigned)min(max(j - 1 + a, 0), ny - 1)) * 4u;
    // only blocks that touch the first or last row of the grid (their halo rows included) pay the boundary-row selects
    const bool edge_block = jb - H * R <= 0 || jb + W * R + H * R - 1 >= ny - 1;
    const unsigned pstride = (unsigned)nx * (unsigned)ny * 4u;
    auto po = [&](int pl) -> unsigned { return pstride * (unsigned)min(max(pl, 0), nz - 1); };
    // LDS rows: slot R wv + a holds row j + a; the neighbours j - 1 and j + R are the slots below and above (the outermost halo
    // row has no neighbour on its far side: it reads a clamped slot, and what it computes from that is never used)
    const int r0 = R * wv, rlo = max(r0 - 1, 0), rhi = min(r0 + R, NS - 1);
    auto put = [&](v4f (*buf)[64], int a, R4 v) { buf[r0 + a][lane] = v4f{v.a.x, v.a.y, v.b.x, v.b.y}; };
    auto get = [&](v4f (*buf)[64], int r) -> R4 { const v4f v = buf[r][lane]; return R4{v2f{v.x, v.y}, v2f{v.z, v.w}}; };

    auto run = [&](auto EDGE_T, auto SM_T) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(EDGE_T)::value;
    constexpr int SM = decltype(SM_T)::value;                       // the levels this wave evaluates (halo waves: fewer than S)
    R4 L0[P][R + 2], D[P][R], Lv[S][P][R];                          // Lv[s]: level s (1 .. S-1) on the wave's own rows
    const R4 zero = R4{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
#pragma unroll
    for (int a = 0; a < P; a++)
#pragma unroll
        for (int c = 0; c < R; c++) {
            D[a][c] = zero;
#pragma unroll
            for (int l = 0; l < S; l++) Lv[l][a][c] = zero;
        }
    int q = kbeg - (S - 1);
#define BQ_SL4(T, d) ((((T) + (d)) % P + P) % P)
#pragma unroll
    for (int d = -1; d <= A - 1; d++) {                             // prologue: planes q-1 .. q+A-1 of p; div of planes q .. q+A-2
        const int sl_ = BQ_SL4(0, d);
        const unsigned pp = po(q + d);
#pragma unroll
        for (int a = 0; a < R + 2; a++) L0[sl_][a] = ld_r4(rp, vo[a], pp);
        if (d >= 0 && d <= A - 2) {
#pragma unroll
            for (int a = 0; a < R; a++) D[sl_][a] = ld_r4(rd, vo[a + 1], pp);
        }
    }
#define BQ_LDS_PHASE(T)                                                                                             \
    {                                                                                                               \
        constexpr int im = BQ_SL4(T, -1), ic = BQ_SL4(T, 0), in_ = BQ_SL4(T, 1), ia = BQ_SL4(T, A), id_ = BQ_SL4(T, A - 1); \
        const unsigned pa = po(q + A), pb = po(q + A - 1);                                                          \
        _Pragma("unroll") for (int a = 0; a < R + 2; a++) L0[ia][a] = ld_r4(rp, vo[a], pa);                           \
        if (S == 3) { _Pragma("unroll") for (int a = 0; a < R; a++) D[id_][a] = ld_r4(rd, vo[a + 1], pb); }           \
        /* the neighbour rows every later level of this step needs were put into LDS before the last barrier: fetch them   \
           all now, ahead of the first level's arithmetic (the compiler cannot hoist them over this step's own puts) */     \
        R4 nlo[S + 1], nhi[S + 1];                                                                                  \
        _Pragma("unroll") for (int s = 2; s <= SM; s++) {                                                            \
            nlo[s] = get(lds[s - 2][(q - (s - 1)) & 1], rlo); nhi[s] = get(lds[s - 2][(q - (s - 1)) & 1], rhi);     \
        }                                                                                                           \
        /* first sweep on plane q, own rows */                                                                       \
        if (q < kA || q >= kB) {                                                                                    \
            _Pragma("unroll") for (int a = 0; a < R; a++) Lv[1][ic][a] = L0[ic][a + 1];                               \
        } else {                                                                                                    \
            _Pragma("unroll") for (int a = 0; a < R; a++) {                                                          \
                D[ic][a].a = alpha * D[ic][a].a; D[ic][a].b = alpha * D[ic][a].b;                                   \
                Lv[1][ic][a] = jac_r4<false, true, true>(L0[ic][a + 1], L0[ic][a], L0[ic][a + 2], L0[im][a + 1], L0[in_][a + 1], D[ic][a], alpha, beta, xlo, xhi); \
                if (EDGE && rowb[a]) Lv[1][ic][a] = L0[ic][a + 1];                                                  \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int a = 0; a < R; a++) put(lds[0][q & 1], a, Lv[1][ic][a]);                            \
        /* sweep s on plane q - (s - 1): rows j-1, j+R of level s-1 on that plane were put into LDS before the last barrier */ \
        _Pragma("unroll") for (int s = 2; s <= S; s++) {                                                             \
            const int ps = q - (s - 1);                                                                             \
            const int cs = BQ_SL4(T, -(s - 1)), us = BQ_SL4(T, -(s - 2)), ds = BQ_SL4(T, -s);                       \
            if (s <= SM && (s < S || (ps >= kbeg && ps < kend))) {   /* (a step without the two wave-uniform plane tests measured slower) */ \
                const bool keep = ps < kA || ps >= kB;                                                              \
                _Pragma("unroll") for (int a = 0; a < R; a++) {                                                      \
                    const R4 ce = Lv[s - 1][cs][a];                                                                 \
                    const R4 fr = a == 0 ? nlo[s] : Lv[s - 1][cs][a == 0 ? 0 : a - 1], bk = a == R - 1 ? nhi[s] : Lv[s - 1][cs][a == R - 1 ? a : a + 1]; \
                    R4 v = jac_r4<false, true, true>(ce, fr, bk, Lv[s - 1][ds][a], Lv[s - 1][us][a], D[cs][a], alpha, beta, xlo, xhi); \
                    if (keep || (EDGE && rowb[a])) v = ce;                                                          \
                    if (s < S) { Lv[s][cs][a] = v; put(lds[s - 1][ps & 1], a, v); }                                 \
                    else if (active[a]) st_r4(v, ro, vo[a + 1], pstride * (unsigned)ps);                            \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        if (S == 4) { _Pragma("unroll") for (int a = 0; a < R; a++) D[id_][a] = ld_r4(rd, vo[a + 1], pb); }           \
        __syncthreads();                                                                                            \
        q++;                                                                                                        \
    }
    while (true) {
        BQ_LDS_PHASE(0)
        if (q > kend + S - 2) break;
        BQ_LDS_PHASE(1)
        if (q > kend + S - 2) break;
        BQ_LDS_PHASE(2)
        if (q > kend + S - 2) break;
        BQ_LDS_PHASE(3)
        if (q > kend + S - 2) break;
    }
    };
    // one instance of the march per number of levels: the waves of a block run different code between the same barriers
    auto go = [&](auto E) __attribute__((always_inline)) {
        if (smax >= S) run(E, std::integral_constant<int, S>{});
        else if (smax == 1) run(E, std::integral_constant<int, 1>{});
        else if (smax == 2) run(E, std::integral_constant<int, 2>{});
        else run(E, std::integral_constant<int, 3>{});
    };
    if (edge_block) go(std::true_type{}); else go(std::false_type{});
#undef BQ_LDS_PHASE
#undef BQ_SL4
}

// ---- THREE sweeps per launch for rows of 260 .. 512 floats: two float4 segments per lane -----------------------------------
// jacobi_lds_kernel needs a row to fit one wave.  Here lane l holds the cells 4l .. 4l+3 (segment A) AND 256 + 4l .. 256 + 4l + 3
// (segment B) of its wave's row, so a 512-float row still belongs to ONE wave and every access is a coalesced 16-byte column;
// the x-neighbours across the seam travel through the adds' DPP operand with a wave rotation (lane 63 receives lane 0's cell
// 256, lane 0 lane 63's cell 255).  Twice the registers per row, so the neighbour rows of ALL levels come out of LDS -- the
// input too: a wave loads its own row only (the outermost halo waves also the one row beyond them) -- and twelve waves of 8
// output rows + two halo rows at either end fit three to a SIMD.  Same arithmetic, same boundary handling, same precondition
// as jacobi_lds_kernel; the fp64 smoother mg_lds3_kernel (bq_mgcg.hip) is this kernel's twin.
struct R8 { R4 a, b; };
__device__ __forceinline__ float add_rol(float from, float own)      // own + `from` of lane + 1 (lane 63: of lane 0)
{
    float r;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_rol:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(from), "v"(own));
    return r;
}
__device__ __forceinline__ float add_ror(float from, float own)      // own + `from` of lane - 1 (lane 0: of lane 63)
{
    float r;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(from), "v"(own));
    return r;
}
// jacobi_kernel's expression on the eight cells of a lane; adv = alpha * div
__device__ __forceinline__ R8 jac_r8(R8 ce, R8 fr, R8 bk, R8 dn, R8 up, R8 adv, float beta, bool xlo, bool xhi, bool first, bool last)
{
    const float uR = first ? ce.b.a.x : ce.a.a.x;                   // lane 63's right neighbour: cell 256, lane 0's segment B
    const float uL = last ? ce.a.b.y : ce.b.b.y;                    // lane 0's left neighbour in segment B: cell 255, lane 63's segment A
    v2f s0 = v2f{add_from_left_lane(ce.a.b.y, ce.a.a.y), ce.a.a.x + ce.a.b.x};
    v2f s1 = v2f{ce.a.a.y + ce.a.b.y, add_rol(uR, ce.a.b.x)};
    v2f t0 = v2f{add_ror(uL, ce.b.a.y), ce.b.a.x + ce.b.b.x};
    v2f t1 = v2f{ce.b.a.y + ce.b.b.y, add_from_right_lane(ce.b.a.x, ce.b.b.x)};
    s0 = s0 + fr.a.a; s1 = s1 + fr.a.b; t0 = t0 + fr.b.a; t1 = t1 + fr.b.b;
    s0 = s0 + bk.a.a; s1 = s1 + bk.a.b; t0 = t0 + bk.b.a; t1 = t1 + bk.b.b;
    s0 = s0 + dn.a.a; s1 = s1 + dn.a.b; t0 = t0 + dn.b.a; t1 = t1 + dn.b.b;
    s0 = s0 + up.a.a; s1 = s1 + up.a.b; t0 = t0 + up.b.a; t1 = t1 + up.b.b;
    s0 = s0 + adv.a.a; s1 = s1 + adv.a.b; t0 = t0 + adv.b.a; t1 = t1 + adv.b.b;
    s0 = s0 * beta; s1 = s1 * beta; t0 = t0 * beta; t1 = t1 * beta;
    if (xlo) s0.x = ce.a.a.x;
    if (xhi) t1.y = ce.b.b.y;
    return R8{R4{s0, s1}, R4{t0, t1}};
}

template <int W>
__global__ __launch_bounds__((W + 4) * 64) void jacobi_lds2seg_kernel(const float *__restrict__ p, const float *__restrict__ div,
                                                                      float *__restrict__ out, int nx, int ny, int nz,
                                                                      int nby, int nblk, int kchunk, float alpha, float beta, Slab sl, PairRanges rg)
{
    constexpr int S = 3, H = 2, NW = W + 2 * H, P = 4;
    __shared__ v4f lds[S][2][NW][2][64];                            // [level][plane parity][row slot][segment][lane]; level 0: the input
    const int per = (int)gridDim.x >> 3;                            // XCD-contiguous block order (grid padded to 8 k blocks)
    const int b = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (b >= nblk) return;
    const int by = b % nby, bz = b / nby;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int jb = by * W;
    const int j = jb + (wv - H);                                    // this wave's row (halo waves: outside the block)
    const bool halo = wv < H || wv >= NW - H;
    const int dn_rows = wv < H ? H - wv : (wv >= NW - H ? wv - (NW - H) + 1 : 0);
    const int smax = S - dn_rows;                                   // a halo wave dn rows outside owes levels 1 .. S - dn
    const bool low_end = wv == 0;
    const int kA = max(1, 1 - sl.koff), kB = min(nz - 1, sl.nkg - 1 - sl.koff);
    const int r0 = bz < rg.nchA ? rg.k0a + bz * kchunk : rg.k0b + (bz - rg.nchA) * kchunk, r1 = bz < rg.nchA ? rg.k1a : rg.k1b;
    const int kbeg = max(kA, r0), kend = min(min(kB, r1), r0 + kchunk);
    if (kbeg >= kend) return;                                       // (block-uniform)
    const int xA = 4 * lane, xBraw = 256 + 4 * lane;
    const bool okB = xBraw < nx;
    const int xB = okB ? xBraw : nx - 4;                            // out-of-range lanes, rows, planes: clamped into the array
    const bool xlo = lane == 0, xhi = okB && xBraw + 3 == nx - 1;
    const bool first = lane == 0, last = lane == 63;
    const bool row_in = !halo && j >= 1 && j <= ny - 2;
    const bool rowb = j <= 0 || j >= ny - 1;
    const unsigned bytes = (unsigned)nx * (unsigned)ny * (unsigned)nz * 4u;
    const v4i rp = make_rsrc4(p, bytes), rd = make_rsrc4(div, bytes), ro = make_rsrc4(out, bytes);
    const unsigned row_own = (unsigned)nx * (unsigned)min(max(j, 0), ny - 1);
    const unsigned row_far = (unsigned)nx * (unsigned)min(max(low_end ? j - 1 : j + 1, 0), ny - 1);
    const unsigned voA = ((unsigned)xA + row_own) * 4u, voB = ((unsigned)xB + row_own) * 4u;
    const unsigned vfA = ((unsigned)xA + row_far) * 4u, vfB = ((unsigned)xB + row_far) * 4u;
    const bool edge_block = jb - H <= 0 || jb + W + H - 1 >= ny - 1;
    const unsigned pstride = (unsigned)nx * (unsigned)ny * 4u;
    auto po = [&](int pl) -> unsigned { return pstride * (unsigned)min(max(pl, 0), nz - 1); };
    const int rlo = max(wv - 1, 0), rhi = min(wv + 1, NW - 1);
    auto put = [&](v4f (*buf)[2][64], R8 v) {
        buf[wv][0][lane] = v4f{v.a.a.x, v.a.a.y, v.a.b.x, v.a.b.y}; buf[wv][1][lane] = v4f{v.b.a.x, v.b.a.y, v.b.b.x, v.b.b.y};
    };
    auto get = [&](v4f (*buf)[2][64], int r) -> R8 {
        const v4f u = buf[r][0][lane], w = buf[r][1][lane];
        return R8{R4{v2f{u.x, u.y}, v2f{u.z, u.w}}, R4{v2f{w.x, w.y}, v2f{w.z, w.w}}};
    };
    const R4 z4 = R4{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    const R8 zero = R8{z4, z4};
    auto ld_own = [&](v4i rs, unsigned pp) -> R8 { return R8{ld_r4(rs, voA, pp), ld_r4(rs, voB, pp)}; };
    auto ld_far = [&](unsigned pp) -> R8 { return R8{ld_r4(rp, vfA, pp), ld_r4(rp, vfB, pp)}; };

    auto run = [&](auto EDGE_T, auto SM_T) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(EDGE_T)::value;
    constexpr int SM = decltype(SM_T)::value;                       // the levels this wave evaluates
    constexpr bool OUTER = SM == 1;                                 // an outermost halo wave
    R8 L0[P], Lx[P], D[P], L1[P], L2[P];
#pragma unroll
    for (int a = 0; a < P; a++) { D[a] = zero; L1[a] = zero; L2[a] = zero; Lx[a] = zero; }
    int q = kbeg - (S - 1);
#define BQ_SL4(T, d) ((((T) + (d)) % P + P) % P)
#pragma unroll
    for (int d = -1; d <= 1; d++) {                                 // prologue: planes q-1, q, q+1 of p; div of plane q
        const int sl_ = BQ_SL4(0, d);
        const unsigned pp = po(q + d);
        L0[sl_] = ld_own(rp, pp);
        if (OUTER && d >= 0) Lx[sl_] = ld_far(pp);
        if (d == 0) D[sl_] = ld_own(rd, pp);
    }
    put(lds[0][q & 1], L0[BQ_SL4(0, 0)]);                           // the first step's centre plane for the neighbours
    __syncthreads();
#define BQ_L2S_PHASE(T)                                                                                             \
    {                                                                                                               \
        constexpr int im = BQ_SL4(T, -1), ic = BQ_SL4(T, 0), in_ = BQ_SL4(T, 1), ia = BQ_SL4(T, 2);                   \
        const unsigned pa = po(q + 2), pb = po(q + 1);                                                              \
        L0[ia] = ld_own(rp, pa);                                                                                    \
        if (OUTER) Lx[ia] = ld_far(pa);                                                                             \
        D[in_] = ld_own(rd, pb);                                                                                    \
        R8 fr0 = zero, bk0 = zero, nlo2 = zero, nhi2 = zero, nlo3 = zero, nhi3 = zero;                              \
        if (!OUTER || !low_end) fr0 = get(lds[0][q & 1], rlo);                                                      \
        if (!OUTER || low_end) bk0 = get(lds[0][q & 1], rhi);                                                       \
        if (OUTER) { if (low_end) fr0 = Lx[ic]; else bk0 = Lx[ic]; }                                                \
        put(lds[0][(q + 1) & 1], L0[in_]);                      /* the next step's centre plane */                   \
        /* first sweep on plane q */                                                                                \
        if (q < kA || q >= kB) {                                                                                    \
            L1[ic] = L0[ic];                                                                                        \
        } else {                                                                                                    \
            D[ic].a.a = alpha * D[ic].a.a; D[ic].a.b = alpha * D[ic].a.b;                                           \
            D[ic].b.a = alpha * D[ic].b.a; D[ic].b.b = alpha * D[ic].b.b;                                           \
            L1[ic] = jac_r8(L0[ic], fr0, bk0, L0[im], L0[in_], D[ic], beta, xlo, xhi, first, last);                 \
            if (EDGE && rowb) L1[ic] = L0[ic];                                                                      \
        }                                                                                                           \
        /* (the later levels' neighbour rows are fetched level by level: six rows in flight at once cost too many registers) */ \
        if (SM >= 2) { nlo2 = get(lds[1][(q - 1) & 1], rlo); nhi2 = get(lds[1][(q - 1) & 1], rhi); }                \
        put(lds[1][q & 1], L1[ic]);                                                                                 \
        if (SM >= 2) {      /* second sweep on plane q - 1 */                                                       \
            constexpr int cs = BQ_SL4(T, -1), us = BQ_SL4(T, 0), ds = BQ_SL4(T, -2);                                 \
            const int ps = q - 1;                                                                                   \
            R8 v = jac_r8(L1[cs], nlo2, nhi2, L1[ds], L1[us], D[cs], beta, xlo, xhi, first, last);                  \
            if (ps < kA || ps >= kB || (EDGE && rowb)) v = L1[cs];                                                  \
            L2[cs] = v;                                                                                             \
            if (SM >= 3) { nlo3 = get(lds[2][(q - 2) & 1], rlo); nhi3 = get(lds[2][(q - 2) & 1], rhi); }            \
            put(lds[2][ps & 1], v);                                                                                 \
        }                                                                                                           \
        if (SM >= 3) {      /* third sweep on plane q - 2 */                                                        \
            constexpr int cs = BQ_SL4(T, -2), us = BQ_SL4(T, -1), ds = BQ_SL4(T, -3);                                \
            const int ps = q - 2;                                                                                   \
            if (ps >= kbeg && ps < kend) {                                                                          \
                R8 v = jac_r8(L2[cs], nlo3, nhi3, L2[ds], L2[us], D[cs], beta, xlo, xhi, first, last);              \
                if (EDGE && rowb) v = L2[cs];                                                                       \
                if (row_in) {                                                                                       \
                    const unsigned pk = pstride * (unsigned)ps;                                                     \
                    st_r4<2>(v.a, ro, voA, pk);                                                                     \
                    if (okB) st_r4<2>(v.b, ro, voB, pk);                                                            \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        __syncthreads();                                                                                            \
        q++;                                                                                                        \
    }
    while (true) {
        BQ_L2S_PHASE(0)
        if (q > kend + S - 2) break;
        BQ_L2S_PHASE(1)
        if (q > kend + S - 2) break;
        BQ_L2S_PHASE(2)
        if (q > kend + S - 2) break;
        BQ_L2S_PHASE(3)
        if (q > kend + S - 2) break;
    }
    };
    auto go = [&](auto E) __attribute__((always_inline)) {
        if (smax >= S) run(E, std::integral_constant<int, S>{});
        else if (smax == 1) run(E, std::integral_constant<int, 1>{});
        else run(E, std::integral_constant<int, 2>{});
    };
    if (edge_block) go(std::true_type{}); else go(std::false_type{});
#undef BQ_L2S_PHASE
#undef BQ_SL4
}

// ---- residual norms (A15 re-specified): r = div - (sum6 p - 6p), sum r^2 and max|r| --------
// update_residual_kernel / calc_poisson_value arithmetic (GPU_kernel.cu:1048-1060,1239-1249);
// the reduction is ours: wave64 shuffles -> one partial per block -> fixed-order final pass.
__global__ __launch_bounds__(256) void residual_partial_kernel(const float *__restrict__ div, const float *__restrict__ p,
                                                               int ni, int nj, int nk,
                                                               double *__restrict__ part_sum, float *__restrict__ part_max,
                                                               Slab sl, int own0, int own1)
{
    const size_t sj = ni, sk = (size_t)ni * nj;
    const size_t total = sk * nk;
    double s = 0.0;
    float mx = 0.f;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const int i = (int)(id % sj), j = (int)((id / sj) % nj), k = (int)(id / sk);
        const int kg = k + sl.koff;
        if (i > 0 && i < ni - 1 && j > 0 && j < nj - 1 && kg > 0 && kg < sl.nkg - 1 && kg >= own0 && kg < own1 && k > 0 && k < nk - 1) {
            float ax = (p[id - 1] + p[id + 1] + p[id - sj] + p[id + sj] + p[id - sk] + p[id + sk]) - p[id] * 6;
            float r = div[id] - ax;
            s += (double)r * (double)r;
            mx = fmaxf(mx, fabsf(r));
        }
    }
    __shared__ double ssum[4];
    __shared__ float smax[4];
    s = wave_sum(s);
    mx = wave_max(mx);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { ssum[wave] = s; smax[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = ((ssum[0] + ssum[1]) + ssum[2]) + ssum[3];
        part_max[blockIdx.x] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

__global__ __launch_bounds__(256) void residual_final_kernel(const double *__restrict__ part_sum, const float *__restrict__ part_max,
                                                             int nparts, double *out_sum, float *out_max,
                                                             float *dbg_sum, float *dbg_max)
{
    double s = 0.0;
    float mx = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) { s += part_sum[i]; mx = fmaxf(mx, part_max[i]); }
    __shared__ double ssum[4];
    __shared__ float smax[4];
    s = wave_sum(s);
    mx = wave_max(mx);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { ssum[wave] = s; smax[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = ((ssum[0] + ssum[1]) + ssum[2]) + ssum[3];
        float m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        if (out_sum) *out_sum = tot;
        if (out_max) *out_max = m;
        if (dbg_sum) *dbg_sum = (float)tot;
        if (dbg_max) *dbg_max = m;
    }
}

// ---- diffuse_field_kernel (GPU_kernel.cu:834-853) ------------------------------------------
__global__ __launch_bounds__(256) void diffuse_kernel(const float *__restrict__ field, const float *__restrict__ in,
                                                      float *__restrict__ out, int ni, int nj, int nk, float coef,
                                                      int koff, int nkg)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    const int kg = k + koff;                    // nkg: GLOBAL plane count of this buffer
    if (!(i > 0 && i < ni - 1 && j > 0 && j < nj - 1 && k > 0 && k < nk - 1 && kg > 0 && kg < nkg - 1)) return;
    const size_t sj = ni, sk = (size_t)ni * nj;
    const size_t id = (size_t)i + sj * j + sk * k;
    float s = in[id - 1] + in[id + 1] + in[id - sj] + in[id + sj] + in[id - sk] + in[id + sk];
    out[id] = (field[id] + coef * s) / (1.0f + 6.0f * coef);
}

static const dim3 kBlock2(64, 4, 1);
static inline dim3 grid2(int a, int b, int c) { return dim3((a + 63) / 64, (b + 3) / 4, c); }

static bool dims_ok(int ni, int nj, int nk, const char *op)
{
    if (ni < 1 || nj < 1 || nk < 1) { latch(FL_ERR_BAD_ARGUMENT, op, "non-positive grid dims"); return false; }
    double bytes = 4.0 * (double)(ni + 1) * (double)(nj + 1) * (double)(nk + 1);
    if (bytes >= 4294967296.0) { latch(FL_ERR_BAD_ARGUMENT, op, "field larger than 4 GiB"); return false; }
    if (nk + 1 > 65535) { latch(FL_ERR_BAD_ARGUMENT, op, "nk too large for grid.z"); return false; }
    return true;
}

// per-context state of this file (bq_host.h: Runtime::project_state), behind the names it always had
struct SweepSpan { hipEvent_t a, b; long long launches, sweeps; };
struct ProjectState {
    int klo = 0, khi = 1 << 30;             // plane range of the next sweep launches (gpu_jacobi_sweep_range)
    const char *last_pair_kernel = "";      // name of the fused sweep kernel launched last (fl_jacobi_kernel_name)
    std::vector<SweepSpan> spans;           // FL_OPT_PROFILE_JACOBI
};
static ProjectState &ps()
{
    Runtime &r = rt();
    if (!r.project_state) r.project_state = new ProjectState();
    return *static_cast<ProjectState *>(r.project_state);
}
#define g_klo (ps().klo)
#define g_khi (ps().khi)
#define g_last_pair_kernel (ps().last_pair_kernel)
#define g_spans (ps().spans)
void project_release_state(Runtime &r)
{
    ProjectState *st = static_cast<ProjectState *>(r.project_state);
    if (!st) return;
    for (const SweepSpan &sp : st->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    delete st;
    r.project_state = nullptr;
}

static inline Slab slab_of(int nk)
{
    const Runtime &r = rt();
    if (r.slab_on) return Slab{r.slab_koff, r.slab_nkg, g_klo, g_khi};
    return Slab{0, nk, g_klo, g_khi};
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// One Jacobi sweep in -> out on the compute stream.
static void jacobi_sweep(const float *in, const float *div, float *out, int ni, int nj, int nk, float alpha, float beta)
{
    if (ni < 3 || nj < 3 || nk < 3) return;         // no interior
    int variant = rt().opt_jacobi_variant;
    const bool tile_ok = (ni % 4 == 0) && ni >= 32 && aligned16(in) && aligned16(div) && aligned16(out);
    if (variant == 0) variant = tile_ok ? 3 : 1;
    if (variant != 1 && !tile_ok) variant = 1;
    hipStream_t st = rt().compute;
    if (variant == 1) {
        jacobi_generic_kernel<<<grid2(ni, nj, nk), kBlock2, 0, st>>>(in, div, out, ni, nj, nk, alpha, beta, slab_of(nk));
        BQ_LAUNCH_CHECK("jacobi_generic_kernel");
        return;
    }
    if (variant == 3) {
        int waves = rt().opt_jacobi_rows;
        if (waves != 4 && waves != 8 && waves != 16) waves = 4;
        const int threads = waves * 64;
        int cw = 16;                                     // float4 columns per tile row: pow2 >= ni/4
        while (cw * 4 < ni && cw < threads) cw *= 2;
        const int rows = threads / cw;
        const int nbx = (ni / 4 + cw - 1) / cw, nby = (nj + rows - 1) / rows;
        // k-chunk: measured optimum at 256^3 is 16 planes (tools/jacobi_tune.py: 4/8/16/32 planes ->
        // 34.2/34.5/31.8/34.0 us); shorter chunks re-read more planes, longer ones leave CUs idle.
        // Keep >= ~1024 blocks when the grid is small in x/y.
        int kchunk = 16;
        while (kchunk > 4 && (long)nbx * nby * ((nk + kchunk - 1) / kchunk) < 1024) kchunk /= 2;
        if (rt().opt_jacobi_kchunk > 0) kchunk = rt().opt_jacobi_kchunk;
        const int nbz = (nk + kchunk - 1) / kchunk;
        const int nblk = nbx * nby * nbz;
#define BQ_JM(W) jacobi_march_kernel<W><<<nblk, W * 64, 0, st>>>(in, div, out, ni, nj, nk, cw, nbx, nby, kchunk, alpha, beta, slab_of(nk))
        if (waves == 4) BQ_JM(4); else if (waves == 8) BQ_JM(8); else BQ_JM(16);
#undef BQ_JM
        BQ_LAUNCH_CHECK("jacobi_march_kernel");
        return;
    }
    // tile geometry: 256-wide rows when the row is long enough, else 128-wide; R float4 per thread
    const bool wide = ni > 128;
    int R = rt().opt_jacobi_rows;
    if (R != 1 && R != 2 && R != 4) R = 4;
    const int TX = wide ? 256 : 128, TY = (wide ? 4 : 8) * R;
    const int bx = (ni + TX - 1) / TX, by = (nj + TY - 1) / TY;
    // k-chunks: enough blocks to fill 256 CUs x 2 resident blocks, at least 8 planes per chunk
    int want = (1024 + bx * by - 1) / (bx * by);
    int kchunk = (nk + want - 1) / want;
    if (kchunk < 8) kchunk = 8;
    if (rt().opt_jacobi_kchunk > 0) kchunk = rt().opt_jacobi_kchunk;
    const int bz = (nk + kchunk - 1) / kchunk;
    dim3 grid(bx, by, bz);
#define BQ_JT(TXV, RR) jacobi_tile_kernel<TXV, RR><<<grid, 256, 0, st>>>(in, div, out, ni, nj, nk, kchunk, alpha, beta, slab_of(nk))
    if (wide) { if (R == 4) BQ_JT(64, 4); else if (R == 2) BQ_JT(64, 2); else BQ_JT(64, 1); }
    else      { if (R == 4) BQ_JT(32, 4); else if (R == 2) BQ_JT(32, 2); else BQ_JT(32, 1); }
#undef BQ_JT
    BQ_LAUNCH_CHECK("jacobi_tile_kernel");
}

// How many k-chunks a fused launch cuts `nkr` planes into when the compute stream does not own the whole chip
// (FL_OPT_RESERVE_CUS): the rules below fill 256 CUs in whole rounds; with another CU count no chunk count divides evenly, so
// take the one that minimises rounds x planes marched per block (chunk + `warm` warm-up planes), nearest to the target
// chunk length among near-equal candidates.  per_cu: resident blocks per CU.
static int chunks_for_cus(int nkr, int nrow, int target, int warm, int ncus, int per_cu)
{
    double best = 1e30; int best_n = 1, best_gap = 1 << 30;
    for (int n = 1; n <= std::max(1, nkr / 4); n++) {
        const int kc = (nkr + n - 1) / n;
        const long blocks = (long)nrow * ((nkr + kc - 1) / kc);
        const long rounds = (blocks + (long)ncus * per_cu - 1) / ((long)ncus * per_cu);
        const double cost = (double)rounds * (kc + warm);
        const int gap = std::abs(kc - target);
        if (cost < best * 0.97 || (cost <= best * 1.03 && gap < best_gap)) { best = std::min(best, cost); best_n = n; best_gap = gap; }
    }
    return best_n;
}


// Two sweeps in one launch (in -> out holds iterate +2) when the fused kernel applies; returns false
// (nothing launched) otherwise.  The caller guarantees that both buffers carry the same boundary layer.
// k0a..k1b: output plane ranges (see PairRanges); the default is the whole array
static bool jacobi_sweep_pair(const float *in, const float *div, float *out, int ni, int nj, int nk, float alpha, float beta,
                              int k0a = 0, int k1a = 1 << 30, int k0b = 0, int k1b = 0)
{
    if (ni < 3 || nj < 3 || nk < 3) return false;
    const int variant = rt().opt_jacobi_variant;
    if (variant != 0 && variant != 3) return false;
    if (!((ni % 4 == 0) && ni >= 32 && ni <= 1024 && aligned16(in) && aligned16(div) && aligned16(out))) return false;
    k0a = std::max(k0a, 0); k1a = std::min(k1a, nk); k0b = std::max(k0b, 0); k1b = std::min(k1b, nk);
    const int lenA = std::max(k1a - k0a, 0), lenB = std::max(k1b - k0b, 0);
    if (lenA + lenB == 0) return true;
    const bool whole = lenB == 0 && lenA == nk;
    const int nkr = lenA + lenB;                         // planes this launch produces
    auto chunks_of = [](int len, int kc) { return len > 0 ? (len + kc - 1) / kc : 0; };
    int cw = 16;
    while (cw * 4 < ni) cw *= 2;                         // float4 lanes per row: <= 64 one wave, 128/256 = 2/4 waves
    const bool wide = cw > 64;
    const int rows = 256 / cw;
    const int nby = (nj + rows - 1) / rows;
    // Two rows per thread (jacobi_march2r_kernel; rows of one wave only).  It has half as many row blocks, runs one
    // 4-wave block per CU best, and like the one-row kernel only pays when the blocks fill the 256 CUs in whole
    // rounds: 256^3 17.1 us per sweep with 8 chunks of 32 planes (256 blocks) against 19.0-19.7 for the one-row
    // kernel, but 21-23 us with chunks of 24-28 and 28 us with chunks of 64; 272 planes 18.0 (8 chunks of 34) against
    // 20.0; 128^3 is slower with it (5.5 vs 4.6: the chunks get too short).  FL_OPT_JACOBI_ROWS: 0 = this rule,
    // 1 = one row, 2 = two rows whenever the kernel applies.  Short plane ranges (the parts of a split launch next to
    // the ghost planes): one chunk per range.
    if (nj >= 4 && rt().opt_jacobi_rows != 1) {
        const int nby2 = (nj + 2 * rows - 1) / (2 * rows);
        int gcd = nby2, rem = 256;
        while (rem) { const int t = gcd % rem; gcd = rem; rem = t; }
        const int quantum = 256 / gcd;                              // chunk counts that make nby2 * nbz a multiple of 256
        // ~32 planes per chunk for rows of one wave; rows of 2-4 waves (WIDE) like longer marches: 512^3 runs 201 us per
        // sweep with 6 chunks of 86 planes, 203-214 with 8 of 64, 226 with 48, 211 with 128 (one-row kernel: 228-238)
        const int target2 = wide ? 80 : 32;
        int nchunks = ((2 * nkr + target2) / (2 * target2) + quantum / 2) / quantum * quantum;
        if (nchunks < quantum) nchunks = quantum;
        const int ncus = rt().num_cus;
        if (ncus != 256) nchunks = chunks_for_cus(nkr, nby2, target2, 2, ncus, 1);
        int kc = (nkr + nchunks - 1) / nchunks;
        // Grids too small to give every CU a chunk of 16 planes (128^3: 8 row blocks): the two-row kernel still wins with the
        // short chunks that fill the chip exactly once -- 128^3: 32 chunks of 4 planes = 256 blocks, 2.99 us per sweep against
        // 4.41 for the one-row kernel with chunks of 8 (chunks of 2 / 3 / 5 planes: 3.34 / 3.73 / 3.24; the three-sweep kernel
        // with chunks of 4: 3.21; gpurun_out/r03f/jacobi_tune_128.txt) -- a march this short is bound by the latency of its
        // 6 plane steps at one wave per SIMD, so what counts is that no CU waits for a second round.
        // Smaller still (64^3 2.21 against 4.11 us per sweep, 96^3 3.15 / 4.28, 160^3 7.34 / 9.85, 192^3 8.30 / 12.95,
        // 256 x 256 x 64 5.49 / 6.74; gpurun_out/r03h/jacobi_small.txt): whole arrays always take the two-row kernel,
        // chunks down to two planes.
        bool pays = kc >= 16 || whole;
        if (!whole && std::max(lenA, lenB) <= 48) {
            // short ranges (the ends of a split launch): as many chunks as fill the 256 CUs once -- a block marches its
            // chunk plus two warm-up planes, so 2 ranges x 32 row blocks x 4 chunks of 3 planes beat 2 x 32 x 1 of 10
            const int nranges = (lenA > 0) + (lenB > 0);
            const int per_range = std::max(1, ncus / std::max(1, nby2 * nranges));
            kc = std::max(2, (std::max(lenA, lenB) + per_range - 1) / per_range);
            pays = true;
        }
        if (rt().opt_jacobi_kchunk2 > 0) kc = rt().opt_jacobi_kchunk2;
        if (kc < 2) kc = 2;
        if (pays || rt().opt_jacobi_rows == 2) {
            const PairRanges rg{k0a, k1a, k0b, k1b, chunks_of(lenA, kc)};
            const int nbz2 = rg.nchA + chunks_of(lenB, kc);
            if (rt().opt_jacobi_rows != 3) {
                // the lean rendering of the same kernel; FL_OPT_JACOBI_ROWS = 3 keeps the older one for A/B timing
                // loads run one plane ahead while p, p', div sit in the 256 MiB Infinity Cache, two planes ahead when they
                // come from HBM (512^3: 199.8 -> 195.6 us per sweep; 256^3 15.75 vs 15.95 the other way round).
                // FL_OPT_JACOBI_KCHUNK = 1 / 2 forces a distance (it has no other meaning for the fused kernels)
                const bool in_cache = 12.0 * (double)ni * (double)nj * (double)nk <= 256.0 * 1048576.0;
                const int forced = rt().opt_jacobi_kchunk;
                const int pf = forced == 1 || forced == 2 ? forced : (in_cache ? 1 : 2);
                const dim3 gr(nby2 * nbz2);
                hipStream_t st = rt().compute;
#define BQ_LEAN2R(W, F) jacobi_lean2r_kernel<W, F><<<gr, 256, 0, st>>>(in, div, out, ni, nj, nk, cw, nby2, kc, alpha, beta, slab_of(nk), rg)
                if (wide) { if (pf == 1) BQ_LEAN2R(true, 1); else BQ_LEAN2R(true, 2); }
                else      { if (pf == 1) BQ_LEAN2R(false, 1); else BQ_LEAN2R(false, 2); }
#undef BQ_LEAN2R
                BQ_LAUNCH_CHECK("jacobi_lean2r_kernel");
                g_last_pair_kernel = "jacobi_lean2r_kernel";
                return true;
            }
            if (wide) jacobi_march2r_kernel<4, true><<<nby2 * nbz2, 256, 0, rt().compute>>>(in, div, out, ni, nj, nk, cw, nby2, kc, alpha, beta, slab_of(nk), rg);
            else      jacobi_march2r_kernel<4, false><<<nby2 * nbz2, 256, 0, rt().compute>>>(in, div, out, ni, nj, nk, cw, nby2, kc, alpha, beta, slab_of(nk), rg);
            BQ_LAUNCH_CHECK("jacobi_march2r_kernel");
            g_last_pair_kernel = "jacobi_march2r_kernel";
            return true;
        }
    }
    // planes per block: ~32 measured best at 256^3 (one wave per row), ~64 at 512^3 (248 vs 254 us per sweep).  What
    // matters more is that the blocks fill the 256 CUs in whole rounds of two blocks per CU: at 256^3, 512 blocks
    // (chunks of 32) run 19.1 us per sweep, 576 or 448 blocks (chunks of 28 or 40) 22.7; a z-slab rank with 272
    // planes runs 25.4 us with chunks of 32 (9 of them) and 20.0 with chunks of 34 (8).  So: the number of chunks is
    // the multiple of 512 / gcd(row blocks, 512) closest to planes / target.
    int kchunk = rt().opt_jacobi_kchunk2;
    if (kchunk <= 0) {
        const int target = wide ? 64 : 32;
        int gcd = nby, rem = 512;
        while (rem) { const int t = gcd % rem; gcd = rem; rem = t; }
        const int quantum = 512 / gcd;                          // chunk counts that make nby * nbz a multiple of 512
        int nchunks = ((2 * nkr + target) / (2 * target) + quantum / 2) / quantum * quantum;
        if (nchunks < quantum) nchunks = quantum;
        if (rt().num_cus != 256) nchunks = chunks_for_cus(nkr, nby, target, 2, rt().num_cus, 2);
        kchunk = (nkr + nchunks - 1) / nchunks;
        if (kchunk < 16) kchunk = target;                       // small grids: no whole round to fill anyway
        if (!whole && std::max(lenA, lenB) <= 48) kchunk = std::max(lenA, lenB);
    }
    while (whole && kchunk > 8 && (long)nby * ((nk + kchunk - 1) / kchunk) < 512) kchunk /= 2;
    const PairRanges rg{k0a, k1a, k0b, k1b, chunks_of(lenA, kchunk)};
    const int nbz = rg.nchA + chunks_of(lenB, kchunk);
    // (loads two planes ahead instead of one measured no better at 256^3: 19.4 vs 19.1 us per sweep)
    if (wide) jacobi_march2_kernel<4, true><<<nby * nbz, 256, 0, rt().compute>>>(in, div, out, ni, nj, nk, cw, nby, kchunk, alpha, beta, slab_of(nk), rg);
    else      jacobi_march2_kernel<4, false><<<nby * nbz, 256, 0, rt().compute>>>(in, div, out, ni, nj, nk, cw, nby, kchunk, alpha, beta, slab_of(nk), rg);
    BQ_LAUNCH_CHECK("jacobi_march2_kernel");
    g_last_pair_kernel = "jacobi_march2_kernel";
    return true;
}

// Three or four sweeps in one launch through jacobi_lds_kernel (neighbour rows of the intermediate levels via LDS); false = not applicable.
// FL_OPT_JACOBI_ROWS: 4 forces it wherever it applies, 5 keeps it off (A/B timing); auto: see jacobi_sweep_triple.
// k0a .. k1b: the OUTPUT planes as up to two ranges (gpu_jacobi_sweep_triple_ranges: the pieces of a z-slab chunk); default: the whole array.
static bool jacobi_sweep_lds(const float *in, const float *div, float *out, int ni, int nj, int nk, float alpha, float beta, int S,
                             int k0a = 0, int k1a = 1 << 30, int k0b = 0, int k1b = 0)
{
    k0a = std::max(k0a, 0); k1a = std::min(k1a, nk); k0b = std::max(k0b, 0); k1b = std::min(k1b, nk);
    const int lenA = std::max(k1a - k0a, 0), lenB = std::max(k1b - k0b, 0);
    if (lenA + lenB == 0) return true;
    const bool whole = lenB == 0 && lenA == nk;
    auto chunks_of = [](int len, int kc) { return len > 0 ? (len + kc - 1) / kc : 0; };
    // chunk length: whole arrays -- as many chunks as fill the CUs once (one block per CU: LDS, registers), refused below 24
    // planes per chunk (2 (S - 1) warm-up planes: the two-sweep kernel wins there) unless the length is forced; plane ranges
    // (a slab chunk's ends and interiors) -- whatever fills the CUs once, down to 2 planes per chunk
    auto chunk_len = [&](int nby, bool &ok) {
        const int ncus = rt().num_cus;
        const int nranges = (lenA > 0) + (lenB > 0);
        const int per_range = std::max(1, ncus / std::max(1, nby * nranges));
        int kc = std::max(2, (std::max(lenA, lenB) + per_range - 1) / per_range);
        if (rt().opt_jacobi_kchunk2 > 0) kc = rt().opt_jacobi_kchunk2;
        ok = !whole || kc >= (rt().opt_jacobi_kchunk2 > 0 ? 8 : 24);
        return kc;
    };
    // block shape: FL_OPT_JACOBI_KCHUNK = 10 R + W selects R rows per wave and W output waves per block for A/B timing
    // (24 / 25 / 26: row pairs, 4 / 5 / 6 of them; 18 / 19: single rows, 8 / 12 of them); default 18 for three sweeps, 24 for four.
    // 256^3, three sweeps, us per sweep (gpurun_out/r03l, r03m): 18 with chunks of 32 planes 10.79, 24 11.06, 25 11.59, 26 with
    // chunks of 26 11.88, 19 11.23 -- against 13.1-13.6 for jacobi_lean3r_kernel and 15.6 for the two-sweep kernel.  A launch
    // then takes 32.4 us for 201 MB of compulsory traffic = 6.2 TB/s: like the two-sweep kernel (31.4 us per launch) it sits on
    // the fabric, so what is left is more sweeps per launch, not a better schedule -- hence S = 4.
    // Later in round 3 (gpurun_out/r03r .. r03v): the l + r stage through v_add_f32_dpp 10.79 -> 10.71; input rings of 5 / 6
    // planes (loads one / two steps further ahead) 11.00 / 11.19; the step's prefetch issued last 11.11, the first level ahead
    // of the LDS reads 10.76; blocks of 4 single rows, two per CU 11.83; halo waves that skip the levels nobody needs: S = 3
    // 10.69, S = 4 in row pairs 11.40 -> 9.98 (39.9 us per launch).  SQ counters: a wave issues 24-28 % of its cycles, is
    // parked on waitcnt / barrier 40 % and stalled at issue 33 % (the L1 path: with the prefetch last the stall moves to the
    // barrier) -- VALU, LDS and L1 path are each 25-40 % busy but take turns between the barriers.
    // rows of 260 .. 512 floats: the two-segment kernel (three sweeps only), 8 output rows per block
    if (S == 3 && ni > 256 && ni <= 512 && ni % 4 == 0 && nj >= 8 && nk >= 12 && aligned16(in) && aligned16(div) && aligned16(out) &&
        g_klo == 0 && g_khi >= nk && (double)ni * nj * nk * 4.0 < 2147483648.0) {
        constexpr int LW = 8;
        const int nby = (nj + LW - 1) / LW;
        bool ok;
        const int kc = chunk_len(nby, ok);
        if (!ok) return false;
        const PairRanges rg{k0a, k1a, k0b, k1b, chunks_of(lenA, kc)};
        const int nbz = rg.nchA + chunks_of(lenB, kc);
        const int nblk = nby * nbz, grid = 8 * ((nblk + 7) / 8);
        jacobi_lds2seg_kernel<LW><<<grid, (LW + 4) * 64, 0, rt().compute>>>(in, div, out, ni, nj, nk, nby, nblk, kc, alpha, beta, slab_of(nk), rg);
        BQ_LAUNCH_CHECK("jacobi_lds2seg_kernel");
        g_last_pair_kernel = "jacobi_lds2seg_kernel";
        return true;
    }
    int shape = rt().opt_jacobi_kchunk;
    if (shape != 24 && shape != 25 && shape != 26 && shape != 18 && shape != 19) shape = S == 4 ? 24 : 18;
    if (S == 4 && shape != 24 && shape != 18) shape = 24;        // (row pairs, 6 of them: 10 waves at 168 registers spill)
    const int R = shape / 10, W = shape == 19 ? 12 : (S == 4 && shape == 18 ? 6 : shape % 10);        // (19: single rows, 12 of them: 16 waves per block)
    const int rows_per_block = W * R;
    if (!((ni % 4 == 0) && ni >= 32 && ni <= 256 && nj >= rows_per_block && nk >= 12 && aligned16(in) && aligned16(div) && aligned16(out))) return false;
    if (g_klo != 0 || g_khi < nk) return false;              // plane ranges: the two-sweep kernels
    const int nby = (nj + rows_per_block - 1) / rows_per_block;
    // 2 (S - 1) warm-up planes per chunk and one block per CU: below ~24 planes per chunk the short-march two-row kernel wins
    // on whole arrays (128^3: 4.5 us per sweep with chunks of 8 against 3.1); a forced chunk length (tests, tuning) may go down to 8
    bool ok;
    const int kc = chunk_len(nby, ok);
    if (!ok) return false;
    const PairRanges rg{k0a, k1a, k0b, k1b, chunks_of(lenA, kc)};
    const int nbz = rg.nchA + chunks_of(lenB, kc);
    const int nblk = nby * nbz, grid = 8 * ((nblk + 7) / 8);
    hipStream_t st = rt().compute;
#define BQ_LDS(WV, RV, SV) jacobi_lds_kernel<WV, RV, SV><<<grid, (WV + 2 * ((SV - 1 + RV - 1) / RV)) * 64, 0, st>>>(in, div, out, ni, nj, nk, nby, nblk, kc, alpha, beta, slab_of(nk), rg)
    if (S == 4) {
        if (shape == 24) BQ_LDS(4, 2, 4); else BQ_LDS(6, 1, 4);      // (18 with four sweeps: 6 single rows + 6 halo waves)
    } else {
        if (shape == 24) BQ_LDS(4, 2, 3); else if (shape == 25) BQ_LDS(5, 2, 3); else if (shape == 26) BQ_LDS(6, 2, 3);
        else if (shape == 19) BQ_LDS(12, 1, 3);
        else BQ_LDS(8, 1, 3);
    }
#undef BQ_LDS
    BQ_LAUNCH_CHECK("jacobi_lds_kernel");
    g_last_pair_kernel = S == 4 ? "jacobi_lds_kernel<4 sweeps>" : "jacobi_lds3_kernel";
    return true;
}

// Four sweeps in one launch (jacobi_lds_kernel<.., 4>): on request only for now (FL_OPT_JACOBI_ROWS = 6)
static bool jacobi_sweep_quad(const float *in, const float *div, float *out, int ni, int nj, int nk, float alpha, float beta)
{
    if (rt().opt_jacobi_rows != 6 || (rt().opt_jacobi_variant != 0 && rt().opt_jacobi_variant != 3)) return false;
    return jacobi_sweep_lds(in, div, out, ni, nj, nk, alpha, beta, 4);
}

// Three sweeps in one launch (in -> out holds iterate +3), whole array, rows of one wave; false = not applicable
static bool jacobi_sweep_triple(const float *in, const float *div, float *out, int ni, int nj, int nk, float alpha, float beta)
{
    if (ni < 3 || nj < 4 || nk < 3) return false;
    if ((rt().opt_jacobi_rows == 4 || rt().opt_jacobi_rows == 6 || rt().opt_jacobi_rows == 0) && (rt().opt_jacobi_variant == 0 || rt().opt_jacobi_variant == 3) &&
        jacobi_sweep_lds(in, div, out, ni, nj, nk, alpha, beta, 3)) return true;
    const int variant = rt().opt_jacobi_variant;
    if ((variant != 0 && variant != 3) || rt().opt_jacobi_rows == 1 || rt().opt_jacobi_rows == 3) return false;
    if (!((ni % 4 == 0) && ni >= 32 && ni <= 256 && aligned16(in) && aligned16(div) && aligned16(out))) return false;
    if (g_klo != 0 || g_khi < nk) return false;              // plane ranges: the two-sweep kernels
    int cw = 16;
    while (cw * 4 < ni) cw *= 2;
    const int rows = 256 / cw;
    const int nby2 = (nj + 2 * rows - 1) / (2 * rows);
    int gcd = nby2, rem = 256;
    while (rem) { const int t = gcd % rem; gcd = rem; rem = t; }
    const int quantum = 256 / gcd;                           // chunk counts that fill the 256 CUs in whole rounds
    const int target = 32;
    int nchunks = ((2 * nk + target) / (2 * target) + quantum / 2) / quantum * quantum;
    if (nchunks < quantum) nchunks = quantum;
    if (rt().num_cus != 256) nchunks = chunks_for_cus(nk, nby2, target, 4, rt().num_cus, 1);
    int kc = (nk + nchunks - 1) / nchunks;
    if (rt().opt_jacobi_kchunk2 > 0) kc = rt().opt_jacobi_kchunk2;
    if (kc < 16 && rt().opt_jacobi_rows != 2) return false;  // chunks too short to pay for four warm-up planes
    if (kc < 4) kc = 4;
    const int nbz = (nk + kc - 1) / kc;
    // FL_OPT_JACOBI_KCHUNK = 1 / 2: how many planes ahead the loads run
    if (rt().opt_jacobi_kchunk == 2) jacobi_lean3r_kernel<2><<<nby2 * nbz, 256, 0, rt().compute>>>(in, div, out, ni, nj, nk, cw, nby2, kc, alpha, beta, slab_of(nk));
    else                             jacobi_lean3r_kernel<1><<<nby2 * nbz, 256, 0, rt().compute>>>(in, div, out, ni, nj, nk, cw, nby2, kc, alpha, beta, slab_of(nk));
    BQ_LAUNCH_CHECK("jacobi_lean3r_kernel");
    g_last_pair_kernel = "jacobi_lean3r_kernel";
    return true;
}

static const int kResidualBlocks = 1024;

static void residual_norms_async(const float *div, const float *p, int ni, int nj, int nk,
                                 double *d_sum, float *d_max, float *dbg_sum, float *dbg_max)
{
    char *ws = (char *)scratch(kResidualBlocks * (sizeof(double) + sizeof(float)) + 64);
    if (!ws) return;
    double *ps = (double *)ws;
    float *pm = (float *)(ws + kResidualBlocks * sizeof(double));
    hipStream_t st = rt().compute;
    {
        const Runtime &r = rt();
        const int own0 = r.slab_on ? r.slab_own0 : 0, own1 = r.slab_on ? r.slab_own1 : nk;
        residual_partial_kernel<<<kResidualBlocks, 256, 0, st>>>(div, p, ni, nj, nk, ps, pm, slab_of(nk), own0, own1);
    }
    BQ_LAUNCH_CHECK("residual_partial_kernel");
    residual_final_kernel<<<1, 256, 0, st>>>(ps, pm, kResidualBlocks, d_sum, d_max, dbg_sum, dbg_max);
    BQ_LAUNCH_CHECK("residual_final_kernel");
    if (comm_ranks() > 1) {                     // owned-plane partials -> global norms
        if (d_sum) comm_allreduce(d_sum, 1, true, false, st);
        if (d_max) comm_allreduce(d_max, 1, false, true, st);
        if (dbg_sum) comm_allreduce(dbg_sum, 1, false, false, st);
        if (dbg_max) comm_allreduce(dbg_max, 1, false, true, st);
    }
}

// hipEvent pairs around the sweep loops of gpu_projection_jacobi (FL_OPT_PROFILE_JACOBI):
// lets bench.py price the dominant kernel inside the timed region, on the launch stream.

bool profile_begin(ProfileSpan &sp)
{
    if (!rt().opt_profile_jacobi) return false;
    if (!BQ_HIP(hipEventCreate(&sp.a)) || !BQ_HIP(hipEventCreate(&sp.b))) return false;
    return BQ_HIP(hipEventRecord(sp.a, rt().compute));
}
void profile_end(ProfileSpan &sp, long long launches, long long sweeps)
{
    if (!sp.a || !sp.b) return;
    BQ_HIP(hipEventRecord(sp.b, rt().compute));
    g_spans.push_back(SweepSpan{sp.a, sp.b, launches, sweeps});
}

} // namespace bq

using namespace bq;

#define BQ_ENTER(op, ...)                                                  \
    if (!ensure_ready(op)) return;                                         \
    if (!dims_ok(ni, nj, nk, op)) return;                                  \
    {                                                                      \
        const void *ptrs_[] = { __VA_ARGS__ };                             \
        for (const void *p_ : ptrs_)                                       \
            if (!p_) { latch(FL_ERR_BAD_ARGUMENT, op, "null device pointer"); return; } \
    }

extern "C" {

void gpu_divergence(const float *u, const float *v, const float *w, float *div, int ni, int nj, int nk, float halfrdx)
{
    BQ_ENTER("gpu_divergence", u, v, w, div)
    divergence_kernel<<<grid2(ni, nj, nk), kBlock2, 0, rt().compute>>>(u, v, w, div, ni, nj, nk, halfrdx, slab_of(nk));
    BQ_LAUNCH_CHECK("divergence_kernel");
}

int gpu_jacobi_sweeps(float *p, const float *div, float *p_temp, int ni, int nj, int nk, int sweeps, float alpha, float beta)
{
    if (!ensure_ready("gpu_jacobi_sweeps")) return 0;
    if (!dims_ok(ni, nj, nk, "gpu_jacobi_sweeps")) return 0;
    if (!p || !div || !p_temp || p == p_temp) { latch(FL_ERR_BAD_ARGUMENT, "gpu_jacobi_sweeps", "null or aliased buffers"); return 0; }
    float *in = p, *out = p_temp;
    int s = 0;
    long long launches = 0;
    ProfileSpan span;
    const bool prof = sweeps > 0 && profile_begin(span);      // FL_OPT_PROFILE_JACOBI (the z-slab projection runs through here)
    // FL_OPT_JACOBI_FUSE == 2: the caller vouches that p and p_temp carry the same boundary layer
    // (FL_OPT_JACOBI_FUSE: 2 = pairs and triples, 4 = pairs only)
    // FL_OPT_JACOBI_ROWS = 6: four sweeps per launch where jacobi_lds_kernel applies (rows 0: auto, see jacobi_sweep_quad)
    while (rt().opt_jacobi_fuse >= 2 && rt().opt_jacobi_fuse != 4 && s + 4 <= sweeps && jacobi_sweep_quad(in, div, out, ni, nj, nk, alpha, beta)) {
        s += 4; launches++;                        // iterate +4 sits in `out`: swap
        float *t = in; in = out; out = t;
    }
    while (rt().opt_jacobi_fuse >= 2 && rt().opt_jacobi_fuse != 4 && s + 3 <= sweeps && jacobi_sweep_triple(in, div, out, ni, nj, nk, alpha, beta)) {
        float *t = in; in = out; out = t;          // iterate +3 sits in the former `out`
        s += 3; launches++;
    }
    while (rt().opt_jacobi_fuse >= 2 && s + 2 <= sweeps && jacobi_sweep_pair(in, div, out, ni, nj, nk, alpha, beta)) {
        float *t = in; in = out; out = t;          // iterate +2 sits in the former `out`
        s += 2; launches++;
    }
    for (; s < sweeps; s++) {
        jacobi_sweep(in, div, out, ni, nj, nk, alpha, beta);
        float *t = in; in = out; out = t;
        launches++;
    }
    if (prof) profile_end(span, launches, sweeps);
    return in == p ? 0 : 1;
}

// one sweep in -> out restricted to the local planes [k_begin, k_end) (clipped to the interior)
void gpu_jacobi_sweep_range(const float *in, const float *div, float *out, int ni, int nj, int nk,
                            int k_begin, int k_end, float alpha, float beta)
{
    BQ_ENTER("gpu_jacobi_sweep_range", in, div, out)
    BQ_REQUIRE(in != out, "gpu_jacobi_sweep_range");
    if (k_begin >= k_end) return;
    g_klo = k_begin; g_khi = k_end;
    jacobi_sweep(in, div, out, ni, nj, nk, alpha, beta);
    g_klo = 0; g_khi = 1 << 30;
}

// Two sweeps in one launch, `out` written on the planes [k0a, k1a) and [k0b, k1b) only (either may be empty): the
// pieces of a chunk's first two sweeps on a z-slab rank -- planes whose two-sweep stencil stays inside the owned
// planes while the ghost planes are in flight, then the rest.  Returns 1 when the fused kernel ran, 0 when it does
// not apply to this grid (nothing launched: the caller sweeps plane ranges one sweep at a time instead).
int gpu_jacobi_sweep_pair_ranges(const float *in, const float *div, float *out, int ni, int nj, int nk,
                                 int k0a, int k1a, int k0b, int k1b, float alpha, float beta)
{
    if (!ensure_ready("gpu_jacobi_sweep_pair_ranges") || !dims_ok(ni, nj, nk, "gpu_jacobi_sweep_pair_ranges")) return 0;
    if (!in || !div || !out || in == out) { latch(FL_ERR_BAD_ARGUMENT, "gpu_jacobi_sweep_pair_ranges", "null or aliased buffers"); return 0; }
    if (rt().opt_jacobi_fuse == 0) return 0;
    return jacobi_sweep_pair(in, div, out, ni, nj, nk, alpha, beta, k0a, k1a, k0b, k1b) ? 1 : 0;
}

// Three sweeps in -> out on the output planes [k0a, k1a) and [k0b, k1b) (either may be empty) through the LDS-exchanged
// kernels: the triple counterpart of gpu_jacobi_sweep_pair_ranges for the chunks of a z-slab rank.  The input must be
// valid three planes beyond each range.  Returns 1 when it ran, 0 when it does not apply (nothing launched).
int gpu_jacobi_sweep_triple_ranges(const float *in, const float *div, float *out, int ni, int nj, int nk,
                                   int k0a, int k1a, int k0b, int k1b, float alpha, float beta)
{
    if (!ensure_ready("gpu_jacobi_sweep_triple_ranges") || !dims_ok(ni, nj, nk, "gpu_jacobi_sweep_triple_ranges")) return 0;
    if (!in || !div || !out || in == out) { latch(FL_ERR_BAD_ARGUMENT, "gpu_jacobi_sweep_triple_ranges", "null or aliased buffers"); return 0; }
    if (rt().opt_jacobi_fuse == 0 || rt().opt_jacobi_fuse == 4 || rt().opt_jacobi_rows == 5) return 0;
    if (rt().opt_jacobi_variant != 0 && rt().opt_jacobi_variant != 3) return 0;
    if (ni < 3 || nj < 4 || nk < 3) return 0;
    return jacobi_sweep_lds(in, div, out, ni, nj, nk, alpha, beta, 3, k0a, k1a, k0b, k1b) ? 1 : 0;
}

void gpu_gradient(float *u, float *v, float *w, const float *p, int ni, int nj, int nk, float halfrdx)
{
    BQ_ENTER("gpu_gradient", u, v, w, p)
    gradient_kernel<<<grid2(ni, nj, nk), kBlock2, 0, rt().compute>>>(u, v, w, p, ni, nj, nk, halfrdx, slab_of(nk));
    BQ_LAUNCH_CHECK("gradient_kernel");
}

void gpu_gradient_delta(float *u, float *v, float *w, const float *p, float *du, float *dv, float *dw,
                        int ni, int nj, int nk, float halfrdx)
{
    BQ_ENTER("gpu_gradient_delta", u, v, w, p, du, dv, dw)
    gradient_delta_kernel<<<grid2(ni + 1, nj + 1, nk + 1), kBlock2, 0, rt().compute>>>(u, v, w, p, du, dv, dw, ni, nj, nk, halfrdx, slab_of(nk));
    BQ_LAUNCH_CHECK("gradient_delta_kernel");
}

void gpu_residual_norms(const float *div, const float *p, int ni, int nj, int nk, double *sum_sq, float *max_abs)
{
    BQ_ENTER("gpu_residual_norms", div, p)
    char *host = (char *)pinned(64);
    char *dev = (char *)scratch(kResidualBlocks * 12 + 64 + 64);
    if (!host || !dev) return;
    double *d_sum = (double *)(dev + kResidualBlocks * 12 + 64);
    float *d_max = (float *)(d_sum + 1);
    residual_norms_async(div, p, ni, nj, nk, d_sum, d_max, nullptr, nullptr);
    BQ_HIP(hipMemcpyAsync(host, d_sum, 16, hipMemcpyDeviceToHost, rt().compute));
    BQ_HIP(hipStreamSynchronize(rt().compute));
    if (sum_sq) *sum_sq = *(double *)host;
    if (max_abs) *max_abs = *(float *)(host + 8);
}

} // extern "C"

// The boundary shell (outermost cell layer) of a pressure buffer is never written by a sweep; the reference
// ping-pongs, so odd iterates carry p_temp's shell and even ones p's (GPU_kernel.cu:1819-1837: interior only).
// A fused two-sweep launch reads the intermediate iterate's shell from its INPUT buffer, which is the same thing
// only when both shells hold the same values.  One compare of the two shells, one flag.
__global__ void __launch_bounds__(256) shell_differs_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                           int ni, int nj, int nk, int *__restrict__ flag)
{
    const long n = (long)ni * nj * nk;
    bool bad = false;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
        const int i = (int)(id % ni), j = (int)((id / ni) % nj), k = (int)(id / ((long)ni * nj));
        const bool shell = i == 0 || j == 0 || k == 0 || i == ni - 1 || j == nj - 1 || k == nk - 1;
        // value equality, with NaN == NaN (a NaN shell propagates identically from either buffer)
        if (shell && !(a[id] == b[id] || (a[id] != a[id] && b[id] != b[id]))) bad = true;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// true when p and p_temp carry the same boundary shell (blocking: one 4-byte read-back)
static bool shells_match(const float *p, const float *p_temp, int ni, int nj, int nk)
{
    int *dflag = (int *)scratch(64);
    int *hflag = (int *)pinned(64);
    if (!dflag || !hflag) return false;
    hipStream_t st = rt().compute;
    if (!BQ_HIP(hipMemsetAsync(dflag, 0, 4, st))) return false;
    // the shell is 6 faces of a box: walk the whole index space only when it is small, else face by face would be
    // cheaper -- but a single pass at HBM rate costs 20 us at 256^3 against a 3.5 ms projection
    const long n = (long)ni * nj * nk;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    shell_differs_kernel<<<blocks, 256, 0, st>>>(p, p_temp, ni, nj, nk, dflag);
    if (!BQ_LAUNCH_CHECK("shell_differs_kernel")) return false;
    if (!BQ_HIP(hipMemcpyAsync(hflag, dflag, 4, hipMemcpyDeviceToHost, st)) || !BQ_HIP(hipStreamSynchronize(st))) return false;
    return *hflag == 0;
}

extern "C" {

// GPU_kernel.cu:1839-1895.  iter sweeps are specified, iterate iter-1 is what the reference
// applies and leaves in p -> run iter-1 sweeps and make sure the result is in p.
// Two sweeps share a launch (FL_OPT_JACOBI_FUSE) only when that cannot change a value: with the option at 1 (default)
// after checking that p and p_temp carry the same boundary shell -- true for the reference's own caller, which clears
// both (gpuMapper::projectionJacobi, GPU_Advection.h:604-606), not guaranteed for any other caller of this symbol
// (a warm-started p, a p_temp with stale contents) --, at 2 on the caller's word, at 0 never.
void gpu_projection_jacobi(float *u, float *v, float *w, float *div, float *p, float *p_temp, float *debugParam,
                           int ni, int nj, int nk, int iter, float halfrdx, float alpha, float beta)
{
    BQ_ENTER("gpu_projection_jacobi", u, v, w, div, p, p_temp)
    BQ_REQUIRE(p != p_temp && iter >= 0, "gpu_projection_jacobi");
    hipStream_t st = rt().compute;
    divergence_kernel<<<grid2(ni, nj, nk), kBlock2, 0, st>>>(u, v, w, div, ni, nj, nk, halfrdx, slab_of(nk));
    BQ_LAUNCH_CHECK("divergence_kernel");
    const int stride = rt().opt_residual_stride;
    const bool dbg = debugParam != nullptr && stride > 0;
    float *in = p, *out = p_temp;
    const bool prof = rt().opt_profile_jacobi && !dbg && iter > 1;
    SweepSpan span{nullptr, nullptr, 0, (long long)(iter - 1)};
    if (prof && BQ_HIP(hipEventCreate(&span.a)) && BQ_HIP(hipEventCreate(&span.b))) BQ_HIP(hipEventRecord(span.a, st));
    long long launches = 0;
    const bool may_fuse = iter > 2 && (rt().opt_jacobi_fuse >= 2 ||
                                      (rt().opt_jacobi_fuse == 1 && !rt().slab_on && shells_match(p, p_temp, ni, nj, nk)));
    if (iter == 0) {
        // the reference's swap loop does not run: p_out is still p_temp, which is copied over p (:1876-1879) and used
        // by the gradient (:1883-1891)
        fl_memcpy_d2d(p, p_temp, (size_t)ni * nj * nk * sizeof(float));
    }
    for (int it = 0; it + 1 < iter; ) {
        if (dbg && it % stride == 0 && it < 2000)
            residual_norms_async(div, in, ni, nj, nk, nullptr, nullptr, debugParam + it, debugParam + 2000 + it);
        // equal boundary shells (checked above): two sweeps may share a launch -- unless residual norms are wanted
        // for the iterate in between
        const bool pair_ok = may_fuse && it + 2 < iter && !(dbg && (it + 1) % stride == 0);
        const bool triple_ok = pair_ok && it + 3 < iter && !(dbg && (it + 2) % stride == 0) && rt().opt_jacobi_fuse != 4;
        if (triple_ok && jacobi_sweep_triple(in, div, out, ni, nj, nk, alpha, beta)) {
            it += 3;
        } else if (pair_ok && jacobi_sweep_pair(in, div, out, ni, nj, nk, alpha, beta)) {
            it += 2;
        } else {
            jacobi_sweep(in, div, out, ni, nj, nk, alpha, beta);
            it += 1;
        }
        launches++;
        float *t = in; in = out; out = t;
    }
    span.launches = launches;
    if (prof && span.a && span.b) { BQ_HIP(hipEventRecord(span.b, st)); g_spans.push_back(span); }
    if (dbg && iter > 0 && (iter - 1) % stride == 0 && iter - 1 < 2000)
        residual_norms_async(div, in, ni, nj, nk, nullptr, nullptr, debugParam + iter - 1, debugParam + 2000 + iter - 1);
    if (in != p) fl_memcpy_d2d(p, in, (size_t)ni * nj * nk * sizeof(float));
    gradient_kernel<<<grid2(ni, nj, nk), kBlock2, 0, st>>>(u, v, w, p, ni, nj, nk, halfrdx, slab_of(nk));
    BQ_LAUNCH_CHECK("gradient_kernel");
}

// Sum of the recorded sweep-loop spans since the last call (blocking); clears the list.
void fl_jacobi_profile(double *total_ms, long long *launches, long long *sweeps)
{
    double ms = 0.0;
    long long n = 0, sw = 0;
    for (SweepSpan &sp : g_spans) {
        float t = 0.f;
        if (BQ_HIP(hipEventSynchronize(sp.b)) && BQ_HIP(hipEventElapsedTime(&t, sp.a, sp.b))) { ms += t; n += sp.launches; sw += sp.sweeps; }
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    g_spans.clear();
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (sweeps) *sweeps = sw;
}

const char *fl_jacobi_kernel_name(void) { return g_last_pair_kernel; }

// GPU_kernel.cu:855-876
void gpu_diffuse_field(float *field, float *fieldTemp0, float *filedTemp1, int ni, int nj, int nk, int iter, float coef)
{
    BQ_ENTER("gpu_diffuse_field", field, fieldTemp0, filedTemp1)
    BQ_REQUIRE(field != fieldTemp0 && field != filedTemp1 && fieldTemp0 != filedTemp1 && iter >= 0, "gpu_diffuse_field");
    const size_t bytes = (size_t)ni * nj * nk * sizeof(float);
    float *in = fieldTemp0, *out = filedTemp1;
    // nk is a BUFFER dim here (nk+1 for w); the slab context carries cell planes: keep the difference
    Slab dsl = slab_of(nk);
    if (rt().slab_on) dsl.nkg += nk - rt().slab_nkl;
    fl_memcpy_d2d(in, field, bytes);
    for (int it = 0; it < iter; it++) {
        diffuse_kernel<<<grid2(ni, nj, nk), kBlock2, 0, rt().compute>>>(field, in, out, ni, nj, nk, coef, dsl.koff, dsl.nkg);
        BQ_LAUNCH_CHECK("diffuse_kernel");
        float *t = out; out = in; in = t;
    }
    fl_memcpy_d2d(field, out, bytes);
}

// `sweeps` sweeps of diffuse_field_kernel ping-ponging in -> out -> in ...; returns 0 when the newest iterate
// sits in `in`, 1 when it sits in `out` (the other buffer holds the iterate before it).  No copies: the pieces
// of gpu_diffuse_field a z-slab host needs to refresh ghost planes between chunks of sweeps.
int gpu_diffuse_sweeps(const float *field, float *in, float *out, int ni, int nj, int nk, int sweeps, float coef)
{
    if (!ensure_ready("gpu_diffuse_sweeps") || !dims_ok(ni, nj, nk, "gpu_diffuse_sweeps")) return 0;
    if (!field || !in || !out || in == out || field == in || field == out || sweeps < 0) {
        latch(FL_ERR_BAD_ARGUMENT, "gpu_diffuse_sweeps", "null or aliased buffers"); return 0;
    }
    Slab dsl = slab_of(nk);
    if (rt().slab_on) dsl.nkg += nk - rt().slab_nkl;        // nk is a BUFFER dim (nk+1 for w)
    float *a = in, *b = out;
    for (int it = 0; it < sweeps; it++) {
        diffuse_kernel<<<grid2(ni, nj, nk), kBlock2, 0, rt().compute>>>(field, a, b, ni, nj, nk, coef, dsl.koff, dsl.nkg);
        float *t = a; a = b; b = t;
    }
    BQ_LAUNCH_CHECK("diffuse_kernel");
    return a == in ? 0 : 1;
}

void gpu_conjugate_gradient(float *, float *, float *, float *, float *, float *, float *, float *, int, int, int, int, float)
{
    latch(FL_ERR_UNSUPPORTED, "gpu_conjugate_gradient", "out of scope: alternative solver compiled out in the reference");
}

// gpu_multi_grid_conjugate_gradient: bq_mgcg.hip

} // extern "C"
