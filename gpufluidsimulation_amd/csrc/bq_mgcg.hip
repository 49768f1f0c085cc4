// bq_mgcg.hip -- fp64 multigrid-corrected CG pressure projection:
// gpu_multi_grid_conjugate_gradient, src/bimocq3D/GPU_kernel.cu:1764-1815 and everything it launches
// (:1043-1338 CG helpers, :1420-1707 smoothing / restriction / prolongation / V_Cycle).
//
// Same buffers, same launch order, same arithmetic as the reference, so the stale boundary entries the
// interior-only kernels leave behind (oracle/mgcg_oracle.c, M5) carry the same values.  The quirks that
// change values are kept (M1 the float-narrowed, mis-indexed block dot product; M2 float lerps inside
// the fp64 transfer operators; M3 alpha*8 on level 1 only; M4 unclamped coarse look-ups, with reads past
// the end of an array returning 0).  What is ours: launch geometry (3-D grids, x fastest), the reductions'
// work split where the order does not matter (calc_max), one stream, no per-call allocation.
//
// Bounds: every kernel here streams fp64 arrays -- 24 B/cell per smoothing sweep (x, b in, x' out).  At
// 256^3 the three level-0 arrays (402 MB) exceed the 256 MiB Infinity Cache, so the level-0 sweeps are
// HBM-bound; levels >= 1 (<= 50 MB) live in the cache.
//
// Kernels by level size (256^3 V-cycle: 256, 127, 63, 31, 15, 7):
//   >= 1 M cells   mg_lean2r_kernel (two sweeps per launch, two rows per thread, buffer descriptors, plane rings; odd rows
//                  too), mg_stencil_lean_kernel (residual and A.dir in the marching form), clears reduced to the faces;
//   <  2^19 cells  mg_smooth_tile_kernel (4 or 2 sweeps per launch on a 16^3 region in LDS, clears folded in);
//   every level    mg_restrict_kernel (3x3x3 register block), mg_prolong_block_kernel (2x2x2 fine cells per thread);
//   fallbacks      mg_smooth_kernel / mg_smooth2_kernel / mg_residual_kernel / mg_poisson_kernel / mg_prolong_kernel
//                  (one thread per cell; FL_OPT_MGCG_TILE = 0, FL_OPT_JACOBI_FUSE = 0, FL_OPT_JACOBI_ROWS = 3 select them).
#include "bq_device.hip.h"
#include "bq_buffer.hip.h"
#include <type_traits>
#include "bq_host.h"

#include <cstdint>

namespace bq {

// ---- index helpers ------------------------------------------------------------------------------
static const dim3 kBlk(64, 4, 1);
static inline dim3 grid_of(int ni, int nj, int nk) { return dim3((ni + 63) / 64, (nj + 3) / 4, nk); }
#define MG_IJK(NI, NJ, NK)                                                                   \
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z; \
    if (i >= (NI) || j >= (NJ) || k >= (NK)) return;
#define MG_INTERIOR(NI, NJ, NK) (i > 0 && i < (NI) - 1 && j > 0 && j < (NJ) - 1 && k > 0 && k < (NK) - 1)

__device__ __forceinline__ size_t id3(int i, int j, int k, int ni, int nj) { return (size_t)i + (size_t)ni * j + (size_t)ni * nj * k; }

// calc_poisson_value (:1048-1060): (l + r + f + b + d + u) - c*6
__device__ __forceinline__ double poisson_value(const double *x, size_t id, size_t sj, size_t sk)
{
    return (x[id - 1] + x[id + 1] + x[id - sj] + x[id + sj] + x[id - sk] + x[id + sk]) - x[id] * 6;
}

// divergence_kernel, double overload (:991-1007): every cell
__global__ __launch_bounds__(256) void mg_divergence_kernel(const float *__restrict__ u, const float *__restrict__ v,
                                                            const float *__restrict__ w, double *__restrict__ div,
                                                            int ni, int nj, int nk, double halfrdx)
{
    MG_IJK(ni, nj, nk)
    const double ul = u[id3(i, j, k, ni + 1, nj)], ur = u[id3(i + 1, j, k, ni + 1, nj)];
    const double vf = v[id3(i, j, k, ni, nj + 1)], vb = v[id3(i, j + 1, k, ni, nj + 1)];
    const double wd = w[id3(i, j, k, ni, nj)],     wu = w[id3(i, j, k + 1, ni, nj)];
    div[id3(i, j, k, ni, nj)] = halfrdx * ((ur - ul) + (vb - vf) + (wu - wd));
}

// calc_poisson_kernel (:1075-1085) and update_residual_kernel (:1251-1261), double, interior only
__global__ __launch_bounds__(256) void mg_poisson_kernel(const double *__restrict__ x, double *__restrict__ b, int ni, int nj, int nk)
{
    MG_IJK(ni, nj, nk)
    if (!MG_INTERIOR(ni, nj, nk)) return;
    const size_t id = id3(i, j, k, ni, nj);
    b[id] = poisson_value(x, id, ni, (size_t)ni * nj);
}
__global__ __launch_bounds__(256) void mg_residual_kernel(double *__restrict__ r, const double *__restrict__ b,
                                                          const double *__restrict__ x, int ni, int nj, int nk)
{
    MG_IJK(ni, nj, nk)
    if (!MG_INTERIOR(ni, nj, nk)) return;
    const size_t id = id3(i, j, k, ni, nj);
    r[id] = b[id] - poisson_value(x, id, ni, (size_t)ni * nj);
}

// dot_vector_kernel (:1087-1126), M1: one block per 256 elements, the reference's summation tree
__global__ __launch_bounds__(256) void mg_dot_kernel(const double *__restrict__ v0, const double *__restrict__ v1,
                                                     double *__restrict__ output, size_t count)
{
    __shared__ double sh[272];
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    sh[threadIdx.x] = id < count ? v0[id] * v1[id] : 0.0;
    __syncthreads();
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        double s = sh[t * 16];
#pragma unroll
        for (int q = 1; q < 16; q++) s = s + sh[t * 16 + q];
        sh[256 + t] = (double)(float)s;                     // `float sum0`
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = sh[256 + 0] + sh[256 + 1] + sh[256 + 2] + sh[3] +
                   sh[256 + 4] + sh[256 + 5] + sh[256 + 6] + sh[7] +
                   sh[256 + 8] + sh[256 + 9] + sh[256 + 10] + sh[11] +
                   sh[256 + 12] + sh[256 + 13] + sh[256 + 14] + sh[15];
        output[blockIdx.x] = (double)(float)s;              // `float sum`
    }
}

// calc_sum_kernel<<<1,256>>> (:1134-1183): thread t sums `per_thread` consecutive partials, then 16x16
__global__ __launch_bounds__(256) void mg_calc_sum_kernel(const double *__restrict__ v, double *__restrict__ output,
                                                          size_t count, size_t per_thread, int iter_index)
{
    __shared__ double sh[272];
    double s = 0;
    const size_t start = (size_t)threadIdx.x * per_thread;
    for (size_t q = 0; q < per_thread; q++)
        if (start + q < count) s += v[start + q];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        double a = sh[t * 16];
#pragma unroll
        for (int q = 1; q < 16; q++) a = a + sh[t * 16 + q];
        sh[256 + t] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = sh[256];
#pragma unroll
        for (int q = 1; q < 16; q++) a = a + sh[256 + q];
        output[iter_index] = a;
    }
}

// The same sums in two launches.  One workgroup whose thread t walks its own 2 KB of partials reads 64 cache lines per
// load instruction: 36 us for the 65 536 partials of a 256^3 dot product, 101 times per step.  Here workgroup g takes the
// reference's threads 16 g .. 16 g + 15: their 16 x per_thread partials are loaded coalesced into LDS (in tiles of 256
// columns), 16 lanes add their row in the reference's order, and a second launch runs the 16 x 16 tree on the 256 row sums.
__global__ __launch_bounds__(256) void mg_calc_sum_rows_kernel(const double *__restrict__ v, double *__restrict__ rows,
                                                               size_t count, size_t per_thread)
{
    __shared__ double tile[16][257];                        // (+1: the 16 summing lanes read one column of 16 rows at a time)
    const int t0 = blockIdx.x * 16;
    double s = 0;
    for (size_t c0 = 0; c0 < per_thread; c0 += 256) {
        const size_t w = per_thread - c0 < 256 ? per_thread - c0 : 256;
        __syncthreads();
        for (int r = 0; r < 16; r++) {
            const size_t id = (size_t)(t0 + r) * per_thread + c0 + threadIdx.x;
            if (threadIdx.x < w) tile[r][threadIdx.x] = id < count ? v[id] : 0.0;
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            const size_t start = (size_t)(t0 + threadIdx.x) * per_thread + c0;
            // partials beyond `count` are not added (the reference's bound check): n of this tile's w columns are
            const size_t n = start >= count ? 0 : (count - start < w ? count - start : w);
            const double *row = tile[threadIdx.x];
            size_t q = 0;
            for (; q + 8 <= n; q += 8) {                    // the LDS reads of a group are independent, the adds stay in order
                double t[8];
#pragma unroll
                for (int e = 0; e < 8; e++) t[e] = row[q + e];
#pragma unroll
                for (int e = 0; e < 8; e++) s += t[e];
            }
            for (; q < n; q++) s += row[q];
        }
    }
    if (threadIdx.x < 16) rows[t0 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void mg_calc_sum_tree_kernel(const double *__restrict__ rows, double *__restrict__ output, int iter_index)
{
    __shared__ double sh[16];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        double a = rows[t * 16];
#pragma unroll
        for (int q = 1; q < 16; q++) a = a + rows[t * 16 + q];
        sh[t] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = sh[0];
#pragma unroll
        for (int q = 1; q < 16; q++) a = a + sh[q];
        output[iter_index] = a;
    }
}

// calc_max (:1185-1237): max(v, 0) over the array -- order-free, so split over many blocks
__global__ __launch_bounds__(256) void mg_max_partial_kernel(const double *__restrict__ v, size_t count, double *__restrict__ part)
{
    double m = 0;
    // pairs of doubles (the arrays are 16-byte aligned allocations; an odd count leaves one element for thread 0)
    const size_t pairs = (((uintptr_t)v & 15u) == 0) ? count / 2 : 0;
    const double2 *v2 = reinterpret_cast<const double2 *>(v);
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < pairs; q += (size_t)gridDim.x * 256) {
        const double2 t = v2[q];
        m = fmax(m, fmax(t.x, t.y));
    }
    for (size_t q = 2 * pairs + (size_t)blockIdx.x * 256 + threadIdx.x; q < count; q += (size_t)gridDim.x * 256) m = fmax(m, v[q]);
    __shared__ double sh[256];
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void mg_max_final_kernel(const double *__restrict__ part, int nparts, double *output, int iter_index)
{
    double m = 0;
    for (int q = threadIdx.x; q < nparts; q += 256) m = fmax(m, part[q]);
    __shared__ double sh[256];
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) output[iter_index] = sh[0];
}

// update_x_kernel / update_dir_kernel / mul_kernel / add_kernel (:1275-1338), double
__global__ __launch_bounds__(256) void mg_update_x_kernel(double *__restrict__ x, const double *__restrict__ dir,
                                                          const double *__restrict__ alpha, size_t count, int r_index, int d_index)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id < count) x[id] += dir[id] * alpha[r_index] / alpha[d_index];
}
__global__ __launch_bounds__(256) void mg_update_dir_kernel(double *__restrict__ dir, const double *__restrict__ residual,
                                                            const double *__restrict__ beta, size_t count, int r_index, int rplus_index)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id < count) dir[id] = residual[id] + dir[id] * beta[rplus_index] / beta[r_index];
}
__global__ __launch_bounds__(256) void mg_mul_kernel(double *__restrict__ result, const double *__restrict__ field, double constant, size_t count)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id < count) result[id] = field[id] * constant;
}
__global__ __launch_bounds__(256) void mg_add_kernel(double *__restrict__ f0, const double *__restrict__ f1, double coef, size_t count)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id < count) f0[id] += f1[id] * coef;
}

// smoothing_jacobi_kernel, double (:1443-1461): interior only
__global__ __launch_bounds__(256) void mg_smooth_kernel(const double *__restrict__ x, const double *__restrict__ b,
                                                        double *__restrict__ out, double alpha, double beta, int ni, int nj, int nk)
{
    MG_IJK(ni, nj, nk)
    if (!MG_INTERIOR(ni, nj, nk)) return;
    const size_t sj = ni, sk = (size_t)ni * nj, id = id3(i, j, k, ni, nj);
    out[id] = ((x[id - 1] + x[id + 1] + x[id - sj] + x[id + sj] + x[id - sk] + x[id + sk]) + alpha * b[id]) * beta;
}

// ---- two smoothing sweeps per launch (temporal fusion, register marching) -----------------------------
// The fp64 counterpart of jacobi_march2_kernel (bq_project.hip): L1 = S(L0), L2 = S(L1) with L1 kept in
// registers.  A thread owns VEC (1 or 2) consecutive cells of row j and marches along k holding L0 of the
// rows j-1, j, j+1 on three planes; it evaluates L1 on those rows (the outer two redundantly) and L2 on
// row j one plane behind.  x-neighbours come from the neighbouring lanes by wave64 shuffle.  A row may
// span several waves (256 doubles = 2 waves of double2): at a wave boundary lane 0 / lane 63 fetch the
// L0 values just outside the wave from memory and evaluate the one L1 value there themselves (a few
// one-lane loads per plane, no LDS, no barrier).  Every value is produced by the reference's expression
// ((l + r + f + b + d + u) + alpha*rhs) * beta, so two launches of mg_smooth_kernel give the same bits.
// Preconditions (checked by the launcher): both ping-pong buffers carry the same boundary layer (V_Cycle
// clears x and temp0 before every smoothing call); VEC == 2 needs an even nx.
template <int VEC> struct DV { double c[VEC]; };

template <int VEC, int THREADS>
__global__ __launch_bounds__(THREADS) void mg_smooth2_kernel(const double *__restrict__ x, const double *__restrict__ rhs,
                                                             double *__restrict__ out, double alpha, double beta,
                                                             int nx, int ny, int nz, int lpr, int nby, int kchunk)
{
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);          // XCD-contiguous block order
    const int by = blk % nby, bz = blk / nby;
    const int rows = THREADS / lpr;
    const int c = threadIdx.x % lpr, r = threadIdx.x / lpr;
    const int xraw = VEC * c, j = by * rows + r;
    const int lane = threadIdx.x & 63;
    const int kbeg = max(1, bz * kchunk), kend = min(nz - 1, bz * kchunk + kchunk);
    if (kbeg >= kend) return;
    const bool xok = xraw < nx;
    const bool active = xok && j >= 1 && j <= ny - 2;
    // Out-of-range rows, planes and lanes are CLAMPED into the array instead of being zero-filled: whatever
    // they load only ever feeds cells that are boundary (kept from L0) or not stored at all.
    const int x0 = xok ? xraw : nx - VEC;
    const long long sj = nx, sk = (long long)nx * ny;
    auto rowoff = [&](int row) -> long long { return (long long)x0 + sj * min(max(row, 0), ny - 1); };
    const long long o_m2 = rowoff(j - 2), o_m1 = rowoff(j - 1), o_0 = rowoff(j), o_p1 = rowoff(j + 1), o_p2 = rowoff(j + 2);
    auto plane = [&](int pl) -> long long { return sk * min(max(pl, 0), nz - 1); };
    // edge lanes: lane 0 looks after the cell just left of the wave, lane 63 after the cell just right of it
    const bool edgeL = lane == 0 && xok && xraw > 0, edgeR = lane == 63 && xraw + VEC < nx;
    const bool edge = edgeL || edgeR;
    const int xe = edgeL ? xraw - 1 : xraw + VEC;                      // the outside cell
    const int xo = edgeL ? xe - 1 : xe + 1;                            // its own outer x-neighbour
    const long long e_m1 = (long long)xe + sj * min(max(j - 1, 0), ny - 1), e_0 = (long long)xe + sj * min(max(j, 0), ny - 1),
                    e_p1 = (long long)xe + sj * min(max(j + 1, 0), ny - 1),
                    e_o = (long long)min(max(xo, 0), nx - 1) + sj * min(max(j, 0), ny - 1);
    const bool rowb_m1 = j - 1 <= 0 || j - 1 >= ny - 1, rowb_0 = j <= 0 || j >= ny - 1, rowb_p1 = j + 1 <= 0 || j + 1 >= ny - 1;
    const bool xe_boundary = xe <= 0 || xe >= nx - 1;

    auto ldv = [&](const double *ptr, long long off) -> DV<VEC> {
        DV<VEC> v;
        if constexpr (VEC == 2) { const double2 t = *reinterpret_cast<const double2 *>(ptr + off); v.c[0] = t.x; v.c[1] = t.y; }
        else v.c[0] = ptr[off];
        return v;
    };
    // one smoothing evaluation on the thread's cells; lo/ro: the values just outside the wave (edge lanes)
    auto jac = [&](const DV<VEC> &ce, const DV<VEC> &fr, const DV<VEC> &bk, const DV<VEC> &dn, const DV<VEC> &up,
                   const DV<VEC> &dv, double outside, bool boundary) -> DV<VEC> {
        double left = lane_up(ce.c[VEC - 1]), right = lane_down(ce.c[0]);
        if (lane == 0) left = outside;
        if (lane == 63) right = outside;
        DV<VEC> o;
        if constexpr (VEC == 2) {
            o.c[0] = ((left + ce.c[1] + fr.c[0] + bk.c[0] + dn.c[0] + up.c[0]) + alpha * dv.c[0]) * beta;
            o.c[1] = ((ce.c[0] + right + fr.c[1] + bk.c[1] + dn.c[1] + up.c[1]) + alpha * dv.c[1]) * beta;
        } else {
            o.c[0] = ((left + right + fr.c[0] + bk.c[0] + dn.c[0] + up.c[0]) + alpha * dv.c[0]) * beta;
        }
        if (boundary) return ce;
        if (x0 == 0) o.c[0] = ce.c[0];
        if (x0 + VEC - 1 == nx - 1) o.c[VEC - 1] = ce.c[VEC - 1];
        return o;
    };

    // L0 of rows j-1, j, j+1 on planes q-1 (Lm), q (Lc), q+1 (Ln); q = plane whose L1 is being built
    DV<VEC> Lm[3], Lc[3], Ln[3], Dv[3];
    int q = kbeg - 1;
    {
        const long long pm = plane(q - 1), pc = plane(q), pn = plane(q + 1);
        Lm[0] = ldv(x, pm + o_m1); Lm[1] = ldv(x, pm + o_0); Lm[2] = ldv(x, pm + o_p1);
        Lc[0] = ldv(x, pc + o_m1); Lc[1] = ldv(x, pc + o_0); Lc[2] = ldv(x, pc + o_p1);
        Ln[0] = ldv(x, pn + o_m1); Ln[1] = ldv(x, pn + o_0); Ln[2] = ldv(x, pn + o_p1);
        Dv[0] = ldv(rhs, pc + o_m1); Dv[1] = ldv(rhs, pc + o_0); Dv[2] = ldv(rhs, pc + o_p1);
    }
    DV<VEC> Hf = ldv(x, plane(q) + o_m2), Hb = ldv(x, plane(q) + o_p2);
    // the outside cell (edge lanes only): L0 on rows j-1..j+1 at plane q (Ec), on row j at planes q-1 (Em)
    // and q+1 (En), its outer x-neighbour at plane q (Eo), its rhs (Eb)
    double Ec[3] = { 0, 0, 0 }, Em = 0, En = 0, Eo = 0, Eb = 0;
    if (edge) {
        const long long pm = plane(q - 1), pc = plane(q), pn = plane(q + 1);
        Ec[0] = x[pc + e_m1]; Ec[1] = x[pc + e_0]; Ec[2] = x[pc + e_p1];
        Em = x[pm + e_0]; En = x[pn + e_0]; Eo = x[pc + e_o]; Eb = rhs[pc + e_0];
    }

    DV<VEC> Mc[3], Mm, Dprev;                    // L1 on plane q-1 (rows j-1..j+1), L1 of row j on plane q-2, rhs of row j on q-1
#pragma unroll
    for (int a = 0; a < VEC; a++) { Mc[0].c[a] = Mc[1].c[a] = Mc[2].c[a] = 0.0; Mm.c[a] = 0.0; Dprev.c[a] = 0.0; }
    double Xp = 0;                               // L1 of the outside cell on row j, plane q-1

    for (; q <= kend; q++) {
        // prefetch what plane q+1 needs (clamped: past the chunk's last plane the values are not used)
        const long long p1 = plane(q + 1), p2 = plane(q + 2);
        DV<VEC> Ln2[3], Dv2[3];
        Ln2[0] = ldv(x, p2 + o_m1); Ln2[1] = ldv(x, p2 + o_0); Ln2[2] = ldv(x, p2 + o_p1);
        Dv2[0] = ldv(rhs, p1 + o_m1); Dv2[1] = ldv(rhs, p1 + o_0); Dv2[2] = ldv(rhs, p1 + o_p1);
        const DV<VEC> Hf2 = ldv(x, p1 + o_m2), Hb2 = ldv(x, p1 + o_p2);
        double E2[3] = { 0, En, 0 }, En2 = 0, Eo2 = 0, Eb2 = 0;
        if (edge) {
            E2[0] = x[p1 + e_m1]; E2[2] = x[p1 + e_p1];
            En2 = x[p2 + e_0]; Eo2 = x[p1 + e_o]; Eb2 = rhs[p1 + e_0];
        }

        // L1 on plane q for rows j-1, j, j+1
        const bool qb = q <= 0 || q >= nz - 1;
        DV<VEC> M[3];
        M[0] = jac(Lc[0], Hf, Lc[1], Lm[0], Ln[0], Dv[0], Ec[0], qb || rowb_m1);
        M[1] = jac(Lc[1], Lc[0], Lc[2], Lm[1], Ln[1], Dv[1], Ec[1], qb || rowb_0);
        M[2] = jac(Lc[2], Lc[1], Hb, Lm[2], Ln[2], Dv[2], Ec[2], qb || rowb_p1);
        // L1 of the outside cell on row j, plane q (edge lanes; a boundary cell keeps L0)
        double X = Ec[1];
        if (edge && !(qb || rowb_0 || xe_boundary)) {
            const double l = edgeL ? Eo : Lc[1].c[VEC - 1], rr = edgeL ? Lc[1].c[0] : Eo;
            X = ((l + rr + Ec[0] + Ec[2] + Em + En) + alpha * Eb) * beta;
        }

        // L2 on plane q-1 for row j
        const int k = q - 1;
        const DV<VEC> o = jac(Mc[1], Mc[0], Mc[2], Mm, M[1], Dprev, Xp, false);
        if (active && k >= kbeg && k < kend) {
            double *dst = out + x0 + sj * j + sk * k;
            if constexpr (VEC == 2) {
                if (x0 >= 2 && x0 + 2 < nx) *reinterpret_cast<double2 *>(dst) = make_double2(o.c[0], o.c[1]);
                else {                              // touches the x boundary: interior cells only
                    if (x0 >= 1) dst[0] = o.c[0];
                    if (x0 + 1 < nx - 1) dst[1] = o.c[1];
                }
            } else {
                if (x0 >= 1 && x0 < nx - 1) dst[0] = o.c[0];
            }
        }
        // rotate
        Mm = Mc[1]; Dprev = Dv[1];
        Xp = X;
        Em = Ec[1]; En = En2; Eo = Eo2; Eb = Eb2;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            Mc[a] = M[a]; Lm[a] = Lc[a]; Lc[a] = Ln[a]; Ln[a] = Ln2[a]; Dv[a] = Dv2[a];
            Ec[a] = E2[a];
        }
        Hf = Hf2; Hb = Hb2;
    }
}

// ---- two sweeps per launch, lean form (the fp64 twin of jacobi_lean2r_kernel, bq_project.hip) ----------------------
// mg_smooth2_kernel rotates its plane registers by moves, computes 64-bit addresses per load and owns one row per
// thread: 130 us per launch at 256^3 = 402 MB of compulsory traffic at 3.1 TB/s.  This is the two-row kernel of the
// fp32 projection on double2 columns: loads and stores through buffer descriptors (row offset in a VGPR, plane offset
// in an SGPR), plane rings with compile-time indices (loop unrolled 3 + PF times, nothing is moved), a thread owns the
// double2 columns of rows j, j+1 and keeps L0 on rows j-1 .. j+2 (+ j-2, j+3 of the centre plane); rows of 2-4 waves
// (nx = 256 doubles is two waves) let the first / last lane of a wave keep the column just outside the wave in a ring
// of its own.  Every value is mg_smooth_kernel's expression on the same operands.  Preconditions as for
// mg_smooth2_kernel, plus: x-boundary columns are stored with their L0 value (what `out` already holds there).
struct D2 { double a, b; };                                 // two consecutive cells of a row
__device__ __forceinline__ D2 ld_d2(v4i rs, unsigned voff, unsigned soff)
{
    const v2d v = __builtin_bit_cast(v2d, bq_buffer_load_x4(rs, (int)voff, (int)soff, 0));
    return D2{v.x, v.y};
}
__device__ __forceinline__ double ld_d(v4i rs, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, bq_buffer_load_x2(rs, (int)voff, (int)soff, 0));
}
template <int AUX = 0>
__device__ __forceinline__ void st_d2(D2 v, v4i rs, unsigned voff, unsigned soff)
{
    bq_buffer_store_x4(__builtin_bit_cast(v4f, v2d{v.a, v.b}), rs, (int)voff, (int)soff, AUX);
}
// smoothing_jacobi_kernel's expression (:1443-1461) on a double2 column; WIDE: `outside` replaces the neighbour lane's
// value at a wave's first / last lane
template <bool WIDE>
__device__ __forceinline__ D2 jac_d2(D2 ce, D2 fr, D2 bk, D2 dn, D2 up, D2 dv, double alpha, double beta, bool xlo, bool xhi,
                                     double outside, bool edgeL, bool edgeR)
{
    double left = lane_up(ce.b), right = lane_down(ce.a);
    if (WIDE) {
        if (edgeL) left = outside;
        if (edgeR) right = outside;
    }
    D2 o;
    o.a = ((left + ce.b + fr.a + bk.a + dn.a + up.a) + alpha * dv.a) * beta;
    o.b = ((ce.a + right + fr.b + bk.b + dn.b + up.b) + alpha * dv.b) * beta;
    if (xlo) o.a = ce.a;
    if (xhi) o.b = ce.b;
    return o;
}

// ZIN: the input field is all zeros and is not read (V_Cycle's first sweep of a level starts from a cleared x)
template <bool WIDE, int PF, bool ZIN = false>
__global__ __launch_bounds__(256) void mg_lean2r_kernel(const double *__restrict__ p, const double *__restrict__ div,
                                                        double *__restrict__ out, int nx, int ny, int nz,
                                                        int cw, int nby, int kchunk, double alpha, double beta)
{
    constexpr int P = 3 + PF;                                       // ring period
    // PF = 2: loads two planes ahead and streaming stores, the form for arrays that come from HBM
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
    const int xraw = 2 * c, j = 2 * (by * rows + r);
    const int kA = 1, kB = nz - 1;
    const int kbeg = max(kA, bz * kchunk), kend = min(kB, bz * kchunk + kchunk);
    if (kbeg >= kend) return;
    const bool xok = xraw < nx;
    // odd rows: the last lane of a row holds the single boundary column nx-1 in .a (its .b is the next row's first cell:
    // loaded, never used, never stored -- the lane stores nothing, `out` already holds the boundary column)
    const bool xlast = xok && xraw == nx - 1;
    const bool active0 = xok && !xlast && j >= 1 && j <= ny - 2, active1 = xok && !xlast && j + 1 >= 1 && j + 1 <= ny - 2;
    const int x = xok ? xraw : 0;                                   // out-of-range lanes, rows, planes: clamped into the array
    const bool xlo = x == 0 || xlast, xhi = x + 1 == nx - 1;
    const unsigned bytes = (unsigned)nx * (unsigned)ny * (unsigned)nz * 8u;
    const v4i rp = make_rsrc4(p, bytes), rd = make_rsrc4(div, bytes), ro = make_rsrc4(out, bytes);
    unsigned vo[6];                                                 // byte offsets of this thread's column in rows j-2 .. j+3
    bool rowb[4];                                                   // rows j-1 .. j+2 are boundary rows (keep L0)
#pragma unroll
    for (int a = 0; a < 6; a++) vo[a] = ((unsigned)x + (unsigned)nx * (unsigned)min(max(j - 2 + a, 0), ny - 1)) * 8u;
#pragma unroll
    for (int a = 0; a < 4; a++) rowb[a] = j - 1 + a <= 0 || j - 1 + a >= ny - 1;
    // only the first and the last row block can hold a boundary row: everybody else runs the loop without the selects
    const bool edge_block = 2 * by * rows - 1 <= 0 || 2 * (by * rows + rows - 1) + 2 >= ny - 1;
    const unsigned pstride = (unsigned)nx * (unsigned)ny * 8u;
    auto po = [&](int pl) -> unsigned { return pstride * (unsigned)min(max(pl, 0), nz - 1); };
    // WIDE: the first / last lane of a wave looks after the column just outside its wave (xe) -- it needs that column's
    // L0 on rows j-1 .. j+2 (x-neighbour of our own first sweep) and its L1 on rows j, j+1 (x-neighbour of our second
    // sweep), which it evaluates itself from that column's own neighbours (outer x-neighbour xo, rows, planes, div).
    // The other 62 lanes run the same eight scalar loads per plane with an offset beyond the descriptor's range: the range
    // check answers 0 and nothing goes to the caches.
    const int lane = threadIdx.x & 63;
    const bool edgeL = WIDE && lane == 0 && xok && xraw > 0, edgeR = WIDE && lane == 63 && xraw + 2 < nx;
    const bool edge = edgeL || edgeR;
    const int xe = edgeL ? xraw - 1 : (edgeR ? xraw + 2 : x), xo = edgeL ? xe - 1 : (edgeR ? xe + 1 : x);
    const bool xe_boundary = xe <= 0 || xe >= nx - 1;
    unsigned ve[4], vx[2];
#pragma unroll
    for (int a = 0; a < 4; a++) ve[a] = ((unsigned)min(max(xe, 0), nx - 1) + (unsigned)nx * (unsigned)min(max(j - 1 + a, 0), ny - 1)) * 8u;
#pragma unroll
    for (int a = 0; a < 2; a++) vx[a] = ((unsigned)min(max(xo, 0), nx - 1) + (unsigned)nx * (unsigned)min(max(j + a, 0), ny - 1)) * 8u;
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
    D2 L0[P][4], H[P][2], L1[P][4], D[P][4];
    double E[P][4], Eo[P][2], Eb[P][2], X[P][2];
    const D2 zero = D2{0.0, 0.0};
#pragma unroll
    for (int a = 0; a < P; a++) {
#pragma unroll
        for (int bb = 0; bb < 4; bb++) L1[a][bb] = zero;
        X[a][0] = 0.0; X[a][1] = 0.0;
    }
    int q = kbeg - 1;
#pragma unroll
    for (int d = -1; d <= PF; d++) {                                // prologue: planes q-1 .. q+PF
        constexpr int dummy = 0; (void)dummy;
        const int sl_ = (d + P) % P;
        const unsigned pp = po(q + d);
#pragma unroll
        for (int a = 0; a < 4; a++) L0[sl_][a] = ZIN ? zero : ld_d2(rp, vo[a + 1], pp);
        if (WIDE) {
#pragma unroll
            for (int a = 0; a < 4; a++) E[sl_][a] = ZIN ? 0.0 : ld_d(rp, ve[a], pp);
        }
        if (d >= 0 && d < PF) {
#pragma unroll
            for (int a = 0; a < 4; a++) D[sl_][a] = ld_d2(rd, vo[a + 1], pp);
            H[sl_][0] = ZIN ? zero : ld_d2(rp, vo[0], pp); H[sl_][1] = ZIN ? zero : ld_d2(rp, vo[5], pp);
            if (WIDE) {
#pragma unroll
                for (int a = 0; a < 2; a++) { Eo[sl_][a] = ZIN ? 0.0 : ld_d(rp, vx[a], pp); Eb[sl_][a] = ld_d(rd, ve[a + 1], pp); }
            }
        }
    }
#define MG_SL(T, d) (((T) + (d) + P) % P)
#define MG_LEAN_PHASE(T)                                                                                            \
    {                                                                                                               \
        constexpr int im = MG_SL(T, -1), ic = MG_SL(T, 0), in_ = MG_SL(T, 1), ia = MG_SL(T, 1 + PF), ha = MG_SL(T, PF); \
        constexpr int mp = MG_SL(T, -1), mpp = MG_SL(T, -2);                                                        \
        const unsigned pa = po(q + 1 + PF), pb = po(q + PF);                                                        \
        _Pragma("unroll") for (int a = 0; a < 4; a++) { L0[ia][a] = ZIN ? zero : ld_d2(rp, vo[a + 1], pa); D[ha][a] = ld_d2(rd, vo[a + 1], pb); } \
        H[ha][0] = ZIN ? zero : ld_d2(rp, vo[0], pb); H[ha][1] = ZIN ? zero : ld_d2(rp, vo[5], pb);                 \
        if (WIDE) {                                                                                                 \
            _Pragma("unroll") for (int a = 0; a < 4; a++) E[ia][a] = ZIN ? 0.0 : ld_d(rp, ve[a], pa);                  \
            _Pragma("unroll") for (int a = 0; a < 2; a++) { Eo[ha][a] = ZIN ? 0.0 : ld_d(rp, vx[a], pb); Eb[ha][a] = ld_d(rd, ve[a + 1], pb); } \
        }                                                                                                           \
        const bool qb = q < kA || q >= kB;                                                                          \
        if (qb) {                                               /* a boundary plane keeps L0 */                      \
            _Pragma("unroll") for (int a = 0; a < 4; a++) L1[ic][a] = L0[ic][a];                                      \
        } else {                                                                                                    \
            L1[ic][0] = jac_d2<WIDE>(L0[ic][0], H[ic][0], L0[ic][1], L0[im][0], L0[in_][0], D[ic][0], alpha, beta, xlo, xhi, E[ic][0], edgeL, edgeR);   \
            L1[ic][1] = jac_d2<WIDE>(L0[ic][1], L0[ic][0], L0[ic][2], L0[im][1], L0[in_][1], D[ic][1], alpha, beta, xlo, xhi, E[ic][1], edgeL, edgeR);  \
            L1[ic][2] = jac_d2<WIDE>(L0[ic][2], L0[ic][1], L0[ic][3], L0[im][2], L0[in_][2], D[ic][2], alpha, beta, xlo, xhi, E[ic][2], edgeL, edgeR);  \
            L1[ic][3] = jac_d2<WIDE>(L0[ic][3], L0[ic][2], H[ic][1], L0[im][3], L0[in_][3], D[ic][3], alpha, beta, xlo, xhi, E[ic][3], edgeL, edgeR);   \
            if (EDGE) {                                                                                             \
                _Pragma("unroll") for (int a = 0; a < 4; a++)                                                        \
                    if (rowb[a]) L1[ic][a] = L0[ic][a];                                                             \
            }                                                                                                       \
        }                                                                                                           \
        if (WIDE) {                                             /* the outside column's own first sweep, rows j, j+1 */ \
            _Pragma("unroll") for (int rr = 0; rr < 2; rr++) {                                                       \
                const double own = edgeL ? L0[ic][rr + 1].a : L0[ic][rr + 1].b;                                       \
                const double l = edgeL ? Eo[ic][rr] : own, rg2 = edgeL ? own : Eo[ic][rr];                           \
                const double v = (l + rg2 + E[ic][rr] + E[ic][rr + 2] + E[im][rr + 1] + E[in_][rr + 1] + alpha * Eb[ic][rr]) * beta; \
                const bool keep = qb || xe_boundary || (j + rr <= 0 || j + rr >= ny - 1);                           \
                X[ic][rr] = keep ? E[ic][rr + 1] : v;                                                               \
            }                                                                                                       \
        }                                                                                                           \
        const int k = q - 1;                                                                                        \
        if (k >= kbeg && k < kend) {                                                                                \
            const D2 o0 = jac_d2<WIDE>(L1[mp][1], L1[mp][0], L1[mp][2], L1[mpp][1], L1[ic][1], D[mp][1], alpha, beta, xlo, xhi, X[mp][0], edgeL, edgeR); \
            const D2 o1 = jac_d2<WIDE>(L1[mp][2], L1[mp][1], L1[mp][3], L1[mpp][2], L1[ic][2], D[mp][2], alpha, beta, xlo, xhi, X[mp][1], edgeL, edgeR); \
            const unsigned pk = pstride * (unsigned)k;                                                              \
            if (active0) st_d2<ST>(o0, ro, vo[2], pk);                                                              \
            if (active1) st_d2<ST>(o1, ro, vo[3], pk);                                                              \
        }                                                                                                           \
        q++;                                                                                                        \
    }
    while (true) {
        MG_LEAN_PHASE(0)
        if (q > kend) break;
        MG_LEAN_PHASE(1)
        if (q > kend) break;
        MG_LEAN_PHASE(2)
        if (q > kend) break;
        MG_LEAN_PHASE(3)
        if (q > kend) break;
        if constexpr (P > 4) {
            MG_LEAN_PHASE(4)
            if (q > kend) break;
        }
        if constexpr (P > 5) {
            MG_LEAN_PHASE(5)
            if (q > kend) break;
        }
        if constexpr (P > 6) {
            MG_LEAN_PHASE(6)
            if (q > kend) break;
        }
    }
    };
    if (edge_block) run(std::true_type{}); else run(std::false_type{});
#undef MG_LEAN_PHASE
}


// ---- THREE sweeps per launch, the intermediate levels' neighbour rows exchanged through LDS ------------------------------
// The fp64 twin of jacobi_lds_kernel<W, 1, 3> (bq_project.hip): a block is W output waves + two halo waves at either end, a
// wave owns ONE row; per plane step it evaluates level 1 on its row (neighbour rows: global loads) and puts it into LDS,
// level 2 one plane behind (neighbour rows of level 1 out of LDS), puts that, level 3 two planes behind and stores it; one
// barrier per step, LDS double-buffered by plane parity.  Rows of 130 .. 256 doubles: lane l holds the cells 2l, 2l+1
// (segment A) AND 128 + 2l, 128 + 2l + 1 (segment B) of its row, so every load is a fully coalesced 16-byte column and the
// row needs no second wave; the x-neighbours across the segment seam travel by wave rotation (lane 63 receives lane 0's
// segment-B cell, lane 0 lane 63's segment-A cell).  Every value is mg_smooth_kernel's expression on the same operands
// (alpha * b is multiplied once and reused by the three levels: the same product); boundary rows / planes / columns keep
// their input through all levels.  Preconditions as for mg_lean2r_kernel (both buffers carry the same boundary layer).
struct D4 { D2 a, b; };                                     // segment A and segment B of a row (two cells each)
__device__ __forceinline__ double lane_rol(double x)        // lane l receives lane l+1's x, lane 63 lane 0's (wave_rol:1)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = (int)b, hi = (int)(b >> 32);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, 0x134, 0xf, 0xf, false);
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xf, 0xf, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)h2 << 32) | l2));
}
__device__ __forceinline__ double lane_ror(double x)        // lane l receives lane l-1's x, lane 0 lane 63's (wave_ror:1)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = (int)b, hi = (int)(b >> 32);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, 0x13C, 0xf, 0xf, false);
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xf, 0xf, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)h2 << 32) | l2));
}
// smoothing_jacobi_kernel's expression (:1443-1461) on the four cells of a lane; adv = alpha * b
__device__ __forceinline__ D4 jac_d4(D4 ce, D4 fr, D4 bk, D4 dn, D4 up, D4 adv, double beta, bool xlo, bool xhi, bool first, bool last)
{
    const double leftA = lane_up(ce.a.b);
    const double rightA = lane_rol(first ? ce.b.a : ce.a.a);    // lane 63: cell 128, which lane 0 holds in segment B
    const double leftB = lane_ror(last ? ce.a.b : ce.b.b);      // lane 0: cell 127, which lane 63 holds in segment A
    const double rightB = lane_down(ce.b.a);
    D4 o;
    o.a.a = ((leftA + ce.a.b + fr.a.a + bk.a.a + dn.a.a + up.a.a) + adv.a.a) * beta;
    o.a.b = ((ce.a.a + rightA + fr.a.b + bk.a.b + dn.a.b + up.a.b) + adv.a.b) * beta;
    o.b.a = ((leftB + ce.b.b + fr.b.a + bk.b.a + dn.b.a + up.b.a) + adv.b.a) * beta;
    o.b.b = ((ce.b.a + rightB + fr.b.b + bk.b.b + dn.b.b + up.b.b) + adv.b.b) * beta;
    if (xlo) o.a.a = ce.a.a;
    if (xhi) o.b.b = ce.b.b;
    return o;
}

// mg_residual_kernel's expression (update_residual_kernel, :1251-1261) on the four cells of a lane
__device__ __forceinline__ D4 res_d4(D4 ce, D4 fr, D4 bk, D4 dn, D4 up, D4 b, bool first, bool last)
{
    const double leftA = lane_up(ce.a.b);
    const double rightA = lane_rol(first ? ce.b.a : ce.a.a);
    const double leftB = lane_ror(last ? ce.a.b : ce.b.b);
    const double rightB = lane_down(ce.b.a);
    D4 o;
    o.a.a = b.a.a - ((leftA + ce.a.b + fr.a.a + bk.a.a + dn.a.a + up.a.a) - ce.a.a * 6);
    o.a.b = b.a.b - ((ce.a.a + rightA + fr.a.b + bk.a.b + dn.a.b + up.a.b) - ce.a.b * 6);
    o.b.a = b.b.a - ((leftB + ce.b.b + fr.b.a + bk.b.a + dn.b.a + up.b.a) - ce.b.a * 6);
    o.b.b = b.b.b - ((ce.b.a + rightB + fr.b.b + bk.b.b + dn.b.b + up.b.b) - ce.b.b * 6);
    return o;
}

// mg_poisson_kernel's expression (calc_poisson_kernel, :1075-1085) on the four cells of a lane
__device__ __forceinline__ D4 poi_d4(D4 ce, D4 fr, D4 bk, D4 dn, D4 up, bool first, bool last)
{
    const double leftA = lane_up(ce.a.b);
    const double rightA = lane_rol(first ? ce.b.a : ce.a.a);
    const double leftB = lane_ror(last ? ce.a.b : ce.b.b);
    const double rightB = lane_down(ce.b.a);
    D4 o;
    o.a.a = (leftA + ce.a.b + fr.a.a + bk.a.a + dn.a.a + up.a.a) - ce.a.a * 6;
    o.a.b = (ce.a.a + rightA + fr.a.b + bk.a.b + dn.a.b + up.a.b) - ce.a.b * 6;
    o.b.a = (leftB + ce.b.b + fr.b.a + bk.b.a + dn.b.a + up.b.a) - ce.b.a * 6;
    o.b.b = (ce.b.a + rightB + fr.b.b + bk.b.b + dn.b.b + up.b.b) - ce.b.b * 6;
    return o;
}

// RES (with S = 3): the third level is not a sweep but the RESIDUAL of the second -- rout = b - A x'' on interior cells
// (mg_residual_kernel's expression), x'' itself is stored to `out` as the level is made: the last launch of V_Cycle's 32 sweeps
// and the residual that follows it in one pass over the arrays.
template <int W, int S, bool ZIN, bool RES = false>
__global__ __launch_bounds__((W + 2 * (S - 1)) * 64) void mg_lds3_kernel(const double *__restrict__ p, const double *__restrict__ div,
                                                               double *__restrict__ out, int nx, int ny, int nz,
                                                               int nby, int nblk, int kchunk, double alpha, double beta,
                                                               double *__restrict__ rout = nullptr)
{
    static_assert(S == 2 || S == 3, "two or three sweeps per launch");
    static_assert(!RES || (S == 3 && !ZIN), "the residual is the third level of a two-sweep launch");
    constexpr int H = S - 1, NW = W + 2 * H, P = 4;
    // [level][plane parity][row slot][segment][lane]; level 0 is the INPUT: a wave loads its own row only and takes the two
    // neighbouring rows of the centre plane out of LDS like those of the later levels (three row loads fewer per plane, 48
    // registers fewer: twelve waves per block fit three to a SIMD).  The outermost halo waves have no neighbour wave on their far
    // side: they fetch that one row from memory themselves.
    __shared__ v2d lds[S][2][NW][2][64];
    const int per = (int)gridDim.x >> 3;                            // XCD-contiguous block order (grid padded to 8 k blocks)
    const int b = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (b >= nblk) return;
    const int by = b % nby, bz = b / nby;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int jb = by * W;
    const int j = jb + (wv - H);                                    // this wave's row (halo waves: outside the block)
    const bool halo = wv < H || wv >= NW - H;
    const int dn = wv < H ? H - wv : (wv >= NW - H ? wv - (NW - H) + 1 : 0);
    const int smax = S - dn;                                        // a halo wave dn rows outside owes levels 1 .. S - dn
    const bool low_end = wv == 0;                                   // (of the two outermost waves)
    const int kA = 1, kB = nz - 1;
    const int kbeg = max(kA, bz * kchunk), kend = min(kB, bz * kchunk + kchunk);
    if (kbeg >= kend) return;                                       // (block-uniform)
    const int xA = 2 * lane, xBraw = 128 + 2 * lane;
    const bool okB = xBraw < nx;
    const int xB = okB ? xBraw : nx - 2;                            // out-of-range lanes, rows, planes: clamped into the array
    const bool xlo = lane == 0, xhi = okB && xBraw + 1 == nx - 1;
    const bool first = lane == 0, last = lane == 63;
    const bool row_in = !halo && j >= 1 && j <= ny - 2;
    const bool rowb = j <= 0 || j >= ny - 1;
    const unsigned bytes = (unsigned)nx * (unsigned)ny * (unsigned)nz * 8u;
    const v4i rp = make_rsrc4(p, bytes), rd = make_rsrc4(div, bytes), ro = make_rsrc4(out, bytes), rr = make_rsrc4(RES ? rout : out, bytes);
    // byte offsets of this lane's columns in its own row and (outermost halo waves) in the row on the far side
    const unsigned row_own = (unsigned)nx * (unsigned)min(max(j, 0), ny - 1);
    const unsigned row_far = (unsigned)nx * (unsigned)min(max(low_end ? j - 1 : j + 1, 0), ny - 1);
    const unsigned voA = ((unsigned)xA + row_own) * 8u, voB = ((unsigned)xB + row_own) * 8u;
    const unsigned vfA = ((unsigned)xA + row_far) * 8u, vfB = ((unsigned)xB + row_far) * 8u;
    const bool edge_block = jb - H <= 0 || jb + W + H - 1 >= ny - 1;
    const unsigned pstride = (unsigned)nx * (unsigned)ny * 8u;
    auto po = [&](int pl) -> unsigned { return pstride * (unsigned)min(max(pl, 0), nz - 1); };
    const int rlo = max(wv - 1, 0), rhi = min(wv + 1, NW - 1);
    auto put = [&](v2d (*buf)[2][64], D4 v) { buf[wv][0][lane] = v2d{v.a.a, v.a.b}; buf[wv][1][lane] = v2d{v.b.a, v.b.b}; };
    auto get = [&](v2d (*buf)[2][64], int r) -> D4 { const v2d u = buf[r][0][lane], w = buf[r][1][lane]; return D4{D2{u.x, u.y}, D2{w.x, w.y}}; };
    const D4 zero = D4{D2{0.0, 0.0}, D2{0.0, 0.0}};
    auto adv = [&](D4 d) -> D4 { return RES ? D4{D2{alpha * d.a.a, alpha * d.a.b}, D2{alpha * d.b.a, alpha * d.b.b}} : d; };
    auto ld_own = [&](v4i rs, unsigned pp) -> D4 { return D4{ld_d2(rs, voA, pp), ld_d2(rs, voB, pp)}; };
    auto ld_far = [&](unsigned pp) -> D4 { return D4{ld_d2(rp, vfA, pp), ld_d2(rp, vfB, pp)}; };

    auto run = [&](auto EDGE_T, auto SM_T) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(EDGE_T)::value;
    constexpr int SM = decltype(SM_T)::value;                       // the levels this wave evaluates
    constexpr bool OUTER = SM == 1;                                 // an outermost halo wave (dn == H)
    D4 L0[P], Lx[P], D[P], L1[P], L2[P];
#pragma unroll
    for (int a = 0; a < P; a++) { D[a] = zero; L1[a] = zero; L2[a] = zero; Lx[a] = zero; }
    int q = kbeg - (S - 1);
#define MG_SL4(T, d) ((((T) + (d)) % P + P) % P)
#pragma unroll
    for (int d = -1; d <= 1; d++) {                                 // prologue: planes q-1, q, q+1 of x; b of plane q
        const int sl_ = MG_SL4(0, d);
        const unsigned pp = po(q + d);
        L0[sl_] = ZIN ? zero : ld_own(rp, pp);
        if (OUTER && d >= 0) Lx[sl_] = ZIN ? zero : ld_far(pp);
        if (d == 0) D[sl_] = ld_own(rd, pp);
    }
    if (!ZIN) { put(lds[0][q & 1], L0[MG_SL4(0, 0)]); __syncthreads(); }    // the first step's centre plane for the neighbours
#define MG_LDS_PHASE(T)                                                                                             \
    {                                                                                                               \
        constexpr int im = MG_SL4(T, -1), ic = MG_SL4(T, 0), in_ = MG_SL4(T, 1), ia = MG_SL4(T, 2);                   \
        const unsigned pa = po(q + 2), pb = po(q + 1);                                                              \
        L0[ia] = ZIN ? zero : ld_own(rp, pa);                                                                       \
        if (OUTER) Lx[ia] = ZIN ? zero : ld_far(pa);                                                                \
        D[in_] = ld_own(rd, pb);                                                                                    \
        D4 fr0 = zero, bk0 = zero, nlo2 = zero, nhi2 = zero, nlo3 = zero, nhi3 = zero;                              \
        if (!ZIN) {                                                                                                 \
            if (!OUTER || !low_end) fr0 = get(lds[0][q & 1], rlo);                                                  \
            if (!OUTER || low_end) bk0 = get(lds[0][q & 1], rhi);                                                   \
            if (OUTER) { if (low_end) fr0 = Lx[ic]; else bk0 = Lx[ic]; }                                            \
        }                                                                                                           \
        if (!ZIN) put(lds[0][(q + 1) & 1], L0[in_]);            /* the next step's centre plane */                   \
        /* first sweep on plane q */                                                                                \
        if (q < kA || q >= kB) {                                                                                    \
            L1[ic] = L0[ic];                                                                                        \
        } else {                                                                                                    \
            if (!RES) {     /* (RES keeps b itself for the residual and multiplies at each use: the same product) */ \
                D[ic].a.a = alpha * D[ic].a.a; D[ic].a.b = alpha * D[ic].a.b;                                       \
                D[ic].b.a = alpha * D[ic].b.a; D[ic].b.b = alpha * D[ic].b.b;                                       \
            }                                                                                                       \
            L1[ic] = jac_d4(L0[ic], fr0, bk0, L0[im], L0[in_], adv(D[ic]), beta, xlo, xhi, first, last);            \
            if (EDGE && rowb) L1[ic] = L0[ic];                                                                      \
        }                                                                                                           \
        /* (the later levels' neighbour rows are fetched level by level: six rows in flight at once cost 25 registers too many) */ \
        if (SM >= 2) { nlo2 = get(lds[1][(q - 1) & 1], rlo); nhi2 = get(lds[1][(q - 1) & 1], rhi); }                \
        put(lds[1][q & 1], L1[ic]);                                                                                 \
        if (SM >= 2) {      /* second sweep on plane q - 1 */                                                       \
            constexpr int cs = MG_SL4(T, -1), us = MG_SL4(T, 0), ds = MG_SL4(T, -2);                                 \
            const int ps = q - 1;                                                                                   \
            D4 v = jac_d4(L1[cs], nlo2, nhi2, L1[ds], L1[us], adv(D[cs]), beta, xlo, xhi, first, last);             \
            if (ps < kA || ps >= kB || (EDGE && rowb)) v = L1[cs];                                                  \
            if constexpr (S == 2 || RES) {     /* the last sweep: store (RES: planes one beyond the chunk feed the residual only) */ \
                if (ps >= kbeg && ps < kend && row_in) {                                                            \
                    const unsigned pk = pstride * (unsigned)ps;                                                     \
                    st_d2<2>(v.a, ro, voA, pk);                                                                     \
                    if (okB) st_d2<2>(v.b, ro, voB, pk);                                                            \
                }                                                                                                   \
            }                                                                                                       \
            if constexpr (S == 3) {                                                                                 \
                L2[cs] = v;                                                                                         \
                if (SM >= 3) { nlo3 = get(lds[S - 1][(q - 2) & 1], rlo); nhi3 = get(lds[S - 1][(q - 2) & 1], rhi); } \
                put(lds[S - 1][ps & 1], v);                                                                         \
            }                                                                                                       \
        }                                                                                                           \
        if (S >= 3 && SM >= 3) {      /* third sweep on plane q - 2 */                                                        \
            constexpr int cs = MG_SL4(T, -2), us = MG_SL4(T, -1), ds = MG_SL4(T, -3);                                \
            const int ps = q - 2;                                                                                   \
            if (ps >= kbeg && ps < kend) {                                                                          \
                if constexpr (!RES) {                                                                               \
                    D4 v = jac_d4(L2[cs], nlo3, nhi3, L2[ds], L2[us], D[cs], beta, xlo, xhi, first, last);          \
                    if (EDGE && rowb) v = L2[cs];                                                                   \
                    if (row_in) {                                                                                   \
                        const unsigned pk = pstride * (unsigned)ps;                                                 \
                        st_d2<2>(v.a, ro, voA, pk);                                                                 \
                        if (okB) st_d2<2>(v.b, ro, voB, pk);                                                        \
                    }                                                                                               \
                } else if (row_in) {    /* r = b - ((l + r + f + b + d + u) - 6 c): interior cells only */          \
                    const D4 rv = res_d4(L2[cs], nlo3, nhi3, L2[ds], L2[us], D[cs], first, last);                   \
                    const unsigned pk = pstride * (unsigned)ps;                                                     \
                    if (!xlo) st_d2<2>(rv.a, rr, voA, pk);                                                          \
                    else bq_buffer_store_x2(__builtin_bit_cast(v2f, rv.a.b), rr, (int)(voA + 8u), (int)pk, 2);      \
                    if (okB) {                                                                                      \
                        if (!xhi) st_d2<2>(rv.b, rr, voB, pk);                                                      \
                        else bq_buffer_store_x2(__builtin_bit_cast(v2f, rv.b.a), rr, (int)voB, (int)pk, 2);         \
                    }                                                                                               \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        __syncthreads();                                                                                            \
        q++;                                                                                                        \
    }
    while (true) {
        MG_LDS_PHASE(0)
        if (q > kend + S - 2) break;
        MG_LDS_PHASE(1)
        if (q > kend + S - 2) break;
        MG_LDS_PHASE(2)
        if (q > kend + S - 2) break;
        MG_LDS_PHASE(3)
        if (q > kend + S - 2) break;
    }
    };
    auto go = [&](auto E) __attribute__((always_inline)) {
        if (smax >= S) run(E, std::integral_constant<int, S>{});
        else if (smax == 1) run(E, std::integral_constant<int, 1>{});
        else run(E, std::integral_constant<int, 2>{});
    };
    if (edge_block) go(std::true_type{}); else go(std::false_type{});
#undef MG_LDS_PHASE
#undef MG_SL4
}

// ---- residual / A x in the marching form ------------------------------------------------------------------------------
// mg_residual_kernel and mg_poisson_kernel give every cell a thread that loads its seven points: 3.8 TB/s at 256^3.  Here
// a thread owns the double2 columns of rows j, j+1 as in mg_lean2r_kernel and marches along z with x of the planes q-1, q,
// q+1 in a ring: per plane it loads x on rows j-1 .. j+2 of the new plane (the outer two for the y neighbours of the plane
// being finished... one plane later) -- 4 + 2 (rhs) loads and 2 stores for four cells.  POISSON: out = A x (calc_poisson_kernel,
// :1075-1085), else out = rhs - A x (update_residual_kernel, :1251-1261); interior cells only, boundary cells of `out` are
// not touched (lanes on an x boundary store the one interior cell of their pair).
template <bool WIDE, bool POISSON>
__global__ __launch_bounds__(256) void mg_stencil_lean_kernel(const double *__restrict__ p, const double *__restrict__ rhs,
                                                              double *__restrict__ out, int nx, int ny, int nz,
                                                              int cw, int nby, int kchunk)
{
    const int nblk = gridDim.x;
    int b = blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);      // XCD-contiguous block order
    const int by = b % nby, bz = b / nby;
    const int rows = 256 / cw;
    const int c = threadIdx.x % cw;
    const int r = cw >= 64 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / cw) : (int)threadIdx.x / cw;
    const int xraw = 2 * c, j = 2 * (by * rows + r);
    const int kbeg = max(1, bz * kchunk), kend = min(nz - 1, bz * kchunk + kchunk);
    if (kbeg >= kend) return;
    const bool xok = xraw < nx;
    const bool xlast = xok && xraw == nx - 1;                       // odd rows: the lane holds the boundary column alone
    const int x = xok ? xraw : 0;
    const bool xlo = x == 0, xhi = x + 1 == nx - 1;
    const bool row0 = xok && !xlast && j >= 1 && j <= ny - 2, row1 = xok && !xlast && j + 1 >= 1 && j + 1 <= ny - 2;
    const unsigned bytes = (unsigned)nx * (unsigned)ny * (unsigned)nz * 8u;
    const v4i rp = make_rsrc4(p, bytes), rd = make_rsrc4(rhs, bytes), ro = make_rsrc4(out, bytes);
    unsigned vo[4];                                                 // rows j-1 .. j+2 (clamped into the array)
#pragma unroll
    for (int a = 0; a < 4; a++) vo[a] = ((unsigned)x + (unsigned)nx * (unsigned)min(max(j - 1 + a, 0), ny - 1)) * 8u;
    const unsigned pstride = (unsigned)nx * (unsigned)ny * 8u;
    auto po = [&](int pl) -> unsigned { return pstride * (unsigned)min(max(pl, 0), nz - 1); };
    // WIDE: the x neighbour of a wave's first / last lane is the column just outside the wave, rows j, j+1
    const int lane = threadIdx.x & 63;
    const bool edgeL = WIDE && lane == 0 && xok && xraw > 0, edgeR = WIDE && lane == 63 && xraw + 2 < nx;
    const int xe = edgeL ? xraw - 1 : (edgeR ? xraw + 2 : x);
    unsigned ve[2];
#pragma unroll
    for (int a = 0; a < 2; a++) ve[a] = (edgeL || edgeR) ? ((unsigned)min(max(xe, 0), nx - 1) + (unsigned)nx * (unsigned)min(max(j + a, 0), ny - 1)) * 8u : 0x80000000u;

    // rings over planes (compile-time slots, the loop is unrolled four times): X[.][0..3] = x on rows j-1 .. j+2 of the
    // planes q-1, q, q+1 and the one arriving; R = rhs rows j, j+1 and E = outside column of plane q and the one arriving
    D2 X[4][4], R[2][2];
    double E[2][2];
    int q = kbeg;
    {
        const unsigned pm = po(q - 1), pc = po(q), pn = po(q + 1);
#pragma unroll
        for (int a = 0; a < 4; a++) { X[3][a] = ld_d2(rp, vo[a], pm); X[0][a] = ld_d2(rp, vo[a], pc); X[1][a] = ld_d2(rp, vo[a], pn); }
        if (!POISSON) { R[0][0] = ld_d2(rd, vo[1], pc); R[0][1] = ld_d2(rd, vo[2], pc); }
        if (WIDE) { E[0][0] = ld_d(rp, ve[0], pc); E[0][1] = ld_d(rp, ve[1], pc); }
    }
#define MG_ST_PHASE(T)                                                                                              \
    {                                                                                                               \
        constexpr int im = (T + 3) & 3, ic = T, in_ = (T + 1) & 3, ia = (T + 2) & 3, rc = T & 1, rn = (T + 1) & 1;     \
        const unsigned pa = po(q + 2), pb = po(q + 1);                                                              \
        _Pragma("unroll") for (int a = 0; a < 4; a++) X[ia][a] = ld_d2(rp, vo[a], pa);                               \
        if (!POISSON) { R[rn][0] = ld_d2(rd, vo[1], pb); R[rn][1] = ld_d2(rd, vo[2], pb); }                         \
        if (WIDE) { E[rn][0] = ld_d(rp, ve[0], pb); E[rn][1] = ld_d(rp, ve[1], pb); }                               \
        _Pragma("unroll") for (int rr = 0; rr < 2; rr++) {                                                           \
            const D2 ce = X[ic][rr + 1], fr = X[ic][rr], bk = X[ic][rr + 2], dn = X[im][rr + 1], up = X[in_][rr + 1]; \
            double left = lane_up(ce.b), right = lane_down(ce.a);                                                   \
            if (WIDE) { if (edgeL) left = E[rc][rr]; if (edgeR) right = E[rc][rr]; }                                \
            D2 o;                                                                                                   \
            o.a = (left + ce.b + fr.a + bk.a + dn.a + up.a) - ce.a * 6;                                             \
            o.b = (ce.a + right + fr.b + bk.b + dn.b + up.b) - ce.b * 6;                                            \
            if (!POISSON) { o.a = R[rc][rr].a - o.a; o.b = R[rc][rr].b - o.b; }                                     \
            if (rr == 0 ? row0 : row1) {                                                                            \
                const unsigned pk = pstride * (unsigned)q;                                                          \
                if (!xlo && !xhi) st_d2<0>(o, ro, vo[rr + 1], pk);                                                  \
                else if (xlo && !xhi) bq_buffer_store_x2(__builtin_bit_cast(v2f, o.b), ro, (int)(vo[rr + 1] + 8u), (int)pk, 0); \
                else if (xhi && !xlo) bq_buffer_store_x2(__builtin_bit_cast(v2f, o.a), ro, (int)vo[rr + 1], (int)pk, 0); \
            }                                                                                                       \
        }                                                                                                           \
        q++;                                                                                                        \
    }
    while (true) {
        MG_ST_PHASE(0)
        if (q >= kend) break;
        MG_ST_PHASE(1)
        if (q >= kend) break;
        MG_ST_PHASE(2)
        if (q >= kend) break;
        MG_ST_PHASE(3)
        if (q >= kend) break;
    }
#undef MG_ST_PHASE
}

// ---- S sweeps per launch on the coarse levels (LDS tiles) ---------------------------------------------
// Levels 1 .. 5 of a 256^3 V-cycle (127^3 .. 7^3) are 176 launches of mg_smooth_kernel per cycle, 2 - 12 us each and
// mostly launch latency: 43 ms of a 229 ms step.  Here a workgroup stages a 16^3 region of x in LDS (thread (tx, ty)
// owns the z-column at (tx, ty): x/y neighbours come from LDS, z neighbours from its own registers, rhs stays in
// registers), runs S sweeps on it and stores the central (16 - 2S)^3 cells -- the cells whose S-sweep dependency cone
// lies inside the region.  Cells outside the array read 0 and, like the array's boundary cells, are never updated, so
// the cone argument only has to hold towards region faces that lie inside the array.  Every value is produced by
// mg_smooth_kernel's expression on the same operands: S launches of that kernel give the same bits.
// Precondition as for mg_smooth2_kernel: in and out carry the same boundary layer.  V_Cycle guarantees it by clearing
// both buffers first; ZIN / ZB fold those clears in: ZIN = the input is all zeros (x is not read), ZB = the boundary
// cells of `out` are written (0 -- what the cleared buffer holds there) along with the interior.
template <int S, bool ZIN, bool ZB>
__global__ __launch_bounds__(256) void mg_smooth_tile_kernel(const double *__restrict__ x, const double *__restrict__ rhs,
                                                             double *__restrict__ out, double alpha, double beta, int ni, int nj, int nk)
{
    constexpr int R = 16, T = R - 2 * S;
    __shared__ double xs[R * R * R];                                                     // [z][y][x]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int gx = (int)blockIdx.x * T - S + tx, gy = (int)blockIdx.y * T - S + ty, gz0 = (int)blockIdx.z * T - S;
    const bool inxy = gx >= 0 && gx < ni && gy >= 0 && gy < nj;
    // swept: interior of the array and not on a face of the region (its four x/y neighbours are in the region)
    const bool sweptxy = gx >= 1 && gx <= ni - 2 && gy >= 1 && gy <= nj - 2 && tx >= 1 && tx <= R - 2 && ty >= 1 && ty <= R - 2;
    const int sk = ni * nj;
    // every address is clamped into the array: what a clamped load returns is replaced by 0 (x) or never used (rhs)
    const int colc = min(max(gx, 0), ni - 1) + ni * min(max(gy, 0), nj - 1);
    double v[R], bb[R];
    unsigned upd = 0;                                                                    // bit z: this cell is swept
#pragma unroll
    for (int z = 0; z < R; z++) {
        const int gz = gz0 + z;
        const int idc = colc + sk * min(max(gz, 0), nk - 1);
        if (!ZIN) { const double t = x[idc]; v[z] = (inxy && gz >= 0 && gz < nk) ? t : 0.0; } else v[z] = 0.0;
        bb[z] = (z >= 1 && z <= R - 2) ? rhs[idc] : 0.0;
        if (sweptxy && z >= 1 && z <= R - 2 && gz >= 1 && gz <= nk - 2) upd |= 1u << z;
    }
    double *mine = xs + (ty * R + tx);
    if (!ZIN) {
#pragma unroll
        for (int z = 0; z < R; z++) mine[z * R * R] = v[z];
    }
#pragma unroll
    for (int s = 0; s < S; s++) {
        if (ZIN && s == 0) {
            // a sweep of the zero field: ((0 + 0 + 0 + 0 + 0 + 0) + alpha b) beta, no neighbour needed
#pragma unroll
            for (int z = 1; z < R - 1; z++) {
                const double t = ((0.0 + 0.0 + 0.0 + 0.0 + 0.0 + 0.0) + alpha * bb[z]) * beta;
                v[z] = ((upd >> z) & 1u) ? t : v[z];
            }
        } else {
            __syncthreads();
            // No branch per cell: threads on a region face read their neighbours' slots of the adjacent row / plane
            // (inside xs for 1 <= z <= 14) and discard the result
            double below = v[0];                                                         // the old value of the cell underneath
#pragma unroll
            for (int z = 1; z < R - 1; z++) {
                const double *c = mine + z * R * R;
                const double cur = v[z];
                const double t = ((c[-1] + c[1] + c[-R] + c[R] + below + v[z + 1]) + alpha * bb[z]) * beta;
                v[z] = ((upd >> z) & 1u) ? t : cur;
                below = cur;
            }
        }
        if (s + 1 < S) {
            if (!(ZIN && s == 0)) __syncthreads();                                       // everybody has read the old values
#pragma unroll
            for (int z = (ZIN && s == 0) ? 0 : 1; z < ((ZIN && s == 0) ? R : R - 1); z++) mine[z * R * R] = v[z];
        }
    }
    if (tx < S || tx >= R - S || ty < S || ty >= R - S || !inxy) return;
    const int col = gx + ni * gy;
#pragma unroll
    for (int z = S; z < R - S; z++) {
        const int gz = gz0 + z;
        if (gz < 0 || gz >= nk) continue;
        if ((upd >> z) & 1u) out[col + sk * gz] = v[z];
        else if (ZB) out[col + sk * gz] = 0.0;
    }
}

// GPU_kernel.cu:22-25 on float operands (M2)
__device__ __forceinline__ float lerp_f(float a, float b, float c)
{
    const float cb = c * b;
    return (float)((1.0 - (double)c) * (double)a + (double)cb);
}

// sample_buffer<double> (:1527-1549): no clamping (M4); `count` = elements of the array.  IDX: the index type -- int for
// arrays below 2^31 elements (every flat index of a sample then fits, negative ones included), long long otherwise
template <typename IDX>
__device__ __forceinline__ double sample_t(const double *b, int nx, int ny, IDX count, float px, float py, float pz)
{
    const int i = (int)floorf(px), j = (int)floorf(py), k = (int)floorf(pz);
    const float a = px - (float)i, bb = py - (float)j, c = pz - (float)k;
    const IDX sj = nx, sk = (IDX)nx * ny;
    const IDX base = (IDX)i + sj * j + sk * k;
    // the eight loads are issued unconditionally from addresses clamped into the array and the out-of-array ones replaced
    // by 0 afterwards: a branch per load (what `cond ? b[id] : 0` compiles to) serialises eight memory latencies
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const IDX id = base + (q & 1) + ((q >> 1) & 1) * sj + ((q >> 2) & 1) * sk;
        const IDX idc = id < 0 ? 0 : (id < count ? id : count - 1);
        const float t = (float)b[idc];
        v[q] = (id >= 0 && id < count) ? t : 0.f;
    }
    return (double)lerp_f(lerp_f(lerp_f(v[0], v[1], a), lerp_f(v[2], v[3], a), bb),
                          lerp_f(lerp_f(v[4], v[5], a), lerp_f(v[6], v[7], a), bb), c);
}

// the trilinear sample of sample_t from a 3x3x3 block of cells held in registers (restriction: the eight samples of a
// coarse cell overlap in 27 fine cells); (ox, oy, oz) in {0, 1}: which 2x2x2 corner of the block
__device__ __forceinline__ double sample_block(const float (&blk)[3][3][3], int ox, int oy, int oz, float a, float bb, float c)
{
    return (double)lerp_f(lerp_f(lerp_f(blk[oz][oy][ox], blk[oz][oy][ox + 1], a), lerp_f(blk[oz][oy + 1][ox], blk[oz][oy + 1][ox + 1], a), bb),
                          lerp_f(lerp_f(blk[oz + 1][oy][ox], blk[oz + 1][oy][ox + 1], a), lerp_f(blk[oz + 1][oy + 1][ox], blk[oz + 1][oy + 1][ox + 1], a), bb), c);
}

// restriction_kernel, double (:1551-1603): every coarse cell = mean of 8 samples at the fine cell centres
__global__ __launch_bounds__(256) void mg_restrict_kernel(const double *__restrict__ residual, double *__restrict__ coarse,
                                                          int ni, int nj, int nk, int ci, int cj, int ck)
{
    MG_IJK(ci, cj, ck)
    const long long count = (long long)ni * nj * nk;
    const float x0 = (float)(2 * i + 0.5), x1 = (float)(2 * i + 1.5);
    const float y0 = (float)(2 * j + 0.5), y1 = (float)(2 * j + 1.5);
    const float z0 = (float)(2 * k + 0.5), z1 = (float)(2 * k + 1.5);
    // sample_t's cell and weights for each of the six coordinates; the eight samples read the 3x3x3 block of fine cells
    // that starts at (floor x0, floor y0, floor z0) -- floor x1 = floor x0 + 1 for every index a float holds exactly
    const int i0 = (int)floorf(x0), j0 = (int)floorf(y0), k0 = (int)floorf(z0);
    const int i1 = (int)floorf(x1), j1 = (int)floorf(y1), k1 = (int)floorf(z1);
    const float ax0 = x0 - (float)i0, ax1 = x1 - (float)i1, ay0 = y0 - (float)j0, ay1 = y1 - (float)j1, az0 = z0 - (float)k0, az1 = z1 - (float)k1;
    if (i1 != i0 + 1 || j1 != j0 + 1 || k1 != k0 + 1) {            // (grids beyond 2^22 cells per axis: the general form)
        const double v0 = sample_t<long long>(residual, ni, nj, count, x0, y0, z0), v1 = sample_t<long long>(residual, ni, nj, count, x0, y0, z1);
        const double v2 = sample_t<long long>(residual, ni, nj, count, x0, y1, z0), v3 = sample_t<long long>(residual, ni, nj, count, x0, y1, z1);
        const double v4 = sample_t<long long>(residual, ni, nj, count, x1, y0, z0), v5 = sample_t<long long>(residual, ni, nj, count, x1, y0, z1);
        const double v6 = sample_t<long long>(residual, ni, nj, count, x1, y1, z0), v7 = sample_t<long long>(residual, ni, nj, count, x1, y1, z1);
        coarse[id3(i, j, k, ci, cj)] = (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7) / 8;
        return;
    }
    const long long sj = ni, sk = (long long)ni * nj;
    const long long base = (long long)i0 + sj * j0 + sk * k0;
    float blk[3][3][3];
#pragma unroll
    for (int dz = 0; dz < 3; dz++)
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const long long id = base + dx + dy * sj + dz * sk;             // flat, unclamped (M4): >= 0 here
                const float t = (float)residual[id < count ? id : count - 1];
                blk[dz][dy][dx] = id < count ? t : 0.f;
            }
    const double v0 = sample_block(blk, 0, 0, 0, ax0, ay0, az0), v1 = sample_block(blk, 0, 0, 1, ax0, ay0, az1);
    const double v2 = sample_block(blk, 0, 1, 0, ax0, ay1, az0), v3 = sample_block(blk, 0, 1, 1, ax0, ay1, az1);
    const double v4 = sample_block(blk, 1, 0, 0, ax1, ay0, az0), v5 = sample_block(blk, 1, 0, 1, ax1, ay0, az1);
    const double v6 = sample_block(blk, 1, 1, 0, ax1, ay1, az0), v7 = sample_block(blk, 1, 1, 1, ax1, ay1, az1);
    coarse[id3(i, j, k, ci, cj)] = (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7) / 8;
}

// prolongation_kernel, double (:1610-1621): fine interior += trilinear sample of the coarse correction
__global__ __launch_bounds__(256) void mg_prolong_kernel(double *__restrict__ x, const double *__restrict__ coarse,
                                                         int ni, int nj, int nk, int ci, int cj, int ck)
{
    MG_IJK(ni, nj, nk)
    if (!MG_INTERIOR(ni, nj, nk)) return;
    const float px = (float)((double)((float)i / 2.f) - 0.5);
    const float py = (float)((double)((float)j / 2.f) - 0.5);
    const float pz = (float)((double)((float)k / 2.f) - 0.5);
    const long long ccount = (long long)ci * cj * ck;
    const double add = ccount <= 0x7fffffffll ? sample_t<int>(coarse, ci, cj, (int)ccount, px, py, pz)
                                              : sample_t<long long>(coarse, ci, cj, ccount, px, py, pz);
    x[id3(i, j, k, ni, nj)] += add;
}

// The same update with one thread per 2x2x2 block of fine cells.  The fine cells 2m+1 and 2m+2 both sample the coarse cells
// m, m+1 (px = m with weight 0, px = m + 0.5 with weight 1/2), so a block shares its eight coarse values and most of its
// lerps: 8 loads and 24 lerps for eight fine cells instead of 64 and 56.  Every lerp is sample_t's, on the same operands
// (weights computed from px exactly as there), so each fine cell receives the same bits.  Arrays below 2^31 elements.
__global__ __launch_bounds__(256) void mg_prolong_block_kernel(double *__restrict__ x, const double *__restrict__ coarse,
                                                               int ni, int nj, int nk, int ci, int cj, int ck)
{
    const int m = blockIdx.x * 64 + threadIdx.x, n = blockIdx.y * 4 + threadIdx.y, l = blockIdx.z;
    if (2 * m + 1 > ni - 2 || 2 * n + 1 > nj - 2 || 2 * l + 1 > nk - 2) return;
    const int count = ci * cj * ck, sj = ci, sk = ci * cj;
    const int base = m + sj * n + sk * l;                               // flat, unclamped (M4); >= 0
    float v[2][2][2];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int id = base + (q & 1) + ((q >> 1) & 1) * sj + ((q >> 2) & 1) * sk;
        const float t = (float)coarse[id < count ? id : count - 1];
        v[(q >> 2) & 1][(q >> 1) & 1][q & 1] = id < count ? t : 0.f;
    }
    // weights of the two fine cells of each axis, from their positions as prolongation_kernel forms them
    float w[3][2];
    bool ok[3][2];
    const int first[3] = { 2 * m + 1, 2 * n + 1, 2 * l + 1 }, lim[3] = { ni - 2, nj - 2, nk - 2 }, cell[3] = { m, n, l };
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int f = first[a] + h;
            const float p = (float)((double)((float)f / 2.f) - 0.5);
            w[a][h] = p - (float)cell[a];                               // floorf(p) == cell[a] for both fine cells
            ok[a][h] = f <= lim[a] && (int)floorf(p) == cell[a];
        }
    float lx[2][2][2];                                                  // [z corner][y corner][fine x]
#pragma unroll
    for (int cz = 0; cz < 2; cz++)
#pragma unroll
        for (int cy = 0; cy < 2; cy++)
#pragma unroll
            for (int h = 0; h < 2; h++) lx[cz][cy][h] = lerp_f(v[cz][cy][0], v[cz][cy][1], w[0][h]);
    float ly[2][2][2];                                                  // [z corner][fine y][fine x]
#pragma unroll
    for (int cz = 0; cz < 2; cz++)
#pragma unroll
        for (int hy = 0; hy < 2; hy++)
#pragma unroll
            for (int hx = 0; hx < 2; hx++) ly[cz][hy][hx] = lerp_f(lx[cz][0][hx], lx[cz][1][hx], w[1][hy]);
#pragma unroll
    for (int hz = 0; hz < 2; hz++)
#pragma unroll
        for (int hy = 0; hy < 2; hy++)
#pragma unroll
            for (int hx = 0; hx < 2; hx++)
                if (ok[0][hx] && ok[1][hy] && ok[2][hz])
                    x[id3(first[0] + hx, first[1] + hy, first[2] + hz, ni, nj)] += (double)lerp_f(ly[0][hy][hx], ly[1][hy][hx], w[2][hz]);
}

// ---- the bottom of the V-cycle in ONE launch ----------------------------------------------------------------------------
// The two coarsest levels of a 256^3 pyramid (15^3 and 7^3) cost ~22 launches per V-cycle -- 8 + 8 + 1 tile-smoother launches,
// residual, restriction, prolongation, clears -- each of them a few microseconds of latency around almost no work.  Both
// levels fit one workgroup's LDS, so one block runs the whole sequence V_Cycle issues between "smooth level A 32 times" and
// "smooth level A 4 times": A's 32 sweeps from a cleared x, its residual, the restriction to B, B's 32 sweeps from a cleared x,
// the prolongation back and A's 4 sweeps, with __syncthreads where the reference has kernel boundaries.  Every value is the
// per-cell expression of the kernel it replaces (mg_smooth_kernel, mg_residual_kernel, mg_restrict_kernel's eight samples,
// mg_prolong_kernel) on the same operands: unclamped flat coarse look-ups that read 0 past the end (M4), float lerps (M2),
// never-written boundary entries of r taken from memory as they are (M5).  Written back: x of both levels, b of B, r of A
// (interior).  temp0 is not touched (V_Cycle clears what the next smoothing call uses of it).
constexpr int kBottomA = 4096, kBottomB = 512;               // LDS capacity in cells of the two levels
__global__ __launch_bounds__(1024) void mg_vbottom_kernel(const double *__restrict__ rhsA, double *__restrict__ xA_g, double *__restrict__ rA_g,
                                                          double *__restrict__ bB_g, double *__restrict__ xB_g,
                                                          int ni, int nj, int nk, int ci, int cj, int ck,
                                                          double alphaA, double betaA, double alphaB, double betaB)
{
    __shared__ double xa[kBottomA], ta[kBottomA], ba[kBottomA], xb[kBottomB], tb[kBottomB], bb[kBottomB];
    const int nA = ni * nj * nk, nB = ci * cj * ck;
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int c = tid; c < nA; c += nth) { ba[c] = rhsA[c]; xa[c] = 0.0; ta[c] = 0.0; }
    for (int c = tid; c < nB; c += nth) { xb[c] = 0.0; tb[c] = 0.0; }
    __syncthreads();
    // which of this thread's cells (tid, tid + 1024, ...) are interior cells: worked out once, not in each of the 68 sweeps
    constexpr int MA = kBottomA / 1024;
    bool innerA[MA];
#pragma unroll
    for (int m = 0; m < MA; m++) {
        const int c = tid + m * 1024;
        const int i = c % ni, j = (c / ni) % nj, k = c / (ni * nj);
        innerA[m] = c < nA && i > 0 && i < ni - 1 && j > 0 && j < nj - 1 && k > 0 && k < nk - 1;
    }
    bool innerB;
    {
        const int i = tid % ci, j = (tid / ci) % cj, k = tid / (ci * cj);
        innerB = tid < nB && i > 0 && i < ci - 1 && j > 0 && j < cj - 1 && k > 0 && k < ck - 1;
    }
    // `iter` sweeps in -> out -> in ... on the interior of a level (smoothing_jacobi: odd counts rounded up)
    auto smoothA = [&](int iter) {
        if (iter % 2 == 1) iter += 1;
        const int sj = ni, sk = ni * nj;
        double *in = xa, *out = ta;
        for (int s = 0; s < iter; s++) {
#pragma unroll
            for (int m = 0; m < MA; m++) {
                const int c = tid + m * 1024;
                if (innerA[m]) out[c] = ((in[c - 1] + in[c + 1] + in[c - sj] + in[c + sj] + in[c - sk] + in[c + sk]) + alphaA * ba[c]) * betaA;
            }
            __syncthreads();
            double *sw = in; in = out; out = sw;
        }
    };
    auto smoothB = [&](int iter) {
        if (iter % 2 == 1) iter += 1;
        const int sj = ci, sk = ci * cj;
        double *in = xb, *out = tb;
        for (int s = 0; s < iter; s++) {
            const int c = tid;
            if (innerB) out[c] = ((in[c - 1] + in[c + 1] + in[c - sj] + in[c + sj] + in[c - sk] + in[c + sk]) + alphaB * bb[c]) * betaB;
            __syncthreads();
            double *sw = in; in = out; out = sw;
        }
    };
    smoothA(32);
    // residual of A (interior) into ta, whose boundary entries take what L[A].r holds in memory (never written: M5)
    {
        const int sj = ni, sk = ni * nj;
        for (int c = tid; c < nA; c += nth) {
            const int i = c % ni, j = (c / ni) % nj, k = c / sk;
            if (i > 0 && i < ni - 1 && j > 0 && j < nj - 1 && k > 0 && k < nk - 1) {
                const double r = ba[c] - ((xa[c - 1] + xa[c + 1] + xa[c - sj] + xa[c + sj] + xa[c - sk] + xa[c + sk]) - xa[c] * 6);
                ta[c] = r;
                rA_g[c] = r;
            } else ta[c] = rA_g[c];
        }
    }
    __syncthreads();
    // restriction A -> B: every coarse cell = mean of eight samples (mg_restrict_kernel's order)
    for (int c = tid; c < nB; c += nth) {
        const int i = c % ci, j = (c / ci) % cj, k = c / (ci * cj);
        const float x0 = (float)(2 * i + 0.5), x1 = (float)(2 * i + 1.5);
        const float y0 = (float)(2 * j + 0.5), y1 = (float)(2 * j + 1.5);
        const float z0 = (float)(2 * k + 0.5), z1 = (float)(2 * k + 1.5);
        const double v0 = sample_t<int>(ta, ni, nj, nA, x0, y0, z0), v1 = sample_t<int>(ta, ni, nj, nA, x0, y0, z1);
        const double v2 = sample_t<int>(ta, ni, nj, nA, x0, y1, z0), v3 = sample_t<int>(ta, ni, nj, nA, x0, y1, z1);
        const double v4 = sample_t<int>(ta, ni, nj, nA, x1, y0, z0), v5 = sample_t<int>(ta, ni, nj, nA, x1, y0, z1);
        const double v6 = sample_t<int>(ta, ni, nj, nA, x1, y1, z0), v7 = sample_t<int>(ta, ni, nj, nA, x1, y1, z1);
        const double v = (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7) / 8;
        bb[c] = v;
        bB_g[c] = v;
    }
    __syncthreads();
    for (int c = tid; c < nA; c += nth) ta[c] = 0.0;             // (V_Cycle clears temp0 before every smoothing call)
    __syncthreads();
    smoothB(32);
    for (int c = tid; c < nB; c += nth) xB_g[c] = xb[c];
    // prolongation B -> A: fine interior += trilinear sample of the coarse correction (mg_prolong_kernel)
    {
        const int sk = ni * nj;
        for (int c = tid; c < nA; c += nth) {
            const int i = c % ni, j = (c / ni) % nj, k = c / sk;
            if (i > 0 && i < ni - 1 && j > 0 && j < nj - 1 && k > 0 && k < nk - 1) {
                const float px = (float)((double)((float)i / 2.f) - 0.5);
                const float py = (float)((double)((float)j / 2.f) - 0.5);
                const float pz = (float)((double)((float)k / 2.f) - 0.5);
                xa[c] += sample_t<int>(xb, ci, cj, nB, px, py, pz);
            }
        }
    }
    __syncthreads();
    smoothA(4);
    for (int c = tid; c < nA; c += nth) xA_g[c] = xa[c];
}

// gradient_kernel, double p (:1009-1023): the three components in one launch
__global__ __launch_bounds__(256) void mg_gradient_kernel(float *__restrict__ u, float *__restrict__ v, float *__restrict__ w,
                                                          const double *__restrict__ p, int ni, int nj, int nk, double halfrdx)
{
    MG_IJK(ni + 1, nj + 1, nk + 1)
    const size_t sj = ni, sk = (size_t)ni * nj;
    // component c: buffer dims (ni+dx, nj+dy, nk+dz); window 2 <= idx < p-dims on every axis
    if (i > 1 && i < ni && j > 1 && j < nj && k > 1 && k < nk) {
        const size_t id = id3(i, j, k, ni, nj);
        const double p0 = p[id];
        u[id3(i, j, k, ni + 1, nj)] -= (float)(halfrdx * (p0 - p[id - 1]));
        v[id3(i, j, k, ni, nj + 1)] -= (float)(halfrdx * (p0 - p[id - sj]));
        w[id3(i, j, k, ni, nj)]     -= (float)(halfrdx * (p0 - p[id - sk]));
    }
}

// ---- host side ----------------------------------------------------------------------------------
static inline unsigned blocks1d(size_t count) { return (unsigned)((count + 255) / 256); }

static void mg_zero(double *p, size_t count) { BQ_HIP(hipMemsetAsync(p, 0, count * sizeof(double), rt().compute)); }

// zeros on the six faces of an array: what a smoothing call needs of a cleared ping-pong buffer when every interior cell
// is about to be overwritten anyway (1.5 % of the bytes of clearing it)
__global__ __launch_bounds__(256) void mg_zero_shell_kernel(double *__restrict__ p, int ni, int nj, int nk)
{
    const long long fz = (long long)ni * nj, fy = (long long)ni * nk, fx = (long long)nj * nk;
    const long long total = 2 * (fz + fy + fx);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        long long e = t;
        int i, j, k;
        if (e < 2 * fz)            { k = e < fz ? 0 : nk - 1; e %= fz; j = (int)(e / ni); i = (int)(e % ni); }
        else if ((e -= 2 * fz) < 2 * fy) { j = e < fy ? 0 : nj - 1; e %= fy; k = (int)(e / ni); i = (int)(e % ni); }
        else                       { e -= 2 * fy; i = e < fx ? 0 : ni - 1; e %= fx; k = (int)(e / nj); j = (int)(e % nj); }
        p[id3(i, j, k, ni, nj)] = 0.0;
    }
}
static void mg_zero_shell(double *p, int ni, int nj, int nk)
{
    const long long total = 2 * ((long long)ni * nj + (long long)ni * nk + (long long)nj * nk);
    mg_zero_shell_kernel<<<(unsigned)std::min<long long>((total + 255) / 256, 2048), 256, 0, rt().compute>>>(p, ni, nj, nk);
    BQ_LAUNCH_CHECK("mg_zero_shell_kernel");
}

// smoothing_jacobi (:1464-1483): odd counts are rounded up, so the newest iterate always ends in x
// clear: the clears V_Cycle issues before the call, left to this function -- bit 0: temp counts as cleared, bit 1: x does.
// Where the lean kernel runs at least two launches they shrink to zeroed faces (every interior cell of both buffers is
// overwritten before it is read) and the first launch does not read x; otherwise both arrays are cleared in full.
// resid (optional): where the caller wants r = b - A x of the smoothed x next; *resid_done tells whether the last launch wrote it
static void mg_smooth(double *x, const double *b, double *temp, double alpha, double beta, int ni, int nj, int nk, int iter, int clear = 0,
                      double *resid = nullptr, bool *resid_done = nullptr)
{
    if (resid_done) *resid_done = false;
    if (iter % 2 == 1) iter += 1;
    const size_t cells = (size_t)ni * nj * nk;
    if (ni < 3 || nj < 3 || nk < 3) {                   // no interior: every sweep is a no-op
        if (clear & 1) mg_zero(temp, cells);
        if (clear & 2) mg_zero(x, cells);
        return;
    }
    double *in = x, *out = temp;
    int s = 0;
    bool cleared = clear == 0;
    rt().mg_smooth_kernel = "";
    // two sweeps per launch where the fused kernel applies (FL_OPT_JACOBI_FUSE != 0): rows of at most 256
    // lanes; double2 lanes when the rows are 16-byte aligned (even nx), else one cell per lane
    // Worth it on large levels only (measured at 256^3: 60 vs 92 us per sweep; at 127^3 and below the
    // k-marching pipeline is too short per block and the plain kernel wins: 11.5 vs 13.5 us, 2.5 vs 3.8 us).
    // FL_OPT_JACOBI_FUSE = 2 forces the fused kernel wherever it applies (tests).
    const bool big = (long long)ni * nj * nk >= (1ll << 21);
    // the lean two-row kernel also pays one level down (127^3: 2.05 M cells)
    const bool mid = (long long)ni * nj * nk >= (1ll << 20);
    if (((rt().opt_jacobi_fuse == 1 && (big || mid)) || rt().opt_jacobi_fuse >= 2) && ni >= 8) {
        const int vec = (ni % 2 == 0 && (((uintptr_t)x | (uintptr_t)temp | (uintptr_t)b) & 15u) == 0) ? 2 : 1;
        // the lean two-row kernel: double2 columns, rows of at most 4 waves, arrays below 2 GiB (32-bit byte offsets).
        // FL_OPT_JACOBI_ROWS = 3 / 8 keep mg_smooth2_kernel (A/B timing)
        // (odd rows: 16-byte loads at 8-byte-aligned addresses, which the memory pipeline splits -- the same mode the
        // gather kernels' dwordx2 loads at 4-byte alignment rely on)
        if ((vec == 2 || ni % 2 == 1) && ni >= 8 && ni <= 512 && nj >= 4 && (double)ni * nj * nk * 8.0 < 2147483648.0 &&
            rt().opt_jacobi_rows != 3 && rt().opt_jacobi_rows != 8) {
            int cw = 16;
            while (cw * 2 < ni) cw *= 2;
            const bool wide = cw > 64;
            const int rows2 = 256 / cw;
            const int nby2 = (nj + 2 * rows2 - 1) / (2 * rows2);
            int gcd = nby2, rem = 256;
            while (rem) { const int t = gcd % rem; gcd = rem; rem = t; }
            const int quantum = 256 / gcd;                              // chunk counts that fill the 256 CUs in whole rounds
            const int target = wide ? 80 : 32;
            int nchunks = ((2 * nk + target) / (2 * target) + quantum / 2) / quantum * quantum;
            if (nchunks < quantum) nchunks = quantum;
            int kc = (nk + nchunks - 1) / nchunks;
            if (rt().opt_jacobi_kchunk2 > 0) kc = rt().opt_jacobi_kchunk2;
            if (kc < 4) kc = 4;
            const int nbz = (nk + kc - 1) / kc;
            const bool in_cache = 24.0 * (double)ni * nj * nk <= 256.0 * 1048576.0;
            const int forced = rt().opt_jacobi_kchunk;
            const int pf = forced == 1 || forced == 2 ? forced : (in_cache ? 1 : 2);
            bool zin = false;
            if (!cleared && iter >= 4) {
                if (clear & 1) mg_zero_shell(temp, ni, nj, nk);
                if (clear & 2) { mg_zero_shell(x, ni, nj, nk); zin = true; }
                cleared = true;
            }
            ProfileSpan span;
            const bool prof = big && iter >= 4 && profile_begin(span);
            const int s_begin = s;
            long long launches = 0;
            bool planned = false;
            // three (and two) sweeps per launch through mg_lds3_kernel where it applies (rows of 130 .. 256 doubles, chunks of >= 24
            // planes at one block per CU; FL_OPT_JACOBI_ROWS = 5 keeps it off for A/B timing): as many triples as leave an even
            // number of launches in total, so that the newest iterate still ends in x, the rest as pairs through the same
            // kernel -- 4 sweeps = 2 pairs.  A call that starts from a cleared x (V_Cycle's way down) is free of the parity rule:
            // its first launch does not read its input, so with an odd number of launches it writes straight into x --
            // 32 sweeps = 10 triples + 1 pair.
            if (vec == 2 && ni >= 130 && ni <= 256 && nj >= 4 && nk >= 12 && rt().opt_jacobi_rows != 5 && cleared) {
                // FL_OPT_JACOBI_KCHUNK = 14: blocks of 4 output rows (8 waves) instead of 8 (12 waves), for A/B timing
                const int LW = rt().opt_jacobi_kchunk == 14 ? 4 : 8;
                const int nbyl = (nj + LW - 1) / LW;
                int nbzl = std::max(1, rt().num_cus / nbyl);
                int kcl = (nk + nbzl - 1) / nbzl;
                if (rt().opt_jacobi_kchunk2 > 0) kcl = rt().opt_jacobi_kchunk2;
                int triples = -1;
                if (kcl >= (rt().opt_jacobi_kchunk2 > 0 ? 8 : 24))
                    for (int a = (iter - s) / 3; a >= 0 && triples < 0; a--) {
                        const int rest = iter - s - 3 * a;
                        if (rest % 2 == 0 && (zin || (rest / 2 + a) % 2 == 0)) triples = a;
                    }
                if (triples >= 0) {
                    const int pairs = (iter - s - 3 * triples) / 2;
                    if (zin && (pairs + triples) % 2 == 1) { double *t2 = in; in = out; out = t2; }
                    nbzl = (nk + kcl - 1) / kcl;
                    const int nblk = nbyl * nbzl, gridl = 8 * ((nblk + 7) / 8);
                    for (int t = 0; t < triples + pairs; t++) {
                        const bool three = t < triples;
                        // the last launch, when it is a pair and the caller wants the residual next: two sweeps + the residual
                        if (!three && t == triples + pairs - 1 && resid && resid_done && LW == 8 && !zin && out == x) {
                            mg_lds3_kernel<8, 3, false, true><<<gridl, 12 * 64, 0, rt().compute>>>(in, b, out, ni, nj, nk, nbyl, nblk, kcl, alpha, beta, resid);
                            *resid_done = true;
                            double *t2 = in; in = out; out = t2;
                            s += 2; launches++;
                            continue;
                        }
#define MG_L3(WV, SV, Z) mg_lds3_kernel<WV, SV, Z><<<gridl, (WV + 2 * (SV - 1)) * 64, 0, rt().compute>>>(in, b, out, ni, nj, nk, nbyl, nblk, kcl, alpha, beta)
                        if (three) {
                            if (LW == 4) { if (zin) MG_L3(4, 3, true); else MG_L3(4, 3, false); }
                            else         { if (zin) MG_L3(8, 3, true); else MG_L3(8, 3, false); }
                        } else {
                            if (LW == 4) { if (zin) MG_L3(4, 2, true); else MG_L3(4, 2, false); }
                            else         { if (zin) MG_L3(8, 2, true); else MG_L3(8, 2, false); }
                        }
#undef MG_L3
                        zin = false;
                        double *t2 = in; in = out; out = t2;
                        s += three ? 3 : 2; launches++;
                    }
                    if (triples + pairs) { BQ_LAUNCH_CHECK("mg_lds3_kernel"); rt().mg_smooth_kernel = "mg_lds3_kernel"; planned = true; }
                }
            }
            // (without triples the pairs come in twos, so that the newest iterate ends in x; with them the plan above has
            // settled the parity and the remaining pairs are counted one by one)
            const int pair_group = planned ? 1 : 2;
            for (; s + 2 * pair_group <= iter; s += 2 * pair_group)
                for (int h = 0; h < pair_group; h++) {
                    launches++;
#define MG_L2(W, F, Z) mg_lean2r_kernel<W, F, Z><<<nby2 * nbz, 256, 0, rt().compute>>>(in, b, out, ni, nj, nk, cw, nby2, kc, alpha, beta)
                    if (zin) {
                        if (wide) { if (pf == 1) MG_L2(true, 1, true); else MG_L2(true, 2, true); }
                        else      { if (pf == 1) MG_L2(false, 1, true); else MG_L2(false, 2, true); }
                        zin = false;
                    } else {
                        if (wide) { if (pf == 1) MG_L2(true, 1, false); else MG_L2(true, 2, false); }
                        else      { if (pf == 1) MG_L2(false, 1, false); else MG_L2(false, 2, false); }
                    }
#undef MG_L2
                    double *t = in; in = out; out = t;
                }
            if (prof) profile_end(span, launches, s - s_begin);
            BQ_LAUNCH_CHECK("mg_lean2r_kernel");
            if (launches && !rt().mg_smooth_kernel[0]) rt().mg_smooth_kernel = "mg_lean2r_kernel";
        }
        if (!cleared) { if (clear & 1) mg_zero(temp, cells); if (clear & 2) mg_zero(x, cells); cleared = true; }
        const int lanes = (ni + vec - 1) / vec;
        const int lpr = ((lanes + 63) / 64) * 64;
        const int threads = rt().opt_jacobi_rows == 8 ? 512 : 256;      // FL_OPT_JACOBI_ROWS: waves per block (4 or 8)
        if (lpr <= threads && (big || rt().opt_jacobi_fuse >= 2)) {      // (slower than one launch per sweep below 2 M cells)
            const int rows = threads / lpr;
            const int nby = (nj + rows - 1) / rows;
            int kchunk = rt().opt_jacobi_kchunk2 > 0 ? rt().opt_jacobi_kchunk2 : 64;
            while (kchunk > 4 && (long)nby * ((nk + kchunk - 1) / kchunk) < 512) kchunk /= 2;
            const int nbz = (nk + kchunk - 1) / kchunk;
            // launches come in pairs (x -> temp -> x) so that the newest iterate still ends where the
            // reference leaves it; 32 and 4 sweeps (V_Cycle) are all pairs
            ProfileSpan span;
            const bool prof = big && iter >= 4 && profile_begin(span);
            const int s_begin = s;
            for (; s + 4 <= iter; s += 4)
                for (int h = 0; h < 2; h++) {
#define MG_S2(V, T) mg_smooth2_kernel<V, T><<<nby * nbz, T, 0, rt().compute>>>(in, b, out, alpha, beta, ni, nj, nk, lpr, nby, kchunk)
                    if (vec == 2) { if (threads == 512) MG_S2(2, 512); else MG_S2(2, 256); }
                    else          { if (threads == 512) MG_S2(1, 512); else MG_S2(1, 256); }
#undef MG_S2
                    double *t = in; in = out; out = t;
                }
            if (prof) profile_end(span, (s - s_begin) / 2, s - s_begin);
            BQ_LAUNCH_CHECK("mg_smooth2_kernel");
        }
    }
    if (!cleared) { if (clear & 1) mg_zero(temp, cells); if (clear & 2) mg_zero(x, cells); }
    for (; s < iter; s++) {
        mg_smooth_kernel<<<grid_of(ni, nj, nk), kBlk, 0, rt().compute>>>(in, b, out, alpha, beta, ni, nj, nk);
        double *t = in; in = out; out = t;
    }
    BQ_LAUNCH_CHECK("mg_smooth_kernel");
}

// The coarse levels of V_Cycle through mg_smooth_tile_kernel: `iter` sweeps of x (newest iterate ends in x, as in
// mg_smooth), with the clears that V_Cycle issues before the call folded in -- temp is never read before it is written
// and x_is_zero says that x would have been cleared too.  false = not applicable (the caller clears and calls mg_smooth).
static bool mg_smooth_tiled(double *x, const double *b, double *temp, double alpha, double beta, int ni, int nj, int nk,
                            int iter, bool x_is_zero)
{
    if (!rt().opt_mgcg_tile || iter < 4 || iter % 4 != 0) return false;
    // 63^3 and coarser by default; FL_OPT_MGCG_TILE = 2 also takes 127^3 (measured slower there: 4096 workgroups that
    // each recompute 3x the cells they store, against sweeps that already stream from the Infinity Cache)
    const long long limit = rt().opt_mgcg_tile >= 2 ? (1ll << 21) : (1ll << 19);
    if ((long long)ni * nj * nk >= limit || ni < 3 || nj < 3 || nk < 3) return false;
    hipStream_t st = rt().compute;
    const int S = iter % 8 == 0 ? 4 : 2, T = 16 - 2 * S, launches = iter / S;            // an even number of launches
    const dim3 grid((ni + T - 1) / T, (nj + T - 1) / T, (nk + T - 1) / T);
    const double *in = x;
    double *out = temp;
    for (int l = 0; l < launches; l++) {
        // launch 0 writes all of temp, launch 1 all of x when x was to be cleared: boundary cells included
        const bool zin = l == 0 && x_is_zero, zb = l == 0 || (l == 1 && x_is_zero);
#define MG_TILE(SS, ZI, ZBB) mg_smooth_tile_kernel<SS, ZI, ZBB><<<grid, 256, 0, st>>>(in, b, out, alpha, beta, ni, nj, nk)
        if (S == 4) { if (zin) MG_TILE(4, true, true); else if (zb) MG_TILE(4, false, true); else MG_TILE(4, false, false); }
        else        { if (zin) MG_TILE(2, true, true); else if (zb) MG_TILE(2, false, true); else MG_TILE(2, false, false); }
#undef MG_TILE
        const double *t = in; in = out; out = (double *)t;
    }
    BQ_LAUNCH_CHECK("mg_smooth_tile_kernel");
    return true;
}

// out = rhs - A x (poisson = false) or out = A x through the marching kernel; false = not applicable to this grid
static bool mg_stencil_lean(double *out, const double *rhs, const double *x, int ni, int nj, int nk, bool poisson)
{
    if (!rt().opt_mgcg_tile || (long long)ni * nj * nk < (1ll << 20)) return false;
    if (ni < 8 || ni > 512 || nj < 4 || nk < 3 || (double)ni * nj * nk * 8.0 >= 2147483648.0) return false;
    if (ni % 2 == 0 && ((((uintptr_t)out | (uintptr_t)x | (uintptr_t)(poisson ? x : rhs)) & 15u) != 0)) return false;
    int cw = 16;
    while (cw * 2 < ni) cw *= 2;
    const bool wide = cw > 64;
    const int rows2 = 256 / cw, nby2 = (nj + 2 * rows2 - 1) / (2 * rows2);
    int gcd = nby2, rem = 256;
    while (rem) { const int t = gcd % rem; gcd = rem; rem = t; }
    const int quantum = 256 / gcd;
    const int target = 32;
    int nchunks = ((2 * nk + target) / (2 * target) + quantum / 2) / quantum * quantum;
    if (nchunks < quantum) nchunks = quantum;
    int kc = (nk + nchunks - 1) / nchunks;
    if (kc < 4) kc = 4;
    const int nbz = (nk + kc - 1) / kc;
    hipStream_t st = rt().compute;
    const double *rp = poisson ? x : rhs;
    if (wide) { if (poisson) mg_stencil_lean_kernel<true, true><<<nby2 * nbz, 256, 0, st>>>(x, rp, out, ni, nj, nk, cw, nby2, kc);
                else         mg_stencil_lean_kernel<true, false><<<nby2 * nbz, 256, 0, st>>>(x, rp, out, ni, nj, nk, cw, nby2, kc); }
    else      { if (poisson) mg_stencil_lean_kernel<false, true><<<nby2 * nbz, 256, 0, st>>>(x, rp, out, ni, nj, nk, cw, nby2, kc);
                else         mg_stencil_lean_kernel<false, false><<<nby2 * nbz, 256, 0, st>>>(x, rp, out, ni, nj, nk, cw, nby2, kc); }
    return BQ_LAUNCH_CHECK("mg_stencil_lean_kernel");
}

static void mg_residual(double *r, const double *b, const double *x, int ni, int nj, int nk)
{
    if (mg_stencil_lean(r, b, x, ni, nj, nk, false)) return;
    mg_residual_kernel<<<grid_of(ni, nj, nk), kBlk, 0, rt().compute>>>(r, b, x, ni, nj, nk);
    BQ_LAUNCH_CHECK("mg_residual_kernel");
}

// calc_sum over the nb block partials of a dot product -> result[iter_index] (the second half of mg_dot; the z-slab solver
// runs it on partials gathered from all ranks)
static void mg_dot_finish(const double *partials, double *result, unsigned nb, int iter_index)
{
    const size_t per_thread = (nb + 255) / 256;
    double *rows = per_thread >= 16 ? (double *)scratch(256 * sizeof(double)) : nullptr;
    if (rows) {
        mg_calc_sum_rows_kernel<<<16, 256, 0, rt().compute>>>(partials, rows, nb, per_thread);
        mg_calc_sum_tree_kernel<<<1, 64, 0, rt().compute>>>(rows, result, iter_index);
    } else {
        mg_calc_sum_kernel<<<1, 256, 0, rt().compute>>>(partials, result, nb, per_thread, iter_index);
    }
    BQ_LAUNCH_CHECK("mg_dot_finish");
}

static void mg_dot(const double *v0, const double *v1, double *partials, double *result, size_t count, int iter_index)
{
    const unsigned nb = blocks1d(count);
    mg_dot_kernel<<<nb, 256, 0, rt().compute>>>(v0, v1, partials, count);
    const size_t per_thread = (nb + 255) / 256;
    double *rows = per_thread >= 16 ? (double *)scratch(256 * sizeof(double)) : nullptr;
    if (rows) {
        mg_calc_sum_rows_kernel<<<16, 256, 0, rt().compute>>>(partials, rows, nb, per_thread);
        mg_calc_sum_tree_kernel<<<1, 64, 0, rt().compute>>>(rows, result, iter_index);
    } else {
        mg_calc_sum_kernel<<<1, 256, 0, rt().compute>>>(partials, result, nb, per_thread, iter_index);
    }
    BQ_LAUNCH_CHECK("mg_dot");
}

static void mg_max(const double *v, double *result, size_t count, int iter_index)
{
    const int nparts = 1024;
    double *part = (double *)scratch(nparts * sizeof(double));
    if (!part) return;
    mg_max_partial_kernel<<<nparts, 256, 0, rt().compute>>>(v, count, part);
    mg_max_final_kernel<<<1, 256, 0, rt().compute>>>(part, nparts, result, iter_index);
    BQ_LAUNCH_CHECK("mg_max");
}

#include "bq_mgcg_fused.hip.inc"

// V_Cycle, multi-level form (:1636-1707), `else` branch
// copy_b: whether level 0's right-hand side is copied into L[0].b first (the reference does, :1640).  The copy only
// matters for what L[0].b holds afterwards, i.e. in the last outer iteration; before that level 0 reads `residual` itself
// (nothing writes it until this function's last launch) and 134 MB of traffic per cycle stay away.
// xin (FL_OPT_MGCG_FUSE): the iterate comes in `xin` (update_x's result) and goes out in `x`; the last two launches -- x += L[0].x,
// residual = b - A x -- and the max / dot of that residual that the caller takes next are one (partials in temp0, the max parts behind)
static void v_cycle(const double *b, double *x, double *residual, const SCoarseLevelInfo *L, double *temp0, int levelnum, bool copy_b = true,
                    const double *xin = nullptr, double *maxpart = nullptr)
{
    double scale[LEVEL_COUNT] = { 1.0, 1.0, 1.0, 1.0, 1.0, 1.0 };
    scale[1] = 8.0;                                                                          // M3
    hipStream_t st = rt().compute;
    const size_t n0 = (size_t)L[0].number;
    if (copy_b) BQ_HIP(hipMemcpyAsync(L[0].b, residual, n0 * sizeof(double), hipMemcpyDeviceToDevice, st));
    auto rhs = [&](int l) -> const double * { return (l == 0 && !copy_b) ? residual : L[l].b; };
    // The reference clears all n0 entries of temp0 before every smoothing call; a level-l smoothing only
    // ever touches the first L[l].number of them, so that is what is cleared.  (What stays behind in the
    // rest of temp0 is later seen only at boundary indices of the level-0 product dir*A(dir), where dir
    // is 0: no value depends on it.)
    // smoothing call of V_Cycle: clear temp0 (and x on the way down), `iter` sweeps
    // resid: the residual array the caller computes next (the way down); returns true when the smoothing's last launch wrote it
    auto smooth_level = [&](int l, int iter, bool clear_x, double *resid = nullptr) -> bool {
        if (mg_smooth_tiled(L[l].x, rhs(l), temp0, L[l].alpha * scale[l], L[l].beta, L[l].ni, L[l].nj, L[l].nk, iter, clear_x)) return false;
        bool done = false;
        mg_smooth(L[l].x, rhs(l), temp0, L[l].alpha * scale[l], L[l].beta, L[l].ni, L[l].nj, L[l].nk, iter, clear_x ? 3 : 1, resid, &done);
        return done;
    };
    // the two coarsest levels in one launch where both fit a workgroup's LDS (FL_OPT_MGCG_BOTTOM)
    const int lb = levelnum - 2;
    const bool bottom = rt().opt_mgcg_bottom && rt().opt_mgcg_tile && levelnum >= 2 && L[lb].number <= kBottomA && L[lb + 1].number <= kBottomB &&
                        L[lb].number > 0 && L[lb + 1].number > 0;
    for (int l = 0; l < (bottom ? lb : levelnum - 1); l++) {
        if (!smooth_level(l, 32, true, rt().opt_mgcg_bottom ? L[l].r : nullptr))
            mg_residual(L[l].r, rhs(l), L[l].x, L[l].ni, L[l].nj, L[l].nk);
        mg_restrict_kernel<<<grid_of(L[l + 1].ni, L[l + 1].nj, L[l + 1].nk), kBlk, 0, st>>>(
            L[l].r, L[l + 1].b, L[l].ni, L[l].nj, L[l].nk, L[l + 1].ni, L[l + 1].nj, L[l + 1].nk);
        BQ_LAUNCH_CHECK("mg_restrict_kernel");
    }
    if (bottom) {
        mg_vbottom_kernel<<<1, 1024, 0, st>>>(rhs(lb), L[lb].x, L[lb].r, L[lb + 1].b, L[lb + 1].x, L[lb].ni, L[lb].nj, L[lb].nk,
                                               L[lb + 1].ni, L[lb + 1].nj, L[lb + 1].nk, L[lb].alpha * scale[lb], L[lb].beta,
                                               L[lb + 1].alpha * scale[lb + 1], L[lb + 1].beta);
        BQ_LAUNCH_CHECK("mg_vbottom_kernel");
    } else smooth_level(levelnum - 1, 32, true);
    for (int l = bottom ? lb - 1 : levelnum - 2; l >= 0; --l) {
        // one thread per 2x2x2 block of fine cells where the coarse array is below 2^31 elements and every fine index is
        // exact in float (always, at these sizes); FL_OPT_MGCG_TILE = 0 keeps the one-cell form
        if (rt().opt_mgcg_tile && (long long)L[l + 1].ni * L[l + 1].nj * L[l + 1].nk < (1ll << 31) && L[l].ni < (1 << 22) &&
            L[l].nj < (1 << 22) && L[l].nk < (1 << 22) && L[l].ni >= 3 && L[l].nj >= 3 && L[l].nk >= 3)
            mg_prolong_block_kernel<<<grid_of((L[l].ni - 1) / 2, (L[l].nj - 1) / 2, (L[l].nk - 1) / 2), kBlk, 0, st>>>(
                L[l].x, L[l + 1].x, L[l].ni, L[l].nj, L[l].nk, L[l + 1].ni, L[l + 1].nj, L[l + 1].nk);
        else
        mg_prolong_kernel<<<grid_of(L[l].ni, L[l].nj, L[l].nk), kBlk, 0, st>>>(
            L[l].x, L[l + 1].x, L[l].ni, L[l].nj, L[l].nk, L[l + 1].ni, L[l + 1].nj, L[l + 1].nk);
        BQ_LAUNCH_CHECK("mg_prolong_kernel");
        smooth_level(l, 4, false);
    }
    if (xin) {
        FuseArgs A{};
        A.base = xin; A.d = L[0].x; A.rhs = b; A.xout = x; A.out = residual; A.mul = 1.0;
        A.partials = temp0; A.maxpart = maxpart;                     // (temp0's other rim cells must stay as the sweeps left them)
        mg_fused<kFuseAddRes>(A, L[0].ni, L[0].nj, L[0].nk);
        return;
    }
    mg_add_kernel<<<blocks1d(n0), 256, 0, st>>>(x, L[0].x, 1.0, n0);
    BQ_LAUNCH_CHECK("mg_add_kernel");
    mg_residual(residual, b, x, L[0].ni, L[0].nj, L[0].nk);
}

// The V-cycle is ~230 launches with the same arguments in every outer iteration, most of them on grids too
// small to hide a launch gap: it is captured once into a hipGraph (stream capture on the compute stream) and
// replayed.  The cache is keyed by everything the launches depend on; FL_OPT_PROFILE_JACOBI (event records
// inside the cycle) and FL_OPT_MGCG_GRAPH = 0 fall back to plain launches.
struct VCycleGraph {
    hipGraphExec_t exec = nullptr;
    const double *b = nullptr; double *x = nullptr, *residual = nullptr, *temp0 = nullptr;
    SCoarseLevelInfo levels[LEVEL_COUNT];
    int levelnum = 0, fuse = 0, rows = 0, kchunk = 0, kchunk2 = 0, tile = 0, cus = 0, bottom = 0;
    const double *xin = nullptr; double *maxpart = nullptr;
};
// the two cached graphs ([copy_b]) of the CURRENT context (bq_host.h: Runtime::mgcg_state)
#include "bq_mgcg_slab.hip.inc"
struct MgcgState { VCycleGraph vcgs[2]; SlabMg slab; };
static MgcgState &ms()
{
    Runtime &r = rt();
    if (!r.mgcg_state) r.mgcg_state = new MgcgState();
    return *static_cast<MgcgState *>(r.mgcg_state);
}
#define g_vcgs (ms().vcgs)

static bool vcg_matches(const VCycleGraph &c, const double *b, double *x, double *residual, const SCoarseLevelInfo *L, double *temp0, int levelnum,
                        const double *xin, double *maxpart)
{
    if (!c.exec || c.xin != xin || c.maxpart != maxpart || c.b != b || c.x != x || c.residual != residual || c.temp0 != temp0 || c.levelnum != levelnum) return false;
    if (c.fuse != rt().opt_jacobi_fuse || c.rows != rt().opt_jacobi_rows || c.kchunk != rt().opt_jacobi_kchunk || c.kchunk2 != rt().opt_jacobi_kchunk2 ||
        c.tile != rt().opt_mgcg_tile || c.cus != rt().num_cus || c.bottom != rt().opt_mgcg_bottom) return false;
    for (int l = 0; l < levelnum; l++) {
        const SCoarseLevelInfo &p = c.levels[l], &q = L[l];
        if (p.ni != q.ni || p.nj != q.nj || p.nk != q.nk || p.number != q.number || p.alpha != q.alpha || p.beta != q.beta ||
            p.b != q.b || p.x != q.x || p.r != q.r) return false;
    }
    return true;
}

static void v_cycle_replayed(const double *b, double *x, double *residual, const SCoarseLevelInfo *L, double *temp0, int levelnum, bool copy_b,
                             const double *xin = nullptr, double *maxpart = nullptr)
{
    if (!rt().opt_mgcg_tile) copy_b = true;
    if (!rt().opt_mgcg_graph || rt().opt_profile_jacobi) { v_cycle(b, x, residual, L, temp0, levelnum, copy_b, xin, maxpart); return; }
    VCycleGraph &g_vcg = g_vcgs[copy_b ? 1 : 0];
    hipStream_t st = rt().compute;
    if (!vcg_matches(g_vcg, b, x, residual, L, temp0, levelnum, xin, maxpart)) {
        if (g_vcg.exec) { (void)hipGraphExecDestroy(g_vcg.exec); g_vcg.exec = nullptr; }
        (void)scratch(64);                                   // no allocation may happen while capturing
        hipGraph_t graph = nullptr;
        bool ok = BQ_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        if (ok) {
            v_cycle(b, x, residual, L, temp0, levelnum, copy_b, xin, maxpart);
            ok = BQ_HIP(hipStreamEndCapture(st, &graph)) && rt().err == FL_OK;
        }
        if (ok) ok = BQ_HIP(hipGraphInstantiate(&g_vcg.exec, graph, nullptr, nullptr, 0));
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) { g_vcg.exec = nullptr; return; }           // the error is latched
        g_vcg.b = b; g_vcg.x = x; g_vcg.residual = residual; g_vcg.temp0 = temp0; g_vcg.levelnum = levelnum; g_vcg.xin = xin; g_vcg.maxpart = maxpart;
        g_vcg.fuse = rt().opt_jacobi_fuse; g_vcg.rows = rt().opt_jacobi_rows; g_vcg.kchunk = rt().opt_jacobi_kchunk; g_vcg.kchunk2 = rt().opt_jacobi_kchunk2; g_vcg.cus = rt().num_cus; g_vcg.bottom = rt().opt_mgcg_bottom;
        g_vcg.tile = rt().opt_mgcg_tile;
        for (int l = 0; l < levelnum; l++) g_vcg.levels[l] = L[l];
    }
    BQ_HIP(hipGraphLaunch(g_vcg.exec, st));
}

void mgcg_release_graph()
{
    if (!rt().mgcg_state) return;
    for (VCycleGraph &g_vcg : g_vcgs)
        if (g_vcg.exec) { (void)hipGraphExecDestroy(g_vcg.exec); g_vcg.exec = nullptr; }
}
void mgcg_release_state(Runtime &r)
{
    MgcgState *st = static_cast<MgcgState *>(r.mgcg_state);
    if (!st) return;
    for (VCycleGraph &g : st->vcgs)
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    slab_mg_release(st->slab);
    delete st;
    r.mgcg_state = nullptr;
}

} // namespace bq

using namespace bq;

extern "C" {

const char *fl_mg_smooth_kernel_name(void) { return rt().mg_smooth_kernel; }
long long fl_mg_fused_launches(void) { const long long n = rt().mg_fused_launches; rt().mg_fused_launches = 0; return n; }

void gpu_smoothing_jacobi(double *x, double *b, double *temp, double alpha, double beta,
                          int ni, int nj, int nk, int iter)
{
    const char *op = "gpu_smoothing_jacobi";
    if (!ensure_ready(op)) return;
    BQ_REQUIRE(x && b && temp && x != temp && ni >= 1 && nj >= 1 && nk >= 1 && nk < 65535 && iter >= 0, op);
    mg_smooth(x, b, temp, alpha, beta, ni, nj, nk, iter);
}

void gpu_multi_grid_conjugate_gradient(float *u, float *v, float *w, double *div, double *p,
                                       double *dir, double *residual, double *temp0, double *temp1,
                                       double *tempResult, struct SCoarseLevelInfo *levels,
                                       int levelNum, int iter, double halfrdx)
{
    const char *op = "gpu_multi_grid_conjugate_gradient";
    if (!ensure_ready(op)) return;
    BQ_REQUIRE(u && v && w && div && p && dir && residual && temp0 && temp1 && tempResult && levels, op);
    BQ_REQUIRE(levelNum >= 1 && levelNum <= LEVEL_COUNT && iter >= 0, op);
    BQ_REQUIRE(2 * iter + 2 < 2000 && 2001 + iter <= 4096, op);         // tempResult layout: [0, 2*iter+2], [2000, 2000+iter]
    if (rt().slab_on) { latch(FL_ERR_UNSUPPORTED, op, "not built for z-slab ranks yet (use the Jacobi projection)"); return; }
    for (int l = 0; l < levelNum; l++) {
        const SCoarseLevelInfo &L = levels[l];
        BQ_REQUIRE(L.ni >= 1 && L.nj >= 1 && L.nk >= 1 && L.nk < 65535 && L.b && L.x && L.r, op);
        BQ_REQUIRE((long long)L.ni * L.nj * L.nk == (long long)L.number, op);
    }
    const int ni = levels[0].ni, nj = levels[0].nj, nk = levels[0].nk;
    const size_t number = (size_t)levels[0].number;
    hipStream_t st = rt().compute;

    mg_divergence_kernel<<<grid_of(ni, nj, nk), kBlk, 0, st>>>(u, v, w, div, ni, nj, nk, halfrdx);
    BQ_LAUNCH_CHECK("mg_divergence_kernel");
    mg_zero(p, number);
    mg_residual(residual, div, p, ni, nj, nk);
    mg_mul_kernel<<<blocks1d(number), 256, 0, st>>>(dir, residual, 1, number);
    BQ_LAUNCH_CHECK("mg_mul_kernel");
    mg_max(residual, tempResult, number, 2000);
    mg_dot(residual, residual, temp0, tempResult, number, 0);                               // r.r

    // FL_OPT_MGCG_FUSE (bq_mgcg_fused.hip.inc): the nine level-0 vector passes of an iteration as three launches.  The fused
    // updates are out of place: p -> temp1 (update_x) -> p (the V-cycle's add), dir <-> levels[0].b (free while the V-cycle
    // reads `residual` itself as its right-hand side, i.e. in every iteration but the last)
    const bool fused = levelNum >= 1 && mg_fuse_ok(ni, nj, nk, { div, p, dir, residual, temp0, temp1, levels[0].b, levels[0].x });
    const FuseGeom fg = fused ? fuse_geom_for(ni, nj, nk) : FuseGeom{};
    // the max parts of the fused residual: in the runtime's scratch, behind the 4 KB that mg_dot_finish uses
    double *maxpart = fused ? (double *)scratch(4096 + (size_t)(fg.nblk + fg.nrim) * sizeof(double)) : nullptr;
    if (fused && !maxpart) return;
    if (maxpart) maxpart += 512;
    double *dcur = dir;                                                                     // where the search direction lives
    for (int it = 0; it < iter; it++) {
        const int off = it * 2;
        const bool last = it == iter - 1;
        // smoothing_conjugate_gradient (:1485-1495): aMulDir = temp0, dotDir = temp1
        if (!fused || it == 0) {
            if (!mg_stencil_lean(temp0, nullptr, dcur, ni, nj, nk, true)) {
                mg_poisson_kernel<<<grid_of(ni, nj, nk), kBlk, 0, st>>>(dcur, temp0, ni, nj, nk);
                BQ_LAUNCH_CHECK("mg_poisson_kernel");
            }
            mg_dot(dcur, temp0, temp1, tempResult, number, off + 1);
        }
        if (!fused) {
            mg_update_x_kernel<<<blocks1d(number), 256, 0, st>>>(p, dir, tempResult, number, off, off + 1);
            BQ_LAUNCH_CHECK("mg_update_x_kernel");
            mg_residual(residual, div, p, ni, nj, nk);

            v_cycle_replayed(div, p, residual, levels, temp0, levelNum, last);
            mg_max(residual, tempResult, number, 2001 + it);

            // updateDir (:1497-1503)
            mg_dot(residual, residual, temp0, tempResult, number, off + 2);
            mg_update_dir_kernel<<<blocks1d(number), 256, 0, st>>>(dir, residual, tempResult, number, off, off + 2);
            BQ_LAUNCH_CHECK("mg_update_dir_kernel");
            continue;
        }
        {   // temp1 = p + dcur a0/a1, residual = div - A temp1
            FuseArgs A{};
            A.base = p; A.d = dcur; A.rhs = div; A.xout = temp1; A.out = residual; A.coef = tempResult; A.num_idx = off; A.den_idx = off + 1;
            mg_fused<kFuseXRes>(A, ni, nj, nk);
        }
        if (last && dcur != dir) {          // the last V-cycle copies its right-hand side into levels[0].b: move the direction out of it
            BQ_HIP(hipMemcpyAsync(dir, dcur, number * sizeof(double), hipMemcpyDeviceToDevice, st));
            dcur = dir;
        }
        v_cycle_replayed(div, p, residual, levels, temp0, levelNum, last, temp1, maxpart);  // ... p = temp1 + L0.x, residual, its max and r.r
        mg_max_final_kernel<<<1, 256, 0, st>>>(maxpart, (int)(fg.nblk + fg.nrim), tempResult, 2001 + it);
        BQ_LAUNCH_CHECK("mg_max_final_kernel");
        mg_dot_finish(temp0, tempResult, fg.npart, off + 2);
        if (last) {
            mg_update_dir_kernel<<<blocks1d(number), 256, 0, st>>>(dir, residual, tempResult, number, off, off + 2);
            BQ_LAUNCH_CHECK("mg_update_dir_kernel");
        } else {                            // updateDir and the next iteration's A dir, dir . A dir
            double *dnext = dcur == dir ? levels[0].b : dir;
            FuseArgs A{};
            A.base = residual; A.d = dcur; A.xout = dnext; A.out = temp0; A.coef = tempResult; A.num_idx = off + 2; A.den_idx = off;
            A.partials = temp1;
            mg_fused<kFuseDirPoi>(A, ni, nj, nk);
            mg_dot_finish(temp1, tempResult, fg.npart, off + 3);
            dcur = dnext;
        }
    }
    mg_gradient_kernel<<<grid_of(ni + 1, nj + 1, nk + 1), kBlk, 0, st>>>(u, v, w, p, ni, nj, nk, halfrdx);
    BQ_LAUNCH_CHECK("mg_gradient_kernel");
}

// 1 when gpu_multi_grid_conjugate_gradient_slab can run this decomposition (bq_mgcg_slab.hip.inc: requirements), else 0
int gpu_mgcg_slab_supported(int ni, int nj, int nkg, int own0, int own1, int ghost, int rank, int nranks)
{
    if (nranks < 2 || ni < 8 || nj < 8 || nkg < 8 || rank < 0 || rank >= nranks) return 0;
    if (((long long)ni * nj) % 256 != 0) return 0;                  // block dot products: slab boundaries must be block boundaries
    if (nkg % nranks != 0 || own0 != rank * (nkg / nranks) || own1 != own0 + nkg / nranks) return 0;
    if (own0 % 8 != 0 || own1 - own0 < 17 || ghost < 8) return 0;   // even ranges on three levels, G = 8 on level 0 inside the velocity's ghosts
    if ((double)ni * nj * (own1 - own0 + 16) * 8.0 >= 2147483648.0) return 0;
    return 1;
}

// gpu_multi_grid_conjugate_gradient (GPU_kernel.cu:1764-1815) on a z-slab rank, the grid's levels shared between the ranks.
//   u, v, w     the rank's LOCAL velocity buffers: global planes [own0 - ghost, own1 + ghost) (w one more), ghost planes correct
//   tempResult  device, 4096 doubles: the reference's residual history, identical on every rank
// Everything else the solver needs it allocates itself, once per geometry and context.  Collective: every rank calls it.
// On return the velocity is projected on the owned planes and on ghost - 1 ghost planes of either side.
void gpu_multi_grid_conjugate_gradient_slab(float *u, float *v, float *w, double *tempResult,
                                            int ni, int nj, int nkg, int own0, int own1, int ghost, int iter, double halfrdx)
{
    const char *op = "gpu_multi_grid_conjugate_gradient_slab";
    if (!ensure_ready(op)) return;
    BQ_REQUIRE(u && v && w && tempResult && iter >= 0, op);
    BQ_REQUIRE(2 * iter + 2 < 2000 && 2001 + iter <= 4096, op);
    const int rank = fl_comm_rank(), nranks = fl_comm_size();
    if (!gpu_mgcg_slab_supported(ni, nj, nkg, own0, own1, ghost, rank, nranks)) { latch(FL_ERR_UNSUPPORTED, op, "geometry not supported (gpu_mgcg_slab_supported)"); return; }
    SlabMg &m = ms().slab;
    if (!slab_mg_setup(m, ni, nj, nkg, own0, own1, rank, nranks)) { if (rt().err == FL_OK) latch(FL_ERR_UNSUPPORTED, op, "levels too thin to share"); return; }
    const SlabLevel &L0 = m.lv[0];
    const int nkl = L0.nkl(), ulo = own0 - ghost;
    const size_t pd = L0.plane(), n0 = pd * (size_t)nkl;
    hipStream_t st = rt().compute;
    const float *uu = u + (size_t)(ni + 1) * nj * (size_t)(L0.lo - ulo);
    const float *vv = v + (size_t)ni * (nj + 1) * (size_t)(L0.lo - ulo);
    const float *ww = w + pd * (size_t)(L0.lo - ulo);

    mg_divergence_kernel<<<grid_of(ni, nj, nkl), kBlk, 0, st>>>(uu, vv, ww, m.div, ni, nj, nkl, halfrdx);
    BQ_LAUNCH_CHECK("mg_divergence_kernel");
    mg_zero(m.p, n0);
    mg_residual(m.residual, m.div, m.p, ni, nj, nkl);
    slab_exchange(L0, m.residual, L0.G);
    mg_mul_kernel<<<blocks1d(n0), 256, 0, st>>>(m.dir, m.residual, 1, n0);
    BQ_LAUNCH_CHECK("mg_mul_kernel");
    slab_max(m, m.residual, tempResult, 2000);
    slab_dot(m, m.residual, m.residual, tempResult, 0);                                      // r.r

    for (int it = 0; it < iter; it++) {
        const int off = it * 2;
        // dir and residual are correct on every stored plane here (exchanged below / above)
        if (!mg_stencil_lean(m.temp0, nullptr, m.dir, ni, nj, nkl, true)) {
            mg_poisson_kernel<<<grid_of(ni, nj, nkl), kBlk, 0, st>>>(m.dir, m.temp0, ni, nj, nkl);
            BQ_LAUNCH_CHECK("mg_poisson_kernel");
        }
        slab_dot(m, m.dir, m.temp0, tempResult, off + 1);
        mg_update_x_kernel<<<blocks1d(n0), 256, 0, st>>>(m.p, m.dir, tempResult, n0, off, off + 1);
        BQ_LAUNCH_CHECK("mg_update_x_kernel");
        mg_residual(m.residual, m.div, m.p, ni, nj, nkl);
        slab_exchange(L0, m.residual, L0.G);                // level 0's right-hand side, ghost planes included

        slab_v_cycle(m, it == iter - 1);
        slab_exchange(L0, m.residual, L0.G);
        slab_max(m, m.residual, tempResult, 2001 + it);

        slab_dot(m, m.residual, m.residual, tempResult, off + 2);
        mg_update_dir_kernel<<<blocks1d(n0), 256, 0, st>>>(m.dir, m.residual, tempResult, n0, off, off + 2);
        BQ_LAUNCH_CHECK("mg_update_dir_kernel");
    }
    // p is correct on every stored plane (the cycle's last exchange + elementwise updates); the gradient needs p(k - 1)
    const int k_first = std::max(L0.lo + 1, 2), k_end = std::min(L0.hi, nkg);
    if (k_end > k_first) {
        mg_gradient_slab_kernel<<<grid_of(ni + 1, nj + 1, k_end - k_first), kBlk, 0, st>>>(u, v, w, m.p, ni, nj, nkg, L0.lo, k_end, ulo, k_first, halfrdx);
        BQ_LAUNCH_CHECK("mg_gradient_slab_kernel");
    }
}

} // extern "C"
