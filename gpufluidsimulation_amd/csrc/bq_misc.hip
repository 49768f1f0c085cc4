// bq_misc.hip -- streaming operators: smoke source, buoyancy,
// axpy-style helpers, identity-map initialisation and the device CFL max-reduction.
#include "bq_device.hip.h"
#include "bq_host.h"

namespace bq {

static const dim3 kBlock3(64, 4, 1);
static inline dim3 grid3(int a, int b, int c) { return dim3((a + 63) / 64, (b + 3) / 4, c); }

// norm3df / hypotf restated as the correctly rounded sqrt of the double sum of squares
// (oracle: norm3, hypot2)
__device__ __forceinline__ float norm3(float x, float y, float z)
{
    return (float)sqrt((double)x * (double)x + (double)y * (double)y + (double)z * (double)z);
}
__device__ __forceinline__ float hypot2(float y, float z)
{
    return (float)sqrt((double)y * (double)y + (double)z * (double)z);
}

/* The oracle's orc_acosf / orc_cosf, operation for operation.  acosf / cosf of the emitter's velocity ring (GPU_kernel.cu:750-752), restated with IEEE double +, -, *, /, sqrt and
 * floor only, so that every compiler and both sides of the parity tests produce the same bits (within 1 ulp of any
 * libm).  acos: asin's Taylor series on |x| <= 1/2 (22 terms, c_k = C(2k,k) / (4^k (2k+1)): truncation < 1e-15),
 * acos(r) = pi/2 - asin(r) for |r| <= 1/2, 2 asin(sqrt((1-|r|)/2)) beyond, reflected for r < 0.  cos: Cody-Waite
 * reduction by pi/2 (two-part constant, exact for the |x| < 2^19 this is specified on; the emitter passes 8 theta <=
 * 8 pi), Taylor series of sin / cos on |y| <= pi/4. */
__device__ __forceinline__ float acos_portable(float rf)
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

__device__ __forceinline__ float cos_portable(float xf)
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

// emit_smoke_velocity_kernel (GPU_kernel.cu:736-758); the u-face offset is used for all three
// components, as in the reference
__global__ __launch_bounds__(256) void emit_velocity_kernel(float *field, float h, int ni, int nj, int nk,
                                                            float cx, float cy, float cz, float radius, float emiter,
                                                            int koff, int nkg)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, kl = blockIdx.z;
    const int k = kl + koff;                    // global plane; nkg = GLOBAL planes of this buffer
    if (!(i > 1 && i < ni - 2 && j > 1 && j < nj - 2 && k > 1 && k < nkg - 2)) return;
    float dx = (float)(((double)(float)i - 0.5) * (double)h - (double)cx);
    float dy = (float)j * h - cy;
    float dz = (float)k * h - cz;
    if (norm3(dx, dy, dz) < radius) {
        float theta = acos_portable(dy / hypot2(dy, dz));
        float c8 = cos_portable((float)(8.0 * (double)theta));
        field[(size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * kl)] = (float)((double)emiter * 0.06 * (1.0 + 0.01 * (double)c8));
    }
}

// emit_smoke_field_kernel (GPU_kernel.cu:760-780)
__global__ __launch_bounds__(256) void emit_field_kernel(float *rho, float *T, float h, int ni, int nj, int nk,
                                                         float cx, float cy, float cz, float radius, float density, float temperature,
                                                         int koff, int nkg)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, kl = blockIdx.z;
    const int k = kl + koff;
    if (!(i > 1 && i < ni - 2 && j > 1 && j < nj - 2 && k > 1 && k < nkg - 2)) return;
    float dx = (float)i * h - cx, dy = (float)j * h - cy, dz = (float)k * h - cz;
    if (norm3(dx, dy, dz) < radius) {
        size_t id = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * kl);
        rho[id] = density;
        T[id] = temperature;
    }
}

// add_buoyancy_kernel (GPU_kernel.cu:804-823) with the intended rho/T indexing
__global__ __launch_bounds__(256) void buoyancy_kernel(float *__restrict__ v, const float *__restrict__ rho, const float *__restrict__ T,
                                                       int ni, int nj, int nk, float alpha, float beta, float dt,
                                                       int koff, int nkg)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= ni || j < 1 || j >= nj || k + koff < 0 || k + koff >= nkg) return;
    const size_t c0 = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * k), c1 = c0 - ni;
    float d0 = rho[c0], T0 = T[c0], d1 = rho[c1], T1 = T[c1];
    float f = (float)(0.5 * (double)dt * (double)(beta * (T0 + T1) - alpha * (d0 + d1)));
    v[(size_t)i + (size_t)ni * ((size_t)j + (size_t)(nj + 1) * k)] += f;
}

// add_kernel / add_field_kernel / mad_kernel (GPU_kernel.cu:560-565, 878-883, 952-957), bounded
__global__ __launch_bounds__(256) void add_kernel(float *__restrict__ f1, const float *__restrict__ f2, float coeff, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) f1[i] += coeff * f2[i];
}
__global__ __launch_bounds__(256) void add_field_kernel(float *out, const float *f1, const float *f2, float coeff, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = f1[i] + coeff * f2[i];
}
__global__ __launch_bounds__(256) void mad_kernel(float *out, const float *f1, const float *f2, float c1, float c2, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = c1 * f1[i] + c2 * f2[i];
}

// MapperBaseGPU::init host loop (Mapping.cpp:310-324) on the device
__global__ __launch_bounds__(256) void init_maps_kernel(float *x, float *y, float *z, float h, int ni, int nj, int nk,
                                                        int koff, int nkg)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= ni || j >= nj) return;
    size_t id = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * k);
    const int kg = k + koff;
    // ghost planes outside the global grid stay 0: that is what a read outside the allocation returns
    const bool in = kg >= 0 && kg < nkg;
    x[id] = in ? (float)i * h : 0.f; y[id] = in ? (float)j * h : 0.f; z[id] = in ? (float)kg * h : 0.f;
}

// getCFL (BimocqGPUSolver.cpp:348-373) as a wave64 max-reduction over the three components
// part[b]: the block's maximum of |x|; flag[b] (may be null): 1 when the block met a NaN or an Inf.  A maximum skips NaNs
// (the reference's host scan `if (fabs(x) > MaxVelocity)` does too), so a field that has gone NaN looks perfectly calm to the
// CFL -- the flag is how a caller finds out (fl_nonfinite_seen): acc += 0 * x turns NaN for a NaN or an Inf and costs nothing
// in a kernel that waits for memory.
__global__ __launch_bounds__(256) void max_abs3_partial_kernel(const float *__restrict__ u, size_t nu,
                                                               const float *__restrict__ v, size_t nv,
                                                               const float *__restrict__ w, size_t nw,
                                                               float *__restrict__ part, float *__restrict__ flag)
{
    float m = 0.f, acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    // 16-byte loads over the aligned bulk of each array (a maximum does not care about the order), scalars for the rest
    auto scan = [&](const float *f, size_t n) {
        const size_t head = min(n, (size_t)((16 - ((uintptr_t)f & 15u)) & 15u) / 4), bulk = (n - head) / 4;
        const float4 *f4 = reinterpret_cast<const float4 *>(f + head);
        for (size_t i = t0; i < bulk; i += stride) {
            const float4 q = f4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
            acc = __builtin_fmaf(0.f, q.x, acc); acc = __builtin_fmaf(0.f, q.y, acc);
            acc = __builtin_fmaf(0.f, q.z, acc); acc = __builtin_fmaf(0.f, q.w, acc);
        }
        for (size_t i = t0; i < head; i += stride) { m = fmaxf(m, fabsf(f[i])); acc = __builtin_fmaf(0.f, f[i], acc); }
        for (size_t i = head + 4 * bulk + t0; i < n; i += stride) { m = fmaxf(m, fabsf(f[i])); acc = __builtin_fmaf(0.f, f[i], acc); }
    };
    scan(u, nu); scan(v, nv); scan(w, nw);
    __shared__ float smax[4];
    __shared__ int sbad[4];
    m = wave_max(m);
    const bool bad = __any(acc != acc);
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = m; sbad[threadIdx.x >> 6] = bad ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        if (flag) flag[blockIdx.x] = (sbad[0] | sbad[1] | sbad[2] | sbad[3]) ? 1.f : 0.f;
    }
}

// out[0] = max(floor, max part); flag != null: out[1] = max flag (0 or 1)
__global__ __launch_bounds__(256) void max_final_kernel(const float *__restrict__ part, int n, float floor_value, float *out,
                                                        const float *__restrict__ flag = nullptr)
{
    float m = 0.f, fl = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { m = fmaxf(m, part[i]); if (flag) fl = fmaxf(fl, flag[i]); }
    __shared__ float smax[4], sfl[4];
    m = wave_max(m);
    fl = wave_max(fl);
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = m; sfl[threadIdx.x >> 6] = fl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        *out = (r > floor_value) ? r : floor_value;     // `if (fabs(x) > MaxVelocity)` starting from 1e-4
        if (flag) out[1] = fmaxf(fmaxf(sfl[0], sfl[1]), fmaxf(sfl[2], sfl[3]));
    }
}

// How far along z the two maps of a set carry their nodes: max over the nodes of the update window (2 <= i, j, kg <= n - 3:
// what forward_kernel / dmc_kernel write; border nodes keep zeros or stale values) of |bz - kg h| and
// |fz - kg h|, per block.  A NaN counts as infinitely far.  Planes [p0, p1) of the local buffers.
__global__ __launch_bounds__(256) void map_travel_z_kernel(const float *__restrict__ bz, const float *__restrict__ fz, float h,
                                                           int ni, int nj, int p0, int koff, int nkg, float *__restrict__ part, int nblocks)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z + p0, kg = k + koff;
    float db = 0.f, df = 0.f;
    if (i > 1 && i < ni - 2 && j > 1 && j < nj - 2 && kg > 1 && kg < nkg - 2) {
        const size_t id = (size_t)i + (size_t)ni * ((size_t)j + (size_t)nj * k);
        const float z = (float)kg * h;
        db = fabsf(bz[id] - z); df = fabsf(fz[id] - z);
        if (db != db) db = __builtin_inff();
        if (df != df) df = __builtin_inff();
    }
    __shared__ float sb[4], sf[4];
    db = wave_max(db); df = wave_max(df);
    if (threadIdx.x == 0) { sb[threadIdx.y] = db; sf[threadIdx.y] = df; }
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        part[b] = fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3]));
        part[nblocks + b] = fmaxf(fmaxf(sf[0], sf[1]), fmaxf(sf[2], sf[3]));
    }
}
__global__ __launch_bounds__(256) void max2_final_kernel(const float *__restrict__ part, int n, float *out)
{
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { a = fmaxf(a, part[i]); b = fmaxf(b, part[n + i]); }
    __shared__ float sa[4], sb[4];
    a = wave_max(a); b = wave_max(b);
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3])); out[1] = fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3])); }
}

// slab context of the library: (koff, nkg); single GPU: (0, nk)
static inline void slab_ctx(int nk, int &koff, int &nkg)
{
    const Runtime &r = rt();
    koff = r.slab_on ? r.slab_koff : 0;
    nkg = r.slab_on ? r.slab_nkg : nk;
}

static inline int stream_blocks(size_t n)
{
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

} // namespace bq

using namespace bq;

#define g_nonfinite_seen (rt().nonfinite_seen)    // sticky, per context: a gpu_max_abs3 met a NaN or an Inf (fl_nonfinite_seen)

extern "C" {

// 1 when some gpu_max_abs3 since the last reset met a NaN or an Inf in u, v or w (on any slab rank); reset != 0 clears it.
// The CFL maximum itself skips NaNs like the reference's host scan, so without this a run that has gone NaN looks calm.
int fl_nonfinite_seen(int reset)
{
    const int v = g_nonfinite_seen;
    if (reset) g_nonfinite_seen = 0;
    return v;
}

void gpu_emit_smoke(float *u, float *v, float *w, float *rho, float *T, float h, int ni, int nj, int nk,
                    float centerX, float centerY, float centerZ, float radius, float density, float temperature, float emiter)
{
    if (!ensure_ready("gpu_emit_smoke")) return;
    BQ_REQUIRE(u && v && w && rho && T && ni > 0 && nj > 0 && nk > 0 && nk < 65535, "gpu_emit_smoke");
    hipStream_t st = rt().compute;
    int koff, nkg;
    slab_ctx(nk, koff, nkg);
    emit_velocity_kernel<<<grid3(ni + 1, nj, nk), kBlock3, 0, st>>>(u, h, ni + 1, nj, nk, centerX, centerY, centerZ, radius, emiter, koff, nkg);
    emit_velocity_kernel<<<grid3(ni, nj + 1, nk), kBlock3, 0, st>>>(v, h, ni, nj + 1, nk, centerX, centerY, centerZ, radius, 0.f, koff, nkg);
    emit_velocity_kernel<<<grid3(ni, nj, nk + 1), kBlock3, 0, st>>>(w, h, ni, nj, nk + 1, centerX, centerY, centerZ, radius, 0.f, koff, nkg + 1);
    emit_field_kernel<<<grid3(ni, nj, nk), kBlock3, 0, st>>>(rho, T, h, ni, nj, nk, centerX, centerY, centerZ, radius, density, temperature, koff, nkg);
    BQ_LAUNCH_CHECK("gpu_emit_smoke");
}

void gpu_add_buoyancy(float *field, float *density, float *temperature, int ni, int nj, int nk, float alpha, float beta, float dt)
{
    if (!ensure_ready("gpu_add_buoyancy")) return;
    BQ_REQUIRE(field && density && temperature && ni > 0 && nj > 0 && nk > 0 && nk < 65535, "gpu_add_buoyancy");
    int koff, nkg;
    slab_ctx(nk, koff, nkg);
    buoyancy_kernel<<<grid3(ni, nj, nk), kBlock3, 0, rt().compute>>>(field, density, temperature, ni, nj, nk, alpha, beta, dt, koff, nkg);
    BQ_LAUNCH_CHECK("buoyancy_kernel");
}

void gpu_add(float *field1, float *field2, float coeff, int number)
{
    if (!ensure_ready("gpu_add")) return;
    BQ_REQUIRE(field1 && field2 && number >= 0, "gpu_add");
    if (!number) return;
    add_kernel<<<stream_blocks(number), 256, 0, rt().compute>>>(field1, field2, coeff, (size_t)number);
    BQ_LAUNCH_CHECK("add_kernel");
}

void gpu_add_field(float *out, float *field1, float *field2, float coeff, int number)
{
    if (!ensure_ready("gpu_add_field")) return;
    BQ_REQUIRE(out && field1 && field2 && number >= 0, "gpu_add_field");
    if (!number) return;
    add_field_kernel<<<stream_blocks(number), 256, 0, rt().compute>>>(out, field1, field2, coeff, (size_t)number);
    BQ_LAUNCH_CHECK("add_field_kernel");
}

void gpu_mad(float *field, float *field1, float *field2, float coeff1, float coeff2, int number)
{
    if (!ensure_ready("gpu_mad")) return;
    BQ_REQUIRE(field && field1 && field2 && number >= 0, "gpu_mad");
    if (!number) return;
    mad_kernel<<<stream_blocks(number), 256, 0, rt().compute>>>(field, field1, field2, coeff1, coeff2, (size_t)number);
    BQ_LAUNCH_CHECK("mad_kernel");
}

void gpu_init_maps(float *x, float *y, float *z, float h, int ni, int nj, int nk)
{
    if (!ensure_ready("gpu_init_maps")) return;
    BQ_REQUIRE(x && y && z && ni > 0 && nj > 0 && nk > 0 && nk < 65535, "gpu_init_maps");
    int koff, nkg;
    slab_ctx(nk, koff, nkg);
    init_maps_kernel<<<grid3(ni, nj, nk), kBlock3, 0, rt().compute>>>(x, y, z, h, ni, nj, nk, koff, nkg);
    BQ_LAUNCH_CHECK("init_maps_kernel");
}

float gpu_max_abs3(const float *u, const float *v, const float *w, int ni, int nj, int nk)
{
    if (!ensure_ready("gpu_max_abs3")) return 0.f;
    if (!u || !v || !w || ni < 1 || nj < 1 || nk < 1) { latch(FL_ERR_BAD_ARGUMENT, "gpu_max_abs3", "bad argument"); return 0.f; }
    const int blocks = 1024;
    float *part = (float *)scratch((2 * blocks + 16) * sizeof(float));
    float *host = (float *)pinned(64);
    if (!part || !host) return 0.f;
    float *flag = part + blocks + 8;
    hipStream_t st = rt().compute;
    // a slab rank reduces the planes it owns (ghost copies may be stale); the last rank also owns w's top plane
    const Runtime &r = rt();
    const int p0 = r.slab_on ? r.slab_own0 - r.slab_koff : 0;
    const int p1 = r.slab_on ? r.slab_own1 - r.slab_koff : nk;
    const int wtop = (!r.slab_on || r.slab_own1 == r.slab_nkg) ? 1 : 0;
    const size_t pu = (size_t)(ni + 1) * nj, pv = (size_t)ni * (nj + 1), pw = (size_t)ni * nj;
    const size_t nu = pu * (p1 - p0), nv = pv * (p1 - p0), nw = pw * (p1 - p0 + wtop);
    max_abs3_partial_kernel<<<blocks, 256, 0, st>>>(u + pu * p0, nu, v + pv * p0, nv, w + pw * p0, nw, part, flag);
    max_final_kernel<<<1, 256, 0, st>>>(part, blocks, 1e-4f, part + blocks, flag);
    BQ_LAUNCH_CHECK("max_abs3");
    comm_allreduce(part + blocks, 2, false, true, st);        // global CFL (and the NaN / Inf flag) over the slab ranks
    BQ_HIP(hipMemcpyAsync(host, part + blocks, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    if (host[1] != 0.f) g_nonfinite_seen = 1;
    return host[0];
}

// max(0, max f) of one field: the host scan of MapperBaseGPU::estimateDistortion (Mapping.cpp:500-516,
// starts from 0 and keeps values that compare greater: NaNs are skipped) as a device reduction; blocking.
float gpu_max_field(const float *field, size_t count)
{
    if (!ensure_ready("gpu_max_field")) return 0.f;
    if (!field) { latch(FL_ERR_BAD_ARGUMENT, "gpu_max_field", "null pointer"); return 0.f; }
    if (rt().slab_on) { latch(FL_ERR_UNSUPPORTED, "gpu_max_field", "not slab-aware (owned planes / all-reduce) yet"); return 0.f; }
    const int blocks = 1024;
    float *part = (float *)scratch((blocks + 16) * sizeof(float));
    float *host = (float *)pinned(64);
    if (!part || !host) return 0.f;
    hipStream_t st = rt().compute;
    max_abs3_partial_kernel<<<blocks, 256, 0, st>>>(field, count, field, 0, field, 0, part, nullptr);    // values are >= 0 or skipped
    max_final_kernel<<<1, 256, 0, st>>>(part, blocks, 0.f, part + blocks);
    BQ_LAUNCH_CHECK("max_field");
    BQ_HIP(hipMemcpyAsync(host, part + blocks, sizeof(float), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    return host[0];
}

// gpu_max_field over the planes this rank owns of a scalar-sized field (ni x nj x nk local planes), all-reduced over the slab
// ranks: estimateDistortion's host scan (Mapping.cpp:500-516) on a z-slab rank.  Single GPU: gpu_max_field of the field.
float gpu_max_field_owned(const float *field, int ni, int nj, int nk)
{
    if (!ensure_ready("gpu_max_field_owned")) return 0.f;
    if (!field || ni < 1 || nj < 1 || nk < 1) { latch(FL_ERR_BAD_ARGUMENT, "gpu_max_field_owned", "bad argument"); return 0.f; }
    const Runtime &r = rt();
    const int p0 = r.slab_on ? r.slab_own0 - r.slab_koff : 0, p1 = r.slab_on ? r.slab_own1 - r.slab_koff : nk;
    const size_t plane = (size_t)ni * nj;
    const int blocks = 1024;
    float *part = (float *)scratch((blocks + 16) * sizeof(float));
    float *host = (float *)pinned(64);
    if (!part || !host) return 0.f;
    hipStream_t st = rt().compute;
    max_abs3_partial_kernel<<<blocks, 256, 0, st>>>(field + plane * p0, plane * (size_t)(p1 - p0), field, 0, field, 0, part, nullptr);
    max_final_kernel<<<1, 256, 0, st>>>(part, blocks, 0.f, part + blocks);
    BQ_LAUNCH_CHECK("max_field_owned");
    comm_allreduce(part + blocks, 1, false, true, st);
    BQ_HIP(hipMemcpyAsync(host, part + blocks, sizeof(float), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    return host[0];
}

// out[0] / out[1]: how many CELLS along z the backward / forward map carries a node at most (max |map_z - z| / h over the
// nodes the map updates write, owned planes, all-reduced over the slab ranks; a NaN counts as infinity).  What a host needs
// to know to size -- or to refuse -- the ghost zone of maps that live for more than one step.  Blocking.
void gpu_map_travel_z(const float *bz, const float *fz, float h, int ni, int nj, int nk, float out[2])
{
    if (out) out[0] = out[1] = 0.f;
    if (!ensure_ready("gpu_map_travel_z")) return;
    if (!bz || !fz || !out || ni < 1 || nj < 1 || nk < 1 || nk >= 65535 || !(h > 0.f)) { latch(FL_ERR_BAD_ARGUMENT, "gpu_map_travel_z", "bad argument"); return; }
    const Runtime &r = rt();
    const int p0 = r.slab_on ? r.slab_own0 - r.slab_koff : 0, p1 = r.slab_on ? r.slab_own1 - r.slab_koff : nk;
    int koff, nkg;
    slab_ctx(nk, koff, nkg);
    const dim3 grid = grid3(ni, nj, p1 - p0);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    float *part = (float *)scratch(((size_t)2 * nblocks + 16) * sizeof(float));
    float *host = (float *)pinned(64);
    if (!part || !host) return;
    hipStream_t st = rt().compute;
    map_travel_z_kernel<<<grid, kBlock3, 0, st>>>(bz, fz, h, ni, nj, p0, koff, nkg, part, nblocks);
    max2_final_kernel<<<1, 256, 0, st>>>(part, nblocks, part + 2 * nblocks);
    BQ_LAUNCH_CHECK("map_travel_z");
    comm_allreduce(part + 2 * nblocks, 2, false, true, st);
    BQ_HIP(hipMemcpyAsync(host, part + 2 * nblocks, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    out[0] = host[0] / h; out[1] = host[1] / h;
}

} // extern "C"
