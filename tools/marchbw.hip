// marchbw.hip -- what the memory system gives a plane-marching triad (tools only).
// The fused Jacobi kernels give every block a few rows and let it march along z: per step a block touches 8 KB of each
// array and then jumps one plane (1 MB at 512^2) ahead.  This measures out = a + s*b with exactly that block -> address
// mapping, and with the march along y instead (consecutive steps touch consecutive rows; the block's "rows" are planes),
// against the contiguous grid-stride triad of tools/membw.hip.
// Build: hipcc --offload-arch=gfx950 -O3 tools/marchbw.hip -o tools/marchbw ; run: tools/marchbw [nx ny nz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// 256 threads = `rows` row pairs of cw float4 lanes; a thread owns two float4 (rows j, j+1 -- or planes in y-march mode).
// inner stride si (elements) between the thread's two "rows", march stride sm, nmarch steps per chunk of kc.
template <int PF>
__global__ __launch_bounds__(256) void march_triad(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ o,
                                                    int nx, int nrow, int nmarch, size_t srow, size_t smarch, int cw, int nby, int kc, float s, int xcd)
{
    const int nblk = gridDim.x;
    int bi = blockIdx.x;
    if (xcd && (nblk & 7) == 0) bi = (bi & 7) * (nblk >> 3) + (bi >> 3);
    const int by = bi % nby, bz = bi / nby;
    const int rows = 256 / cw;
    const int c = threadIdx.x % cw, r = threadIdx.x / cw;
    const int x = 4 * c, j = 2 * (by * rows + r);
    if (x >= nx || j + 1 >= nrow) return;
    const int k0 = bz * kc, k1 = min(nmarch, k0 + kc);
    const size_t base = (size_t)x + srow * (size_t)j;
    float4 A[PF + 1][2], B[PF + 1][2];
#pragma unroll
    for (int d = 0; d < PF; d++) {
        const size_t at = base + smarch * (size_t)min(k0 + d, nmarch - 1);
        A[d][0] = *(const float4 *)(a + at); A[d][1] = *(const float4 *)(a + at + srow);
        B[d][0] = *(const float4 *)(b + at); B[d][1] = *(const float4 *)(b + at + srow);
    }
    for (int k = k0; k < k1; k += PF + 1) {
#pragma unroll
        for (int t = 0; t < PF + 1; t++) {
            if (k + t < k1) {
                const int slot_new = (t + PF) % (PF + 1);
                const size_t an = base + smarch * (size_t)min(k + t + PF, nmarch - 1);
                A[slot_new][0] = *(const float4 *)(a + an); A[slot_new][1] = *(const float4 *)(a + an + srow);
                B[slot_new][0] = *(const float4 *)(b + an); B[slot_new][1] = *(const float4 *)(b + an + srow);
                const size_t at = base + smarch * (size_t)(k + t);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float4 p = A[t][h], q = B[t][h];
                    *(float4 *)(o + at + srow * h) = make_float4(p.x + s * q.x, p.y + s * q.y, p.z + s * q.z, p.w + s * q.w);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void triad(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ o, size_t n4, float s)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 x = a[i], y = b[i];
        o[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}

int main(int argc, char **argv)
{
    const int nx = argc > 3 ? atoi(argv[1]) : 512, ny = argc > 3 ? atoi(argv[2]) : 512, nz = argc > 3 ? atoi(argv[3]) : 512;
    const size_t N = (size_t)nx * ny * nz;
    float *a, *b, *c;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&c, N * 4));
    CK(hipMemset(a, 0, N * 4)); CK(hipMemset(b, 0, N * 4)); CK(hipMemset(c, 0, N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 30;
    int cw = 16;
    while (cw * 4 < nx) cw *= 2;
    const int rows = 256 / cw;
    auto time_it = [&](auto launch, const char *name) {
        float ms = 0;
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0));
            float *in = a, *out = c;
            for (int r = 0; r < reps; r++) { launch(in, out); std::swap(in, out); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double us = ms * 1e3 / reps;
        printf("%-44s %8.2f us  %7.1f GB/s\n", name, us, 12.0 * N / us / 1e3);
    };
    printf("grid %d x %d x %d\n", nx, ny, nz);
    time_it([&](float *in, float *out) { triad<<<4096, 256>>>((const float4 *)in, (const float4 *)b, (float4 *)out, N / 4, 0.5f); }, "contiguous triad, 4096 blocks");
    char name[128];
    for (int xcd = 0; xcd < 2; xcd++)
        for (int kc : { 32, 64, 86, 128, 512 }) {
            {   // march along z: rows = y, march stride = plane
                const int nby = (ny / 2 + rows - 1) / rows, nbz = (nz + kc - 1) / kc;
                snprintf(name, sizeof name, "z-march kc=%3d xcd=%d pf=1 (%d blocks)", kc, xcd, nby * nbz);
                time_it([&](float *in, float *out) { march_triad<1><<<nby * nbz, 256>>>(in, b, out, nx, ny, nz, (size_t)nx, (size_t)nx * ny, cw, nby, kc, 0.5f, xcd); }, name);
                snprintf(name, sizeof name, "z-march kc=%3d xcd=%d pf=2", kc, xcd);
                time_it([&](float *in, float *out) { march_triad<2><<<nby * nbz, 256>>>(in, b, out, nx, ny, nz, (size_t)nx, (size_t)nx * ny, cw, nby, kc, 0.5f, xcd); }, name);
            }
            {   // march along y: "rows" = planes (stride nx*ny), march stride = one row
                const int kcy = std::min(kc, ny);
                const int nby = (nz / 2 + rows - 1) / rows, nbz = (ny + kcy - 1) / kcy;
                snprintf(name, sizeof name, "y-march kc=%3d xcd=%d pf=1 (%d blocks)", kcy, xcd, nby * nbz);
                time_it([&](float *in, float *out) { march_triad<1><<<nby * nbz, 256>>>(in, b, out, nx, nz, ny, (size_t)nx * ny, (size_t)nx, cw, nby, kcy, 0.5f, xcd); }, name);
                snprintf(name, sizeof name, "y-march kc=%3d xcd=%d pf=2", kcy, xcd);
                time_it([&](float *in, float *out) { march_triad<2><<<nby * nbz, 256>>>(in, b, out, nx, nz, ny, (size_t)nx * ny, (size_t)nx, cw, nby, kcy, 0.5f, xcd); }, name);
            }
        }
    return 0;
}
