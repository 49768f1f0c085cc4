// membw.hip -- streaming ceilings for the Jacobi traffic shape on this GPU (tools only).
//   triad:  out = a + s*b            2 reads + 1 write of N floats  (= the 12 B/voxel shape)
//   copy:   out = a                  1 read + 1 write
//   read2:  sum(a + b) per thread    2 reads
// Build: hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void triad(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ o, size_t n4, float s)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 x = a[i], y = b[i];
        o[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}
__global__ __launch_bounds__(256) void copyk(const float4 *__restrict__ a, float4 *__restrict__ o, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
__global__ __launch_bounds__(256) void read2(const float4 *__restrict__ a, const float4 *__restrict__ b, float *__restrict__ o, size_t n4)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 x = a[i], y = b[i];
        acc += x.x + y.x + x.y + y.y + x.z + y.z + x.w + y.w;
    }
    if (acc == 123.456f) o[0] = acc;
}

int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 256;
    size_t N = (size_t)n * n * n, n4 = N / 4;
    float *a, *b, *c;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&c, N * 4));
    CK(hipMemset(a, 0, N * 4)); CK(hipMemset(b, 0, N * 4)); CK(hipMemset(c, 0, N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    int grids[] = { 256, 512, 1024, 2048, 4096, 8192, 16384, 65536 };
    for (int g : grids) {
        float ms;
        // ping-pong like the Jacobi loop: out of sweep s is the input of sweep s+1
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0));
            float *in = a, *out = c;
            for (int r = 0; r < reps; r++) { triad<<<g, 256>>>((float4 *)in, (float4 *)b, (float4 *)out, n4, 0.5f); std::swap(in, out); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        double us = ms * 1e3 / reps;
        printf("triad  grid %6d: %7.2f us  %7.1f GB/s\n", g, us, 12.0 * N / us / 1e3);
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0));
            float *in = a, *out = c;
            for (int r = 0; r < reps; r++) { copyk<<<g, 256>>>((float4 *)in, (float4 *)out, n4); std::swap(in, out); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        us = ms * 1e3 / reps;
        printf("copy   grid %6d: %7.2f us  %7.1f GB/s\n", g, us, 8.0 * N / us / 1e3);
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) read2<<<g, 256>>>((float4 *)a, (float4 *)b, c, n4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        us = ms * 1e3 / reps;
        printf("read2  grid %6d: %7.2f us  %7.1f GB/s\n", g, us, 8.0 * N / us / 1e3);
    }
    return 0;
}
