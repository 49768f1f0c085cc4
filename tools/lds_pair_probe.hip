// lds_pair_probe.hip -- how should a gather kernel read the (x, x+1) corner pair of a trilinear cell out of an LDS window?
// Lane l of a wave reads the two dwords at dword address base + l + s (neighbouring lanes overlap by one dword, as the
// corner pairs of 64 consecutive nodes do), in four forms:
//   A  ds_read_b64 at a 4-byte aligned address (unaligned for every second lane)     -- 1 instruction
//   B  ds_read2_b32 offset0:0 offset1:1                                               -- 1 instruction, two dword passes
//   C  two ds_read_b32                                                                -- 2 instructions
//   D  ds_read_b64 at 8-byte aligned addresses 2l (reference: no overlap)
// Checks that A returns the right dwords on gfx950 (LDS unaligned access mode) and prints LDS cycles per wave-instruction
// per CU for each form (256 CUs x 4 waves/SIMD busy).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_pair_probe.hip -o /tmp/lds_pair_probe && /tmp/lds_pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 2048

template <int FORM>
__global__ __launch_bounds__(256) void probe(float *out, int *bad, int shift)
{
    __shared__ float win[4096];
    for (int a = threadIdx.x; a < 4096; a += 256) win[a] = (float)a;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // a "row" per wave, lanes on consecutive dwords, plus a per-iteration wobble so that nothing is hoisted
    unsigned addr = (unsigned)(wv * 512 + lane + shift) * 4u;
    if (FORM == 3) addr = (unsigned)(wv * 512 + 2 * lane) * 4u;
    const unsigned lds0 = (unsigned)(size_t)win;     // LDS byte address of the array (group segment offset)
    float acc0 = 0.f, acc1 = 0.f;
    int wrong = 0;
    for (int it = 0; it < ITER; it++) {
        const unsigned a = lds0 + addr + (unsigned)((it & 7) * 288);
        float v0, v1;
        if (FORM == 0 || FORM == 3) {
            double d;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory");
            const unsigned long long b = __builtin_bit_cast(unsigned long long, d);
            v0 = __builtin_bit_cast(float, (unsigned)b); v1 = __builtin_bit_cast(float, (unsigned)(b >> 32));
        } else if (FORM == 1) {
            double d;
            asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory");
            const unsigned long long b = __builtin_bit_cast(unsigned long long, d);
            v0 = __builtin_bit_cast(float, (unsigned)b); v1 = __builtin_bit_cast(float, (unsigned)(b >> 32));
        } else {
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a) : "memory");
        }
        const float e0 = (float)((a - lds0) / 4u), e1 = e0 + 1.f;
        wrong += (v0 != e0) | (v1 != e1);
        acc0 += v0; acc1 += v1;
    }
    if (wrong) atomicAdd(bad, wrong);
    if (acc0 + acc1 == -1.f) out[0] = acc0;
}

// throughput form: 8 independent reads in flight, no per-read wait
template <int FORM>
__global__ __launch_bounds__(256) void rate(float *out, int shift)
{
    __shared__ float win[8192];
    for (int a = threadIdx.x; a < 8192; a += 256) win[a] = (float)a;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned addr = (unsigned)(size_t)win + (unsigned)(wv * 1024 + lane + shift) * 4u;
    if (FORM == 3) addr = (unsigned)(size_t)win + (unsigned)(wv * 1024 + 2 * lane) * 4u;
    double s = 0.0;
    for (int it = 0; it < ITER; it++) {
        double d[8];
        if (FORM == 0 || FORM == 3) {
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:288\n\tds_read_b64 %2, %8 offset:576\n\tds_read_b64 %3, %8 offset:864\n\t"
                         "ds_read_b64 %4, %8 offset:1152\n\tds_read_b64 %5, %8 offset:1440\n\tds_read_b64 %6, %8 offset:1728\n\tds_read_b64 %7, %8 offset:2016\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]) : "v"(addr) : "memory");
        } else if (FORM == 1) {
            asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:1\n\tds_read2_b32 %1, %8 offset0:72 offset1:73\n\tds_read2_b32 %2, %8 offset0:144 offset1:145\n\t"
                         "ds_read2_b32 %3, %8 offset0:216 offset1:217\n\tds_read2_b32 %4, %8 offset0:32 offset1:33\n\tds_read2_b32 %5, %8 offset0:104 offset1:105\n\t"
                         "ds_read2_b32 %6, %8 offset0:176 offset1:177\n\tds_read2_b32 %7, %8 offset0:248 offset1:249\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]) : "v"(addr) : "memory");
        } else {
            float f[16];
            asm volatile("ds_read_b32 %0, %16\n\tds_read_b32 %1, %16 offset:4\n\tds_read_b32 %2, %16 offset:288\n\tds_read_b32 %3, %16 offset:292\n\t"
                         "ds_read_b32 %4, %16 offset:576\n\tds_read_b32 %5, %16 offset:580\n\tds_read_b32 %6, %16 offset:864\n\tds_read_b32 %7, %16 offset:868\n\t"
                         "ds_read_b32 %8, %16 offset:1152\n\tds_read_b32 %9, %16 offset:1156\n\tds_read_b32 %10, %16 offset:1440\n\tds_read_b32 %11, %16 offset:1444\n\t"
                         "ds_read_b32 %12, %16 offset:1728\n\tds_read_b32 %13, %16 offset:1732\n\tds_read_b32 %14, %16 offset:2016\n\tds_read_b32 %15, %16 offset:2020\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7]),
                           "=&v"(f[8]), "=&v"(f[9]), "=&v"(f[10]), "=&v"(f[11]), "=&v"(f[12]), "=&v"(f[13]), "=&v"(f[14]), "=&v"(f[15]) : "v"(addr) : "memory");
            for (int q = 0; q < 8; q++) d[q] = (double)f[2 * q] + (double)f[2 * q + 1];
        }
        for (int q = 0; q < 8; q++) s += d[q];
        addr ^= (unsigned)((it & 1) << 2);          // (keeps the loop body from being hoisted)
    }
    if (s == -1.0) out[0] = (float)s;
}

template <int FORM> static double time_rate(float *out, int shift, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    rate<FORM><<<blocks, 256>>>(out, shift);
    hipEventRecord(a);
    rate<FORM><<<blocks, 256>>>(out, shift);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *out; int *bad;
    hipMalloc(&out, 64); hipMalloc(&bad, 4);
    const char *names[4] = { "ds_read_b64 4-byte aligned (overlapping pairs)", "ds_read2_b32 0/1", "2 x ds_read_b32", "ds_read_b64 8-byte aligned" };
    for (int shift = 0; shift < 2; shift++) {
        int h[4];
        for (int f = 0; f < 4; f++) {
            hipMemset(bad, 0, 4);
            if (f == 0) probe<0><<<64, 256>>>(out, bad, shift);
            if (f == 1) probe<1><<<64, 256>>>(out, bad, shift);
            if (f == 2) probe<2><<<64, 256>>>(out, bad, shift);
            if (f == 3) probe<3><<<64, 256>>>(out, bad, shift);
            hipMemcpy(&h[f], bad, 4, hipMemcpyDeviceToHost);
            printf("shift %d  %-50s wrong values: %d\n", shift, names[f], h[f]);
        }
    }
    // 256 CUs x 4 blocks of 4 waves: 16 waves per CU.  pair-reads per CU = 16 waves x ITER x 8.
    const int blocks = 256 * 4;
    for (int shift = 0; shift < 2; shift++) {
        double ms[4] = { time_rate<0>(out, shift, blocks), time_rate<1>(out, shift, blocks), time_rate<2>(out, shift, blocks), time_rate<3>(out, shift, blocks) };
        for (int f = 0; f < 4; f++) {
            const double pairs_per_cu = 16.0 * ITER * 8.0;
            printf("shift %d  %-50s %.3f ms  = %.2f ns per pair-read per CU = %.2f cycles at 2.4 GHz\n", shift, names[f], ms[f],
                   ms[f] * 1e6 / pairs_per_cu, ms[f] * 1e6 / pairs_per_cu * 2.4);
        }
    }
    return 0;
}
