// valu_rate.hip -- issue rate of the VALU instructions the gather kernels are made of (gfx950).
// Each kernel runs ITER x 16 independent instances of one instruction per wave; every SIMD holds
// 2 waves so dependent-issue latency is hidden.  Prints cycles per wave64 instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 4096
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, DECL, BODY, SINK)                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed)                  \
    {                                                                                    \
        DECL                                                                             \
        for (int it = 0; it < ITER; it++) { REP16(BODY) }                                \
        SINK                                                                             \
    }

#define D_F32 float a[16]; for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x;
#define D_F64 double a[16]; for (int i = 0; i < 16; i++) a[i] = (double)seed + i + threadIdx.x;
#define D_MIX float a[16]; double d[16]; for (int i = 0; i < 16; i++) { a[i] = seed + i + threadIdx.x; d[i] = a[i]; }
#define S_F32 float s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345.f) out[0] = s;
#define S_F64 double s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345.0) out[0] = (float)s;
#define S_MIX float s = 0; for (int i = 0; i < 16; i++) s += a[i] + (float)d[i]; if (s == 12345.f) out[0] = s;

#define B_FMA32(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
#define B_MUL32(i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));
#define B_MUL64(i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(a[i]));
#define B_ADD64(i) asm volatile("v_add_f64 %0, %0, %0" : "+v"(a[i]));
#define B_CVT_64_32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define B_CVT_32_64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define B_FLOOR32(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define B_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));

#define D_I32 int a[16]; for (int i = 0; i < 16; i++) a[i] = (int)seed + i + threadIdx.x;
#define S_I32 int s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345) out[0] = (float)s;
#define D_FI float a[16]; int d[16]; for (int i = 0; i < 16; i++) { a[i] = seed + i + threadIdx.x * 0.37f; d[i] = i; }
#define S_FI float s = 0; for (int i = 0; i < 16; i++) s += a[i] + (float)d[i]; if (s == 12345.f) out[0] = s;
#define B_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(a[i]));
#define B_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(a[i]));
#define B_MADU32(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(dd[i]) : "v"(a[i]) : "vcc");
#define B_ADDU32(i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a[i]));
#define B_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %0" : "+v"(a[i]));
#define B_FRACT(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
#define B_CVTFLR(i) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define B_CVTI2F(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define B_MED3(i) asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(a[i]));
#define B_SUB32(i) asm volatile("v_sub_f32 %0, %0, %0" : "+v"(a[i]));
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a[i]));
#define B_MOV(i) asm volatile("v_mov_b32 %0, %0" : "+v"(a[i]));
#define B_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
KERNEL(k_mullo, D_I32, B_MULLO, S_I32)
KERNEL(k_mad24, D_I32, B_MAD24, S_I32)
KERNEL(k_addu32, D_I32, B_ADDU32, S_I32)
KERNEL(k_lshladd, D_I32, B_LSHLADD, S_I32)
KERNEL(k_fract, D_F32, B_FRACT, S_F32)
KERNEL(k_cvtflr, D_FI, B_CVTFLR, S_FI)
KERNEL(k_cvti2f, D_FI, B_CVTI2F, S_FI)
KERNEL(k_med3, D_F32, B_MED3, S_F32)
KERNEL(k_sub32, D_F32, B_SUB32, S_F32)
KERNEL(k_cndmask, D_F32, B_CNDMASK, S_F32)
KERNEL(k_mov, D_F32, B_MOV, S_F32)
KERNEL(k_dpp, D_F32, B_DPP, S_F32)
KERNEL(k_fma32, D_F32, B_FMA32, S_F32)
KERNEL(k_mul32, D_F32, B_MUL32, S_F32)
KERNEL(k_fma64, D_F64, B_FMA64, S_F64)
KERNEL(k_mul64, D_F64, B_MUL64, S_F64)
KERNEL(k_add64, D_F64, B_ADD64, S_F64)
KERNEL(k_cvt6432, D_MIX, B_CVT_64_32, S_MIX)
KERNEL(k_cvt3264, D_MIX, B_CVT_32_64, S_MIX)
KERNEL(k_floor32, D_F32, B_FLOOR32, S_F32)
KERNEL(k_pkfma32, D_F64, B_PKFMA32, S_F64)

static int g_wps = 2;
template <class K> static void run(const char *name, K k, double clock_ghz, int cus)
{
    float *out; hipMalloc(&out, 4);
    const int blocks = cus * g_wps;              // g_wps blocks x 4 waves per CU = g_wps waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(out, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)g_wps * ITER * 16;
    const double cycles = ms * 1e-3 * clock_ghz * 1e9;
    printf("%-14s %8.3f ms  %6.2f cycles per wave64 instruction\n", name, ms, cycles / instr_per_simd);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    for (int wps : {2, 5}) {
    g_wps = wps;
    printf("---- %d wave(s) per SIMD\n", wps);
    run("v_fma_f32", k_fma32, ghz, p.multiProcessorCount);
    run("v_mul_f32", k_mul32, ghz, p.multiProcessorCount);
    run("v_pk_fma_f32", k_pkfma32, ghz, p.multiProcessorCount);
    run("v_fma_f64", k_fma64, ghz, p.multiProcessorCount);
    run("v_mul_f64", k_mul64, ghz, p.multiProcessorCount);
    run("v_add_f64", k_add64, ghz, p.multiProcessorCount);
    run("v_cvt_f64_f32", k_cvt6432, ghz, p.multiProcessorCount);
    run("v_cvt_f32_f64", k_cvt3264, ghz, p.multiProcessorCount);
    run("v_floor_f32", k_floor32, ghz, p.multiProcessorCount);
    run("v_fract_f32", k_fract, ghz, p.multiProcessorCount);
    run("v_cvt_flr_i32_f32", k_cvtflr, ghz, p.multiProcessorCount);
    run("v_cvt_f32_i32", k_cvti2f, ghz, p.multiProcessorCount);
    run("v_med3_f32", k_med3, ghz, p.multiProcessorCount);
    run("v_sub_f32", k_sub32, ghz, p.multiProcessorCount);
    run("v_cndmask_b32", k_cndmask, ghz, p.multiProcessorCount);
    run("v_mov_b32", k_mov, ghz, p.multiProcessorCount);
    run("v_mov_b32_dpp", k_dpp, ghz, p.multiProcessorCount);
    run("v_mul_lo_u32", k_mullo, ghz, p.multiProcessorCount);
    run("v_mad_u32_u24", k_mad24, ghz, p.multiProcessorCount);
    run("v_add_u32", k_addu32, ghz, p.multiProcessorCount);
    run("v_lshl_add_u32", k_lshladd, ghz, p.multiProcessorCount);
    }
    return 0;
}
