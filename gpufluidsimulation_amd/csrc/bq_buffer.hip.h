// bq_buffer.hip.h -- 16-byte buffer loads / stores through resource descriptors, for the marching kernels
// (bq_project.hip: jacobi_lean*_kernel, bq_mgcg.hip: mg_lean2r_kernel).
// Resource descriptor in SGPRs, a 32-bit byte offset per thread (VGPR), a wave-uniform byte offset (SGPR): no 64-bit
// address arithmetic per load, and the range check of the descriptor answers 0 for offsets beyond the array.
// hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128 lowers to a ONE-dword load, so the LLVM intrinsics are declared
// directly (asm labels).
#pragma once
#include <hip/hip_runtime.h>

namespace bq {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ v4f bq_buffer_load_x4(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ v2f bq_buffer_load_x2(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ float bq_buffer_load_x1(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void bq_buffer_store_x4(v4f data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ void bq_buffer_store_x2(v2f data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");

__device__ __forceinline__ v4i make_rsrc4(const void *ptr, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)ptr;
    return v4i{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

} // namespace bq
