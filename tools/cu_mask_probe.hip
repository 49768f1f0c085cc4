// tools/cu_mask_probe.hip -- which compute units does a hipExtStreamCreateWithCUMask stream really use?
//   hipcc --offload-arch=gfx950 -O2 -o build/cu_mask_probe tools/cu_mask_probe.hip && build/cu_mask_probe [reserved=8]
// Launches many long-lived workgroups on a stream whose mask has the top `reserved` bits cleared; every workgroup records
// its XCC_ID and HW_ID (SE / CU).  Prints the CUs seen per XCD with and without the mask: FL_OPT_RESERVE_CUS relies on
// "clearing the top k bits takes k / 8 CUs from every XCD" (bq_runtime.hip: create_compute_stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned *out, int spin)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the CU busy for a while so that the whole grid spreads over everything the mask allows
    float a = (float)threadIdx.x;
    for (int s = 0; s < spin; s++) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
    if (a == 12345.f) out[0] = 0;
}

static void survey(hipStream_t st, const char *label)
{
    const int blocks = 8192;
    unsigned *d = nullptr;
    hipMalloc(&d, blocks * 2 * sizeof(unsigned));
    where_kernel<<<blocks, 256, 0, st>>>(d, 20000);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(blocks * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < blocks; b++) {
        const unsigned hw = h[2 * b + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[h[2 * b]].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%s:", label);
    for (auto &kv : per_xcc) { printf("  xcc%u=%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  total=%d CUs\n", total);
    hipFree(d);
}

int main(int argc, char **argv)
{
    const int reserved = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int total = prop.multiProcessorCount;
    printf("device: %s, %d CUs; reserving %d\n", prop.gcnArchName, total, reserved);
    hipStream_t plain, masked;
    hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    uint32_t mask[16] = { 0 };
    for (int b = 0; b < total - reserved; b++) mask[b / 32] |= 1u << (b % 32);
    if (hipExtStreamCreateWithCUMask(&masked, (total + 31) / 32, mask) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
    survey(plain, "plain stream ");
    survey(masked, "masked stream");
    return 0;
}
