// How long does a grid-wide barrier take inside ONE launch on MI355X (8 XCDs, L2 per XCD)?  The question behind it: the
// multigrid V-cycle's levels 63^3 and below are ~36 launches of 5-11 us per cycle, bound by launch latency inside the graph
// (DESIGN.md section 8); one persistent launch that separates its sweeps by barriers pays off only if a barrier -- agent-scope
// release + acquire, i.e. an L2 write-back and invalidate on every XCD -- costs clearly less than a graph node.
// Each block owns 2 KB of doubles; per round it rewrites them from its LEFT neighbour block's values of the previous round
// (so the data really crosses XCDs: consecutive blocks sit on different XCDs) and the grid meets at a sense-reversing barrier.
//   hipcc --offload-arch=gfx950 -O3 -o build/grid_barrier_probe tools/grid_barrier_probe.hip && build/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// (every wait is bounded: a barrier that does not complete within ~2^22 polls sets `gave_up` and lets the kernel end)
__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned *gen, unsigned nblocks, unsigned *gave_up)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                            // release: this block's stores leave its XCD's L2
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (atomicAdd(count, 1u) == nblocks - 1) {
            atomicExch(count, 0u);
            __threadfence();
            atomicAdd(gen, 1u);
        } else {
            unsigned polls = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > (1u << 22)) { atomicExch(gave_up, 1u); break; }
            }
        }
        __threadfence();                                            // acquire: stale lines of the other XCDs' data are dropped
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void probe(double *a, double *b, unsigned *count, unsigned *gen, int rounds, int with_barrier)
{
    const unsigned nb = gridDim.x, me = blockIdx.x, left = (me + nb - 1) % nb;
    double *src = a, *dst = b;
    for (int r = 0; r < rounds; r++) {
        dst[me * 256 + threadIdx.x] = __builtin_nontemporal_load(&src[left * 256 + threadIdx.x]) + 1.0;
        if (with_barrier) grid_barrier(count, gen, nb, gen + 8);
        if (with_barrier && __hip_atomic_load(gen + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        double *t = src; src = dst; dst = t;
    }
}

int main()
{
    const int rounds = 200;
    for (int nb : { 32, 64, 128, 216, 256 }) {
        double *a, *b; unsigned *sync;
        hipMalloc(&a, nb * 256 * sizeof(double)); hipMalloc(&b, nb * 256 * sizeof(double)); hipMalloc(&sync, 256);
        hipMemset(a, 0, nb * 256 * sizeof(double)); hipMemset(b, 0, nb * 256 * sizeof(double)); hipMemset(sync, 0, 256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms[2] = { 0, 0 };
        for (int wb = 0; wb < 2; wb++) {
            probe<<<nb, 256>>>(a, b, sync, sync + 32, rounds, wb);          // warm-up
            hipDeviceSynchronize();
            hipMemset(a, 0, nb * 256 * sizeof(double)); hipMemset(b, 0, nb * 256 * sizeof(double));
            hipEventRecord(e0);
            probe<<<nb, 256>>>(a, b, sync, sync + 32, rounds, wb);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[wb], e0, e1);
        }
        // with the barrier every element has been incremented once per round along the ring: value = rounds
        std::vector<double> h(nb * 256);
        hipMemcpy(h.data(), rounds % 2 ? b : a, h.size() * sizeof(double), hipMemcpyDeviceToHost);
        int bad = 0;
        for (double v : h) bad += v != (double)rounds;
        unsigned gave_up = 0;
        hipMemcpy(&gave_up, sync + 40, sizeof(unsigned), hipMemcpyDeviceToHost);
        printf("blocks %3d: %7.3f us per round with the barrier, %6.3f without; %d of %zu values wrong%s\n", nb,
               ms[1] * 1e3 / rounds, ms[0] * 1e3 / rounds, bad, h.size(), gave_up ? "  (A BARRIER GAVE UP)" : "");
        hipFree(a); hipFree(b); hipFree(sync);
    }
    return 0;
}
