// bq_host.h -- host-side state shared by the C-ABI translation units (runtime, launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../../include/bimocq_gpu.h"

namespace bq {

struct Runtime {
    bool        ready = false;
    int         device = -1;
    hipStream_t compute = nullptr;      // every operator launches here
    hipStream_t halo = nullptr;         // ghost-plane exchange (multi-GPU), overlapped with interior work
    hipStream_t copy = nullptr;         // device -> host downloads of the dump path, overlapped with the next step
    // fl_aux_*: a second compute stream for an operator that is independent of the ones that follow it (the forward-map
    // update beside the backward one): while a section is open `compute` IS the auxiliary stream
    hipStream_t aux = nullptr, compute_main = nullptr;
    hipEvent_t  aux_fork = nullptr, aux_done = nullptr;
    bool        aux_pending = false;
    int         err = FL_OK;
    char        err_text[256] = {0};
    int         opt_residual_stride = 0;
    int         opt_skip_unit_blend = 1;
    int         opt_jacobi_variant = 0;
    int         opt_profile_jacobi = 0;
    int         opt_structured_maps = 1;
    int         opt_jacobi_fuse = 1;        // 0 never, 1 inside gpu_projection_jacobi, 2 also in gpu_jacobi_sweeps
    int         opt_jacobi_kchunk2 = 0;     // planes per block of the fused kernel (0 = auto)    // structured (compile-time taps) map look-up on power-of-two spacing
    int         opt_jacobi_kchunk = 0;      // 0 = auto
    int         opt_fused_housekeeping = 0; // FL_OPT_FUSED_HOUSEKEEPING bit mask
    int         opt_fast_lerp = 0;          // gather kernels: one fp32 fma per lerp instead of the double-evaluated one
    int         opt_map_quarter_fp32 = 0;   // FL_OPT_MAP_QUARTER_FP32: the caller vouches for the maps (gpu_maps_quarter_safe)
    // map-value guard (fl_map_guard_*): gpu_solve_backwardDMC ORs word 0, gpu_solve_forward word 1 of this device array
    // when a map value they store fails tile_value_ok; nullptr while the guard is off
    int        *map_guard = nullptr;
    bool        map_guard_on = false;
    int         opt_mgcg_graph = 1;         // replay the multigrid V-cycle from a captured hipGraph
    int         opt_mgcg_tile = 1;          // FL_OPT_MGCG_TILE: LDS tile smoother on the coarse levels of the V-cycle
    int         opt_profile_comm = 0;       // FL_OPT_PROFILE_COMM: time the compute stream's waits on the halo stream
    int         opt_mgcg_bottom = 1;        // FL_OPT_MGCG_BOTTOM: the two coarsest V-cycle levels in one launch
    int         opt_mgcg_fuse = -1;         // FL_OPT_MGCG_FUSE: level-0 vector updates inside the stencil passes (bq_mgcg_fused.hip.inc); -1 = not vouched for: off
    int         opt_comm_check = 0;         // FL_OPT_COMM_CHECK: ledger of communicator calls (bq_halo.hip)
    int         opt_field_window = -1;      // FL_OPT_FIELD_WINDOW: -1 auto (on with FL_OPT_FAST_LERP), 0 off, 1 on, k > 1 planes per block
    int         opt_reserve_cus = 0;        // FL_OPT_RESERVE_CUS: CUs the compute stream leaves to the halo stream's RCCL kernels
    int         device_cus = 0;             // CUs of the device (hipDeviceProp_t::multiProcessorCount), set by fl_init
    int         num_cus = 256;              // CUs the compute stream may use (device CUs - opt_reserve_cus)
    int         opt_jacobi_rows = 0;        // float4 rows per thread in the tiled kernel (0 = auto)
    // z-slab context (fl_set_slab): local plane k is global plane k + slab_koff of slab_nkg planes;
    // this rank owns global planes [slab_own0, slab_own1) (reductions count only those)
    // plane window (fl_set_plane_window): the map operators that honour it produce the local planes [win_k0, win_k1) only
    bool        win_on = false;
    int         win_k0 = 0, win_k1 = 0;
    bool        slab_on = false;
    int         slab_koff = 0, slab_nkg = 0, slab_own0 = 0, slab_own1 = 0, slab_nkl = 0;  // nkl: local cell planes
    // persistent workspace (replaces the cudaMalloc/cudaFree pair inside the reference's
    // gpu_projection_jacobi, GPU_kernel.cu:1847-1850,1893-1894)
    void  *scratch = nullptr;           // device: reduction partials
    size_t scratch_bytes = 0;
    void  *pinned = nullptr;            // host-pinned mirror for blocking reductions
    size_t pinned_bytes = 0;
    // state the other translation units keep PER CONTEXT (fl_context_*): the communicator and its event ring (bq_halo.hip),
    // the sweep-profile spans and plane range (bq_project.hip), the cached V-cycle graphs (bq_mgcg.hip).  Allocated on first
    // use, released by fl_shutdown through the *_release hooks below.
    void  *halo_state = nullptr, *project_state = nullptr, *mgcg_state = nullptr;
    // tables of the structured map look-up on spacings that are not a power of two (bq_advect.hip: map_tabs): device
    // arrays frac / rel, what they were built for, and which (axis, stagger) pairs conform (bit 2 * axis + S)
    float *map_tab_dev = nullptr;
    float  map_tab_h = 0.f;
    int    map_tab_dims[3] = {0, 0, 0}, map_tab_stride = 0, map_tab_ok = 0;
    int    nonfinite_seen = 0;          // sticky: a gpu_max_abs3 met a NaN or an Inf (fl_nonfinite_seen)
    long long   mg_fused_launches = 0;  // fl_mg_fused_launches
    const char *mg_smooth_kernel = ""; // the fused kernel the last fp64 smoothing call launched first (fl_mg_smooth_kernel_name)
};

Runtime &rt();
void latch(int code, const char *what, const char *detail);
bool ensure_ready(const char *op);      // lazily fl_init(current device); false -> latched
void *scratch(size_t bytes);            // device scratch of at least `bytes` (grows, never shrinks)
void *pinned(size_t bytes);
// bq_halo.hip: in-stream all-reduce of device values across slab ranks (no-op on one rank)
bool comm_allreduce(void *dev, size_t count, bool is_double, bool is_max, hipStream_t st);
int  comm_ranks();
void mgcg_release_graph();              // bq_mgcg.hip: drop the cached V-cycle graphs of the current context (fl_free / fl_shutdown)
void mgcg_release_state(Runtime &r);    // ... and free the per-context state itself (fl_shutdown)
void halo_release_state(Runtime &r);    // bq_halo.hip
void halo_abandon_comm(Runtime &r);     // bq_halo.hip: forget the communicator without destroying it (process exit)
void project_release_state(Runtime &r); // bq_project.hip
// bq_project.hip: FL_OPT_PROFILE_JACOBI spans -- an event pair around a loop of sweep launches on the compute
// stream, summed by fl_jacobi_profile().  profile_begin returns false when profiling is off.
struct ProfileSpan { hipEvent_t a = nullptr, b = nullptr; };
bool profile_begin(ProfileSpan &sp);
void profile_end(ProfileSpan &sp, long long launches, long long sweeps);

inline bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    latch(FL_ERR_HIP, what, hipGetErrorString(e));
    return false;
}

} // namespace bq

#define BQ_HIP(call) ::bq::hip_ok((call), #call)
#define BQ_LAUNCH_CHECK(name) ::bq::hip_ok(hipGetLastError(), name)

// argument validation shared by the launchers: positive dims, non-null pointers
#define BQ_REQUIRE(cond, op)                                                        \
    do {                                                                            \
        if (!(cond)) { ::bq::latch(FL_ERR_BAD_ARGUMENT, op, #cond); return; }       \
    } while (0)
