// bq_runtime.hip -- the fl_* runtime mini-ABI (include/bimocq_gpu.h, section 2).
// Replaces the raw CUDA runtime calls of the reference's gpuMapper
// (src/bimocq3D/GPU_Advection.h:214-326): device selection, zero-filled allocation,
// copies, event timing -- plus the error latch the reference lacks.
#include "bq_host.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace bq {

// The library's state lives in contexts (include/bimocq_gpu.h: fl_context_*).  The DEFAULT context is what every call used
// before contexts existed and what a program that never creates one keeps using; fl_context_make_current switches the calling
// THREAD to another one (thread-local, like a HIP device): one process can then drive several devices, or several solvers on
// one device, side by side behind the same C ABI.
static Runtime g_default_rt;
static thread_local Runtime *t_current = nullptr;
static std::mutex g_mu;

// Every context that holds streams (fl_init succeeded, fl_shutdown not yet run).  Process exit releases them through
// shutdown_all_at_exit below: the HIP runtime's own exit-time teardown of a stream created with
// hipExtStreamCreateWithCUMask (the copy stream; the compute stream under FL_OPT_RESERVE_CUS) ran after the profiler's
// finalisation and crashed inside __cxa_finalize in round 3 (two traces, both from processes that left the masked stream
// alive).  The handler is registered with atexit() at the end of the FIRST successful fl_init, i.e. after the runtime's
// lazily created singletons that stream creation touches have registered their destructors -- exit handlers run in reverse
// order of registration, so the library's streams, events, graphs and workspaces are gone while the runtime is still whole.
static std::vector<Runtime *> g_live;
static std::once_flag g_exit_once;
static bool g_exiting = false;          // set by the exit handler: nothing re-creates streams behind it
static void note_live(Runtime *r, bool live)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t a = 0; a < g_live.size(); a++)
        if (g_live[a] == r) { if (!live) g_live.erase(g_live.begin() + a); return; }
    if (live) g_live.push_back(r);
}

Runtime &rt() { return t_current ? *t_current : g_default_rt; }
#define g_rt (::bq::rt())

void latch(int code, const char *what, const char *detail)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rt.err != FL_OK) return;      // keep the FIRST failure
    g_rt.err = code;
    snprintf(g_rt.err_text, sizeof g_rt.err_text, "%s: %s", what ? what : "?", detail ? detail : "");
}

bool ensure_ready(const char *op)
{
    if (g_rt.ready) {
        // the context's device must be the thread's HIP device: a context of another device made on this thread
        // (fl_context_create) or a foreign hipSetDevice would otherwise silently misplace allocations and launches
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != g_rt.device) (void)hipSetDevice(g_rt.device);
        return true;
    }
    if (g_exiting) { latch(FL_ERR_NO_DEVICE, op, "the process is exiting: the library has released its streams"); return false; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (fl_init(dev) != FL_OK) { (void)op; return false; }
    return true;
}

void *scratch(size_t bytes)
{
    if (g_rt.scratch_bytes >= bytes) return g_rt.scratch;
    if (g_rt.scratch) { (void)hipStreamSynchronize(g_rt.compute); (void)hipFree(g_rt.scratch); g_rt.scratch = nullptr; }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    if (!BQ_HIP(hipMalloc(&g_rt.scratch, want))) { g_rt.scratch_bytes = 0; return nullptr; }
    g_rt.scratch_bytes = want;
    return g_rt.scratch;
}

void *pinned(size_t bytes)
{
    if (g_rt.pinned_bytes >= bytes) return g_rt.pinned;
    if (g_rt.pinned) { (void)hipHostFree(g_rt.pinned); g_rt.pinned = nullptr; }
    size_t want = bytes < (1u << 16) ? (1u << 16) : bytes;
    if (!BQ_HIP(hipHostMalloc(&g_rt.pinned, want, hipHostMallocDefault))) { g_rt.pinned_bytes = 0; return nullptr; }
    g_rt.pinned_bytes = want;
    return g_rt.pinned;
}

// The compute stream.  FL_OPT_RESERVE_CUS = k > 0: created with a CU mask that leaves k compute units unused, so that the
// RCCL send/recv kernels of the halo stream find free CUs the moment an exchange is issued instead of queueing behind
// compute workgroups that occupy all of them (the fused Jacobi kernels run one 4-wave block per CU for the whole launch).
// The mask's bits are dealt round-robin over the 8 XCDs (KFD: bit i -> XCD i % 8; tools/cu_mask_probe.hip prints what a
// mask really excludes), so clearing the top k bits takes k / 8 CUs from every XCD.
static bool create_compute_stream()
{
    const int total = g_rt.device_cus > 0 ? g_rt.device_cus : 256;
    int k = g_rt.opt_reserve_cus;
    if (k < 0) k = 0;
    if (k > total / 2) k = total / 2;
    g_rt.num_cus = total - k;
    if (k == 0) return BQ_HIP(hipStreamCreateWithFlags(&g_rt.compute, hipStreamNonBlocking));
    uint32_t mask[16] = { 0 };
    const int words = (total + 31) / 32;
    for (int b = 0; b < total - k; b++) mask[b / 32] |= 1u << (b % 32);
    return BQ_HIP(hipExtStreamCreateWithCUMask(&g_rt.compute, (uint32_t)words, mask));
}

} // namespace bq


extern "C" {

int fl_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        bq::latch(FL_ERR_NO_DEVICE, "fl_init", "no HIP device visible (this library has no CPU fallback)");
        return FL_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        bq::latch(FL_ERR_BAD_ARGUMENT, "fl_init", "device index out of range");
        return FL_ERR_BAD_ARGUMENT;
    }
    if (g_rt.ready && g_rt.device == device) return FL_OK;
    if (g_rt.ready) fl_shutdown();
    if (!BQ_HIP(hipSetDevice(device))) return FL_ERR_HIP;
    hipDeviceProp_t prop;
    if (!BQ_HIP(hipGetDeviceProperties(&prop, device))) return FL_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        bq::latch(FL_ERR_NO_DEVICE, "fl_init: kernels are built for gfx950 only, device is", prop.gcnArchName);
        return FL_ERR_NO_DEVICE;
    }
    g_rt.device_cus = prop.multiProcessorCount;
    if (!bq::create_compute_stream()) return FL_ERR_HIP;
    if (!BQ_HIP(hipStreamCreateWithFlags(&g_rt.halo, hipStreamNonBlocking))) return FL_ERR_HIP;
    // The copy stream carries the dump's device -> host download.  The runtime executes a large copy to pinned memory with
    // a blit KERNEL whose workgroups, stalled on PCIe, occupy every CU they are given: at 1024 x 1024 x 80 planes a 335 MB
    // download ran 6.1 ms and the compute kernels beside it crawled (the CFL reduction 6.1 ms instead of 0.2;
    // gpurun_out/r03a).  PCIe needs no more than a few CUs, so the stream is created with a mask of eight -- one per XCD
    // (KFD deals the mask's bits round-robin over the XCDs, tools/cu_mask_probe.hip) -- and falls back to a plain stream
    // where the runtime refuses masks.
    {
        // BQ_COPY_STREAM_CUS = n (environment; default 8, 0 = a plain stream): how many CUs the copy stream may use
        const char *env = getenv("BQ_COPY_STREAM_CUS");
        int ncopy = env && *env ? atoi(env) : 8;
        if (ncopy < 0) ncopy = 0;
        if (ncopy > 32) ncopy = 32;
        uint32_t mask[16] = { 0 };
        mask[0] = ncopy >= 32 ? 0xffffffffu : ((1u << ncopy) - 1u);
        const int words = (prop.multiProcessorCount + 31) / 32;
        if (ncopy == 0 || hipExtStreamCreateWithCUMask(&g_rt.copy, (uint32_t)words, mask) != hipSuccess) {
            (void)hipGetLastError();
            if (!BQ_HIP(hipStreamCreateWithFlags(&g_rt.copy, hipStreamNonBlocking))) return FL_ERR_HIP;
        }
    }
    {
        // BQ_FIELD_WINDOW = k (environment): initial value of FL_OPT_FIELD_WINDOW for every context of the process -- lets a
        // whole program (the C++ drivers, the multi-process tests) run the z-marching gather kernels without a code change
        const char *env = getenv("BQ_FIELD_WINDOW");
        if (env && *env) g_rt.opt_field_window = atoi(env) < 0 ? -1 : atoi(env);
    }
    g_rt.device = device;
    g_rt.ready = true;
    bq::note_live(&g_rt, true);
    std::call_once(bq::g_exit_once, [] { (void)atexit([] { bq::g_exiting = true; fl_shutdown_all(); }); });
    return FL_OK;
}

// Release every live context (the default one and those of fl_context_create): streams, events, cached graphs, workspaces.
// Registered with atexit() by the first fl_init; a host may call it itself (idempotent).  A communicator still alive at this
// point is abandoned, not destroyed: its peers may be gone already and ncclCommDestroy could wait for them.
void fl_shutdown_all(void)
{
    std::vector<bq::Runtime *> live;
    { std::lock_guard<std::mutex> lk(bq::g_mu); live = bq::g_live; }
    bq::Runtime *prev = bq::t_current;
    for (bq::Runtime *r : live) {
        bq::t_current = r == &bq::g_default_rt ? nullptr : r;
        if (!g_rt.ready) continue;
        (void)hipSetDevice(g_rt.device);
        bq::halo_abandon_comm(g_rt);
        fl_shutdown();
    }
    bq::t_current = prev;
}

void fl_shutdown(void)
{
    if (!g_rt.ready) return;
    (void)hipStreamSynchronize(g_rt.compute);
    (void)hipStreamSynchronize(g_rt.halo);
    (void)hipStreamSynchronize(g_rt.copy);
    // ORDER MATTERS: a stream created with a CU mask (the copy stream by default; the compute stream with FL_OPT_RESERVE_CUS) is
    // destroyed while the unmasked streams still exist.  Destroyed after them, hipStreamDestroy of the masked stream can sit in
    // the runtime's queue-destroy ioctl for ever while the runtime's event thread waits for a lock the call holds (ROCm 7.2; seen
    // at the exit of tools/jacobi_tune.py --n 128 with seven variants, `tools/batches/r04_zu.sh`: masked stream last -> hang,
    // first -> exit in 1 s, no mask -> exit in 1 s; tests/test_gpu_runtime.py keeps the case).
    (void)hipStreamDestroy(g_rt.copy);
    g_rt.copy = nullptr;
    if (g_rt.opt_reserve_cus > 0 && !g_rt.compute_main && !g_rt.aux) { (void)hipStreamDestroy(g_rt.compute); g_rt.compute = nullptr; }
    bq::mgcg_release_state(g_rt);
    bq::halo_release_state(g_rt);
    bq::project_release_state(g_rt);
    if (g_rt.scratch) (void)hipFree(g_rt.scratch);
    if (g_rt.pinned) (void)hipHostFree(g_rt.pinned);
    if (g_rt.map_guard) (void)hipFree(g_rt.map_guard);
    if (g_rt.map_tab_dev) (void)hipFree(g_rt.map_tab_dev);
    g_rt.map_tab_dev = nullptr; g_rt.map_tab_h = 0.f; g_rt.map_tab_ok = 0;
    g_rt.map_guard = nullptr; g_rt.map_guard_on = false;
    if (g_rt.compute_main) { g_rt.compute = g_rt.compute_main; g_rt.compute_main = nullptr; }
    if (g_rt.aux) { (void)hipStreamSynchronize(g_rt.aux); (void)hipStreamDestroy(g_rt.aux); g_rt.aux = nullptr; }
    if (g_rt.aux_fork) { (void)hipEventDestroy(g_rt.aux_fork); g_rt.aux_fork = nullptr; }
    if (g_rt.aux_done) { (void)hipEventDestroy(g_rt.aux_done); g_rt.aux_done = nullptr; }
    g_rt.aux_pending = false;
    if (g_rt.compute) (void)hipStreamDestroy(g_rt.compute);
    (void)hipStreamDestroy(g_rt.halo);
    g_rt.scratch = nullptr; g_rt.scratch_bytes = 0;
    g_rt.pinned = nullptr; g_rt.pinned_bytes = 0;
    g_rt.compute = nullptr; g_rt.halo = nullptr;
    g_rt.ready = false; g_rt.device = -1;
    bq::note_live(&g_rt, false);
}

// ---- contexts -------------------------------------------------------------------------------------------------------
struct fl_context { bq::Runtime rt; };

// back on the caller's context: the thread's HIP device follows it (fl_init / fl_shutdown of another context moved it)
static void restore_device_of_current()
{
    const bq::Runtime &r = bq::rt();
    if (r.ready) (void)hipSetDevice(r.device);
}

fl_context *fl_context_create(int device)
{
    fl_context *c = new fl_context();
    bq::Runtime *prev = bq::t_current;
    int dev_before = -1;
    if (hipGetDevice(&dev_before) != hipSuccess) dev_before = -1;
    bq::t_current = &c->rt;
    const int rc = fl_init(device);
    if (rc != FL_OK) {
        // report on the caller's context, where it will look
        char text[256];
        snprintf(text, sizeof text, "%s", c->rt.err_text);
        bq::t_current = prev;
        if (bq::rt().ready) restore_device_of_current(); else if (dev_before >= 0) (void)hipSetDevice(dev_before);
        bq::latch(rc, "fl_context_create", text);
        delete c;
        return nullptr;
    }
    bq::t_current = prev;
    // the caller's context stays current, so its device does too (a caller whose context holds no device yet keeps the HIP
    // device it had)
    if (bq::rt().ready) restore_device_of_current(); else if (dev_before >= 0) (void)hipSetDevice(dev_before);
    return c;
}

void fl_context_make_current(fl_context *c)
{
    bq::t_current = c ? &c->rt : nullptr;
    bq::Runtime &r = bq::rt();
    if (r.ready) (void)hipSetDevice(r.device);
}

fl_context *fl_context_current(void)
{
    // (the Runtime is the first member of fl_context)
    return bq::t_current ? reinterpret_cast<fl_context *>(bq::t_current) : nullptr;
}

void fl_context_destroy(fl_context *c)
{
    if (!c) return;
    bq::Runtime *prev = bq::t_current == &c->rt ? nullptr : bq::t_current;
    bq::t_current = &c->rt;
    fl_comm_destroy();
    fl_shutdown();
    bq::t_current = prev;
    restore_device_of_current();            // prev == nullptr: the default context's device, when it holds one
    delete c;
}

void *fl_malloc(size_t bytes)
{
    if (!bq::ensure_ready("fl_malloc")) return nullptr;
    void *p = nullptr;
    if (!BQ_HIP(hipMalloc(&p, bytes ? bytes : 4))) return nullptr;
    if (!BQ_HIP(hipMemsetAsync(p, 0, bytes ? bytes : 4, g_rt.compute))) { (void)hipFree(p); return nullptr; }
    return p;
}

void fl_free(void *p)
{
    if (!p) return;
    bq::mgcg_release_graph();                  // a cached graph may hold this pointer
    if (g_rt.ready) (void)hipStreamSynchronize(g_rt.compute);
    BQ_HIP(hipFree(p));
}

void fl_memset(void *dst, int value, size_t bytes)
{
    if (!bq::ensure_ready("fl_memset")) return;
    BQ_REQUIRE(dst != nullptr || bytes == 0, "fl_memset");
    if (bytes) BQ_HIP(hipMemsetAsync(dst, value, bytes, g_rt.compute));
}

void fl_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
    if (!bq::ensure_ready("fl_memcpy_h2d")) return;
    BQ_REQUIRE((dst && src) || bytes == 0, "fl_memcpy_h2d");
    if (!bytes) return;
    BQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_rt.compute));
    BQ_HIP(hipStreamSynchronize(g_rt.compute));
}

void fl_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
    if (!bq::ensure_ready("fl_memcpy_d2h")) return;
    BQ_REQUIRE((dst && src) || bytes == 0, "fl_memcpy_d2h");
    if (!bytes) return;
    BQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_rt.compute));
    BQ_HIP(hipStreamSynchronize(g_rt.compute));
}

void fl_memcpy_d2d(void *dst, const void *src, size_t bytes)
{
    if (!bq::ensure_ready("fl_memcpy_d2d")) return;
    BQ_REQUIRE((dst && src) || bytes == 0, "fl_memcpy_d2d");
    if (bytes && dst != src) BQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_rt.compute));
}

void fl_sync(void)
{
    if (!g_rt.ready) return;
    BQ_HIP(hipStreamSynchronize(g_rt.compute));
    if (g_rt.compute_main) BQ_HIP(hipStreamSynchronize(g_rt.compute_main));
    if (g_rt.aux && g_rt.aux_pending) BQ_HIP(hipStreamSynchronize(g_rt.aux));
    BQ_HIP(hipStreamSynchronize(g_rt.halo));
}

// ---- an auxiliary compute stream for ONE independent operator (round 4) -----------------------------------------------------
// fl_aux_begin(): operators launched from here on go to a second stream, which first waits for everything queued on the
// compute stream; fl_aux_end(): launches go to the compute stream again while the auxiliary one keeps running;
// fl_aux_join(): the compute stream waits for what the section launched.  The caller vouches that the section's operators
// neither read nor write anything the operators between end and join write (the host solver: the forward-map update, which
// reads the velocity and updates its own three arrays, beside the backward one's sub-steps).  No blocking call inside a section.
void fl_aux_begin(void)
{
    if (!bq::ensure_ready("fl_aux_begin") || g_rt.compute_main) return;
    if (!g_rt.aux && !BQ_HIP(hipStreamCreateWithFlags(&g_rt.aux, hipStreamNonBlocking))) { g_rt.aux = nullptr; return; }
    if (!g_rt.aux_fork && !BQ_HIP(hipEventCreateWithFlags(&g_rt.aux_fork, hipEventDisableTiming))) return;
    if (!g_rt.aux_done && !BQ_HIP(hipEventCreateWithFlags(&g_rt.aux_done, hipEventDisableTiming))) return;
    if (g_rt.aux_pending) fl_aux_join();
    BQ_HIP(hipEventRecord(g_rt.aux_fork, g_rt.compute));
    BQ_HIP(hipStreamWaitEvent(g_rt.aux, g_rt.aux_fork, 0));
    g_rt.compute_main = g_rt.compute;
    g_rt.compute = g_rt.aux;
}

void fl_aux_end(void)
{
    if (!g_rt.compute_main) return;
    g_rt.compute = g_rt.compute_main;
    g_rt.compute_main = nullptr;
    g_rt.aux_pending = true;
}

void fl_aux_join(void)
{
    if (g_rt.compute_main) fl_aux_end();
    if (!g_rt.aux_pending) return;
    BQ_HIP(hipEventRecord(g_rt.aux_done, g_rt.aux));
    BQ_HIP(hipStreamWaitEvent(g_rt.compute, g_rt.aux_done, 0));
    g_rt.aux_pending = false;
}

// ---- asynchronous downloads (the dump path) ---------------------------------------
void *fl_malloc_host(size_t bytes)
{
    if (!bq::ensure_ready("fl_malloc_host")) return nullptr;
    void *p = nullptr;
    if (!BQ_HIP(hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault))) return nullptr;
    return p;
}

void fl_free_host(void *p)
{
    if (p) BQ_HIP(hipHostFree(p));
}

// Copy `bytes` device -> pinned host on the copy stream, ordered after everything queued on the compute
// stream so far; later compute work does not wait for it.  Returns a ticket for fl_download_wait().
void *fl_download_begin(void *host_dst, const void *dev_src, size_t bytes)
{
    if (!bq::ensure_ready("fl_download_begin")) return nullptr;
    if (!host_dst || !dev_src) { bq::latch(FL_ERR_BAD_ARGUMENT, "fl_download_begin", "null pointer"); return nullptr; }
    hipEvent_t ready = nullptr, done = nullptr;
    if (!BQ_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming)) || !BQ_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming))) return nullptr;
    bool ok = BQ_HIP(hipEventRecord(ready, g_rt.compute)) && BQ_HIP(hipStreamWaitEvent(g_rt.copy, ready, 0));
    ok = ok && BQ_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, g_rt.copy));
    ok = ok && BQ_HIP(hipEventRecord(done, g_rt.copy));
    (void)hipEventDestroy(ready);
    if (!ok) { (void)hipEventDestroy(done); return nullptr; }
    return (void *)done;
}

// Block the calling host thread (any thread) until the download has landed; releases the ticket.
int fl_download_wait(void *ticket)
{
    if (!ticket) return FL_ERR_BAD_ARGUMENT;
    const bool ok = BQ_HIP(hipEventSynchronize((hipEvent_t)ticket));
    (void)hipEventDestroy((hipEvent_t)ticket);
    return ok ? FL_OK : FL_ERR_HIP;
}

void *fl_event_create(void)
{
    if (!bq::ensure_ready("fl_event_create")) return nullptr;
    hipEvent_t ev = nullptr;
    if (!BQ_HIP(hipEventCreate(&ev))) return nullptr;
    return (void *)ev;
}

void fl_event_record(void *ev)
{
    if (!ev || !g_rt.ready) return;
    BQ_HIP(hipEventRecord((hipEvent_t)ev, g_rt.compute));
}

float fl_event_elapsed_ms(void *start, void *stop)
{
    if (!start || !stop) return -1.f;
    float ms = -1.f;
    if (!BQ_HIP(hipEventSynchronize((hipEvent_t)stop))) return -1.f;
    if (!BQ_HIP(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop))) return -1.f;
    return ms;
}

void fl_event_destroy(void *ev)
{
    if (ev) BQ_HIP(hipEventDestroy((hipEvent_t)ev));
}

int fl_last_error(void) { return g_rt.err; }
const char *fl_last_error_string(void) { return g_rt.err == FL_OK ? "" : g_rt.err_text; }
void fl_clear_error(void) { g_rt.err = FL_OK; g_rt.err_text[0] = 0; }
void fl_report_error(int code, const char *text) { bq::latch(code, "host", text ? text : ""); }
void *fl_compute_stream(void) { return bq::ensure_ready("fl_compute_stream") ? (void *)g_rt.compute : nullptr; }

void fl_set_option(int option, int value)
{
    switch (option) {
    case FL_OPT_RESIDUAL_STRIDE: g_rt.opt_residual_stride = value < 0 ? 0 : value; break;
    case FL_OPT_SKIP_UNIT_BLEND: g_rt.opt_skip_unit_blend = (value == 1 || value == 2) ? value : 0; break;
    case FL_OPT_JACOBI_VARIANT:  g_rt.opt_jacobi_variant = value; break;
    case FL_OPT_PROFILE_JACOBI:  g_rt.opt_profile_jacobi = value ? 1 : 0; break;
    case FL_OPT_JACOBI_KCHUNK:   g_rt.opt_jacobi_kchunk = value < 0 ? 0 : value; break;
    case FL_OPT_JACOBI_ROWS:     g_rt.opt_jacobi_rows = value; break;
    case FL_OPT_STRUCTURED_MAPS: g_rt.opt_structured_maps = value ? 1 : 0; break;
    case FL_OPT_JACOBI_FUSE:     g_rt.opt_jacobi_fuse = value; break;
    case FL_OPT_JACOBI_KCHUNK2:  g_rt.opt_jacobi_kchunk2 = value < 0 ? 0 : value; break;
    case FL_OPT_MGCG_GRAPH:      g_rt.opt_mgcg_graph = value != 0; break;
    case FL_OPT_FAST_LERP:       g_rt.opt_fast_lerp = value != 0; break;
    case FL_OPT_FUSED_HOUSEKEEPING: g_rt.opt_fused_housekeeping = value & 15; break;
    case FL_OPT_MAP_QUARTER_FP32: g_rt.opt_map_quarter_fp32 = value != 0; break;
    case FL_OPT_MGCG_TILE:       g_rt.opt_mgcg_tile = value < 0 ? 0 : value; break;
    case FL_OPT_PROFILE_COMM:    g_rt.opt_profile_comm = value != 0; break;
    case FL_OPT_MGCG_BOTTOM:     g_rt.opt_mgcg_bottom = value != 0; break;
    case FL_OPT_MGCG_FUSE:       g_rt.opt_mgcg_fuse = value < 0 ? -1 : (value > 3 ? 3 : value); break;
    case FL_OPT_FIELD_WINDOW:    g_rt.opt_field_window = value < 0 ? -1 : value; break;
    case FL_OPT_COMM_CHECK:      g_rt.opt_comm_check = value != 0; break;
    case FL_OPT_RESERVE_CUS: {
        const int k = value < 0 ? 0 : value;
        if (k == g_rt.opt_reserve_cus) break;
        g_rt.opt_reserve_cus = k;
        if (!g_rt.ready) break;                 // applied by fl_init
        // swap the compute stream for one with the new mask: everything queued has to finish first, and a cached
        // graph captured on the old stream is dropped
        fl_sync();
        bq::mgcg_release_graph();
        (void)hipStreamDestroy(g_rt.compute);
        g_rt.compute = nullptr;
        if (!bq::create_compute_stream()) g_rt.ready = false;
        break;
    }
    default: bq::latch(FL_ERR_BAD_ARGUMENT, "fl_set_option", "unknown option");
    }
}

int fl_get_option(int option)
{
    switch (option) {
    case FL_OPT_RESIDUAL_STRIDE: return g_rt.opt_residual_stride;
    case FL_OPT_SKIP_UNIT_BLEND: return g_rt.opt_skip_unit_blend;
    case FL_OPT_JACOBI_VARIANT:  return g_rt.opt_jacobi_variant;
    case FL_OPT_PROFILE_JACOBI:  return g_rt.opt_profile_jacobi;
    case FL_OPT_JACOBI_KCHUNK:   return g_rt.opt_jacobi_kchunk;
    case FL_OPT_JACOBI_ROWS:     return g_rt.opt_jacobi_rows;
    case FL_OPT_STRUCTURED_MAPS: return g_rt.opt_structured_maps;
    case FL_OPT_JACOBI_FUSE:     return g_rt.opt_jacobi_fuse;
    case FL_OPT_JACOBI_KCHUNK2:  return g_rt.opt_jacobi_kchunk2;
    case FL_OPT_MGCG_GRAPH:      return g_rt.opt_mgcg_graph;
    case FL_OPT_FAST_LERP:       return g_rt.opt_fast_lerp;
    case FL_OPT_FUSED_HOUSEKEEPING: return g_rt.opt_fused_housekeeping;
    case FL_OPT_MAP_QUARTER_FP32: return g_rt.opt_map_quarter_fp32;
    case FL_OPT_MGCG_TILE:       return g_rt.opt_mgcg_tile;
    case FL_OPT_PROFILE_COMM:    return g_rt.opt_profile_comm;
    case FL_OPT_MGCG_BOTTOM:     return g_rt.opt_mgcg_bottom;
    case FL_OPT_MGCG_FUSE:       return g_rt.opt_mgcg_fuse;
    case FL_OPT_FIELD_WINDOW:    return g_rt.opt_field_window;
    case FL_OPT_COMM_CHECK:      return g_rt.opt_comm_check;
    case FL_OPT_RESERVE_CUS:     return g_rt.opt_reserve_cus;
    default: return -1;
    }
}

} // extern "C"
