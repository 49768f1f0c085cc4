// bq_halo.hip -- z-slab context and ghost-plane exchange over RCCL.
//
// One process per GPU.  Rank r owns the global cell planes [own0, own1) of a grid with nkg planes and
// stores the planes [own0 - G, own1 + G) of every field (G ghost planes per side; ghost planes that
// fall outside the global grid stay zero, which is what a read outside the allocation returns on a
// single GPU).  A plane is one contiguous nx*ny block, so an exchange is four contiguous transfers:
// ncclSend/ncclRecv to/from rank-1 and rank+1 inside one ncclGroup, on the library's halo stream.
// xGMI is point-to-point: a slab neighbour exchange uses exactly one link per neighbour and there is
// no bulk collective anywhere on this path (the only all-reduces carry 1-2 scalars).
//
// RCCL is dlopen'ed on first use, so single-GPU runs never load it.
#include "bq_host.h"

#include <dlfcn.h>
#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <vector>

namespace bq {

// the slice of the RCCL API this file uses (rccl/rccl.h), resolved with dlsym
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat = 7, ncclDouble = 8 };       // ncclDataType_t
enum { ncclSum = 0, ncclMax = 2 };            // ncclRedOp_t

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;         // optional (reports only)
    int (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr;   // optional: the scalar all-reduces' own communicator
};

static Rccl g_rccl;                          // the dlopen'ed library: one per process
static void allreduce_span(hipEvent_t a, hipEvent_t b);
// Everything else is per CONTEXT (bq_host.h: Runtime::halo_state): communicator, rank, transport hooks, the event ring,
// traffic counters, FL_OPT_PROFILE_COMM spans.  The g_* names below are kept as accessors of the current context's state.
struct EvPair { hipEvent_t ready = nullptr, done = nullptr; };
static constexpr int kEvRing = 16;
struct WaitSpan { hipEvent_t a, b; };
// FL_OPT_COMM_CHECK: what a rank remembers of the communicator calls it has issued since the last check (fl_comm_check).
//   p2p    every ncclSend / ncclRecv is a token (src, dst, position in the src -> dst sequence, element count); a send and the
//          receive that will match it produce the SAME token, so over all ranks the sum of the send tokens' hashes equals the
//          sum of the receive tokens' hashes exactly when every message has a partner of the same size at the same position
//   coll   the all-reduces must come in the same order with the same shape on every rank: an order-dependent chain
struct CommLedger {
    unsigned long long p2p_sent = 0, p2p_received = 0;      // sums of 40-bit token hashes
    unsigned long long coll_chain = 0x9e3779b97f4a7c15ull;
    unsigned seq_to[64] = {0}, seq_from[64] = {0};          // per peer (ranks beyond 64 share counters: still symmetric)
    long long calls = 0;
};

struct HaloState {
    ncclComm_t comm = nullptr;
    // The in-stream scalar all-reduces (CFL maximum, norms, map guard, NaN flag) run on the COMPUTE stream while ghost-plane
    // exchanges are in flight on the halo stream.  RCCL serialises the operations of ONE communicator in host issue order
    // whatever stream they are on, so with a single communicator an all-reduce would make the compute stream wait for the
    // exchange it was meant to overlap -- and correctness would rest on every rank interleaving the two kinds identically.
    // They therefore get a communicator of their own (ncclCommSplit of the first, same ranks): nullptr = not available, the
    // one communicator serves both (RCCL without ncclCommSplit, BQ_SINGLE_COMM=1).
    ncclComm_t comm_red = nullptr;
    CommLedger ledger;
    int rank = 0, nranks = 1;
    // optional host-side transport (fl_comm_set_custom): used instead of RCCL when set
    fl_exchange_cb custom_exchange = nullptr;
    fl_allreduce_cb custom_allreduce = nullptr;
    fl_p2p_cb custom_p2p = nullptr;
    // Every exchange gets its own (ready, done) event pair from a small ring: an exchange that is still in flight when the
    // next one is issued keeps its events (the ring is far longer than the number of exchanges a step ever overlaps).
    EvPair ev[kEvRing];
    int ev_next = 0;
    hipEvent_t ev_pending = nullptr;        // `done` of the newest exchange nobody has waited for yet (fl_halo_wait)
    bool null_transport = false;            // fl_comm_set_null: exchanges and all-reduces are skipped (timing aid)
    // traffic counters (fl_comm_stats): exchanges issued, bytes this rank sent in them, wall-sheet message groups, bytes sent in them
    long long stat[4] = { 0, 0, 0, 0 };
    // FL_OPT_PROFILE_COMM: every place where the COMPUTE stream waits for the halo stream is bracketed by two timing events
    // on the compute stream.  The first completes when the compute stream has run dry up to the wait, the second when the
    // wait has been satisfied: their distance is communication time that no kernel hid ("exposed").  Summed by
    // fl_comm_profile().  A host-side transport blocks the host instead; that wall time is counted the same way.
    std::vector<WaitSpan> wait_spans, allreduce_spans;
    double wait_host_ms = 0.0;
    long long wait_host_count = 0;
};
static HaloState &hs()
{
    Runtime &r = rt();
    if (!r.halo_state) r.halo_state = new HaloState();
    return *static_cast<HaloState *>(r.halo_state);
}
#define g_comm (hs().comm)
#define g_rank (hs().rank)
#define g_nranks (hs().nranks)
#define g_custom_exchange (hs().custom_exchange)
#define g_custom_allreduce (hs().custom_allreduce)
#define g_custom_p2p (hs().custom_p2p)
#define g_ev (hs().ev)
#define g_ev_next (hs().ev_next)
#define g_ev_pending (hs().ev_pending)
#define g_null_transport (hs().null_transport)
#define g_stat (hs().stat)
#define g_wait_spans (hs().wait_spans)
#define g_allreduce_spans (hs().allreduce_spans)
#define g_wait_host_ms (hs().wait_host_ms)
#define g_wait_host_count (hs().wait_host_count)

void halo_release_state(Runtime &r)
{
    HaloState *h = static_cast<HaloState *>(r.halo_state);
    if (!h) return;
    if (h->comm_red && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm_red);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (EvPair &e : h->ev) { if (e.ready) (void)hipEventDestroy(e.ready); if (e.done) (void)hipEventDestroy(e.done); }
    for (std::vector<WaitSpan> *v : { &h->wait_spans, &h->allreduce_spans })
        for (const WaitSpan &sp : *v) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    delete h;
    r.halo_state = nullptr;
}

void halo_abandon_comm(Runtime &r)
{
    HaloState *h = static_cast<HaloState *>(r.halo_state);
    if (h) { h->comm = nullptr; h->comm_red = nullptr; }
}

// ---- FL_OPT_COMM_CHECK ledger ---------------------------------------------------------------------------------------
static inline unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
static constexpr unsigned long long kMask40 = (1ull << 40) - 1;
static void ledger_p2p(bool send, int peer, size_t count)
{
    if (!rt().opt_comm_check) return;
    HaloState &h = hs();
    CommLedger &L = h.ledger;
    const int src = send ? h.rank : peer, dst = send ? peer : h.rank;
    const unsigned seq = send ? ++L.seq_to[peer & 63] : ++L.seq_from[peer & 63];
    const unsigned long long tok = mix64(((unsigned long long)src << 52) ^ ((unsigned long long)dst << 44) ^ ((unsigned long long)seq << 20)) ^ mix64((unsigned long long)count + 0x51ull);
    (send ? L.p2p_sent : L.p2p_received) += mix64(tok) & kMask40;
    L.calls++;
}
static void ledger_coll(size_t count, bool is_double, bool is_max)
{
    if (!rt().opt_comm_check) return;
    CommLedger &L = hs().ledger;
    L.coll_chain = mix64(L.coll_chain ^ ((unsigned long long)count * 4 + (is_double ? 2 : 0) + (is_max ? 1 : 0)));
    L.calls++;
}

static void compute_waits_for(hipEvent_t done)
{
    Runtime &r = rt();
    if (!r.opt_profile_comm) { BQ_HIP(hipStreamWaitEvent(r.compute, done, 0)); return; }
    WaitSpan sp{nullptr, nullptr};
    if (!BQ_HIP(hipEventCreate(&sp.a)) || !BQ_HIP(hipEventCreate(&sp.b))) { BQ_HIP(hipStreamWaitEvent(r.compute, done, 0)); return; }
    BQ_HIP(hipEventRecord(sp.a, r.compute));
    BQ_HIP(hipStreamWaitEvent(r.compute, done, 0));
    BQ_HIP(hipEventRecord(sp.b, r.compute));
    g_wait_spans.push_back(sp);
}

static void allreduce_span(hipEvent_t a, hipEvent_t b) { g_allreduce_spans.push_back(WaitSpan{a, b}); }

struct HostWaitTimer {                      // wall time of a blocking host-side transport call
    std::chrono::steady_clock::time_point t0;
    bool on;
    HostWaitTimer() : on(rt().opt_profile_comm != 0) { if (on) t0 = std::chrono::steady_clock::now(); }
    ~HostWaitTimer()
    {
        if (!on) return;
        g_wait_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        g_wait_host_count++;
    }
};

static EvPair *next_events()
{
    EvPair &e = g_ev[g_ev_next];
    g_ev_next = (g_ev_next + 1) % kEvRing;
    if (!e.ready) {
        if (!BQ_HIP(hipEventCreateWithFlags(&e.ready, hipEventDisableTiming)) ||
            !BQ_HIP(hipEventCreateWithFlags(&e.done, hipEventDisableTiming))) return nullptr;
    }
    return &e;
}

static bool load_rccl()
{
    if (g_rccl.handle) return true;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    void *h = nullptr;
    // BQ_RCCL_LIBRARY: load this library instead (another RCCL build; the tests' multi-process stand-in for one-GPU boxes,
    // tests/fake_rccl).  No fallback to the default names when it is set.
    const char *forced = getenv("BQ_RCCL_LIBRARY");
    if (forced && *forced) h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    else for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { latch(FL_ERR_COMM, "dlopen(librccl)", dlerror()); return false; }
    g_rccl.handle = h;
#define BQ_SYM(field, name) *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) { latch(FL_ERR_COMM, "dlsym", name); return false; }
    BQ_SYM(GetUniqueId, "ncclGetUniqueId")
    BQ_SYM(CommInitRank, "ncclCommInitRank")
    BQ_SYM(CommDestroy, "ncclCommDestroy")
    BQ_SYM(Send, "ncclSend")
    BQ_SYM(Recv, "ncclRecv")
    BQ_SYM(AllReduce, "ncclAllReduce")
    BQ_SYM(GroupStart, "ncclGroupStart")
    BQ_SYM(GroupEnd, "ncclGroupEnd")
    BQ_SYM(GetErrorString, "ncclGetErrorString")
#undef BQ_SYM
    *(void **)(&g_rccl.GetVersion) = dlsym(h, "ncclGetVersion");
    *(void **)(&g_rccl.CommSplit) = dlsym(h, "ncclCommSplit");
    return true;
}

static bool nccl_ok(int rc, const char *what)
{
    if (rc == ncclSuccess) return true;
    latch(FL_ERR_COMM, what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error");
    return false;
}
#define BQ_NCCL(call) nccl_ok(g_rccl.call, #call)

// In-stream all-reduce of `count` values already on the device (used by gpu_max_abs3 and the
// residual norms when a communicator exists).  No-op on a single rank.
bool comm_allreduce(void *dev, size_t count, bool is_double, bool is_max, hipStream_t st)
{
    if (g_nranks <= 1 || g_null_transport) return true;
    if (g_custom_allreduce) {
        const size_t bytes = count * (is_double ? 8 : 4);
        void *host = pinned(bytes < 64 ? 64 : bytes);
        if (!host) return false;
        if (!BQ_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st)) || !BQ_HIP(hipStreamSynchronize(st))) return false;
        g_custom_allreduce(host, (int)count, is_double ? 1 : 0, is_max ? 1 : 0);
        return BQ_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st)) && BQ_HIP(hipStreamSynchronize(st));
    }
    if (!g_comm) return true;
    ledger_coll(count, is_double, is_max);
    // the all-reduce runs IN the stream it is given (the compute stream): its whole duration is exposed; timed separately
    hipEvent_t a = nullptr, b = nullptr;
    const bool timed = rt().opt_profile_comm && BQ_HIP(hipEventCreate(&a)) && BQ_HIP(hipEventCreate(&b)) && BQ_HIP(hipEventRecord(a, st));
    ncclComm_t comm = hs().comm_red ? hs().comm_red : g_comm;
    const bool ok = BQ_NCCL(AllReduce(dev, dev, count, is_double ? ncclDouble : ncclFloat, is_max ? ncclMax : ncclSum, comm, st));
    if (timed) { BQ_HIP(hipEventRecord(b, st)); allreduce_span(a, b); }
    return ok;
}
int comm_ranks() { return g_nranks; }

} // namespace bq

using namespace bq;

extern "C" {

void fl_set_slab(int koff, int nk_global, int own0, int own1, int nk_local)
{
    Runtime &r = rt();
    if (nk_global <= 0) { r.slab_on = false; return; }
    r.slab_on = true;
    r.slab_koff = koff; r.slab_nkg = nk_global; r.slab_own0 = own0; r.slab_own1 = own1; r.slab_nkl = nk_local;
}

// Restrict the map operators (gpu_solve_forward/_backwardDMC, gpu_advect_*, gpu_compensate_error_*, gpu_accumulate_*) to
// the local CELL planes [k0, k1): they produce exactly the nodes of those planes (a buffer with one plane more than
// cells -- the w component -- gets its extra plane with the window that reaches the last cell plane).  k0 < 0: off.
// Lets a z-slab host run an operator on the planes that need no ghost data while those are in flight.  Returns 1
// (supported).
int fl_set_plane_window(int k0, int k1)
{
    Runtime &r = rt();
    if (k0 < 0) { r.win_on = false; return 1; }
    r.win_on = true; r.win_k0 = k0; r.win_k1 = k1 < k0 ? k0 : k1;
    return 1;
}

int fl_comm_rank(void) { return g_rank; }
int fl_comm_size(void) { return g_nranks; }

int fl_comm_unique_id(void *id128)
{
    if (!id128) return FL_ERR_BAD_ARGUMENT;
    if (!ensure_ready("fl_comm_unique_id") || !load_rccl()) return fl_last_error();
    ncclUniqueId id;
    if (!BQ_NCCL(GetUniqueId(&id))) return FL_ERR_COMM;
    memcpy(id128, &id, sizeof id);
    return FL_OK;
}

int fl_comm_init(const void *id128, int rank, int nranks)
{
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) { latch(FL_ERR_BAD_ARGUMENT, "fl_comm_init", "bad rank/size"); return FL_ERR_BAD_ARGUMENT; }
    if (!ensure_ready("fl_comm_init")) return fl_last_error();
    g_rank = rank; g_nranks = nranks;
    if (nranks == 1) return FL_OK;
    if (!load_rccl()) return FL_ERR_COMM;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    if (!BQ_NCCL(CommInitRank(&g_comm, nranks, id, rank))) return FL_ERR_COMM;
    // the scalar all-reduces' own communicator (HaloState::comm_red): same ranks, split off the first -- a collective call,
    // every rank makes it here.  BQ_SINGLE_COMM=1 keeps the single communicator of rounds 1-3 (A/B on real links).
    const char *single = getenv("BQ_SINGLE_COMM");
    if (g_rccl.CommSplit && !(single && atoi(single) != 0)) {
        // (a failure here is not fatal: the first communicator serves both, as in rounds 1-3 -- the run must not lose its RCCL
        // transport over an optimisation)
        ncclComm_t red = nullptr;
        const int rc = g_rccl.CommSplit(g_comm, 0, rank, &red, nullptr);
        if (rc == ncclSuccess && red) hs().comm_red = red;
        else fprintf(stderr, "[bimocq] ncclCommSplit failed (%s): one communicator serves exchanges and all-reduces\n",
                     g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    }
    hs().ledger = CommLedger();
    return fl_last_error();
}

/* 2 when the scalar all-reduces have a communicator of their own, 1 with a single communicator, 0 without any */
int fl_comm_count(void) { return g_comm ? (hs().comm_red ? 2 : 1) : 0; }

// FL_OPT_COMM_CHECK: compare what the ranks have issued since the last call (CommLedger above).  Two small all-reduces on
// the compute stream + one read-back: a debugging aid, called by the host solver at the end of every step while the option
// is on.  Returns FL_OK, or FL_ERR_COMM (latched) when a send has no receive of the same size at the same position of its
// pair's sequence or the ranks' all-reduce sequences differ.  `perturb` != 0 (tests): this rank's ledger is falsified first.
int fl_comm_check(int perturb)
{
    if (g_nranks <= 1 || g_null_transport || (!g_comm && !g_custom_allreduce)) return FL_OK;
    Runtime &r = rt();
    CommLedger &L = hs().ledger;
    if (perturb) { L.p2p_sent += 12345; L.coll_chain ^= 0x77; }
    const unsigned long long diff = (L.p2p_sent - L.p2p_received) & kMask40;       // in [0, 2^40): 8 ranks' sum stays below 2^53
    const unsigned long long chain = L.coll_chain & ((1ull << 48) - 1);
    L = CommLedger();                               // (before the two all-reduces below, which every rank issues alike)
    double *dev = (double *)scratch(256);
    double *host = (double *)pinned(256);
    if (!dev || !host) return fl_last_error();
    hipStream_t st = r.compute;
    const int saved = r.opt_comm_check;
    r.opt_comm_check = 0;                           // the check's own collectives are not part of the ledger
    host[0] = (double)diff;
    host[1] = (double)chain; host[2] = -(double)chain;
    bool ok = BQ_HIP(hipMemcpyAsync(dev, host, 24, hipMemcpyHostToDevice, st)) && BQ_HIP(hipStreamSynchronize(st));
    ok = ok && comm_allreduce(dev, 1, true, false, st) && comm_allreduce(dev + 1, 2, true, true, st);
    ok = ok && BQ_HIP(hipMemcpyAsync(host, dev, 24, hipMemcpyDeviceToHost, st)) && BQ_HIP(hipStreamSynchronize(st));
    r.opt_comm_check = saved;
    if (!ok) return fl_last_error();
    const unsigned long long total = (unsigned long long)host[0];
    if ((total & kMask40) != 0) { latch(FL_ERR_COMM, "fl_comm_check", "a send has no receive of the same size at the same place of its pair's sequence (or the reverse)"); return FL_ERR_COMM; }
    if (host[1] != -host[2]) { latch(FL_ERR_COMM, "fl_comm_check", "the ranks issued different sequences of all-reduces"); return FL_ERR_COMM; }
    return FL_OK;
}

void fl_comm_destroy(void)
{
    if (g_comm) {
        fl_sync();
        if (hs().comm_red) { g_rccl.CommDestroy(hs().comm_red); hs().comm_red = nullptr; }
        g_rccl.CommDestroy(g_comm); g_comm = nullptr;
    }
    g_custom_exchange = nullptr; g_custom_allreduce = nullptr; g_custom_p2p = nullptr;
    g_null_transport = false;
    g_ev_pending = nullptr;
    g_rank = 0; g_nranks = 1;
}

// Timing aid: this process plays rank `rank` of `nranks` with a transport that moves nothing and never waits, so
// that the compute-side cost of the z-slab path can be measured on one GPU (bench.py --emulate-slab).  The ghost
// planes keep whatever they held: results near the slab boundary are meaningless.
void fl_comm_set_null(int rank, int nranks)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) { latch(FL_ERR_BAD_ARGUMENT, "fl_comm_set_null", "bad rank/size"); return; }
    g_rank = rank; g_nranks = nranks; g_null_transport = nranks > 1;
    g_custom_exchange = nullptr; g_custom_allreduce = nullptr;
}

void fl_comm_set_custom(int rank, int nranks, fl_exchange_cb exchange, fl_allreduce_cb allreduce)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && (!exchange || !allreduce))) {
        latch(FL_ERR_BAD_ARGUMENT, "fl_comm_set_custom", "bad rank/size/callbacks"); return;
    }
    g_rank = rank; g_nranks = nranks;
    g_custom_exchange = exchange; g_custom_allreduce = allreduce;
}

// Refresh `depth` ghost planes on both sides of n fields in one RCCL group.
//   fields[f]      local buffer of (nk_local + extra[f]) planes of plane_elems[f] floats
//   extra[f]       1 for the w component (nk+1 planes; its high ghost takes depth+1 planes), else 0
//   nk_local       local CELL planes = owned + 2*G
// Ordering: the halo stream first waits for everything queued on the compute stream; with
// wait != 0 the compute stream then waits for the exchange (otherwise call fl_halo_wait() before
// the ghost planes are read or the sent planes are overwritten).
void fl_halo_exchange(int n, float *const *fields, const size_t *plane_elems, const int *extra,
                      int nk_local, int G, int depth, int wait)
{
    if (g_nranks <= 1) return;
    Runtime &r = rt();
    const int own = nk_local - 2 * G;
    if (n <= 0 || !fields || !plane_elems || !extra || depth < 1 || depth > G || own < depth + 1) {
        latch(FL_ERR_BAD_ARGUMENT, "fl_halo_exchange", "bad depth/ghost/owned plane counts"); return;
    }
    {
        const int lo_ = g_rank - 1, hi_ = g_rank + 1;
        long long bytes = 0;
        for (int f = 0; f < n; f++) bytes += (long long)plane_elems[f] * 4 * ((lo_ >= 0 ? depth + extra[f] : 0) + (hi_ < g_nranks ? depth : 0));
        g_stat[0] += 1; g_stat[1] += bytes;
    }
    if (g_null_transport) return;               // timing aid: nothing moves, nothing waits
    if (g_custom_exchange) {                    // host-side transport: everything queued so far must be done
        BQ_HIP(hipStreamSynchronize(r.compute));
        HostWaitTimer timer;
        g_custom_exchange(n, fields, plane_elems, extra, nk_local, G, depth);
        (void)wait;
        return;
    }
    if (!g_comm) { latch(FL_ERR_COMM, "fl_halo_exchange", "no communicator (fl_comm_init)"); return; }
    EvPair *ev = next_events();
    if (!ev) return;
    BQ_HIP(hipEventRecord(ev->ready, r.compute));
    BQ_HIP(hipStreamWaitEvent(r.halo, ev->ready, 0));
    const int lo = g_rank - 1, hi = g_rank + 1;
    if (!BQ_NCCL(GroupStart())) return;
    for (int f = 0; f < n; f++) {
        float *b = fields[f];
        const size_t pe = plane_elems[f];
        const int ex = extra[f];
        if (lo >= 0) {
            // my bottom owned planes [G, G+depth+ex) become rank-1's high ghost; its top owned planes fill my low ghost
            ledger_p2p(true, lo, pe * (size_t)(depth + ex)); ledger_p2p(false, lo, pe * (size_t)depth);
            BQ_NCCL(Send(b + pe * (size_t)G, pe * (size_t)(depth + ex), ncclFloat, lo, g_comm, r.halo));
            BQ_NCCL(Recv(b + pe * (size_t)(G - depth), pe * (size_t)depth, ncclFloat, lo, g_comm, r.halo));
        }
        if (hi < g_nranks) {
            ledger_p2p(true, hi, pe * (size_t)depth); ledger_p2p(false, hi, pe * (size_t)(depth + ex));
            BQ_NCCL(Send(b + pe * (size_t)(G + own - depth), pe * (size_t)depth, ncclFloat, hi, g_comm, r.halo));
            BQ_NCCL(Recv(b + pe * (size_t)(G + own), pe * (size_t)(depth + ex), ncclFloat, hi, g_comm, r.halo));
        }
    }
    if (!BQ_NCCL(GroupEnd())) return;
    BQ_HIP(hipEventRecord(ev->done, r.halo));
    if (wait) compute_waits_for(ev->done);
    else g_ev_pending = ev->done;
}

} // extern "C"

// ---- wall sheets (include/bimocq_gpu.h, section 4): box gather / scatter and point-to-point messages -------------
namespace {
constexpr int kBoxChunk = 64;
struct BoxChunk {
    int n;
    int x0[kBoxChunk], y0[kBoxChunk], z0[kBoxChunk], wx[kBoxChunk], wy[kBoxChunk];
    long long off[kBoxChunk + 1];       // element offset of each box inside this chunk's packed range
};
}

// one thread per packed element; MODE 0: packed <- field, 1: field <- packed, 2: field <- NaN,
// 3: packed[same global cell] <- field, `packed` being a second field that holds the planes [koff2, ...)
template <int MODE>
__global__ __launch_bounds__(256) void box_copy_kernel(float *__restrict__ field, float *__restrict__ packed, BoxChunk c,
                                                       int nbi, int nbj, int koff, int koff2 = 0)
{
    const long long total = c.off[c.n];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        int b = 0;
        while (b + 1 < c.n && t >= c.off[b + 1]) b++;
        const long long e = t - c.off[b];
        const int x = (int)(e % c.wx[b]);
        const long long r = e / c.wx[b];
        const int y = (int)(r % c.wy[b]), z = (int)(r / c.wy[b]);
        const size_t id = (size_t)(c.x0[b] + x) + (size_t)nbi * ((size_t)(c.y0[b] + y) + (size_t)nbj * (size_t)(c.z0[b] + z - koff));
        if (MODE == 0) packed[t] = field[id];
        else if (MODE == 1) field[id] = packed[t];
        else if (MODE == 2) field[id] = __builtin_nanf("");
        else packed[(size_t)(c.x0[b] + x) + (size_t)nbi * ((size_t)(c.y0[b] + y) + (size_t)nbj * (size_t)(c.z0[b] + z - koff2))] = field[id];
    }
}

template <int MODE>
static void box_copy(float *field, int nbi, int nbj, int nk_field, int koff, const fl_box *boxes, int nboxes, float *packed, const char *op,
                     int nk2 = 0, int koff2 = 0)
{
    if (!ensure_ready(op)) return;
    if (nboxes <= 0) return;
    if (!field || !boxes || (MODE != 2 && !packed) || nbi < 1 || nbj < 1 || nk_field < 1) { latch(FL_ERR_BAD_ARGUMENT, op, "null pointer or bad dims"); return; }
    long long base = 0;
    for (int first = 0; first < nboxes; first += kBoxChunk) {
        BoxChunk c;
        c.n = 0; c.off[0] = 0;
        for (int b = first; b < nboxes && c.n < kBoxChunk; b++) {
            const fl_box &q = boxes[b];
            if (q.x0 < 0 || q.y0 < 0 || q.z0 < koff || q.x1 > nbi || q.y1 > nbj || q.z1 > koff + nk_field || q.x1 < q.x0 || q.y1 < q.y0 || q.z1 < q.z0) {
                latch(FL_ERR_BAD_ARGUMENT, op, "box outside the field"); return;
            }
            if (MODE == 3 && (q.z0 < koff2 || q.z1 > koff2 + nk2)) { latch(FL_ERR_BAD_ARGUMENT, op, "box outside the destination"); return; }
            const long long vol = (long long)(q.x1 - q.x0) * (q.y1 - q.y0) * (q.z1 - q.z0);
            if (vol == 0) continue;
            c.x0[c.n] = q.x0; c.y0[c.n] = q.y0; c.z0[c.n] = q.z0; c.wx[c.n] = q.x1 - q.x0; c.wy[c.n] = q.y1 - q.y0;
            c.off[c.n + 1] = c.off[c.n] + vol;
            c.n++;
        }
        const long long total = c.off[c.n];
        if (total == 0) continue;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
        box_copy_kernel<MODE><<<blocks, 256, 0, rt().compute>>>(field, MODE == 2 ? nullptr : (MODE == 3 ? packed : packed + base), c, nbi, nbj, koff, koff2);
        if (!BQ_LAUNCH_CHECK("box_copy_kernel")) return;
        base += total;
    }
}

extern "C" {

void fl_box_pack(const float *field, int nbi, int nbj, int nk_field, int koff, const fl_box *boxes, int nboxes, float *packed)
{
    box_copy<0>(const_cast<float *>(field), nbi, nbj, nk_field, koff, boxes, nboxes, packed, "fl_box_pack");
}

void fl_box_unpack(float *field, int nbi, int nbj, int nk_field, int koff, const fl_box *boxes, int nboxes, const float *packed)
{
    if (packed) box_copy<1>(field, nbi, nbj, nk_field, koff, boxes, nboxes, const_cast<float *>(packed), "fl_box_unpack");
    else        box_copy<2>(field, nbi, nbj, nk_field, koff, boxes, nboxes, nullptr, "fl_box_unpack");
}

void fl_box_copy(const float *src, int nbi, int nbj, int nk_src, int koff_src, float *dst, int nk_dst, int koff_dst,
                 const fl_box *boxes, int nboxes)
{
    box_copy<3>(const_cast<float *>(src), nbi, nbj, nk_src, koff_src, boxes, nboxes, dst, "fl_box_copy", nk_dst, koff_dst);
}

void fl_comm_set_custom_p2p(fl_p2p_cb p2p) { g_custom_p2p = p2p; }

static void p2p_exchange(int n, const int *peers, float *const *send, const size_t *send_count,
                         float *const *recv, const size_t *recv_count, bool wait)
{
    if (g_nranks <= 1 || n <= 0) return;
    Runtime &r = rt();
    if (!peers || !send || !send_count || !recv || !recv_count) { latch(FL_ERR_BAD_ARGUMENT, "fl_p2p_exchange", "null argument"); return; }
    for (int m = 0; m < n; m++)
        if (peers[m] < 0 || peers[m] >= g_nranks || peers[m] == g_rank || (send_count[m] && !send[m]) || (recv_count[m] && !recv[m])) {
            latch(FL_ERR_BAD_ARGUMENT, "fl_p2p_exchange", "bad peer or null buffer"); return;
        }
    {
        long long bytes = 0;
        for (int m = 0; m < n; m++) bytes += (long long)send_count[m] * 4;
        g_stat[2] += 1; g_stat[3] += bytes;
    }
    if (g_null_transport) return;
    if (g_custom_exchange || g_custom_p2p) {
        if (!g_custom_p2p) { latch(FL_ERR_COMM, "fl_p2p_exchange", "the custom transport has no point-to-point callback (fl_comm_set_custom_p2p)"); return; }
        BQ_HIP(hipStreamSynchronize(r.compute));
        HostWaitTimer timer;
        g_custom_p2p(n, peers, send, send_count, recv, recv_count);
        return;
    }
    if (!g_comm) { latch(FL_ERR_COMM, "fl_p2p_exchange", "no communicator (fl_comm_init)"); return; }
    EvPair *ev = next_events();
    if (!ev) return;
    BQ_HIP(hipEventRecord(ev->ready, r.compute));
    BQ_HIP(hipStreamWaitEvent(r.halo, ev->ready, 0));
    if (!BQ_NCCL(GroupStart())) return;
    for (int m = 0; m < n; m++) {
        if (send_count[m]) { ledger_p2p(true, peers[m], send_count[m]); BQ_NCCL(Send(send[m], send_count[m], ncclFloat, peers[m], g_comm, r.halo)); }
        if (recv_count[m]) { ledger_p2p(false, peers[m], recv_count[m]); BQ_NCCL(Recv(recv[m], recv_count[m], ncclFloat, peers[m], g_comm, r.halo)); }
    }
    if (!BQ_NCCL(GroupEnd())) return;
    BQ_HIP(hipEventRecord(ev->done, r.halo));
    if (wait) compute_waits_for(ev->done);
    else g_ev_pending = ev->done;
}

void fl_p2p_exchange(int n, const int *peers, float *const *send, const size_t *send_count,
                     float *const *recv, const size_t *recv_count)
{
    p2p_exchange(n, peers, send, send_count, recv, recv_count, true);
}

void fl_p2p_exchange_begin(int n, const int *peers, float *const *send, const size_t *send_count,
                           float *const *recv, const size_t *recv_count)
{
    p2p_exchange(n, peers, send, send_count, recv, recv_count, false);
}

// Exercises every RCCL entry point this file binds on a throw-away ONE-rank communicator: unique id,
// init, in-place all-reduces with the data types/operators comm_allreduce uses, a grouped
// ncclSend/ncclRecv pair (to self) on the halo stream, destroy.  Lets a single-GPU box verify the
// dlopen'ed binding (symbols, enum values, call signatures) that the multi-GPU path depends on.
int fl_comm_selftest(void)
{
    if (!ensure_ready("fl_comm_selftest") || !load_rccl()) return fl_last_error();
    Runtime &r = rt();
    ncclUniqueId id;
    ncclComm_t comm = nullptr;
    if (!BQ_NCCL(GetUniqueId(&id)) || !BQ_NCCL(CommInitRank(&comm, 1, id, 0))) return FL_ERR_COMM;
    const int n = 1024;
    float *buf = (float *)fl_malloc(sizeof(float) * 2 * n + sizeof(double) * 2);
    int rc = FL_OK;
    if (buf) {
        float host[2 * n];
        for (int i = 0; i < n; i++) { host[i] = 0.25f * i - 3.f; host[n + i] = -1.f; }
        double hd[2] = { 1.0 / 3.0, -7.5 };
        double *dd = (double *)(buf + 2 * n);
        // on the compute stream, behind fl_malloc's asynchronous clear of the buffer (a copy on the null stream is not ordered
        // against it: the clear could land on top of the test pattern), and complete before the halo stream touches the data
        bool ok = BQ_HIP(hipMemcpyAsync(buf, host, sizeof host, hipMemcpyHostToDevice, r.compute)) &&
                  BQ_HIP(hipMemcpyAsync(dd, hd, sizeof hd, hipMemcpyHostToDevice, r.compute)) && BQ_HIP(hipStreamSynchronize(r.compute));
        ok = ok && nccl_ok(g_rccl.AllReduce(dd, dd, 2, ncclDouble, ncclSum, comm, r.halo), "selftest AllReduce(double,sum)");
        ok = ok && nccl_ok(g_rccl.AllReduce(buf, buf, 8, ncclFloat, ncclMax, comm, r.halo), "selftest AllReduce(float,max)");
        ok = ok && BQ_NCCL(GroupStart());
        ok = ok && nccl_ok(g_rccl.Send(buf, n, ncclFloat, 0, comm, r.halo), "selftest Send");
        ok = ok && nccl_ok(g_rccl.Recv(buf + n, n, ncclFloat, 0, comm, r.halo), "selftest Recv");
        ok = ok && BQ_NCCL(GroupEnd());
        ok = ok && BQ_HIP(hipStreamSynchronize(r.halo));
        float back[2 * n]; double bd[2];
        // (on the halo stream too, then one sync: no null-stream call anywhere -- a CU-masked stream is a BLOCKING stream with
        // respect to the legacy null stream, so a null-stream copy next to a dump or an exchange would serialise with it)
        ok = ok && BQ_HIP(hipMemcpyAsync(back, buf, sizeof back, hipMemcpyDeviceToHost, r.halo)) &&
             BQ_HIP(hipMemcpyAsync(bd, dd, sizeof bd, hipMemcpyDeviceToHost, r.halo)) && BQ_HIP(hipStreamSynchronize(r.halo));
        if (ok) {
            for (int i = 0; i < n && ok; i++) ok = back[i] == host[i] && back[n + i] == host[i];
            ok = ok && bd[0] == hd[0] && bd[1] == hd[1];
            if (!ok) latch(FL_ERR_COMM, "fl_comm_selftest", "data mismatch after all-reduce / send-recv");
        }
        if (!ok) rc = FL_ERR_COMM;
        fl_free(buf);
    } else rc = fl_last_error();
    g_rccl.CommDestroy(comm);
    return rc;
}

void fl_comm_stats(long long out[4], int reset)
{
    if (out) for (int a = 0; a < 4; a++) out[a] = g_stat[a];
    if (reset) for (int a = 0; a < 4; a++) g_stat[a] = 0;
}

void fl_halo_wait(void)
{
    if (g_nranks <= 1 || !g_ev_pending) return;
    compute_waits_for(g_ev_pending);
    g_ev_pending = nullptr;
}

// FL_OPT_PROFILE_COMM.  ms[0] / n[0]: milliseconds the compute stream spent blocked on the halo stream (or the host inside a
// host-side transport) and the number of such waits; ms[1] / n[1]: the in-stream scalar all-reduces (CFL maximum, map
// guard, norms), which run ON the compute stream.  Since the last reset; blocking (synchronises the compute stream).
void fl_comm_profile(double ms[2], long long n[2], int reset)
{
    double m[2] = { g_wait_host_ms, 0.0 };
    long long c[2] = { g_wait_host_count, 0 };
    if (rt().ready && (!g_wait_spans.empty() || !g_allreduce_spans.empty())) {
        BQ_HIP(hipStreamSynchronize(rt().compute));
        const std::vector<WaitSpan> *sets[2] = { &g_wait_spans, &g_allreduce_spans };
        for (int a = 0; a < 2; a++)
            for (const WaitSpan &sp : *sets[a]) {
                float t = 0.f;
                if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess && t > 0.f) m[a] += (double)t;
                c[a]++;
            }
    }
    if (ms) { ms[0] = m[0]; ms[1] = m[1]; }
    if (n) { n[0] = c[0]; n[1] = c[1]; }
    if (reset) {
        for (std::vector<WaitSpan> *v : { &g_wait_spans, &g_allreduce_spans }) {
            for (const WaitSpan &sp : *v) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
            v->clear();
        }
        g_wait_host_ms = 0.0; g_wait_host_count = 0;
    }
}

// ncclGetVersion of the loaded RCCL (e.g. 22105), 0 when none is loaded or the library does not export it
int fl_comm_rccl_version(void)
{
    int v = 0;
    if (g_rccl.handle && g_rccl.GetVersion && g_rccl.GetVersion(&v) == ncclSuccess) return v;
    return 0;
}

} // extern "C"
