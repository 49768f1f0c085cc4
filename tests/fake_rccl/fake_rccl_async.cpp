// fake_rccl_async.cpp -- TEST INFRASTRUCTURE: the stream-ordered variant of the librccl stand-in (see fake_rccl.cpp).
//
// fake_rccl.cpp synchronises the stream it is handed and moves data with blocking copies, so it cannot see a missing
// stream dependency on the caller's side.  This one never blocks the host: every ncclSend / ncclRecv becomes, on the
// stream the caller named,
//     send:  wait (stream memory op) until the peer has emptied mailbox slot s % K   -> async copy of the buffer into the
//            peer's device mailbox (opened through hipIpc*: the ranks are processes sharing ONE GPU) -> write "full = s"
//     recv:  wait until "full >= s"  -> async copy mailbox -> buffer  -> write "empty = s"
// i.e. data leaves a send buffer and lands in a receive buffer exactly when the stream reaches the operation, as with
// RCCL.  A kernel that overwrites planes still being sent, or reads ghost planes before the receive has run, now gets
// what it would get on real hardware.  Flags live in a POSIX shared-memory segment registered with hipHostRegister.
// Matching is by issue order per ordered pair, message sizes are checked on the host at enqueue time (both sides keep a
// size log in the segment).  ncclAllReduce: the contributions go to the segment (async D2H), a host function enqueued
// on the stream reduces them once every rank has arrived, the result is copied back (async H2D).
//
// BQ_FAKE_RCCL_DELAY_MB = n: every transfer is preceded by an n-MB fill on the same stream, so that it starts long after
// the call returned (the tests run with and without).
//
// Limits (checked): a message must fit one mailbox slot (BQ_FAKE_RCCL_SLOT_MB, default 16); operations of a group run in
// issue order on the stream, so a pattern that needs RCCL's concurrent progress to avoid a cycle longer than the K = 8
// slots in flight per pair would stall here -- the library's exchanges (send, recv per field and peer, in the same order
// on both sides) do not.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 8;
constexpr int kSlots = 8;                            // mailbox slots per ordered pair
constexpr int kLog = 4096;                           // size log entries per ordered pair (ring)
constexpr size_t kRedMax = 4096;
constexpr double kTimeout = 120.0;

struct Pair {
    unsigned full;                                   // sequence number of the newest message the sender has delivered
    unsigned empty;                                  // sequence number of the newest message the receiver has consumed
    std::atomic<unsigned> posted;                    // host side: messages whose size the sender has logged
    unsigned long long bytes[kLog];                  // size of message s at [s % kLog]
    char pad[64];
};

struct Segment {
    std::atomic<int> arrived, ready, left;
    hipIpcMemHandle_t box[kMaxRanks];                // each rank's mailbox allocation: nranks x kSlots slots, [src][slot]
    Pair pair[kMaxRanks * kMaxRanks];                // [src * kMaxRanks + dst]
    unsigned red_arrive[kMaxRanks], red_done[kMaxRanks];
    unsigned char red[kMaxRanks][kRedMax];
    unsigned char red_out[kMaxRanks][kRedMax];
};

struct Comm {
    Segment *seg = nullptr, *dseg = nullptr;         // host mapping, device-visible alias of the same memory
    int rank = 0, nranks = 1;
    size_t slot_bytes = 0;
    char *mybox = nullptr;                           // device: my mailboxes
    char *delay_buf = nullptr; size_t delay_bytes = 0;  // BQ_FAKE_RCCL_DELAY_MB: a fill this large precedes every transfer
    char *peerbox[kMaxRanks] = {};                   // device: the peers' mailboxes (IPC)
    unsigned send_seq[kMaxRanks] = {}, recv_seq[kMaxRanks] = {};
    unsigned red_seq = 0;
    std::string name;
    int splits = 0;                                  // ncclCommSplit calls made on this communicator (names the children)
};

thread_local int g_group_depth = 0;
const char *g_last_error = "no error";
enum { kSuccess = 0, kUnhandledCuda = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int fail(int code, const char *what) { g_last_error = what; fprintf(stderr, "[fake_rccl_async] %s\n", what); return code; }
#define HIPCK(x, msg) do { if ((x) != hipSuccess) return fail(kUnhandledCuda, msg); } while (0)

size_t type_size(int dtype)
{
    switch (dtype) { case 0: case 1: return 1; case 2: case 3: return 4; case 4: case 5: return 8; case 6: return 2; case 7: return 4; case 8: return 8; default: return 0; }
}

template <typename T> T *dev(Comm *c, T *host_field) { return (T *)((char *)c->dseg + ((char *)host_field - (char *)c->seg)); }

struct RedJob { Comm *c; size_t count; int dtype, op; };

template <typename T>
void reduce(Segment *s, int nranks, int me, size_t count, int op)
{
    T *out = (T *)s->red_out[me];
    for (size_t i = 0; i < count; i++) {
        T acc = ((const T *)s->red[0])[i];
        for (int r = 1; r < nranks; r++) {
            const T v = ((const T *)s->red[r])[i];
            switch (op) { case 0: acc = acc + v; break; case 1: acc = acc * v; break; case 2: acc = v > acc ? v : acc; break; case 3: acc = v < acc ? v : acc; break; default: break; }
        }
        out[i] = acc;
    }
}

void reduce_host_fn(void *p)
{
    RedJob *j = (RedJob *)p;
    if (j->dtype == 7) reduce<float>(j->c->seg, j->c->nranks, j->c->rank, j->count, j->op);
    else reduce<double>(j->c->seg, j->c->nranks, j->c->rank, j->count, j->op);
    delete j;
}

int transfer(bool send, void *buf, size_t count, int dtype, int peer, Comm *c, hipStream_t st)
{
    if (!c || !c->seg) return fail(kInvalidArgument, "null communicator");
    const size_t ts = type_size(dtype);
    if (!ts) return fail(kInvalidArgument, "unknown data type");
    if (peer < 0 || peer >= c->nranks) return fail(kInvalidArgument, "peer out of range");
    const size_t bytes = count * ts;
    if (bytes > c->slot_bytes) return fail(kInvalidArgument, "message larger than a mailbox slot (BQ_FAKE_RCCL_SLOT_MB)");
    if (bytes && !buf) return fail(kInvalidArgument, "null buffer");
    const int src = send ? c->rank : peer, dst = send ? peer : c->rank;
    Pair *hp = &c->seg->pair[src * kMaxRanks + dst], *dp = dev(c, hp);
    // widen the window in which a missing dependency on the caller's side would show: the transfer itself starts late
    if (c->delay_bytes) HIPCK(hipMemsetAsync(c->delay_buf, 0, c->delay_bytes, st), "hipMemsetAsync (delay) failed");
    if (send) {
        const unsigned s = ++c->send_seq[peer];
        hp->bytes[s % kLog] = bytes;                                  // size log first, then the count of posted messages
        hp->posted.store(s);
        if (s > (unsigned)kSlots) HIPCK(hipStreamWaitValue32(st, &dp->empty, s - kSlots, hipStreamWaitValueGte, 0xffffffffu), "hipStreamWaitValue32 failed");
        char *slot = c->peerbox[peer] + ((size_t)c->rank * kSlots + s % kSlots) * c->slot_bytes;
        if (bytes) HIPCK(hipMemcpyAsync(slot, buf, bytes, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync into the peer's mailbox failed");
        HIPCK(hipStreamWriteValue32(st, &dp->full, s, 0), "hipStreamWriteValue32 failed");
    } else {
        const unsigned s = ++c->recv_seq[peer];
        // the size check needs the sender's log entry: wait (host side, bounded) until it has been posted
        const double t0 = now();
        while (hp->posted.load() < s) {
            if (now() - t0 > kTimeout) return fail(kInternalError, "timeout: a receive found no matching send (mismatched exchange)");
            std::this_thread::yield();
        }
        if (hp->bytes[s % kLog] != bytes) {
            fprintf(stderr, "[fake_rccl_async] rank %d: ncclRecv #%u from %d expects %zu bytes, the matching ncclSend has %llu\n", c->rank, s, peer, bytes, hp->bytes[s % kLog]);
            return fail(kInvalidArgument, "send / receive sizes do not match");
        }
        HIPCK(hipStreamWaitValue32(st, &dp->full, s, hipStreamWaitValueGte, 0xffffffffu), "hipStreamWaitValue32 failed");
        char *slot = c->mybox + ((size_t)peer * kSlots + s % kSlots) * c->slot_bytes;
        if (bytes) HIPCK(hipMemcpyAsync(buf, slot, bytes, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync out of the mailbox failed");
        HIPCK(hipStreamWriteValue32(st, &dp->empty, s, 0), "hipStreamWriteValue32 failed");
    }
    return kSuccess;
}

} // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return fail(kInvalidArgument, "null id");
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/bq_fake_rccl_a_%d_%llx", (int)getpid(),
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return kSuccess;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return fail(kInvalidArgument, "bad rank / size");
    id.internal[sizeof id.internal - 1] = 0;
    if (id.internal[0] != '/') return fail(kInvalidArgument, "not an id from this library's ncclGetUniqueId");
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return fail(kSystemError, "shm_open failed");
    if (ftruncate(fd, (off_t)sizeof(Segment)) != 0) { close(fd); return fail(kSystemError, "ftruncate failed"); }
    void *p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(kSystemError, "mmap failed");
    Comm *c = new Comm;
    c->seg = (Segment *)p; c->rank = rank; c->nranks = nranks; c->name = id.internal;
    const char *mb = getenv("BQ_FAKE_RCCL_SLOT_MB");
    c->slot_bytes = (size_t)(mb && atoi(mb) > 0 ? atoi(mb) : 16) << 20;
    HIPCK(hipHostRegister(c->seg, sizeof(Segment), hipHostRegisterMapped), "hipHostRegister of the shared segment failed");
    HIPCK(hipHostGetDevicePointer((void **)&c->dseg, c->seg, 0), "hipHostGetDevicePointer failed");
    HIPCK(hipMalloc((void **)&c->mybox, (size_t)nranks * kSlots * c->slot_bytes), "hipMalloc of the mailboxes failed");
    const char *dl = getenv("BQ_FAKE_RCCL_DELAY_MB");
    if (dl && atoi(dl) > 0) {
        c->delay_bytes = (size_t)atoi(dl) << 20;
        HIPCK(hipMalloc((void **)&c->delay_buf, c->delay_bytes), "hipMalloc of the delay buffer failed");
    }
    if (nranks > 1) HIPCK(hipIpcGetMemHandle(&c->seg->box[rank], c->mybox), "hipIpcGetMemHandle failed");
    c->seg->arrived.fetch_add(1);
    const double t0 = now();
    while (c->seg->arrived.load() < nranks) {
        if (now() - t0 > kTimeout) return fail(kInternalError, "timeout: not every rank reached ncclCommInitRank");
        std::this_thread::yield();
    }
    for (int r = 0; r < nranks; r++) {
        if (r == rank) { c->peerbox[r] = c->mybox; continue; }
        HIPCK(hipIpcOpenMemHandle((void **)&c->peerbox[r], c->seg->box[r], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle failed");
    }
    c->seg->ready.fetch_add(1);
    while (c->seg->ready.load() < nranks) {
        if (now() - t0 > kTimeout) return fail(kInternalError, "timeout while the ranks opened each other's mailboxes");
        std::this_thread::yield();
    }
    *comm = c;
    return kSuccess;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return kSuccess;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->nranks; r++)
        if (r != c->rank && c->peerbox[r]) (void)hipIpcCloseMemHandle(c->peerbox[r]);
    // nobody may free a mailbox another rank still has open
    c->seg->left.fetch_add(1);
    const double t0 = now();
    while (c->seg->left.load() < c->nranks && now() - t0 < 30.0) std::this_thread::yield();
    (void)hipFree(c->mybox);
    if (c->delay_buf) (void)hipFree(c->delay_buf);
    (void)hipHostUnregister(c->seg);
    const bool unlink_it = c->rank == 0;
    munmap(c->seg, sizeof(Segment));
    if (unlink_it) shm_unlink(c->name.c_str());
    delete c;
    return kSuccess;
}

// ncclCommSplit with one colour for everybody (what bq_halo.hip asks for: a second communicator over the same ranks, for the
// in-stream scalar all-reduces): a child segment named after the parent's and the number of splits made on it so far --
// every rank splits in the same order, so the names agree.  `key` must be the caller's rank (ranks keep their numbers).
int ncclCommSplit(void *comm, int color, int key, void **newcomm, void *config)
{
    (void)config;
    Comm *c = (Comm *)comm;
    if (!c || !newcomm) return fail(kInvalidArgument, "null communicator");
    if (color != 0 || key != c->rank) return fail(kInvalidArgument, "the stand-in splits with colour 0 and key = rank only");
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    snprintf(id.internal, sizeof id.internal, "%s_s%d", c->name.c_str(), c->splits++);
    return ncclCommInitRank(newcomm, c->nranks, id, c->rank);
}

int ncclGroupStart(void) { g_group_depth++; return kSuccess; }
int ncclGroupEnd(void)
{
    if (g_group_depth <= 0) return fail(kInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    --g_group_depth;                                  // operations were enqueued as they were issued
    return kSuccess;
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st)
{
    return transfer(true, const_cast<void *>(buf), count, dtype, peer, (Comm *)comm, st);
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st)
{
    return transfer(false, buf, count, dtype, peer, (Comm *)comm, st);
}

int ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, int dtype, int op, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || !c->seg) return fail(kInvalidArgument, "null communicator");
    const size_t bytes = count * type_size(dtype);
    if ((dtype != 7 && dtype != 8) || op < 0 || op > 3) return fail(kInvalidArgument, "the stand-in reduces float / double with sum, prod, max, min");
    if (bytes > kRedMax) return fail(kInvalidArgument, "all-reduce larger than the stand-in's buffer");
    const unsigned s = ++c->red_seq;
    Segment *h = c->seg, *d = c->dseg;
    // nobody overwrites its contribution before every rank has finished reading the previous round
    if (s > 1)
        for (int r = 0; r < c->nranks; r++) HIPCK(hipStreamWaitValue32(st, &d->red_done[r], s - 1, hipStreamWaitValueGte, 0xffffffffu), "hipStreamWaitValue32 failed");
    HIPCK(hipMemcpyAsync(h->red[c->rank], sendbuf, bytes, hipMemcpyDeviceToHost, st), "hipMemcpyAsync D2H failed");
    HIPCK(hipStreamWriteValue32(st, &d->red_arrive[c->rank], s, 0), "hipStreamWriteValue32 failed");
    for (int r = 0; r < c->nranks; r++) HIPCK(hipStreamWaitValue32(st, &d->red_arrive[r], s, hipStreamWaitValueGte, 0xffffffffu), "hipStreamWaitValue32 failed");
    HIPCK(hipLaunchHostFunc(st, reduce_host_fn, new RedJob{c, count, dtype, op}), "hipLaunchHostFunc failed");
    HIPCK(hipMemcpyAsync(recvbuf, h->red_out[c->rank], bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync H2D failed");
    HIPCK(hipStreamWriteValue32(st, &d->red_done[c->rank], s, 0), "hipStreamWriteValue32 failed");
    return kSuccess;
}

const char *ncclGetErrorString(int) { return g_last_error; }

} // extern "C"
