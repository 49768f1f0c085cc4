// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so that moves data between PROCESSES THAT SHARE ONE GPU.
//
// Real RCCL refuses two ranks on the same device, so on a one-GPU box the library's RCCL code path (csrc/bq_halo.hip:
// unique id -> ncclCommInitRank -> grouped ncclSend / ncclRecv between neighbours and arbitrary peers -> in-stream
// ncclAllReduce) could only ever run with one rank.  This library exports the nine symbols bq_halo.hip binds and
// implements them over a POSIX shared-memory segment, so that 2-6 ranks on one card exercise exactly that code path:
// the bootstrap, the send/receive matching (per ordered pair, in issue order, equal counts -- checked), the box and
// plane offsets, the all-reduce call sites.  It is selected with BQ_RCCL_LIBRARY=<path> (bq_halo.hip: load_rccl).
//
// What it does NOT reproduce: asynchrony.  ncclGroupEnd / ncclAllReduce first synchronise the stream they were given
// and then move the data with blocking copies through host memory, so stream-ordering mistakes on the caller's side
// are not found here -- matching, sizes, peers and deadlocks are.  Every wait has a timeout and fails loudly.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kChunk = 1u << 20;                 // bytes per mailbox slot
constexpr size_t kRedMax = 4096;                    // bytes per rank in the all-reduce area
constexpr double kTimeout = 120.0;                  // seconds

struct Mailbox {
    std::atomic<uint64_t> written, consumed;        // chunks
    std::atomic<uint64_t> msg_bytes;                // size of the message the sender is on (header; checked by the receiver)
    char pad[64 - 3 * sizeof(std::atomic<uint64_t>)];
};

struct Segment {
    std::atomic<int> arrived;                       // ncclCommInitRank
    std::atomic<int> bar_count;
    std::atomic<int> bar_gen;
    std::atomic<int> left;                          // ncclCommDestroy
    Mailbox box[kMaxRanks * kMaxRanks];             // [src * kMaxRanks + dst]
    unsigned char red[kMaxRanks][kRedMax];
    unsigned char data[kMaxRanks * kMaxRanks][kChunk];
};

struct Comm {
    Segment *seg = nullptr;
    int rank = 0, nranks = 1;
    std::string name;
    int splits = 0;                                  // ncclCommSplit calls made on this communicator (names the children)
};

struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t stream; Comm *comm; size_t done = 0; bool header = false; };

thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;
const char *g_last_error = "no error";

enum { kSuccess = 0, kUnhandledCuda = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int fail(int code, const char *what)
{
    g_last_error = what;
    fprintf(stderr, "[fake_rccl] %s\n", what);
    return code;
}

size_t type_size(int dtype)
{
    switch (dtype) {
    case 0: case 1: return 1;       // int8 / uint8
    case 2: case 3: return 4;       // int32 / uint32
    case 4: case 5: return 8;       // int64 / uint64
    case 6: return 2;               // half
    case 7: return 4;               // float
    case 8: return 8;               // double
    default: return 0;
    }
}

bool barrier(Comm *c)
{
    Segment *s = c->seg;
    const int gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == c->nranks) {
        s->bar_count.store(0);
        s->bar_gen.fetch_add(1);
        return true;
    }
    const double t0 = now();
    while (s->bar_gen.load() == gen) {
        if (now() - t0 > kTimeout) return false;
        std::this_thread::yield();
    }
    return true;
}

// all queued operations of a group make progress together: a rank that sends a long message to a peer which is itself
// sending one to us must keep consuming while it produces
int run_ops(std::vector<Op> &ops)
{
    if (ops.empty()) return kSuccess;
    for (Op &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return fail(kUnhandledCuda, "hipStreamSynchronize failed before a group");
    // per ordered pair the operations are matched in issue order: only the first unfinished one of a pair may move
    const double t0 = now();
    size_t remaining = ops.size();
    while (remaining) {
        bool progress = false;
        bool busy[2][kMaxRanks] = {};
        for (Op &o : ops) {
            if (o.done == o.bytes && o.header) continue;
            bool &pair_busy = busy[o.send ? 1 : 0][o.peer];
            if (pair_busy) continue;
            pair_busy = true;
            Comm *c = o.comm;
            const int src = o.send ? c->rank : o.peer, dst = o.send ? o.peer : c->rank;
            Mailbox &m = c->seg->box[src * kMaxRanks + dst];
            unsigned char *slot = c->seg->data[src * kMaxRanks + dst];
            if (o.send) {
                if (m.written.load() != m.consumed.load()) continue;            // the slot is still full
                if (!o.header) { m.msg_bytes.store(o.bytes); o.header = true; }
                const size_t n = std::min(kChunk, o.bytes - o.done);
                if (n && hipMemcpy(slot, (char *)o.buf + o.done, n, hipMemcpyDeviceToHost) != hipSuccess)
                    return fail(kUnhandledCuda, "hipMemcpy D2H failed in ncclSend");
                o.done += n;
                m.written.fetch_add(1);
                progress = true;
            } else {
                if (m.written.load() == m.consumed.load()) continue;            // nothing there yet
                if (!o.header) {
                    if (m.msg_bytes.load() != o.bytes) {
                        fprintf(stderr, "[fake_rccl] rank %d: ncclRecv from %d expects %zu bytes, the matching ncclSend has %llu\n",
                                c->rank, o.peer, o.bytes, (unsigned long long)m.msg_bytes.load());
                        return fail(kInvalidArgument, "send / receive sizes do not match");
                    }
                    o.header = true;
                }
                const size_t n = std::min(kChunk, o.bytes - o.done);
                if (n && hipMemcpy((char *)o.buf + o.done, slot, n, hipMemcpyHostToDevice) != hipSuccess)
                    return fail(kUnhandledCuda, "hipMemcpy H2D failed in ncclRecv");
                o.done += n;
                m.consumed.fetch_add(1);
                progress = true;
            }
            if (o.done == o.bytes) remaining--;
        }
        if (!progress) {
            if (now() - t0 > kTimeout) {
                for (const Op &o : ops)
                    if (o.done != o.bytes || !o.header)
                        fprintf(stderr, "[fake_rccl] rank %d: stuck %s peer %d, %zu of %zu bytes\n", o.comm->rank,
                                o.send ? "ncclSend to" : "ncclRecv from", o.peer, o.done, o.bytes);
                return fail(kInternalError, "timeout: a send or receive found no partner (deadlock or mismatched group)");
            }
            std::this_thread::yield();
        }
    }
    return kSuccess;
}

int enqueue(bool send, void *buf, size_t count, int dtype, int peer, Comm *c, hipStream_t st)
{
    if (!c || !c->seg) return fail(kInvalidArgument, "null communicator");
    const size_t ts = type_size(dtype);
    if (!ts) return fail(kInvalidArgument, "unknown data type");
    if (peer < 0 || peer >= c->nranks) return fail(kInvalidArgument, "peer out of range");
    if (count && !buf) return fail(kInvalidArgument, "null buffer");
    g_ops.push_back(Op{send, buf, count * ts, peer, st, c});
    if (count == 0) g_ops.back().bytes = 0;
    if (g_group_depth == 0) {
        std::vector<Op> ops;
        ops.swap(g_ops);
        return run_ops(ops);
    }
    return kSuccess;
}

template <typename T>
void reduce(Segment *s, int nranks, size_t count, int op, T *out)
{
    for (size_t i = 0; i < count; i++) {
        T acc = ((const T *)s->red[0])[i];
        for (int r = 1; r < nranks; r++) {
            const T v = ((const T *)s->red[r])[i];
            switch (op) {
            case 0: acc = acc + v; break;
            case 1: acc = acc * v; break;
            case 2: acc = v > acc ? v : acc; break;
            case 3: acc = v < acc ? v : acc; break;
            default: break;
            }
        }
        out[i] = acc;
    }
}

} // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return fail(kInvalidArgument, "null id");
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/bq_fake_rccl_%d_%llx", (int)getpid(),
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
    c->seg->arrived.fetch_add(1);
    const double t0 = now();
    while (c->seg->arrived.load() < nranks) {
        if (now() - t0 > kTimeout) { delete c; return fail(kInternalError, "timeout: not every rank reached ncclCommInitRank"); }
        std::this_thread::yield();
    }
    *comm = c;
    return kSuccess;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return kSuccess;
    const bool last = c->seg->left.fetch_add(1) + 1 == c->nranks;
    munmap(c->seg, sizeof(Segment));
    if (last) shm_unlink(c->name.c_str());
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
    if (--g_group_depth) return kSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st)
{
    return enqueue(true, const_cast<void *>(buf), count, dtype, peer, (Comm *)comm, st);
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st)
{
    return enqueue(false, buf, count, dtype, peer, (Comm *)comm, st);
}

int ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, int dtype, int op, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    if (!c || !c->seg) return fail(kInvalidArgument, "null communicator");
    if (g_group_depth) return fail(kInvalidUsage, "ncclAllReduce inside a group is not supported by the stand-in");
    const size_t ts = type_size(dtype), bytes = count * ts;
    if ((dtype != 7 && dtype != 8) || op < 0 || op > 3) return fail(kInvalidArgument, "the stand-in reduces float / double with sum, prod, max, min");
    if (bytes > kRedMax) return fail(kInvalidArgument, "all-reduce larger than the stand-in's buffer");
    if (hipStreamSynchronize(st) != hipSuccess) return fail(kUnhandledCuda, "hipStreamSynchronize failed");
    if (hipMemcpy(c->seg->red[c->rank], sendbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(kUnhandledCuda, "hipMemcpy D2H failed");
    if (!barrier(c)) return fail(kInternalError, "timeout in the all-reduce (not every rank called it)");
    unsigned char out[kRedMax];
    if (dtype == 7) reduce<float>(c->seg, c->nranks, count, op, (float *)out);
    else reduce<double>(c->seg, c->nranks, count, op, (double *)out);
    if (!barrier(c)) return fail(kInternalError, "timeout in the all-reduce");      // nobody overwrites red[] before all have read it
    if (hipMemcpy(recvbuf, out, bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(kUnhandledCuda, "hipMemcpy H2D failed");
    return kSuccess;
}

const char *ncclGetErrorString(int) { return g_last_error; }

} // extern "C"
