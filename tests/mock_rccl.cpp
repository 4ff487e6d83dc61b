// tests/mock_rccl.cpp -- TEST INFRASTRUCTURE.  A stand-in for librccl.so with
// the nccl* entry points libsdp_hip.so binds (ncclGetUniqueId, CommInitRank,
// CommDestroy, AllGather, Broadcast, AllReduce, Send, Recv, GroupStart/End, GetErrorString),
// implemented over a POSIX shared-memory segment and blocking host-staged
// copies, so that SEVERAL RANKS CAN SHARE ONE GPU: RCCL itself refuses two
// ranks on one device, and the test box has one.  Loaded through
// SDP_RCCL_LIBRARY; lets tests/test_gpu_dist.py drive the library's multi-rank
// code path (phase partition, in-place gather addresses, events, policy
// gather) with nranks = 2 on real kernels.  Not a collective library.
//
// Two builds.  Default: every call synchronises the stream and the ranks (the
// host blocks until the collective is complete).  -DSDP_MOCK_ASYNC: like the
// real library the call only ENQUEUES work on the stream and returns -- the
// staging copies are hipMemcpyAsync to / from the page-locked shared segment
// and the rendezvous between the ranks are host functions in stream order
// (hipLaunchHostFunc) -- so a missing event / stream dependency in the caller
// (a kernel that reads J before the gather has landed, a host read before the
// communicator stream was joined) is no longer hidden by the stand-in.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sched.h>

namespace {
#ifdef SDP_MOCK_SLOT_MB
constexpr size_t SLOT = (size_t)SDP_MOCK_SLOT_MB << 20;   // (tools/mock8_bench.py: 512^3 slabs)
#elif defined(SDP_MOCK_ASYNC)
constexpr size_t SLOT = (size_t)16 << 20;          // staging bytes per rank (the segment is page-locked)
#else
constexpr size_t SLOT = (size_t)96 << 20;          // staging bytes per rank
#endif
constexpr int MAXR = 8;
struct Header {
    std::atomic<int> arrived;
    std::atomic<int> sense;
    std::atomic<int> attached;
};
struct Comm {
    int rank, n;
    int local_sense;
    char name[64];
    Header *hdr;
    char *slots;
    size_t bytes;
};
const size_t dtype_size[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};

void barrier(Comm *c)
{
    c->local_sense ^= 1;
    if (c->hdr->arrived.fetch_add(1) == c->n - 1) {
        c->hdr->arrived.store(0);
        c->hdr->sense.store(c->local_sense);
    } else {
        while (c->hdr->sense.load() != c->local_sense) sched_yield();
    }
}
char *slot(Comm *c, int r) { return c->slots + (size_t)r * SLOT; }
// the communicator and stream of this thread's last call: a group that records nothing (a rank with nothing to send or
// receive in a phase) still has to meet the others at ncclGroupEnd
thread_local Comm *g_last_comm = nullptr;
thread_local hipStream_t g_last_stream = nullptr;
}  // namespace

extern "C" {

int ncclGetUniqueId(char *id)
{
    memset(id, 0, 128);
    unsigned long long tag = ((unsigned long long)getpid() << 20) ^ (unsigned long long)rand();
    snprintf(id, 128, "/sdp_mock_%llx", tag);
    return 0;
}

struct uid128 { char internal[128]; };

int ncclCommInitRank(void **out, int nranks, uid128 id, int rank)
{
    if (nranks > MAXR) return 4;
    Comm *c = new Comm();
    c->rank = rank; c->n = nranks; c->local_sense = 0;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->bytes = 4096 + (size_t)nranks * SLOT;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return 2; }
    if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return 2; }
    void *m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return 2; }
    c->hdr = (Header *)m;                 // a fresh segment is zero-filled: counters start at 0
    c->slots = (char *)m + 4096;
#ifdef SDP_MOCK_ASYNC
    // page-locked: the staging copies are truly asynchronous
    if (hipHostRegister(m, c->bytes, hipHostRegisterDefault) != hipSuccess) { munmap(m, c->bytes); delete c; return 1; }
#endif
    c->hdr->attached.fetch_add(1);
    while (c->hdr->attached.load() < nranks) sched_yield();
    *out = c;
    return 0;
}

int ncclCommDestroy(void *h)
{
    Comm *c = (Comm *)h;
    if (!c) return 0;
#ifdef SDP_MOCK_ASYNC
    (void)hipDeviceSynchronize();
    (void)hipHostUnregister((void *)c->hdr);
#endif
    munmap((void *)c->hdr, c->bytes);
    shm_unlink(c->name);
    delete c;
    return 0;
}

#ifdef SDP_MOCK_ASYNC
// ---- asynchronous build ----------------------------------------------------
namespace {
void barrier_cb(void *h) { barrier((Comm *)h); }
struct ReduceJob { Comm *c; int dtype; };
void reduce_cb(void *p)
{
    ReduceJob *j = (ReduceJob *)p;
    Comm *c = j->c;
    char *out = slot(c, c->rank) + 64;                       // result next to this rank's operand
    if (j->dtype == 2) {
        int best; memcpy(&best, slot(c, 0), 4);
        for (int r = 1; r < c->n; ++r) { int o; memcpy(&o, slot(c, r), 4); if (o > best) best = o; }
        memcpy(out, &best, 4);
    } else {
        double best; memcpy(&best, slot(c, 0), 8);
        for (int r = 1; r < c->n; ++r) { double o; memcpy(&o, slot(c, r), 8); if (o > best) best = o; }
        memcpy(out, &best, 8);
    }
    delete j;
}
#define TRY(e) do { if ((e) != hipSuccess) return 1; } while (0)
}  // namespace

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *h, hipStream_t stream)
{
    Comm *c = (Comm *)h;
    g_last_comm = c; g_last_stream = stream;
    const size_t bytes = count * dtype_size[dtype];
    if (bytes > SLOT) return 4;
    TRY(hipMemcpyAsync(slot(c, c->rank), send, bytes, hipMemcpyDeviceToHost, stream));
    TRY(hipLaunchHostFunc(stream, barrier_cb, c));            // every rank has staged its part
    for (int r = 0; r < c->n; ++r) {
        char *dst = (char *)recv + (size_t)r * bytes;
        if (r == c->rank && dst == (const char *)send) continue;      // in place
        TRY(hipMemcpyAsync(dst, slot(c, r), bytes, hipMemcpyHostToDevice, stream));
    }
    TRY(hipLaunchHostFunc(stream, barrier_cb, c));            // every rank has read: the slots are free
    return 0;
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *h, hipStream_t stream)
{
    Comm *c = (Comm *)h;
    const size_t bytes = count * dtype_size[dtype];
    if (bytes > SLOT) return 4;
    if (c->rank == root) TRY(hipMemcpyAsync(slot(c, root), send, bytes, hipMemcpyDeviceToHost, stream));
    TRY(hipLaunchHostFunc(stream, barrier_cb, c));
    if (c->rank != root) TRY(hipMemcpyAsync(recv, slot(c, root), bytes, hipMemcpyHostToDevice, stream));
    else if (recv != send) TRY(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, stream));
    TRY(hipLaunchHostFunc(stream, barrier_cb, c));
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *h, hipStream_t stream)
{
    Comm *c = (Comm *)h;
    if ((dtype != 8 && dtype != 2) || op != 2 || count != 1) return 4;   // max of one float64 / int32
    const size_t bytes = dtype_size[dtype];
    TRY(hipMemcpyAsync(slot(c, c->rank), send, bytes, hipMemcpyDeviceToHost, stream));
    TRY(hipLaunchHostFunc(stream, barrier_cb, c));
    TRY(hipLaunchHostFunc(stream, reduce_cb, new ReduceJob{c, dtype}));
    TRY(hipMemcpyAsync(recv, slot(c, c->rank) + 64, bytes, hipMemcpyHostToDevice, stream));
    TRY(hipLaunchHostFunc(stream, barrier_cb, c));            // operands may be overwritten
    return 0;
}
#undef TRY
#else
// ---- synchronous build -----------------------------------------------------
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *h, hipStream_t stream)
{
    Comm *c = (Comm *)h;
    g_last_comm = c; g_last_stream = stream;
    const size_t bytes = count * dtype_size[dtype];
    if (bytes > SLOT) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (hipMemcpy(slot(c, c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    barrier(c);
    for (int r = 0; r < c->n; ++r) {
        char *dst = (char *)recv + (size_t)r * bytes;
        if (r == c->rank && dst == (const char *)send) continue;      // in place
        if (hipMemcpy(dst, slot(c, r), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    barrier(c);
    return 0;
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *h, hipStream_t stream)
{
    Comm *c = (Comm *)h;
    const size_t bytes = count * dtype_size[dtype];
    if (bytes > SLOT) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (c->rank == root && hipMemcpy(slot(c, root), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    barrier(c);
    if (c->rank != root) {
        if (hipMemcpy(recv, slot(c, root), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    } else if (recv != send) {
        if (hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    }
    barrier(c);
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *h, hipStream_t stream)
{
    Comm *c = (Comm *)h;
    if ((dtype != 8 && dtype != 2) || op != 2 || count != 1) return 4;   // max of one float64 / int32
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (dtype == 2) {
        int v;
        if (hipMemcpy(&v, send, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        memcpy(slot(c, c->rank), &v, 4);
        barrier(c);
        int best = v;
        for (int r = 0; r < c->n; ++r) { int o; memcpy(&o, slot(c, r), 4); if (o > best) best = o; }
        barrier(c);
        if (hipMemcpy(recv, &best, 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
        return 0;
    }
    double v;
    if (hipMemcpy(&v, send, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    memcpy(slot(c, c->rank), &v, 8);
    barrier(c);
    double best = v;
    for (int r = 0; r < c->n; ++r) { double o; memcpy(&o, slot(c, r), 8); if (o > best) best = o; }
    barrier(c);
    if (hipMemcpy(recv, &best, 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

#endif  // SDP_MOCK_ASYNC

// ---- grouped point-to-point (both builds): ncclSend / ncclRecv between ncclGroupStart and ncclGroupEnd are recorded
// and carried out at ncclGroupEnd -- every rank stages what it sends to rank q in sub-slot q of its own slot, the
// ranks meet, every rank fetches sub-slot `me` of each sender's slot, the ranks meet again.  EVERY rank must close
// the same number of groups (the library does: one per phase, with or without transfers); one send and one receive
// per pair of ranks and group, each below SLOT / nranks bytes.
namespace {
struct P2P { bool send; void *buf; size_t bytes; int peer; Comm *c; hipStream_t stream; };
thread_local P2P g_ops[4 * MAXR];
thread_local int g_nops = 0, g_depth = 0;
}  // namespace
int ncclGroupStart(void) { if (g_depth++ == 0) g_nops = 0; return 0; }
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *h, hipStream_t stream)
{
    if (g_depth == 0 || g_nops >= 4 * MAXR) return 4;
    g_ops[g_nops++] = P2P{true, (void *)buf, count * dtype_size[dtype], peer, (Comm *)h, stream};
    return 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *h, hipStream_t stream)
{
    if (g_depth == 0 || g_nops >= 4 * MAXR) return 4;
    g_ops[g_nops++] = P2P{false, buf, count * dtype_size[dtype], peer, (Comm *)h, stream};
    return 0;
}
int ncclGroupEnd(void)
{
    if (--g_depth > 0) return 0;
    Comm *c = g_nops ? g_ops[0].c : g_last_comm;
    hipStream_t stream = g_nops ? g_ops[0].stream : g_last_stream;
    if (!c) return 4;
    g_last_comm = c; g_last_stream = stream;
    const size_t sub = SLOT / (size_t)c->n;
    for (int k = 0; k < g_nops; ++k) if (g_ops[k].bytes > sub || g_ops[k].c != c) return 4;
#ifdef SDP_MOCK_ASYNC
    for (int k = 0; k < g_nops; ++k)
        if (g_ops[k].send && hipMemcpyAsync(slot(c, c->rank) + (size_t)g_ops[k].peer * sub, g_ops[k].buf, g_ops[k].bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
    if (hipLaunchHostFunc(stream, barrier_cb, c) != hipSuccess) return 1;
    for (int k = 0; k < g_nops; ++k)
        if (!g_ops[k].send && hipMemcpyAsync(g_ops[k].buf, slot(c, g_ops[k].peer) + (size_t)c->rank * sub, g_ops[k].bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    if (hipLaunchHostFunc(stream, barrier_cb, c) != hipSuccess) return 1;
#else
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    for (int k = 0; k < g_nops; ++k)
        if (g_ops[k].send && hipMemcpy(slot(c, c->rank) + (size_t)g_ops[k].peer * sub, g_ops[k].buf, g_ops[k].bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    barrier(c);
    for (int k = 0; k < g_nops; ++k)
        if (!g_ops[k].send && hipMemcpy(g_ops[k].buf, slot(c, g_ops[k].peer) + (size_t)c->rank * sub, g_ops[k].bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    barrier(c);
#endif
    g_nops = 0;
    return 0;
}
const char *ncclGetErrorString(int r)
{
    switch (r) {
    case 0: return "success";
    case 1: return "mock: HIP call failed";
    case 2: return "mock: shared memory failed";
    case 4: return "mock: unsupported argument";
    default: return "mock: error";
    }
}
}  // extern "C"
