// tests/mock_rccl.cpp -- TEST INFRASTRUCTURE.  A stand-in for librccl.so with
// the nccl* entry points libsdp_hip.so binds (ncclGetUniqueId, CommInitRank,
// CommDestroy, AllGather, Broadcast, AllReduce, GroupStart/End, GetErrorString),
// implemented over a POSIX shared-memory segment and blocking host-staged
// copies, so that SEVERAL RANKS CAN SHARE ONE GPU: RCCL itself refuses two
// ranks on one device, and the test box has one.  Loaded through
// SDP_RCCL_LIBRARY; lets tests/test_gpu_dist.py drive the library's multi-rank
// code path (phase partition, in-place gather addresses, events, policy
// gather) with nranks = 2 on real kernels.  Not a collective library: every
// call synchronises the stream and the ranks.
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
constexpr size_t SLOT = (size_t)96 << 20;          // staging bytes per rank
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
    c->hdr->attached.fetch_add(1);
    while (c->hdr->attached.load() < nranks) sched_yield();
    *out = c;
    return 0;
}

int ncclCommDestroy(void *h)
{
    Comm *c = (Comm *)h;
    if (!c) return 0;
    munmap((void *)c->hdr, c->bytes);
    shm_unlink(c->name);
    delete c;
    return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *h, hipStream_t stream)
{
    Comm *c = (Comm *)h;
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

int ncclGroupStart(void) { return 0; }
int ncclGroupEnd(void) { return 0; }
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
