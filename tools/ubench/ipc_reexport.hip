// ipc_reexport.hip -- what happens when a device buffer that was exported over HIP IPC is freed and a new
// allocation of the same size (usually at the same address) is exported again?  (round 4
// parked such buffers instead of freeing them, on the evidence of 24 clean runs and a guessed mechanism.)
//
//   hipcc --offload-arch=gfx950 -O2 -o ipc_reexport ipc_reexport.hip && ./ipc_reexport [rounds] [MiB] [mode]
//
// Two processes on ONE device (forked before either touches HIP).  Per round:
//   A: malloc -> fill -> hipIpcGetMemHandle -> handle to B
//   B: hipIpcOpenMemHandle -> read back / write through the mapping -> (mode 0: hipIpcCloseMemHandle) -> ack
//   A: check B's writes -> hipFree -> next round (the allocator usually hands the same address out again)
// mode 0: B closes its mapping before A frees (the order the library keeps)
// mode 1: B closes its mapping only AFTER A has freed and re-exported (a stale mapping at the moment of the export)
// mode 2: B never closes (mappings pile up)
// Every HIP status that is not hipSuccess is printed with the round and the step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <vector>

__global__ void k_fill(uint32_t *p, size_t n, uint32_t v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (uint32_t)i;
}
__global__ void k_check(const uint32_t *p, size_t n, uint32_t v, unsigned long long *bad)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != v + (uint32_t)i) atomicAdd(bad, 1ull);
}

static int g_fail = 0;
#define CK(who, round, expr)                                                                              \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) {                                                                          \
            printf("%s round %d: %s -> %s\n", who, round, #expr, hipGetErrorString(e__));                 \
            fflush(stdout);                                                                               \
            ++g_fail;                                                                                     \
            (void)hipGetLastError();                                                                      \
        }                                                                                                 \
    } while (0)

static unsigned long long count_bad(const uint32_t *p, size_t n, uint32_t v)
{
    unsigned long long *d = nullptr, h = 0;
    if (hipMalloc(&d, 8) != hipSuccess) return ~0ull;
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k_check, dim3(256), dim3(256), 0, 0, p, n, v, d);
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) h = ~0ull;
    (void)hipFree(d);
    return h;
}

struct Msg { hipIpcMemHandle_t h; uint64_t addr; int round; int ok; };

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 128) << 20;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const size_t n = bytes / 4;
    int a2b[2], b2a[2];
    if (pipe(a2b) || pipe(b2a)) return 2;
    const pid_t child = fork();                               // before any HIP call in either process
    if (child == 0) {
        // ---- B, the importer
        std::vector<void *> open_maps;
        void *stale = nullptr;
        for (int r = 0; r < rounds; ++r) {
            Msg m;
            if (read(a2b[0], &m, sizeof(m)) != (ssize_t)sizeof(m)) return 3;
            if (mode == 1 && stale) { CK("B", r, hipIpcCloseMemHandle(stale)); stale = nullptr; }   // (A has freed and re-exported)
            void *q = nullptr;
            int ok = 1;
            if (m.ok) {
                CK("B", r, hipIpcOpenMemHandle(&q, m.h, hipIpcMemLazyEnablePeerAccess));
                if (q) {
                    const unsigned long long bad = count_bad((const uint32_t *)q, n, 1000u * (uint32_t)r);
                    if (bad) { printf("B round %d: %llu words of A's fill read wrong through the mapping\n", r, bad); ok = 0; ++g_fail; }
                    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (uint32_t *)q, n, 7u + 1000u * (uint32_t)r);
                    CK("B", r, hipDeviceSynchronize());
                    if (mode == 0) CK("B", r, hipIpcCloseMemHandle(q));
                    else if (mode == 1) stale = q;
                    else open_maps.push_back(q);
                } else ok = 0;
            }
            Msg back;
            memset(&back, 0, sizeof(back));
            back.round = r; back.ok = ok;
            if (write(b2a[1], &back, sizeof(back)) != (ssize_t)sizeof(back)) return 3;
        }
        printf("B: %d failure(s)\n", g_fail);
        return g_fail ? 1 : 0;
    }
    // ---- A, the exporter
    uint64_t first_addr = 0;
    int same_addr = 0;
    for (int r = 0; r < rounds; ++r) {
        void *p = nullptr;
        CK("A", r, hipMalloc(&p, bytes));
        Msg m;
        memset(&m, 0, sizeof(m));
        m.round = r; m.addr = (uint64_t)p;
        if (r == 0) first_addr = m.addr; else if (m.addr == first_addr) ++same_addr;
        if (p) {
            hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (uint32_t *)p, n, 1000u * (uint32_t)r);
            CK("A", r, hipDeviceSynchronize());
            hipError_t e = hipIpcGetMemHandle(&m.h, p);
            if (e != hipSuccess) { printf("A round %d: hipIpcGetMemHandle(%p) -> %s\n", r, p, hipGetErrorString(e)); ++g_fail; (void)hipGetLastError(); }
            m.ok = e == hipSuccess;
        }
        if (write(a2b[1], &m, sizeof(m)) != (ssize_t)sizeof(m)) return 3;
        Msg back;
        if (read(b2a[0], &back, sizeof(back)) != (ssize_t)sizeof(back)) return 3;
        if (p && m.ok && back.ok) {
            const unsigned long long bad = count_bad((const uint32_t *)p, n, 7u + 1000u * (uint32_t)r);
            if (bad) { printf("A round %d: %llu words of B's writes did not arrive\n", r, bad); ++g_fail; }
        }
        if (p) CK("A", r, hipFree(p));
    }
    int st = 0;
    waitpid(child, &st, 0);
    printf("A: mode %d, %d rounds of %zu MiB, %d allocation(s) at the first address again, %d failure(s); B exit status %d\n",
           mode, rounds, bytes >> 20, same_addr, g_fail, WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    return g_fail || st ? 1 : 0;
}
