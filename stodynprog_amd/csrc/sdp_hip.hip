// sdp_hip.hip -- libsdp_hip.so: C ABI (include/sdp_hip.h) + built-in gfx950
// kernels (stand-alone multilinear interpolation, tabulated backup,
// relative-DP shift) + launch of the generated-model code objects + RCCL glue.
// Plain HIP runtime; no PyTorch, no CUDA compatibility layer.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>
#include <utility>

#include "../../include/sdp_hip.h"
#include "sdp_device.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess)                                                         \
            return fail(e__ == hipErrorOutOfMemory ? SDP_ENOMEM : SDP_EHIP,            \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),        \
                        __FILE__, __LINE__);                                           \
    } while (0)

extern "C" const char *sdp_last_error(void) { return g_err; }

static size_t real_size(int dtype) { return dtype == SDP_F32 ? 4 : 8; }

// ---------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------
extern "C" int sdp_device_count(int *count)
{
    if (!count) return fail(SDP_EINVAL, "count is NULL");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return fail(SDP_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return SDP_OK;
}

extern "C" int sdp_set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return SDP_OK;
}

extern "C" int sdp_device_info(int device, char *name, int *compute_units, int64_t *hbm_bytes,
                               char *gcn_arch)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name) { strncpy(name, prop.name, 255); name[255] = 0; }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (gcn_arch) { strncpy(gcn_arch, prop.gcnArchName, 63); gcn_arch[63] = 0; }
    return SDP_OK;
}

extern "C" int sdp_synchronize(void)
{
    HIP_TRY(hipDeviceSynchronize());
    return SDP_OK;
}

static int device_cus()
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
}

// ---------------------------------------------------------------------------
// built-in kernel: stand-alone multilinear interpolation
// (multilinear_cython.pyx:17-49 dispatcher, 51-300 point kernels).
// One lane per query point: the d coordinate rows s[k][i] are read coalesced,
// the 2^d vertex gathers go through L1/L2, the value rows are looped in-lane
// so the cell location is computed once per point.
// ---------------------------------------------------------------------------
template <typename real, int D>
__global__ void __launch_bounds__(256) k_mlinterp(SdpInterpArgs a)
{
    SdpGrid<real, D> grid;
    real smin[D], smax[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { smin[k] = (real)a.smin[k]; smax[k] = (real)a.smax[k]; }
    sdp_make_grid<real, D>(grid, a.orders, smin, smax);
    const real *__restrict__ s = (const real *)a.s;
    const real *__restrict__ values = (const real *)a.values;
    real *__restrict__ out = (real *)a.out;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_s; i += stride) {
        SdpCell<real, D, double> c;
#pragma unroll
        for (int k = 0; k < D; ++k) sdp_locate_axis<real, D, double>(grid, k, s[k * a.n_s + i], c);
        for (int v = 0; v < a.n_v; ++v)
            out[v * a.n_s + i] = (real)SdpLerp<real, D, double, 0>::eval(values + v * a.S, grid, c, 0);
    }
}

template <typename real>
static int launch_mlinterp(const SdpInterpArgs &a, hipStream_t stream)
{
    if (a.n_s == 0 || a.n_v == 0) return SDP_OK;
    int64_t blocks = (a.n_s + 255) / 256;
    const int64_t cap = (int64_t)device_cus() * 16;
    if (blocks > cap) blocks = cap;
    dim3 g((unsigned)blocks), b(256);
    switch (a.d) {
    case 1: hipLaunchKernelGGL((k_mlinterp<real, 1>), g, b, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((k_mlinterp<real, 2>), g, b, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((k_mlinterp<real, 3>), g, b, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((k_mlinterp<real, 4>), g, b, 0, stream, a); break;
    default: return fail(SDP_EDIM, "Can't interpolate in dimension strictly greater than 5");
    }
    HIP_TRY(hipGetLastError());
    return SDP_OK;
}

// Buffers that were exported to other processes (hipIpcGetMemHandle) are not handed back to the allocator when their
// problem goes: a later allocation at the same address gets an IPC handle the peers may still hold a stale mapping for
// -- seen as `hipIpcGetMemHandle: invalid argument`, and as GPU page faults in a peer's stores a few plans later,
// roughly once in seven runs of bench.py's exchange tuning (nine problems with mappings, one after the other) even
// with "every rank unmaps, barrier, then free" (tools/ubench/ipc_reexport.hip is the two-process probe of that
// sequence; DESIGN section 5 has what it showed).  They rest here instead, and the NEXT problem that needs a buffer of
// that size on that device takes one back (the same live allocation exports to the same handle: nothing stale about
// it), so a process that re-plans with one grid holds two such buffers however often it re-plans.  Bounded: beyond
// PARK_CAP_BYTES the oldest are freed; an allocation that fails for lack of memory frees all of them and tries again.
struct Parked { void *p; size_t bytes; int device; };
static std::mutex g_park_mu;
static std::vector<Parked> g_parked;
static size_t g_parked_bytes = 0;
static const size_t PARK_CAP_BYTES = (size_t)4 << 30;

static int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev;
}

// free every parked buffer of the current device (all of them with any_device); returns the bytes released
static size_t drain_parked(bool any_device = false)
{
    std::lock_guard<std::mutex> lock(g_park_mu);
    const int dev = current_device();
    size_t freed = 0;
    std::vector<Parked> keep;
    for (auto &q : g_parked) {
        if (any_device || q.device == dev) { (void)hipFree(q.p); freed += q.bytes; }
        else keep.push_back(q);
    }
    g_parked.swap(keep);
    g_parked_bytes -= freed;
    return freed;
}

struct DevBuf {
    void *p = nullptr;
    bool ever_exported = false;            // handed to another process once (this allocation): park it, never free it early
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        ever_exported = false;
        if (bytes == 0) bytes = 8;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipErrorOutOfMemory && drain_parked() > 0) {          // (parked buffers are the first to go)
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); return fail(SDP_ENOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); }
        return SDP_OK;
    }
    // a buffer that may be exported later (V, J of a problem): a parked one of this size if there is one.
    // Never less than EXPORT_MIN_BYTES: the runtime carves small allocations out of shared 2 MiB blocks, and
    // a HIP IPC handle of such a fragment shares the whole BLOCK with the other process -- V, J and whatever
    // else lives there, mapped once per handle.  A buffer of its own block is exported alone.
    static constexpr size_t EXPORT_MIN_BYTES = (size_t)4 << 20;
    int alloc_exportable(size_t bytes)
    {
        if (bytes < EXPORT_MIN_BYTES) bytes = EXPORT_MIN_BYTES;
        if (p) { (void)hipFree(p); p = nullptr; }
        {
            std::lock_guard<std::mutex> lock(g_park_mu);
            const int dev = current_device();
            for (size_t k = 0; k < g_parked.size(); ++k)
                if (g_parked[k].bytes == bytes && g_parked[k].device == dev) {
                    p = g_parked[k].p;
                    ever_exported = true;
                    g_parked_bytes -= bytes;
                    g_parked.erase(g_parked.begin() + (long)k);
                    return SDP_OK;
                }
        }
        return alloc(bytes);
    }
};

// Mappings of other processes' buffers (hipIpcOpenMemHandle), ONE per remote allocation and process, counted.
// Problems come and go in an order the ranks do not share (a Python object waiting for its garbage collector keeps
// its mappings for a while), and a rank that re-plans takes its parked buffers back and exports them again: a second
// problem here then maps what a first one still holds.  A second hipIpcOpenMemHandle of the same handle and the
// first problem's hipIpcCloseMemHandle of it are exactly what corrupted things in round 5's first runs (the same
// remote buffer at one address twice: the close of the old mapping took the new one with it -- wrong J, GPU page
// faults, and `hipIpcGetMemHandle: invalid argument` once the runtime's own table of allocations was off).  So the
// library opens a handle once, hands the same address to whoever asks again, and closes it with its last user.
struct IpcMap { hipIpcMemHandle_t h; void *ptr; int refs; };
static std::mutex g_ipc_mu;
static std::vector<IpcMap> g_ipc_maps;

static void ipc_trace(const char *what, const void *ptr, const hipIpcMemHandle_t *h, int refs)
{
#ifdef SDP_TEST_HOOKS
    static const int on = getenv("SDP_IPC_TRACE") != nullptr;
    if (!on) return;
    unsigned long long sig = 0;
    if (h) for (size_t k = 0; k < sizeof(*h); ++k) sig = sig * 1099511628211ull + ((const unsigned char *)h)[k];
    fprintf(stderr, "[ipc %d] %s ptr %p handle %016llx refs %d\n", (int)getpid(), what, ptr, sig, refs);
#else
    (void)what; (void)ptr; (void)h; (void)refs;
#endif
}

static hipError_t ipc_open(void **out, const hipIpcMemHandle_t &h)
{
    std::lock_guard<std::mutex> lock(g_ipc_mu);
    for (auto &m : g_ipc_maps)
        if (memcmp(&m.h, &h, sizeof(h)) == 0) {
            ++m.refs;
            *out = m.ptr;
            ipc_trace("reuse", m.ptr, &h, m.refs);
            return hipSuccess;
        }
    void *q = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { *out = nullptr; ipc_trace("open FAILED", nullptr, &h, 0); return e; }
    g_ipc_maps.push_back({h, q, 1});
    *out = q;
    ipc_trace("open", q, &h, 1);
    return hipSuccess;
}

static void ipc_close(void *ptr)
{
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(g_ipc_mu);
    for (size_t k = 0; k < g_ipc_maps.size(); ++k)
        if (g_ipc_maps[k].ptr == ptr) {
            if (--g_ipc_maps[k].refs > 0) { ipc_trace("release", ptr, &g_ipc_maps[k].h, g_ipc_maps[k].refs); return; }
            ipc_trace("close", ptr, &g_ipc_maps[k].h, 0);
            (void)hipIpcCloseMemHandle(ptr);
            g_ipc_maps.erase(g_ipc_maps.begin() + (long)k);
            return;
        }
    ipc_trace("close of an unknown mapping", ptr, nullptr, 0);
}

static void park_exported(DevBuf &b, size_t bytes)
{
    if (!b.p || !b.ever_exported) return;                  // (never exported: its destructor frees it)
    if (bytes < DevBuf::EXPORT_MIN_BYTES) bytes = DevBuf::EXPORT_MIN_BYTES;
    std::lock_guard<std::mutex> lock(g_park_mu);
    g_parked.push_back({b.p, bytes, current_device()});
    g_parked_bytes += bytes;
    b.p = nullptr;
    size_t k = 0;
    while (k + 1 < g_parked.size() && g_parked_bytes > PARK_CAP_BYTES) {     // (the newest stays)
        (void)hipFree(g_parked[k].p);
        g_parked_bytes -= g_parked[k].bytes;
        ++k;
    }
    g_parked.erase(g_parked.begin(), g_parked.begin() + (long)k);
}

static int upload(DevBuf &b, const void *host, size_t bytes)
{
    int rc = b.alloc(bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpy(b.p, host, bytes, hipMemcpyHostToDevice));
    return SDP_OK;
}

template <typename real>
static int mlinterp_host(int d, const real *smin, const real *smax, const int64_t *orders,
                         const real *values, int64_t n_v, const real *s, int64_t n_s, real *out)
{
    if (d < 1 || d > SDP_MAXD)                                    // pyx:46-47
        return fail(SDP_EDIM, "Can't interpolate in dimension strictly greater than 5");
    if (!smin || !smax || !orders || (!values && n_v) || (!s && n_s) || (!out && n_v && n_s))
        return fail(SDP_EINVAL, "NULL argument");
    if (n_v < 0 || n_s < 0) return fail(SDP_EINVAL, "negative size");
    SdpInterpArgs a;
    memset(&a, 0, sizeof(a));
    int64_t S = 1;
    for (int k = 0; k < d; ++k) {
        if (orders[k] < 2) return fail(SDP_EINVAL, "orders[%d] = %lld: need at least 2 points per axis", k, (long long)orders[k]);
        a.orders[k] = (int32_t)orders[k];
        a.smin[k] = (double)smin[k];
        a.smax[k] = (double)smax[k];
        S *= orders[k];
        if (S >= (int64_t)1 << 31) return fail(SDP_EINVAL, "grid too large: 32-bit vertex indices (pyx:164-165)");
    }
    if (n_v == 0 || n_s == 0) return SDP_OK;
    DevBuf dv, ds, dout;
    int rc;
    if ((rc = dv.alloc(sizeof(real) * S * n_v))) return rc;
    if ((rc = ds.alloc(sizeof(real) * d * n_s))) return rc;
    if ((rc = dout.alloc(sizeof(real) * n_v * n_s))) return rc;
    HIP_TRY(hipMemcpy(dv.p, values, sizeof(real) * S * n_v, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ds.p, s, sizeof(real) * d * n_s, hipMemcpyHostToDevice));
    a.values = dv.p; a.s = ds.p; a.out = dout.p;
    a.n_s = n_s; a.S = S; a.n_v = (int32_t)n_v; a.d = d;
    if ((rc = launch_mlinterp<real>(a, 0))) return rc;
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(real) * n_v * n_s, hipMemcpyDeviceToHost));
    return SDP_OK;
}

extern "C" int sdp_mlinterp_f64(int d, const double *smin, const double *smax,
                                const int64_t *orders, const double *values, int64_t n_v,
                                const double *s, int64_t n_s, double *out)
{
    return mlinterp_host<double>(d, smin, smax, orders, values, n_v, s, n_s, out);
}

extern "C" int sdp_mlinterp_f32(int d, const float *smin, const float *smax,
                                const int64_t *orders, const float *values, int64_t n_v,
                                const float *s, int64_t n_s, float *out)
{
    return mlinterp_host<float>(d, smin, smax, orders, values, n_v, s, n_s, out);
}

// ---------------------------------------------------------------------------
// interpolator handle: the value rows stay on the device between evaluations
// (MlinInterpolator.set_values once, __call__ many times -- e.g. the policy
// look-ups of a simulation loop, reference storage_control.py:242-251)
// ---------------------------------------------------------------------------
struct sdp_interp {
    int dtype = SDP_F64, d = 0;
    int64_t S = 0, n_v = 0;
    int32_t orders[SDP_MAXD] = {0, 0, 0, 0};
    double smin[SDP_MAXD], smax[SDP_MAXD];
    DevBuf values, pts, out;
    size_t pts_bytes = 0, out_bytes = 0;
};

extern "C" int sdp_interp_create(int dtype, int d, const double *smin, const double *smax,
                                 const int64_t *orders, const void *host_values, int64_t n_v,
                                 sdp_interp **out)
{
    if (dtype != SDP_F64 && dtype != SDP_F32) return fail(SDP_EINVAL, "dtype must be SDP_F64 or SDP_F32");
    if (d < 1 || d > SDP_MAXD) return fail(SDP_EDIM, "Can't interpolate in dimension strictly greater than 5");
    if (!smin || !smax || !orders || !host_values || !out || n_v < 1) return fail(SDP_EINVAL, "bad argument");
    std::unique_ptr<sdp_interp> h(new sdp_interp());
    h->dtype = dtype; h->d = d; h->n_v = n_v;
    int64_t S = 1;
    for (int k = 0; k < d; ++k) {
        if (orders[k] < 2) return fail(SDP_EINVAL, "orders[%d] = %lld: need at least 2 points per axis", k, (long long)orders[k]);
        h->orders[k] = (int32_t)orders[k]; h->smin[k] = smin[k]; h->smax[k] = smax[k];
        S *= orders[k];
        if (S >= (int64_t)1 << 31) return fail(SDP_EINVAL, "grid too large: 32-bit vertex indices (pyx:164-165)");
    }
    h->S = S;
    int rc = upload(h->values, host_values, (size_t)S * n_v * real_size(dtype));
    if (rc) return rc;
    *out = h.release();
    return SDP_OK;
}

extern "C" int sdp_interp_destroy(sdp_interp *h)
{
    delete h;
    return SDP_OK;
}

extern "C" int sdp_interp_eval(sdp_interp *h, const void *host_s, int64_t n_s, void *host_out)
{
    if (!h || n_s < 0 || (n_s && (!host_s || !host_out))) return fail(SDP_EINVAL, "bad argument");
    if (n_s == 0) return SDP_OK;
    const size_t rs = real_size(h->dtype);
    const size_t pb = (size_t)h->d * n_s * rs, ob = (size_t)h->n_v * n_s * rs;
    int rc;
    if (h->pts_bytes < pb) { if ((rc = h->pts.alloc(pb))) return rc; h->pts_bytes = pb; }
    if (h->out_bytes < ob) { if ((rc = h->out.alloc(ob))) return rc; h->out_bytes = ob; }
    HIP_TRY(hipMemcpy(h->pts.p, host_s, pb, hipMemcpyHostToDevice));
    SdpInterpArgs a;
    memset(&a, 0, sizeof(a));
    for (int k = 0; k < h->d; ++k) { a.orders[k] = h->orders[k]; a.smin[k] = h->smin[k]; a.smax[k] = h->smax[k]; }
    a.values = h->values.p; a.s = h->pts.p; a.out = h->out.p;
    a.n_s = n_s; a.S = h->S; a.n_v = (int32_t)h->n_v; a.d = h->d;
    rc = h->dtype == SDP_F32 ? launch_mlinterp<float>(a, 0) : launch_mlinterp<double>(a, 0);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(host_out, h->out.p, ob, hipMemcpyDeviceToHost));
    return SDP_OK;
}

// ---------------------------------------------------------------------------
// built-in kernels: relative-DP shift (stodynprog.py:523-525, 760-762)
// ---------------------------------------------------------------------------
template <typename real>
__global__ void k_pick_ref(const real *__restrict__ J, int64_t ref, double *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (double)J[ref];
}

template <typename real>
__global__ void __launch_bounds__(256) k_shift(real *__restrict__ J, int64_t n,
                                               const double *__restrict__ ref)
{
    const real r = (real)*ref;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        J[i] = J[i] - r;
}

// ---------------------------------------------------------------------------
// built-in kernels: tabulated backup (stodynprog.py:677-690)
//   k_tab_cells : one lane per lattice cell, coalesced x_next / g reads,
//                 jc = g + interp(V, x_next)
//   k_tab_reduce: one wavefront per node, lanes stride over controls, in-lane
//                 sequential expectation over w, butterfly first-index argmin
// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256) k_tab_cells(SdpTabArgs a, double *__restrict__ jc)
{
    SdpGrid<double, D> grid;
    double smin[D], smax[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { smin[k] = a.smin[k]; smax[k] = a.smax[k]; }
    sdp_make_grid<double, D>(grid, a.orders, smin, smax);
    const double *__restrict__ xn = (const double *)a.x_next;
    const double *__restrict__ g = (const double *)a.g;
    const double *__restrict__ V = (const double *)a.V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_cells; i += stride) {
        double pt[D];
#pragma unroll
        for (int k = 0; k < D; ++k) pt[k] = xn[k * a.n_cells + i];
        jc[i] = g[i] + sdp_interp_point<double, D>(V, grid, pt);
    }
}

__global__ void __launch_bounds__(256) k_tab_reduce(SdpTabArgs a, const double *__restrict__ jc,
                                                    int64_t *__restrict__ idx_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const double *__restrict__ proba = (const double *)a.proba;
    const int W = a.W > 0 ? a.W : 1;
    for (int64_t n = wave; n < a.n_nodes; n += n_waves) {
        const int64_t off = a.cell_off[n];
        const int U = (int)((a.cell_off[n + 1] - off) / W);
        double best = INFINITY;
        int ibest = INT_MAX;
        for (int c = lane; c < U; c += 64) {
            double acc;
            if (a.W > 0) {
                acc = 0.0;
                for (int w = 0; w < W; ++w) acc = acc + jc[off + (int64_t)c * W + w] * proba[w];
            } else {
                acc = jc[off + c];
            }
            if (ibest == INT_MAX || sdp_better_seq(acc, best)) { best = acc; ibest = c; }
        }
        sdp_seg_argmin<double, 64>(best, ibest);
        if (lane == 0) { ((double *)a.J)[n] = best; idx_out[n] = ibest; }
    }
}

struct sdp_tab {
    int d = 0;
    int64_t S = 0;
    int32_t orders[SDP_MAXD] = {0, 0, 0, 0};
    double smin[SDP_MAXD], smax[SDP_MAXD];
    DevBuf V;
    // per-call buffers, kept and grown: x_next, g, cell_off, proba, jc, J, idx
    DevBuf buf[7];
    size_t cap[7] = {0, 0, 0, 0, 0, 0, 0};
    int reserve(int k, size_t bytes)
    {
        if (cap[k] >= bytes) return SDP_OK;
        int rc = buf[k].alloc(bytes + bytes / 4);
        cap[k] = rc ? 0 : bytes + bytes / 4;
        return rc;
    }
    int put(int k, const void *host, size_t bytes)
    {
        int rc = reserve(k, bytes);
        if (rc) return rc;
        if (bytes) HIP_TRY(hipMemcpy(buf[k].p, host, bytes, hipMemcpyHostToDevice));
        return SDP_OK;
    }
};

extern "C" int sdp_tab_create(int d, const double *smin, const double *smax,
                              const int64_t *orders, const double *host_V, sdp_tab **out)
{
    if (d < 1 || d > SDP_MAXD) return fail(SDP_EDIM, "Can't interpolate in dimension strictly greater than 5");
    if (!smin || !smax || !orders || !host_V || !out) return fail(SDP_EINVAL, "NULL argument");
    std::unique_ptr<sdp_tab> t(new sdp_tab());
    t->d = d;
    int64_t S = 1;
    for (int k = 0; k < d; ++k) {
        if (orders[k] < 2) return fail(SDP_EINVAL, "state axis %d has %lld points: at least 2 needed", k, (long long)orders[k]);
        t->orders[k] = (int32_t)orders[k]; t->smin[k] = smin[k]; t->smax[k] = smax[k];
        S *= orders[k];
        if (S >= (int64_t)1 << 31) return fail(SDP_EINVAL, "state grid too large");
    }
    t->S = S;
    int rc = upload(t->V, host_V, S * 8);
    if (rc) return rc;
    *out = t.release();
    return SDP_OK;
}

extern "C" int sdp_tab_destroy(sdp_tab *t)
{
    delete t;
    return SDP_OK;
}

extern "C" int sdp_tab_backup(sdp_tab *t, int64_t n_nodes, const int64_t *cell_off, int64_t W,
                              const double *proba, const double *x_next, const double *g,
                              double *J_out, int64_t *idx_out)
{
    if (!t || !cell_off || !J_out || !idx_out) return fail(SDP_EINVAL, "NULL argument");
    if (n_nodes < 0 || W < 0) return fail(SDP_EINVAL, "negative size");
    if (n_nodes == 0) return SDP_OK;
    const int64_t n_cells = cell_off[n_nodes];
    if (n_cells <= 0 || !x_next || !g) return fail(SDP_EINVAL, "empty lattice");
    if (W > 0 && !proba) return fail(SDP_EINVAL, "perturbation weights missing");
    // the handle keeps its device buffers from call to call (a caller that backs nodes
    // up one at a time -- DPSolver._value_at_state_vect -- allocates nothing here)
    DevBuf &dx = t->buf[0], &dg = t->buf[1], &doff = t->buf[2], &dp = t->buf[3], &djc = t->buf[4],
           &dJ = t->buf[5], &didx = t->buf[6];
    int rc;
    if ((rc = t->put(0, x_next, (size_t)t->d * n_cells * 8))) return rc;
    if ((rc = t->put(1, g, (size_t)n_cells * 8))) return rc;
    if ((rc = t->put(2, cell_off, (size_t)(n_nodes + 1) * 8))) return rc;
    if (W > 0 && (rc = t->put(3, proba, (size_t)W * 8))) return rc;
    if ((rc = t->reserve(4, (size_t)n_cells * 8))) return rc;
    if ((rc = t->reserve(5, (size_t)n_nodes * 8))) return rc;
    if ((rc = t->reserve(6, (size_t)n_nodes * 8))) return rc;
    SdpTabArgs a;
    memset(&a, 0, sizeof(a));
    a.V = t->V.p; a.x_next = dx.p; a.g = dg.p; a.cell_off = (const int64_t *)doff.p;
    a.proba = dp.p; a.J = dJ.p; a.n_nodes = n_nodes; a.n_cells = n_cells;
    a.W = (int32_t)W; a.d = t->d;
    for (int k = 0; k < t->d; ++k) { a.orders[k] = t->orders[k]; a.smin[k] = t->smin[k]; a.smax[k] = t->smax[k]; }
    const int cus = device_cus();
    int64_t blocks = (n_cells + 255) / 256;
    if (blocks > (int64_t)cus * 16) blocks = (int64_t)cus * 16;
    dim3 gc((unsigned)blocks), b(256);
    switch (t->d) {
    case 1: hipLaunchKernelGGL(k_tab_cells<1>, gc, b, 0, 0, a, (double *)djc.p); break;
    case 2: hipLaunchKernelGGL(k_tab_cells<2>, gc, b, 0, 0, a, (double *)djc.p); break;
    case 3: hipLaunchKernelGGL(k_tab_cells<3>, gc, b, 0, 0, a, (double *)djc.p); break;
    default: hipLaunchKernelGGL(k_tab_cells<4>, gc, b, 0, 0, a, (double *)djc.p); break;
    }
    HIP_TRY(hipGetLastError());
    int64_t rblocks = (n_nodes + 3) / 4;
    if (rblocks > (int64_t)cus * 8) rblocks = (int64_t)cus * 8;
    hipLaunchKernelGGL(k_tab_reduce, dim3((unsigned)rblocks), b, 0, 0, a, (const double *)djc.p, (int64_t *)didx.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(J_out, dJ.p, (size_t)n_nodes * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(idx_out, didx.p, (size_t)n_nodes * 8, hipMemcpyDeviceToHost));
    return SDP_OK;
}

// ---------------------------------------------------------------------------
// built-in kernel: layout conversion at the API boundary.  Host arrays are
// always in the reference's C order (last state axis fastest); a handle in
// SDP_LAYOUT_COLUMNS keeps its per-node arrays with axis 0 fastest.  Both
// directions are the transpose of an [R][C] matrix of elements of WORDS 32-bit
// words, done through a 32x33 LDS tile so reads and writes are both coalesced.
// ---------------------------------------------------------------------------
template <int WORDS>
__global__ void __launch_bounds__(256) k_transpose(const uint32_t *__restrict__ in,
                                                   uint32_t *__restrict__ out, int64_t R, int64_t C)
{
    __shared__ uint32_t tile[WORDS][32][33];
    const int64_t tiles_c = (C + 31) / 32, tiles_r = (R + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads
    for (int64_t t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x) {
        const int64_t r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int64_t r = r0 + j, c = c0 + tx;
            if (r < R && c < C)
#pragma unroll
                for (int k = 0; k < WORDS; ++k) tile[k][j][tx] = in[(r * C + c) * WORDS + k];
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int64_t c = c0 + j, r = r0 + tx;
            if (r < R && c < C)
#pragma unroll
                for (int k = 0; k < WORDS; ++k) out[(c * R + r) * WORDS + k] = tile[k][tx][j];
        }
    }
}

static int launch_transpose(const void *in, void *out, int64_t R, int64_t C, int words,
                            hipStream_t stream)
{
    if (R == 0 || C == 0) return SDP_OK;
    int64_t tiles = ((R + 31) / 32) * ((C + 31) / 32);
    unsigned blocks = (unsigned)(tiles < 65536 ? tiles : 65536);
    const uint32_t *i = (const uint32_t *)in;
    uint32_t *o = (uint32_t *)out;
    switch (words) {
    case 1: hipLaunchKernelGGL(k_transpose<1>, dim3(blocks), dim3(256), 0, stream, i, o, R, C); break;
    case 2: hipLaunchKernelGGL(k_transpose<2>, dim3(blocks), dim3(256), 0, stream, i, o, R, C); break;
    case 3: hipLaunchKernelGGL(k_transpose<3>, dim3(blocks), dim3(256), 0, stream, i, o, R, C); break;
    case 4: hipLaunchKernelGGL(k_transpose<4>, dim3(blocks), dim3(256), 0, stream, i, o, R, C); break;
    case 6: hipLaunchKernelGGL(k_transpose<6>, dim3(blocks), dim3(256), 0, stream, i, o, R, C); break;
    case 8: hipLaunchKernelGGL(k_transpose<8>, dim3(blocks), dim3(256), 0, stream, i, o, R, C); break;
    default: return fail(SDP_EINVAL, "unsupported element size for the layout conversion");
    }
    HIP_TRY(hipGetLastError());
    return SDP_OK;
}

// ---------------------------------------------------------------------------
// RCCL, loaded lazily so single-GPU use never needs librccl
// ---------------------------------------------------------------------------
typedef struct { char internal[128]; } nccl_uid;
typedef void *nccl_comm_t;
enum { NCCL_INT8 = 0, NCCL_INT32 = 2, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_MAX = 2 };

struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    // point-to-point, grouped: the sparse exchange that needs nothing but the collective library (sendrecv_phase)
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)(void) = nullptr;
    int (*GroupEnd)(void) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static std::string g_rccl_name;

extern "C" const char *sdp_comm_library(void) { return g_rccl_name.c_str(); }
extern "C" int sdp_test_hooks(void)
{
#ifdef SDP_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

static int rccl_load()
{
    if (g_rccl.h) return SDP_OK;
#ifdef SDP_TEST_HOOKS
    // TEST BUILD ONLY (-DSDP_TEST_HOOKS, never the product library): SDP_RCCL_LIBRARY names another
    // library with the same nccl* entry points (the tests load a host-staged stand-in to run
    // several ranks on the one GPU of the test box)
    const char *override_path = getenv("SDP_RCCL_LIBRARY");
#else
    const char *override_path = nullptr;
#endif
    const char *names[] = {override_path && *override_path ? override_path : "librccl.so",
                           "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) g_rccl_name = n;
        if (h || (override_path && *override_path)) break;     // an explicit path must load
    }
    if (!h) return fail(SDP_ECOMM, "cannot load librccl.so: %s", dlerror());
#define SYM(field, name)                                                         \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                  \
    if (!g_rccl.field) return fail(SDP_ECOMM, "librccl: missing symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.h = h;
    return SDP_OK;
}

#define NCCL_TRY(expr)                                                                 \
    do {                                                                               \
        int r__ = (expr);                                                              \
        if (r__ != 0) return fail(SDP_ECOMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r__)); \
    } while (0)

struct sdp_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, nranks = 1;
    hipStream_t stream = nullptr;
    double *d_scalar = nullptr;
};

extern "C" int sdp_comm_unique_id(char id[128])
{
    int rc = rccl_load();
    if (rc) return rc;
    nccl_uid uid;
    NCCL_TRY(g_rccl.GetUniqueId(&uid));
    memcpy(id, uid.internal, 128);
    return SDP_OK;
}

extern "C" int sdp_comm_create(int rank, int nranks, const char id[128], sdp_comm **out)
{
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(SDP_EINVAL, "bad communicator arguments");
    int rc = rccl_load();
    if (rc) return rc;
    sdp_comm *c = new sdp_comm();
    c->rank = rank; c->nranks = nranks;
    nccl_uid uid;
    memcpy(uid.internal, id, 128);
    int r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
    if (r != 0) { delete c; return fail(SDP_ECOMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // numerically lowest = highest priority
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipMalloc((void **)&c->d_scalar, 8) != hipSuccess) {
        delete c; return fail(SDP_EHIP, "communicator stream/scratch allocation failed");
    }
    *out = c;
    return SDP_OK;
}

extern "C" int sdp_comm_destroy(sdp_comm *c)
{
    if (!c) return SDP_OK;
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    delete c;
    return SDP_OK;
}

extern "C" int sdp_comm_allreduce_max(sdp_comm *c, double *inout)
{
    if (!c || !inout) return fail(SDP_EINVAL, "NULL argument");
    HIP_TRY(hipMemcpyAsync(c->d_scalar, inout, 8, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(g_rccl.AllReduce(c->d_scalar, c->d_scalar, 1, NCCL_FLOAT64, NCCL_MAX, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(inout, c->d_scalar, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDP_OK;
}

extern "C" int sdp_comm_barrier(sdp_comm *c)
{
    double v = 0;
    return sdp_comm_allreduce_max(c, &v);
}

// ---------------------------------------------------------------------------
// problem handle
// ---------------------------------------------------------------------------
struct sdp_problem {
    int dtype = SDP_F64, d = 0, nu = 0, W = 0, lanes = 64, box_per_node = 0, layout = 0, variant = 0;
    int stg_threads = 0, col_seg = 0;
    int64_t stg_tiles = 0;
    int64_t S = 0, node_begin = 0, node_end = 0;
    int32_t orders[SDP_MAXD] = {0, 0, 0, 0};
    int32_t axis_off[SDP_MAXD] = {0, 0, 0, 0};
    DevBuf axes, wgrid, proba, box_lo, box_hi, box_n, V, J, pol, idx, pol_in, refs, scratch, stamps, claim, tail;
    DevBuf gstage;                         // gather buffer of phases with uneven parts (gather_phase_of)
    size_t gstage_bytes = 0;
    DevBuf stage[3];                       // layout-conversion buffers of the fused host call (J, pol, idx)
    size_t stage_bytes[3] = {0, 0, 0};
    // peer-write exchange (sdp_problem_enable_peer_exchange): the other ranks' value / J buffers
    // mapped into this process, one copy stream per peer
    bool peer_exchange = false;
    std::vector<void *> peer_V, peer_J;    // [nranks] (own entry: the local buffer)
    std::vector<hipStream_t> peer_stream;  // [nranks] (own entry unused)
    std::vector<hipEvent_t> peer_done;     // [nranks]
    int *d_flag = nullptr;                 // word for the rendezvous all-reduces
    hipEvent_t ev_enter = nullptr, ev_fence = nullptr;
    bool peer_fence = true;                // next backup is the first of an API call: see open_pushes
    hipStream_t copy_stream = nullptr;     // downloads of finished phases, under the next phase's kernel
    hipEvent_t ev_host[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_copy = nullptr;
    int host_overlap = 1;                  // sdp_problem_set_host_overlap
    int64_t stamp_words = 0;
    size_t scratch_bytes = 0;
    hipModule_t mod = nullptr;
    hipFunction_t f_sweep = nullptr, f_evalpol = nullptr, f_simulate = nullptr;
    // several controlled state variables (csrc/sdp_lead_kernel.h): the kernel that reduces V over w, launched
    // before every sweep, and its outputs (A[S] and a copy of V, both plane-major; E[nodes per trailing block];
    // bits of max |V|)
    hipFunction_t f_lead_reduce = nullptr;
    DevBuf lead_a, lead_v, lead_e, lead_vmax;
    hipDeviceptr_t prm_dev = nullptr;     // `sdp_model_prm` of the code object (lifted model constants)
    size_t prm_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    double last_kernel_ms = 0;
    sdp_comm *comm = nullptr;
    std::vector<int64_t> parts;            // [n_phases][nranks+1] node bounds (device order)
    std::vector<hipEvent_t> ev_phase;
    hipEvent_t ev_comm = nullptr;
    int n_phases = 0;
    bool comm_pending = false;
    // sparse peer exchange (sdp_problem_set_peer_needs): need[q] = sorted, disjoint node ranges of J
    // that rank q reads in a backup; J / V are then complete on a rank only there
    std::vector<std::vector<std::pair<int64_t, int64_t>>> need;
    bool sparse = false, J_partial = false, V_partial = false;
    // direct exchange (sdp_problem_set_direct_exchange): the backup kernels themselves store J into the
    // peers' mapped buffers (SdpSweepArgs.peer_J); sparse: per column the ranks that read it
    bool direct = false, mask_valid = false, send_everything = false;
    bool sendrecv = false;                 // sparse exchange by grouped ncclSend / ncclRecv (no mapped buffers): sdp_problem_set_sendrecv_exchange
    DevBuf peer_mask;
    // reduced-array sweep (csrc/sdp_lead_kernel.h), sharded: rows of the first stock the controls of a node
    // reach on either side (a guess of the host: too small costs time, not correctness); < 0: reduce everything
    int64_t lead_halo = -1;
    // what the plane-major arrays currently hold: the value buffer they were reduced from, its generation
    // (bumped whenever the contents of V change), and the lead indices covered
    const void *red_V = nullptr;
    uint64_t red_gen = 0, V_gen = 1;
    int red_parity = 0;                    // line kernel: which of the two slots of lead_vmax the last reduction used
    int64_t red_begin = 0, red_end = 0;
    double red_t = 0;
    int cus = 256;
    int refs_cap = 0;
    int col_threads = 512;
    int col_occupancy = 8;                 // workgroups of sdp_sweep_col a CU holds at once
    int32_t meta[SDP_META_WORDS] = {0};    // `sdp_meta` of the code object
    int peer_me = -1;                      // this rank, as of sdp_problem_enable_peer_exchange
    void release_peers()
    {
        // (the rank is the one recorded when the mappings were made: the communicator is not
        // owned by the problem and may already be gone)
        const int me = peer_me;
        for (size_t r = 0; r < peer_V.size(); ++r) {
            if ((int)r == me) continue;
            ipc_close(peer_V[r]);
            ipc_close(peer_J[r]);
        }
        peer_V.clear(); peer_J.clear();
        for (auto &st : peer_stream) if (st) (void)hipStreamDestroy(st);
        for (auto &e : peer_done) if (e) (void)hipEventDestroy(e);
        peer_stream.clear(); peer_done.clear();
        if (d_flag) { (void)hipFree(d_flag); d_flag = nullptr; }
        if (ev_enter) { (void)hipEventDestroy(ev_enter); ev_enter = nullptr; }
        if (ev_fence) { (void)hipEventDestroy(ev_fence); ev_fence = nullptr; }
        peer_exchange = false;
        direct = false;
        peer_me = -1;
    }
    ~sdp_problem()
    {
        {
            const size_t bytes = (size_t)S * (dtype == SDP_F64 ? 8 : 4);
            park_exported(V, bytes);                       // (no-ops for buffers that never left this process)
            park_exported(J, bytes);
        }
        if (mod) (void)hipModuleUnload(mod);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev2) (void)hipEventDestroy(ev2);
        if (ev3) (void)hipEventDestroy(ev3);
        if (ev_comm) (void)hipEventDestroy(ev_comm);
        release_peers();
        if (ev_copy) (void)hipEventDestroy(ev_copy);
        for (auto &e : ev_host) if (e) (void)hipEventDestroy(e);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto &e : ev_phase) if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static int64_t lead_trailing_nodes(const sdp_problem *p);

extern "C" int sdp_problem_create(const sdp_problem_desc *desc, sdp_problem **out)
{
    if (!desc || !out) return fail(SDP_EINVAL, "NULL argument");
    if (desc->dtype != SDP_F64 && desc->dtype != SDP_F32) return fail(SDP_EINVAL, "dtype must be SDP_F64 or SDP_F32");
    if (desc->d < 1 || desc->d > SDP_MAXD)
        return fail(SDP_EDIM, "state dimension %d: multilinear interpolation supports 1..4", desc->d);
    if (desc->nu < 1 || desc->nu > SDP_MAXU) return fail(SDP_EINVAL, "number of controls %d outside 1..%d", desc->nu, SDP_MAXU);
    if (desc->W < 0) return fail(SDP_EINVAL, "negative perturbation count");
    if (!desc->module_path) return fail(SDP_EMODULE, "no model code object given");
    if (!desc->box_lo || !desc->box_hi || !desc->box_n) return fail(SDP_EINVAL, "control box arrays missing");
    sdp_problem *p = new sdp_problem();
    std::unique_ptr<sdp_problem> guard(p);
    p->dtype = desc->dtype; p->d = desc->d; p->nu = desc->nu; p->W = desc->W;
    p->lanes = desc->lanes_per_node; p->box_per_node = desc->box_per_node;
    if (p->lanes < 1 || p->lanes > 64 || (p->lanes & (p->lanes - 1))) return fail(SDP_EINVAL, "lanes_per_node must be a power of two in 1..64");
    const size_t rs = real_size(p->dtype);
    int64_t S = 1, total = 0;
    for (int k = 0; k < p->d; ++k) {
        if (desc->orders[k] < 2) return fail(SDP_EINVAL, "state axis %d has %lld points: at least 2 needed to interpolate", k, (long long)desc->orders[k]);
        if (!desc->axes[k]) return fail(SDP_EINVAL, "state axis %d missing", k);
        p->orders[k] = (int32_t)desc->orders[k];
        p->axis_off[k] = (int32_t)total;
        total += desc->orders[k];
        S *= desc->orders[k];
        if (S >= (int64_t)1 << 31) return fail(SDP_EINVAL, "state grid too large: 32-bit vertex indices (pyx:164-165)");
    }
    p->S = S;
    p->node_begin = desc->node_begin;
    p->node_end = desc->node_end;
    if (p->node_begin < 0 || p->node_end > S || p->node_begin > p->node_end) return fail(SDP_EINVAL, "node slab [%lld,%lld) outside [0,%lld)", (long long)p->node_begin, (long long)p->node_end, (long long)S);
    p->cus = device_cus();

    // concatenated axes
    std::vector<char> ax(total * rs);
    for (int k = 0; k < p->d; ++k)
        memcpy(ax.data() + (size_t)p->axis_off[k] * rs, desc->axes[k], (size_t)desc->orders[k] * rs);
    int rc;
    if ((rc = upload(p->axes, ax.data(), ax.size()))) return rc;
    if (p->W > 0) {
        if (!desc->wgrid || !desc->proba) return fail(SDP_EINVAL, "perturbation grid/weights missing");
        if ((rc = upload(p->wgrid, desc->wgrid, p->W * rs))) return rc;
        if ((rc = upload(p->proba, desc->proba, p->W * rs))) return rc;
    }
    const size_t nbox = (size_t)p->nu * (p->box_per_node ? (size_t)S : 1);
    if (p->box_per_node && desc->layout == SDP_LAYOUT_COLUMNS) {
        // [nu][N0][P] (reference order) -> [nu][P][N0] (axis 0 fastest)
        const int64_t n0 = desc->orders[0], P = S / n0;
        auto reorder = [&](const void *src, size_t es, DevBuf &dst) -> int {
            std::vector<char> tmp(nbox * es);
            for (int c = 0; c < p->nu; ++c)
                for (int64_t i = 0; i < n0; ++i)
                    for (int64_t q = 0; q < P; ++q)
                        memcpy(tmp.data() + ((size_t)c * S + q * n0 + i) * es,
                               (const char *)src + ((size_t)c * S + i * P + q) * es, es);
            return upload(dst, tmp.data(), tmp.size());
        };
        if ((rc = reorder(desc->box_lo, rs, p->box_lo))) return rc;
        if ((rc = reorder(desc->box_hi, rs, p->box_hi))) return rc;
        if ((rc = reorder(desc->box_n, 4, p->box_n))) return rc;
    } else {
        if ((rc = upload(p->box_lo, desc->box_lo, nbox * rs))) return rc;
        if ((rc = upload(p->box_hi, desc->box_hi, nbox * rs))) return rc;
        if ((rc = upload(p->box_n, desc->box_n, nbox * 4))) return rc;
    }
    if ((rc = p->V.alloc_exportable(S * rs))) return rc;       // (a parked, once-exported buffer of this size if there is one)
    if ((rc = p->J.alloc_exportable(S * rs))) return rc;
    if ((rc = p->pol.alloc((size_t)S * p->nu * rs))) return rc;
    if ((rc = p->idx.alloc(S * 4))) return rc;
    HIP_TRY(hipMemset(p->V.p, 0, S * rs));
    HIP_TRY(hipMemset(p->J.p, 0, S * rs));
    HIP_TRY(hipMemset(p->pol.p, 0, (size_t)S * p->nu * rs));
    HIP_TRY(hipMemset(p->idx.p, 0, S * 4));

    hipError_t e = hipModuleLoad(&p->mod, desc->module_path);
    if (e != hipSuccess) { p->mod = nullptr; return fail(SDP_EMODULE, "hipModuleLoad(%s): %s", desc->module_path, hipGetErrorString(e)); }
    p->layout = desc->layout;
    if (p->layout != SDP_LAYOUT_NODES && p->layout != SDP_LAYOUT_COLUMNS) return fail(SDP_EINVAL, "unknown layout %d", p->layout);
    if (p->layout == SDP_LAYOUT_COLUMNS) {
        if (p->d < 2) return fail(SDP_EINVAL, "column layout needs at least two state axes");
        if (p->node_begin % p->orders[0] || p->node_end % p->orders[0])
            return fail(SDP_EINVAL, "column layout: the node slab must consist of whole columns");
    }
    p->variant = desc->variant;
    p->col_seg = desc->col_seg_nodes > 0 ? desc->col_seg_nodes : 0;
    if (p->variant != SDP_VARIANT_DIRECT && p->variant != SDP_VARIANT_STAGED) return fail(SDP_EINVAL, "unknown kernel variant %d", p->variant);
    if (p->variant == SDP_VARIANT_STAGED) {
        if (p->layout != SDP_LAYOUT_NODES) return fail(SDP_EINVAL, "the staged kernel works on the node layout");
        p->stg_tiles = 1;
        for (int k = 0; k < p->d; ++k) {
            if (desc->tile[k] < 1) return fail(SDP_EINVAL, "staged kernel: tile[%d] = %d", k, (int)desc->tile[k]);
            p->stg_tiles *= (p->orders[k] + desc->tile[k] - 1) / desc->tile[k];
        }
    }
    // what the code object was generated for must be THIS problem: the kernels carry table
    // sizes, dimensions and the real type as compile-time constants
    {
        hipDeviceptr_t mptr = nullptr;
        size_t mbytes = 0;
        if (hipModuleGetGlobal(&mptr, &mbytes, p->mod, "sdp_meta") != hipSuccess || mbytes != sizeof(p->meta)) {
            (void)hipGetLastError();
            return fail(SDP_EMODULE, "code object %s does not declare `sdp_meta` (%d words, sdp_kernel_args.h): "
                        "cannot check that it was built for this problem", desc->module_path, SDP_META_WORDS);
        }
        HIP_TRY(hipMemcpyDtoH(p->meta, mptr, sizeof(p->meta)));
        const int32_t *m = p->meta;
        const int64_t controls = [&] {
            if (p->box_per_node) return (int64_t)-1;
            int64_t t = 1;
            for (int c = 0; c < p->nu; ++c) t *= ((const int32_t *)desc->box_n)[c];
            return t;
        }();
        const char *what = nullptr;
        char detail[160] = "";
#define META_CHECK(cond, ...) if (!what && !(cond)) { what = #cond; snprintf(detail, sizeof(detail), __VA_ARGS__); }
        META_CHECK(m[SDP_META_MAGIC_AT] == SDP_META_MAGIC, "bad magic 0x%x", (unsigned)m[SDP_META_MAGIC_AT]);
        META_CHECK(m[SDP_META_REAL_BYTES] == (int)rs, "built for %d-byte reals, the problem has %d-byte reals", m[SDP_META_REAL_BYTES], (int)rs);
        META_CHECK(m[SDP_META_D] == p->d, "built for %d state variables, the problem has %d", m[SDP_META_D], p->d);
        META_CHECK(m[SDP_META_NU] == p->nu, "built for %d controls, the problem has %d", m[SDP_META_NU], p->nu);
        META_CHECK((m[SDP_META_HAS_W] != 0) == (p->W > 0), "built %s a perturbation, the problem has W = %d", m[SDP_META_HAS_W] ? "with" : "without", p->W);
        // (every unit also carries the node-order kernels sdp_sweep / sdp_evalpol: a column or staged unit
        // may serve a node-layout, direct-variant problem; the reverse is what must be refused)
        META_CHECK(p->layout != SDP_LAYOUT_COLUMNS || m[SDP_META_LAYOUT] == SDP_LAYOUT_COLUMNS, "the column layout was asked of a code object without column kernels");
        META_CHECK(p->variant != SDP_VARIANT_STAGED || (m[SDP_META_FLAGS] & SDP_META_F_STAGED), "the staged-tile variant was asked of a code object without sdp_sweep_lds");
        if (p->layout == SDP_LAYOUT_COLUMNS) {
            META_CHECK(m[SDP_META_COL_N0] == p->orders[0], "column table built for %d points along axis 0, the grid has %d", m[SDP_META_COL_N0], p->orders[0]);
            META_CHECK(m[SDP_META_COL_W] == (p->W > 0 ? p->W : 1), "column table built for %d perturbation points, the problem has %d", m[SDP_META_COL_W], p->W);
            META_CHECK(((m[SDP_META_FLAGS] & SDP_META_F_WINDOW) != 0) == (p->col_seg > 0) || (m[SDP_META_FLAGS] & SDP_META_F_TRAIL_HAS_U), "row-window kernel %s but col_seg_nodes = %d", (m[SDP_META_FLAGS] & SDP_META_F_WINDOW) ? "present" : "absent", p->col_seg);
            META_CHECK(m[SDP_META_UTAB] == 0 || (controls >= 1 && controls <= m[SDP_META_UTAB_N]), "control table built for one lattice of at most %d controls at every node, the problem has %s%lld", m[SDP_META_UTAB_N], controls < 0 ? "per-node boxes " : "", (long long)controls);
        }
#undef META_CHECK
        if (what)
            return fail(SDP_EMODULE, "code object %s was not built for this problem: %s", desc->module_path, detail);
    }
    const char *k_sweep = p->layout == SDP_LAYOUT_COLUMNS ? "sdp_sweep_col"
                          : (p->variant == SDP_VARIANT_STAGED ? "sdp_sweep_lds" : "sdp_sweep");
    const char *k_eval = p->layout == SDP_LAYOUT_COLUMNS ? "sdp_evalpol_col" : "sdp_evalpol";
    e = hipModuleGetFunction(&p->f_sweep, p->mod, k_sweep);
    if (e != hipSuccess) return fail(SDP_EMODULE, "code object %s has no %s kernel: %s", desc->module_path, k_sweep, hipGetErrorString(e));
    if (p->layout == SDP_LAYOUT_COLUMNS) {
        // the workgroup size the column kernels were compiled for (__launch_bounds__)
        int mt = 0;
        if (hipFuncGetAttribute(&mt, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, p->f_sweep) == hipSuccess && mt >= 64)
            p->col_threads = mt > 1024 ? 1024 : mt;
        {
            int occ = 0;
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, p->f_sweep, p->col_threads, 0) == hipSuccess && occ >= 1)
                p->col_occupancy = occ > 16 ? 16 : occ;
            else
                (void)hipGetLastError();
        }
        // unit counters of the filtered column kernel (SdpSweepArgs.claim): zero once, the kernel
        // leaves them zero
        int rc = p->claim.alloc(4 * (8 * 32 + 32));
        if (rc) return rc;
        HIP_TRY(hipMemset(p->claim.p, 0, 4 * (8 * 32 + 32)));
        // the tail of every workgroup's table (SdpSweepArgs.tail): such kernels claim their units, so their grid is
        // what the chip holds at once (column_grid)
        if (p->meta[SDP_META_TAIL_BYTES] > 0) {
            if (!(p->meta[SDP_META_FLAGS] & SDP_META_F_CLAIMS))
                return fail(SDP_EMODULE, "code object %s keeps the tail of its table in global memory but does not claim its units", desc->module_path);
            const size_t blocks = (((size_t)p->cus * (size_t)p->col_occupancy + 7) / 8) * 8;
            rc = p->tail.alloc(blocks * (size_t)p->meta[SDP_META_TAIL_BYTES]);
            if (rc) return rc;
        }
    }
    if (p->layout != SDP_LAYOUT_COLUMNS && p->variant != SDP_VARIANT_STAGED && (p->meta[SDP_META_FLAGS] & SDP_META_F_LEAD)) {
        e = hipModuleGetFunction(&p->f_lead_reduce, p->mod, "sdp_lead_reduce");
        if (e != hipSuccess) return fail(SDP_EMODULE, "code object %s has no sdp_lead_reduce kernel: %s", desc->module_path, hipGetErrorString(e));
        // (one state variable, sdp_line_kernel.h: the filter on the shifted lattice -- several lanes per node, and aux_a holds
        // (A', B', C') triplets for the positions of a lattice of 2 S + 64 rows at most; no copy of V)
        const bool line = p->d == 1 && (p->meta[SDP_META_FLAGS] & SDP_META_F_SHIFT);
        if (line && p->meta[SDP_META_COL_W] != p->W)
            return fail(SDP_EMODULE, "code object %s was built for %d perturbation points, the problem has %d", desc->module_path, p->meta[SDP_META_COL_W], p->W);
        if (!line && p->lanes != 1) return fail(SDP_EINVAL, "the reduced-array sweep takes one lane per node (lanes = %d)", p->lanes);
        int rc = p->lead_a.alloc(line ? (size_t)(2 * p->S + 64) * 24 : (size_t)p->S * 8);   // (the reduced array: 8-byte sums whatever the reals are, sdp_lead_kernel.h)
        if (!rc) rc = p->lead_v.alloc(line ? 8 : (size_t)p->S * rs);
        if (!rc) rc = p->lead_e.alloc((size_t)lead_trailing_nodes(p) * rs);            // (a value per trailing index)
        if (!rc) rc = p->lead_vmax.alloc(32);        // (the line kernel: two slots of two words, used in turn)
        if (!rc) HIP_TRY(hipMemset(p->lead_vmax.p, 0, 32));
        if (rc) return rc;
    }
    if (p->variant == SDP_VARIANT_STAGED) {
        int mt = 0;
        if (hipFuncGetAttribute(&mt, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, p->f_sweep) != hipSuccess || mt < 64)
            return fail(SDP_EMODULE, "cannot read the workgroup size of sdp_sweep_lds");
        p->stg_threads = mt;
        int64_t tile_nodes = 1;
        for (int k = 0; k < p->d; ++k) tile_nodes *= desc->tile[k];
        if (tile_nodes != mt) return fail(SDP_EINVAL, "staged kernel: tile of %lld nodes but workgroups of %d threads", (long long)tile_nodes, mt);
    }
    e = hipModuleGetFunction(&p->f_evalpol, p->mod, k_eval);
    if (e != hipSuccess) return fail(SDP_EMODULE, "code object %s has no %s kernel: %s", desc->module_path, k_eval, hipGetErrorString(e));
    if (hipModuleGetFunction(&p->f_simulate, p->mod, "sdp_simulate") != hipSuccess) {
        (void)hipGetLastError();
        p->f_simulate = nullptr;
    }
    // lifted model constants (codegen: `__constant__ sdp_real sdp_model_prm[]`), if any
    if (hipModuleGetGlobal(&p->prm_dev, &p->prm_bytes, p->mod, "sdp_model_prm") != hipSuccess) {
        (void)hipGetLastError();
        p->prm_dev = nullptr;
        p->prm_bytes = 0;
    }
    HIP_TRY(hipStreamCreate(&p->stream));
    HIP_TRY(hipEventCreate(&p->ev0));
    HIP_TRY(hipEventCreate(&p->ev1));
    HIP_TRY(hipEventCreate(&p->ev2));
    HIP_TRY(hipEventCreate(&p->ev3));
    guard.release();
    *out = p;
    return SDP_OK;
}

extern "C" int sdp_problem_set_params(sdp_problem *p, const void *values, int32_t n)
{
    if (!p || n < 0 || (n > 0 && !values)) return fail(SDP_EINVAL, "sdp_problem_set_params: bad arguments");
    const size_t bytes = (size_t)n * real_size(p->dtype);
    if (bytes != p->prm_bytes)
        return fail(SDP_EINVAL, "sdp_problem_set_params: the code object declares %zu parameter(s), %d given",
                    p->prm_bytes / real_size(p->dtype), (int)n);
    if (!bytes) return SDP_OK;
    // on the problem's stream: ordered after the launches that still read the old values
    HIP_TRY(hipMemcpyAsync(p->prm_dev, values, bytes, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));        // `values` may be reused by the caller
    p->red_V = nullptr;                               // (the reduced array was made with the old constants)
    return SDP_OK;
}

extern "C" int sdp_problem_destroy(sdp_problem *p)
{
    if (p) { (void)hipDeviceSynchronize(); delete p; }
    return SDP_OK;
}

static int ensure_scratch(sdp_problem *p, size_t bytes)
{
    if (p->scratch_bytes >= bytes) return SDP_OK;
    int rc = p->scratch.alloc(bytes);
    if (rc) { p->scratch_bytes = 0; return rc; }
    p->scratch_bytes = bytes;
    return SDP_OK;
}

// host (reference C order) -> device buffer in the handle's layout
static int upload_nodes(sdp_problem *p, void *dev, const void *host, size_t elem_bytes)
{
    const size_t bytes = (size_t)p->S * elem_bytes;
    if (p->layout != SDP_LAYOUT_COLUMNS) {
        HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, p->stream));
    } else {
        int rc = ensure_scratch(p, bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(p->scratch.p, host, bytes, hipMemcpyHostToDevice, p->stream));
        // [N0][P] -> [P][N0]
        if ((rc = launch_transpose(p->scratch.p, dev, p->orders[0], p->S / p->orders[0],
                                   (int)(elem_bytes / 4), p->stream))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    return SDP_OK;
}

// device buffer in the handle's layout -> host (reference C order)
static int download_nodes(sdp_problem *p, void *host, const void *dev, size_t elem_bytes)
{
    const size_t bytes = (size_t)p->S * elem_bytes;
    if (p->layout != SDP_LAYOUT_COLUMNS) {
        HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, p->stream));
    } else {
        int rc = ensure_scratch(p, bytes);
        if (rc) return rc;
        // [P][N0] -> [N0][P]
        if ((rc = launch_transpose(dev, p->scratch.p, p->S / p->orders[0], p->orders[0],
                                   (int)(elem_bytes / 4), p->stream))) return rc;
        HIP_TRY(hipMemcpyAsync(host, p->scratch.p, bytes, hipMemcpyDeviceToHost, p->stream));
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    return SDP_OK;
}

extern "C" int sdp_problem_set_value(sdp_problem *p, const void *host_V)
{
    if (!p || !host_V) return fail(SDP_EINVAL, "NULL argument");
    p->V_partial = false;
    ++p->V_gen;                              // (whatever was derived from the old contents is stale)
    return upload_nodes(p, p->V.p, host_V, real_size(p->dtype));
}

extern "C" int sdp_problem_set_policy(sdp_problem *p, const void *host_pol)
{
    if (!p || !host_pol) return fail(SDP_EINVAL, "NULL argument");
    const size_t bytes = (size_t)p->S * p->nu * real_size(p->dtype);
    int rc;
    if (!p->pol_in.p && (rc = p->pol_in.alloc(bytes))) return rc;
    return upload_nodes(p, p->pol_in.p, host_pol, (size_t)p->nu * real_size(p->dtype));
}

static void fill_args(const sdp_problem *p, SdpSweepArgs &a, double t_k, int64_t nb, int64_t ne)
{
    memset(&a, 0, sizeof(a));
    a.V = p->V.p; a.J = p->J.p; a.pol = p->pol.p; a.idx = (int32_t *)p->idx.p;
    a.axes = p->axes.p; a.wgrid = p->wgrid.p; a.proba = p->proba.p;
    a.box_lo = p->box_lo.p; a.box_hi = p->box_hi.p; a.box_n = (const int32_t *)p->box_n.p;
    a.pol_in = p->pol_in.p;
    a.node_begin = nb; a.node_end = ne; a.S = p->S; a.t_k = t_k;
    for (int k = 0; k < SDP_MAXD; ++k) { a.orders[k] = p->orders[k]; a.axis_off[k] = p->axis_off[k]; }
    a.W = p->W; a.box_per_node = p->box_per_node;
    a.shift_index = -1; a.ref_out = nullptr;
    a.stamps = (unsigned long long *)p->stamps.p;
    a.claim = (unsigned int *)p->claim.p;
    a.tail = p->tail.p;
    a.aux_a = p->lead_a.p; a.aux_v = p->lead_v.p; a.aux_e = p->lead_e.p; a.aux_vmax = (unsigned long long *)p->lead_vmax.p;
    a.aux_begin = p->red_begin; a.aux_end = p->red_end;
    if (p->direct && p->comm && p->comm->nranks > 1 && p->peer_exchange) {
        const int n = p->comm->nranks;
        a.n_peer = n;
        for (int q = 0; q < n && q < SDP_MAX_PEERS; ++q) a.peer_J[q] = q == p->comm->rank ? nullptr : p->peer_J[(size_t)q];
        a.peer_mask = p->send_everything ? nullptr : (const unsigned char *)(p->sparse ? p->peer_mask.p : nullptr);
    }
    if (p->layout == SDP_LAYOUT_COLUMNS) {
        a.n_lead = p->orders[0];
        a.col_begin = nb / p->orders[0];
        a.col_end = ne / p->orders[0];
        a.col_splits = 1;
    }
}

static int launch_module(hipFunction_t f, SdpSweepArgs &a, unsigned blocks, unsigned threads,
                         hipStream_t stream)
{
    size_t size = sizeof(a);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size,
                     HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(f, blocks, 1, 1, threads, 1, 1, 0, stream, nullptr, extra));
    return SDP_OK;
}

// Column kernels: a unit of work is (column, split).  Splitting a column over
// several workgroups repeats its table build, so split only as far as needed
// to give every CU a few workgroups; `min_nodes` nodes per split at least.
static unsigned column_grid(const sdp_problem *p, SdpSweepArgs &a, int min_nodes, bool sweep_kernel)
{
    const int64_t cols = a.col_end - a.col_begin;
    const int n0 = p->orders[0];
    int64_t want = (int64_t)p->cus * 4;
    int splits = (int)((want + cols - 1) / cols);
    int max_splits = n0 / (min_nodes > 0 ? min_nodes : 1);
    if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    // row window: a workgroup's table covers the rows reachable from col_seg nodes
    if (p->col_seg > 0 && splits < (n0 + p->col_seg - 1) / p->col_seg) splits = (n0 + p->col_seg - 1) / p->col_seg;
    a.col_splits = splits;
    int64_t units = cols * splits;
    // Single GPU: a bounded grid whose workgroups stride over the units.  With
    // a communicator: one workgroup per unit, so CUs are handed back all the
    // time and the RCCL kernels of the previous phase's all-gather (on their
    // own, higher-priority stream) get scheduled beside the sweep instead of
    // behind it.
    // (kernels whose workgroups CLAIM their units are persistent whatever the grid: a workgroup per
    // unit would only add thousands of workgroups that start, find nothing left and leave)
    // Their grid is what the chip holds at once (occupancy of the kernel: LDS, registers, wave slots).
    // (sdp_evalpol_col strides over the units: it takes the bounded grid of the other kernels)
    const bool claims = sweep_kernel && (p->meta[SDP_META_FLAGS] & SDP_META_F_CLAIMS) != 0;
    int64_t blocks = (p->comm && p->comm->nranks > 1 && !claims) ? units
                     : (int64_t)p->cus * (claims ? p->col_occupancy : 8);
    if (blocks > units) blocks = units;
    blocks = ((blocks + 7) / 8) * 8;
    if (blocks < 8) blocks = 8;
    return (unsigned)blocks;
}

// workgroups for the sweep: a multiple of 8 (one share per XCD), enough to
// fill every CU several times over, never more than there are node tiles
static unsigned sweep_blocks(const sdp_problem *p, int64_t nodes)
{
    // (with a communicator the bounded grid is kept: the generic kernel leaves
    // most of the LDS and half of the wave slots free for the RCCL kernels)
    // (the line kernel, one state variable on the shifted lattice: a workgroup's four waves share a tile of 64 / lanes nodes)
    const bool line = p->d == 1 && (p->meta[SDP_META_FLAGS] & SDP_META_F_LEAD) && (p->meta[SDP_META_FLAGS] & SDP_META_F_SHIFT);
    const int64_t tile = (64 / p->lanes) * (line ? 1 : 4);
    int64_t tiles = (nodes + tile - 1) / tile;
    int64_t blocks = (int64_t)p->cus * 8;
    if (blocks > tiles) blocks = tiles;
    blocks = ((blocks + 7) / 8) * 8;
    if (blocks < 8) blocks = 8;
    return (unsigned)blocks;
}

// reduced-array sweep: nodes per block of trailing coordinates, and the lead indices [lb, le) whose reduced
// values a launch over the nodes [nb, ne) may read
static int64_t lead_trailing_nodes(const sdp_problem *p)
{
    int64_t ts = 1;
    for (int k = p->meta[SDP_META_LEAD_AXES]; k < p->d; ++k) ts *= p->orders[(p->meta[SDP_META_LEAD_PERM] >> (4 * k)) & 15];
    return ts;
}
static bool lead_permuted(const sdp_problem *p)
{
    for (int k = 0; k < p->d; ++k)
        if (((p->meta[SDP_META_LEAD_PERM] >> (4 * k)) & 15) != k) return true;
    return false;
}
static void lead_reduce_range(const sdp_problem *p, int64_t nb, int64_t ne, int64_t &lb, int64_t &le)
{
    const int64_t ts = lead_trailing_nodes(p), ls = p->S / ts;
    lb = 0; le = ls;
    // (stocks not listed first: a slab of nodes is no range of lead indices -- everything is reduced)
    if (p->lead_halo < 0 || !p->comm || p->comm->nranks < 2 || lead_permuted(p)) return;
    // this rank's nodes over all phases of the backup (one contiguous slab with dist.slab_partition)
    int64_t lo = nb, hi = ne;
    const int n = p->comm->nranks, me = p->comm->rank;
    for (int ph = 0; ph < p->n_phases; ++ph) {
        const int64_t *b = p->parts.data() + (size_t)ph * (n + 1);
        if (b[me + 1] > b[me]) { lo = b[me] < lo ? b[me] : lo; hi = b[me + 1] > hi ? b[me + 1] : hi; }
    }
    const int64_t row = ls / p->orders[0];                  // lead indices per row of the first stock
    lb = lo / ts - (p->lead_halo + 1) * row;
    le = (hi + ts - 1) / ts + (p->lead_halo + 2) * row;
    if (lb < 0) lb = 0;
    if (le > ls) le = ls;
}

static int launch_sweep(sdp_problem *p, double t_k, int64_t nb, int64_t ne)
{
    if (ne <= nb) return SDP_OK;
    SdpSweepArgs a;
    fill_args(p, a, t_k, nb, ne);
    if (p->layout == SDP_LAYOUT_COLUMNS) {
        // 512-thread workgroups: 8 waves share one LDS table, one lane per node
        const unsigned threads = (unsigned)p->col_threads;   // SDP_COL_THREADS of the code object
        const unsigned blocks = column_grid(p, a, 64, true);
        return launch_module(p->f_sweep, a, blocks, threads, p->stream);
    }
    if (p->variant == SDP_VARIANT_STAGED) {
        // bounded grid of tile-walking workgroups (LDS admits two per CU; a few per slot
        // even out tiles of different cost), a multiple of 8 for the XCD-contiguous walk
        int64_t blocks = (int64_t)p->cus * 4;
        if (blocks > p->stg_tiles) blocks = p->stg_tiles;
        blocks = ((blocks + 7) / 8) * 8;
        return launch_module(p->f_sweep, a, (unsigned)blocks, (unsigned)p->stg_threads, p->stream);
    }
    if (p->f_lead_reduce) {
        // The array reduced over w (every node's first pass reads it at the nodes its controls reach): from
        // the whole of V on one GPU; sharded, from this rank's rows of the first stock plus `lead_halo` rows
        // on either side -- once per value array, however many launches (phases) the backup takes.
        if (p->V_partial) return fail(SDP_EINVAL, "the reduced-array sweep needs the whole cost-to-go array on this device");
        int64_t lb, le;
        lead_reduce_range(p, nb, ne, lb, le);
        if (!(p->red_V == p->V.p && p->red_gen == p->V_gen && p->red_t == t_k && p->red_begin <= lb && le <= p->red_end)) {
            const bool line = p->d == 1 && (p->meta[SDP_META_FLAGS] & SDP_META_F_SHIFT);
            // (the line kernel keeps its maximum in one of two slots and clears the other for the next reduction itself:
            // no fill launch between the kernels of a sweep that takes tens of microseconds)
            if (line) p->red_parity ^= 1;
            else HIP_TRY(hipMemsetAsync(p->lead_vmax.p, 0, 16, p->stream));
            const int64_t ts = lead_trailing_nodes(p);
            SdpSweepArgs r = a;
            if (line) {
                r.aux_vmax = (unsigned long long *)p->lead_vmax.p + 2 * p->red_parity;
                r.aux_e = (unsigned long long *)p->lead_vmax.p + 2 * (p->red_parity ^ 1);
            }
            r.n_peer = 0;
            r.node_begin = lb * ts; r.node_end = le * ts;
            r.aux_begin = lb; r.aux_end = le;
            int rc = launch_module(p->f_lead_reduce, r, sweep_blocks(p, r.node_end - r.node_begin), 256, p->stream);
            if (rc) return rc;
            p->red_V = p->V.p; p->red_gen = p->V_gen; p->red_t = t_k; p->red_begin = lb; p->red_end = le;
        }
        a.aux_begin = p->red_begin; a.aux_end = p->red_end;
        if (p->d == 1 && (p->meta[SDP_META_FLAGS] & SDP_META_F_SHIFT)) a.aux_vmax = (unsigned long long *)p->lead_vmax.p + 2 * p->red_parity;
    }
    return launch_module(p->f_sweep, a, sweep_blocks(p, ne - nb), 256, p->stream);
}

static int launch_evalpol(sdp_problem *p, double t_k, int64_t nb, int64_t ne,
                          int64_t shift_index, double *ref_out)
{
    if (ne <= nb) return SDP_OK;
    SdpSweepArgs a;
    fill_args(p, a, t_k, nb, ne);
    a.shift_index = shift_index;
    a.ref_out = ref_out;
    if (p->layout == SDP_LAYOUT_COLUMNS) {
        const unsigned blocks = column_grid(p, a, 64, false);      // one lane per node
        const int per_split = (p->orders[0] + a.col_splits - 1) / a.col_splits;
        unsigned threads = (unsigned)((per_split + 63) / 64) * 64;
        if (threads > (unsigned)p->col_threads) threads = (unsigned)p->col_threads;
        return launch_module(p->f_evalpol, a, blocks, threads, p->stream);
    }
    const int64_t nodes = ne - nb;
    int64_t blocks = (nodes + 255) / 256;
    if (blocks > (int64_t)p->cus * 16) blocks = (int64_t)p->cus * 16;
    return launch_module(p->f_evalpol, a, (unsigned)blocks, 256, p->stream);
}

// Exchange of one phase of a per-node array (in place) on the communicator's
// stream: rank r owns nodes [b[r], b[r+1]) of the phase, elem_bytes bytes per
// node.  Equal parts: one in-place ncclAllGather.  Uneven parts (a column count
// that is no multiple of ranks x phases -- Searev's 61 x 61 columns on 8 GPUs --, tapered
// plans): every part padded to the longest one in a gather buffer, ONE
// ncclAllGather there, and the parts copied to their places ("pad, or
// use grouped ncclBroadcast" -- round 4 used the grouped broadcasts, one per rank
// and phase; with 8 ranks a tapered plan stalled in 3 of 12 full-size runs under the
// stand-in, cause not found: the path is gone).  Moved as raw bytes, so it serves J,
// the policy and the index.
static int gather_phase_of(sdp_problem *p, int phase, void *buffer, size_t elem_bytes)
{
    const int n = p->comm->nranks, me = p->comm->rank;
    const int64_t *b = p->parts.data() + (size_t)phase * (n + 1);
    bool even = true;
    const int64_t len0 = b[1] - b[0];
    int64_t longest = 0;
    for (int r = 0; r < n; ++r) {
        even = even && (b[r + 1] - b[r] == len0);
        longest = std::max(longest, b[r + 1] - b[r]);
    }
    if (b[n] == b[0]) return SDP_OK;
    char *base = (char *)buffer;
    hipStream_t cs = p->comm->stream;
    if (even) {
        NCCL_TRY(g_rccl.AllGather(base + b[me] * elem_bytes, base + b[0] * elem_bytes,
                                  (size_t)len0 * elem_bytes, NCCL_INT8, p->comm->comm, cs));
        return SDP_OK;
    }
    const size_t slot = (size_t)longest * elem_bytes;
    if (p->gstage_bytes < slot * (size_t)n) {
        // (sdp_problem_attach_comm sizes the buffer for the J exchange of every phase; only the policy
        // gather -- on request, nothing in flight -- can come here.  hipFree waits for the device.)
        int rc = p->gstage.alloc(slot * (size_t)n);
        if (rc) { p->gstage_bytes = 0; return rc; }
        p->gstage_bytes = slot * (size_t)n;
    }
    char *g = (char *)p->gstage.p;
    const size_t mine = (size_t)(b[me + 1] - b[me]) * elem_bytes;
    if (mine) HIP_TRY(hipMemcpyAsync(g + (size_t)me * slot, base + b[me] * elem_bytes, mine, hipMemcpyDeviceToDevice, cs));
    NCCL_TRY(g_rccl.AllGather(g + (size_t)me * slot, g, slot, NCCL_INT8, p->comm->comm, cs));
    for (int r = 0; r < n; ++r) {
        const size_t cnt = (size_t)(b[r + 1] - b[r]) * elem_bytes;
        if (r == me || cnt == 0) continue;
        HIP_TRY(hipMemcpyAsync(base + b[r] * elem_bytes, g + (size_t)r * slot, cnt, hipMemcpyDeviceToDevice, cs));
    }
    return SDP_OK;
}

// Peer-write form of the exchange of one phase of J: this rank's rows of the phase go straight
// into every other rank's J buffer (mapped through HIP IPC) as device-to-device copies on one
// stream per peer -- copy engines over xGMI, no compute units, all links at once -- ordered
// behind the phase's kernel by `after`.
static int push_phase(sdp_problem *p, int phase, hipEvent_t after, bool everything)
{
    const int n = p->comm->nranks, me = p->comm->rank;
    const int64_t *b = p->parts.data() + (size_t)phase * (n + 1);
    const size_t rs = real_size(p->dtype);
    const int64_t lo = b[me], hi = b[me + 1];
    if (hi <= lo) return SDP_OK;
    for (int k = 1; k < n; ++k) {
        const int q = (me + k) % n;                       // start with a different peer on every rank
        if (!p->sparse || everything) {
            HIP_TRY(hipStreamWaitEvent(p->peer_stream[q], after, 0));
            HIP_TRY(hipMemcpyAsync((char *)p->peer_J[q] + (size_t)lo * rs, (const char *)p->J.p + (size_t)lo * rs,
                                   (size_t)(hi - lo) * rs, hipMemcpyDeviceToDevice, p->peer_stream[q]));
            continue;
        }
        // only the rows the peer reads (sdp_problem_set_peer_needs)
        const auto &iv = p->need[q];
        auto it = std::lower_bound(iv.begin(), iv.end(), lo,
                                   [](const std::pair<int64_t, int64_t> &r, int64_t v) { return r.second <= v; });
        bool waited = false;
        for (; it != iv.end() && it->first < hi; ++it) {
            const int64_t s0 = it->first > lo ? it->first : lo, s1 = it->second < hi ? it->second : hi;
            if (s1 <= s0) continue;
            if (!waited) { HIP_TRY(hipStreamWaitEvent(p->peer_stream[q], after, 0)); waited = true; }
            HIP_TRY(hipMemcpyAsync((char *)p->peer_J[q] + (size_t)s0 * rs, (const char *)p->J.p + (size_t)s0 * rs,
                                   (size_t)(s1 - s0) * rs, hipMemcpyDeviceToDevice, p->peer_stream[q]));
        }
    }
    return SDP_OK;
}

// Start of the first peer-write backup of an API call.  Writes into a peer's J are one-sided:
// nothing in them waits for the peer, which may still be reading that buffer on behalf of its
// PREVIOUS call (relative-DP shift, download to the host) when a faster rank starts the next
// one.  So the first pushes of a call wait behind a tiny all-reduce that every rank joins only
// once its previous call has returned.  It runs on the communicator stream under the first
// phase's kernel.  Inside a chain of backups the end-of-backup rendezvous is enough (pushes of
// step k+1 go to the buffer step k only read, and every rank's step-k kernels precede it).
static int open_pushes(sdp_problem *p)
{
    const int n = p->comm->nranks, me = p->comm->rank;
    hipStream_t cs = p->comm->stream;
    HIP_TRY(hipEventRecord(p->ev_enter, p->stream));
    HIP_TRY(hipStreamWaitEvent(cs, p->ev_enter, 0));
    NCCL_TRY(g_rccl.AllReduce(p->d_flag, p->d_flag, 1, NCCL_INT32, NCCL_MAX, p->comm->comm, cs));
    HIP_TRY(hipEventRecord(p->ev_fence, cs));
    for (int q = 0; q < n; ++q)
        if (q != me) HIP_TRY(hipStreamWaitEvent(p->peer_stream[q], p->ev_fence, 0));
    // direct exchange: the kernels themselves write into the peers
    if (p->direct) HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_fence, 0));
    p->peer_fence = false;
    return SDP_OK;
}

// End of a peer-write backup: when the tiny all-reduce below completes on this rank, every
// rank has passed its own copies (stream order), so all rows have landed everywhere.
static int finish_pushes(sdp_problem *p, hipEvent_t last_kernel)
{
    const int n = p->comm->nranks, me = p->comm->rank;
    hipStream_t cs = p->comm->stream;
    // The rendezvous must also say "this rank has finished READING V": once it completes, a peer may
    // start the next step and write rows into this rank's next J buffer -- the buffer this step's
    // kernels read as V.  With the dense exchange every phase's pushes wait for that phase's kernel,
    // so the copy streams carry the order; with the sparse one a rank whose last phases send nothing
    // would join the rendezvous with kernels still running.
    if (last_kernel) HIP_TRY(hipStreamWaitEvent(cs, last_kernel, 0));
    for (int q = 0; q < n; ++q) {
        if (q == me) continue;
        HIP_TRY(hipEventRecord(p->peer_done[q], p->peer_stream[q]));
        HIP_TRY(hipStreamWaitEvent(cs, p->peer_done[q], 0));
    }
    NCCL_TRY(g_rccl.AllReduce(p->d_flag, p->d_flag, 1, NCCL_INT32, NCCL_MAX, p->comm->comm, cs));
    return SDP_OK;
}

// direct sparse exchange: per column of the column layout, the ranks that read it (need lists of
// sdp_problem_set_peer_needs) as a bit mask the kernel looks up when it stores J
static int build_peer_mask(sdp_problem *p)
{
    if (p->mask_valid) return SDP_OK;
    if (p->layout != SDP_LAYOUT_COLUMNS) return fail(SDP_EINVAL, "the sparse direct exchange works on whole columns (column layout)");
    const int64_t n0 = p->orders[0], cols = p->S / n0;
    std::vector<unsigned char> mask((size_t)cols, 0);
    const int n = p->comm->nranks, me = p->comm->rank;
    for (int q = 0; q < n && q < SDP_MAX_PEERS; ++q) {
        if (q == me) continue;
        for (const auto &iv : p->need[(size_t)q])
            for (int64_t c = iv.first / n0; c < iv.second / n0; ++c) mask[(size_t)c] |= (unsigned char)(1u << q);
    }
    int rc = p->peer_mask.alloc((size_t)cols);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(p->peer_mask.p, mask.data(), (size_t)cols, hipMemcpyHostToDevice));
    p->mask_valid = true;
    return SDP_OK;
}

// Sparse exchange of one phase of J through the collective library alone (the path to a sparse
// exchange that rests on RCCL, not on stores into HIP-IPC mappings).  A rank reads the cost-to-go array only where the
// trailing next states of its columns fall (need lists: sdp_problem_set_peer_needs); rank q therefore gets, of this
// rank's rows of the phase, the BOUNDING range of what it reads there -- one ncclSend / ncclRecv per pair of ranks
// and phase inside one group (the need lists of a contracting exogenous process are runs of whole rows of the second
// axis: the bounding range is little more than their union; for anything else it is at worst the dense exchange).
// Both sides work the range out from the same lists and partition, so the counts match by construction.  Two-sided:
// nothing lands in a rank's J before that rank has posted its receive, which it does behind the phase's own kernel
// on the communicator stream -- no rendezvous beside the sends and receives themselves.
static void bounding_need(const std::vector<std::pair<int64_t, int64_t>> &iv, int64_t lo, int64_t hi, int64_t &s0, int64_t &s1)
{
    s0 = s1 = 0;
    if (hi <= lo) return;
    auto it = std::lower_bound(iv.begin(), iv.end(), lo,
                               [](const std::pair<int64_t, int64_t> &r, int64_t v) { return r.second <= v; });
    if (it == iv.end() || it->first >= hi) return;
    s0 = it->first > lo ? it->first : lo;
    auto last = std::lower_bound(iv.begin(), iv.end(), hi,
                                 [](const std::pair<int64_t, int64_t> &r, int64_t v) { return r.first < v; });
    --last;                                                // the last range that starts below hi
    s1 = last->second < hi ? last->second : hi;
}
// (slab partitions -- dist.slab_partition: nranks x G phases in node order, phase r G + g owned by rank r alone -- are
// exchanged G times a backup, not nranks x G times: when this rank's g-th phase is done it sends its rows of that phase
// and receives every peer's rows of THEIR g-th phase in one group; the phases it does not own pass)
static bool slab_phases(const sdp_problem *p, int &per_rank)
{
    const int n = p->comm->nranks;
    per_rank = 0;
    if (p->n_phases % n) return false;
    const int g = p->n_phases / n;
    for (int ph = 0; ph < p->n_phases; ++ph) {
        const int64_t *b = p->parts.data() + (size_t)ph * (n + 1);
        for (int r = 0; r < n; ++r)
            if (b[r + 1] > b[r] && r != ph / g) return false;
    }
    per_rank = g;
    return true;
}
static int sendrecv_phase(sdp_problem *p, int phase)
{
    const int n = p->comm->nranks, me = p->comm->rank;
    int per = 0;
    const bool slab = slab_phases(p, per);
    if (slab && phase / per != me) return SDP_OK;
    const int g = slab ? phase % per : 0;
    const int64_t *b = p->parts.data() + (size_t)phase * (n + 1);
    const size_t rs = real_size(p->dtype);
    hipStream_t cs = p->comm->stream;
    char *J = (char *)p->J.p;
    NCCL_TRY(g_rccl.GroupStart());
    for (int k = 1; k < n; ++k) {
        const int to = (me + k) % n, from = (me - k + n) % n;
        const int64_t *bf = slab ? p->parts.data() + (size_t)(from * per + g) * (n + 1) : b;
        int64_t s0, s1;
        bounding_need(p->need[(size_t)to], b[me], b[me + 1], s0, s1);
        if (s1 > s0) NCCL_TRY(g_rccl.Send(J + (size_t)s0 * rs, (size_t)(s1 - s0) * rs, NCCL_INT8, to, p->comm->comm, cs));
        bounding_need(p->need[(size_t)me], bf[from], bf[from + 1], s0, s1);
        if (s1 > s0) NCCL_TRY(g_rccl.Recv(J + (size_t)s0 * rs, (size_t)(s1 - s0) * rs, NCCL_INT8, from, p->comm->comm, cs));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return SDP_OK;
}

static int gather_phase(sdp_problem *p, int phase)
{
    return gather_phase_of(p, phase, p->J.p, real_size(p->dtype));
}

// One backup (Bellman sweep or fixed-policy evaluation) over the nodes this
// handle owns, V -> J.  Single GPU: one launch.  With a communicator the node
// range is cut into phases, every phase is shared out over the ranks, and the
// all-gather of phase k (communicator stream) overlaps the kernel of phase k+1
// (problem stream); the problem stream then waits for the last gather, so
// whatever follows (relative-DP shift, next sweep) sees the complete J.
static int run_backup(sdp_problem *p, bool evalpol, double t_k, int64_t shift_index = -1,
                      double *ref_out = nullptr, bool everything = false)
{
    int rc;
    if (!p->comm) {
        return evalpol ? launch_evalpol(p, t_k, p->node_begin, p->node_end, shift_index, ref_out)
                       : launch_sweep(p, t_k, p->node_begin, p->node_end);
    }
    const int n = p->comm->nranks, rank = p->comm->rank;
    if (n > 1 && p->peer_exchange && p->peer_fence && (rc = open_pushes(p))) return rc;
    const bool direct = n > 1 && p->peer_exchange && p->direct;
    if (direct && p->sparse && !everything && (rc = build_peer_mask(p))) return rc;
    p->send_everything = everything;
    for (int ph = 0; ph < p->n_phases; ++ph) {
        const int64_t *b = p->parts.data() + (size_t)ph * (n + 1);
        rc = evalpol ? launch_evalpol(p, t_k, b[rank], b[rank + 1], shift_index, ref_out)
                     : launch_sweep(p, t_k, b[rank], b[rank + 1]);
        if (rc) return rc;
        if (n > 1) {
            HIP_TRY(hipEventRecord(p->ev_phase[ph], p->stream));
            if (direct) {
                // (the kernel has stored its rows into the peers that read them: nothing to copy)
            } else if (p->peer_exchange) {
                if ((rc = push_phase(p, ph, p->ev_phase[ph], everything))) return rc;
            } else if (p->sendrecv && p->sparse && !everything) {
                HIP_TRY(hipStreamWaitEvent(p->comm->stream, p->ev_phase[ph], 0));
                if ((rc = sendrecv_phase(p, ph))) return rc;
            } else {
                HIP_TRY(hipStreamWaitEvent(p->comm->stream, p->ev_phase[ph], 0));
                if ((rc = gather_phase(p, ph))) return rc;
            }
        }
    }
    if (n > 1) {
        if (p->peer_exchange && (rc = finish_pushes(p, p->n_phases ? p->ev_phase[p->n_phases - 1] : nullptr))) return rc;
        HIP_TRY(hipEventRecord(p->ev_comm, p->comm->stream));
        p->comm_pending = true;
        p->J_partial = (p->peer_exchange || p->sendrecv) && p->sparse && !everything;
    }
    return SDP_OK;
}

// make the problem stream wait for the exchange of the last backup
static int join_comm(sdp_problem *p)
{
    if (p->comm_pending) {
        HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_comm, 0));
        p->comm_pending = false;
    }
    return SDP_OK;
}

// Sparse peer exchange: before J leaves the device (or is read whole), every rank sends the rows
// it computed to every peer.  Collective, like the backups.
static int complete_J(sdp_problem *p)
{
    if (!p->comm || p->comm->nranks == 1 || !(p->peer_exchange || p->sendrecv) || !p->J_partial) return SDP_OK;
    int rc;
    if ((rc = join_comm(p))) return rc;
    if (p->sendrecv) {                                     // every rank's rows to every rank: the all-gather of each phase
        if (!p->ev_enter) HIP_TRY(hipEventCreateWithFlags(&p->ev_enter, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(p->ev_enter, p->stream));
        HIP_TRY(hipStreamWaitEvent(p->comm->stream, p->ev_enter, 0));
        for (int ph = 0; ph < p->n_phases; ++ph)
            if ((rc = gather_phase(p, ph))) return rc;
        HIP_TRY(hipEventRecord(p->ev_comm, p->comm->stream));
        p->comm_pending = true;
        if ((rc = join_comm(p))) return rc;
        p->J_partial = false;
        return SDP_OK;
    }
    p->peer_fence = true;                                  // nobody may still be reading its J
    if ((rc = open_pushes(p))) return rc;
    HIP_TRY(hipEventRecord(p->ev_enter, p->stream));
    for (int ph = 0; ph < p->n_phases; ++ph)
        if ((rc = push_phase(p, ph, p->ev_enter, true))) return rc;
    if ((rc = finish_pushes(p, p->ev_enter))) return rc;
    HIP_TRY(hipEventRecord(p->ev_comm, p->comm->stream));
    p->comm_pending = true;
    if ((rc = join_comm(p))) return rc;
    p->J_partial = false;
    return SDP_OK;
}

static int rel_shift(sdp_problem *p, int64_t ref_index, int slot)
{
    double *ref = (double *)p->refs.p + slot;
    unsigned blocks = (unsigned)((p->S + 255) / 256);
    if (blocks > (unsigned)p->cus * 16) blocks = p->cus * 16;
    if (p->dtype == SDP_F32) {
        hipLaunchKernelGGL(k_pick_ref<float>, dim3(1), dim3(64), 0, p->stream, (const float *)p->J.p, ref_index, ref);
        hipLaunchKernelGGL(k_shift<float>, dim3(blocks), dim3(256), 0, p->stream, (float *)p->J.p, p->S, ref);
    } else {
        hipLaunchKernelGGL(k_pick_ref<double>, dim3(1), dim3(64), 0, p->stream, (const double *)p->J.p, ref_index, ref);
        hipLaunchKernelGGL(k_shift<double>, dim3(blocks), dim3(256), 0, p->stream, (double *)p->J.p, p->S, ref);
    }
    HIP_TRY(hipGetLastError());
    return SDP_OK;
}

static int ensure_refs(sdp_problem *p, int n)
{
    if (n < 16) n = 16;
    if (p->refs_cap >= n) return SDP_OK;
    int rc = p->refs.alloc((size_t)n * 8);
    if (rc) return rc;
    p->refs_cap = n;
    return SDP_OK;
}

// ref_index arrives as a flat index in the reference's C order (like every
// host-side quantity of the API) and is turned into the handle's device order
static int check_ref(const sdp_problem *p, int rel_dp, int64_t &ref_index)
{
    if (!rel_dp) return SDP_OK;
    if (ref_index < 0 || ref_index >= p->S) return fail(SDP_EINVAL, "reference node %lld outside the grid", (long long)ref_index);
    if (p->layout == SDP_LAYOUT_COLUMNS) {
        const int64_t n0 = p->orders[0], P = p->S / n0;
        ref_index = (ref_index % P) * n0 + ref_index / P;
    }
    if (p->sparse && p->comm && p->comm->nranks > 1) {
        // sparse exchange: the shift reads J[ref] / V[ref] on every rank, so the node must be one this
        // rank computes or receives (the host puts the reference column into every need list)
        const int me = p->comm->rank;
        bool have = false;
        for (int ph = 0; ph < p->n_phases && !have; ++ph) {
            const int64_t *b = p->parts.data() + (size_t)ph * (p->comm->nranks + 1);
            have = ref_index >= b[me] && ref_index < b[me + 1];
        }
        for (size_t k = 0; k < p->need[(size_t)me].size() && !have; ++k)
            have = ref_index >= p->need[(size_t)me][k].first && ref_index < p->need[(size_t)me][k].second;
        if (!have)
            return fail(SDP_EINVAL, "sparse exchange: reference node %lld is neither computed nor received by rank %d "
                        "(it must be in every rank's need list)", (long long)ref_index, me);
    }
    return SDP_OK;
}

extern "C" int sdp_problem_vi_sweep(sdp_problem *p, double t_k, int rel_dp, int64_t ref_index,
                                    double *J_ref_out)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    p->peer_fence = true;                // first backup of a call (open_pushes)
    int rc;
    if ((rc = check_ref(p, rel_dp, ref_index))) return rc;
    if ((rc = ensure_refs(p, 1))) return rc;
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    if ((rc = run_backup(p, false, t_k))) return rc;
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    if ((rc = join_comm(p))) return rc;
    if (rel_dp && (rc = rel_shift(p, ref_index, 0))) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->last_kernel_ms = ms;
    if (rel_dp && J_ref_out) HIP_TRY(hipMemcpy(J_ref_out, p->refs.p, 8, hipMemcpyDeviceToHost));
    return SDP_OK;
}

// V <-> J ping-pong.  Every rank swaps in lockstep, so the mapped peer buffers swap too.
static void swap_buffers(sdp_problem *p)
{
    std::swap(p->V.p, p->J.p);
    std::swap(p->V.ever_exported, p->J.ever_exported);
    std::swap(p->V_partial, p->J_partial);
    ++p->V_gen;
    if (p->peer_exchange) p->peer_V.swap(p->peer_J);
}

extern "C" int sdp_problem_swap(sdp_problem *p)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    swap_buffers(p);
    return SDP_OK;
}

extern "C" int sdp_problem_eval_policy(sdp_problem *p, int32_t n_iter, int rel_dp,
                                       int64_t ref_index, double *J_ref_out)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    p->peer_fence = true;                // first backup of a call (open_pushes)
    if (n_iter < 0) return fail(SDP_EINVAL, "negative iteration count");
    if (!p->pol_in.p) return fail(SDP_EINVAL, "no policy set (sdp_problem_set_policy)");
    int rc;
    if ((rc = check_ref(p, rel_dp, ref_index))) return rc;
    if ((rc = ensure_refs(p, n_iter))) return rc;
    if (n_iter == 0) {                   // result = the starting value (stodynprog.py:714)
        HIP_TRY(hipMemcpyAsync(p->J.p, p->V.p, p->S * real_size(p->dtype), hipMemcpyDeviceToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        return SDP_OK;
    }
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    // Relative DP (stodynprog.py:760-762) without writing the shifted array out
    // between steps: step k leaves the raw J_k in the J buffer; step k+1 reads
    // every vertex as V - V[ref] (same single rounding as the reference's
    // in-place `J_pol -= J_ref[k]`) and records J_ref[k] = V[ref]; only the last
    // step is followed by the explicit shift kernels.  One launch per step
    // instead of three.
    double *refs = (double *)p->refs.p;
    for (int k = 0; k < n_iter; ++k) {
        if (k > 0) swap_buffers(p);
        const bool fused = rel_dp && k > 0;
        if ((rc = run_backup(p, true, 0.0, fused ? ref_index : -1, fused ? refs + (k - 1) : nullptr))) return rc;
        if ((rc = join_comm(p))) return rc;
    }
    if (rel_dp && (rc = rel_shift(p, ref_index, n_iter - 1))) return rc;
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->last_kernel_ms = ms;
    if (rel_dp && J_ref_out) HIP_TRY(hipMemcpy(J_ref_out, p->refs.p, (size_t)n_iter * 8, hipMemcpyDeviceToHost));
    return SDP_OK;
}

// ---------------------------------------------------------------------------
// pinned host memory for the arrays that cross the API every call
// (DPSolver.value_iteration takes and returns numpy arrays, stodynprog.py:466,
// 494-498, 530-533): copies to and from page-locked memory run as one DMA at
// PCIe rate and asynchronously; pageable memory is staged by the runtime
// ---------------------------------------------------------------------------
extern "C" int sdp_host_alloc(size_t bytes, void **out)
{
    if (!out) return fail(SDP_EINVAL, "NULL argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(SDP_ENOMEM, "hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    return SDP_OK;
}

extern "C" int sdp_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return SDP_OK;
}

// device buffer in the handle's layout -> host (reference C order), enqueued on the
// problem stream WITHOUT a synchronisation; `slot` picks the conversion buffer
static int download_nodes_async(sdp_problem *p, void *host, const void *dev, size_t elem_bytes, int slot)
{
    const size_t bytes = (size_t)p->S * elem_bytes;
    if (p->layout != SDP_LAYOUT_COLUMNS) {
        HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, p->stream));
        return SDP_OK;
    }
    if (p->stage_bytes[slot] < bytes) {
        int rc = p->stage[slot].alloc(bytes);
        if (rc) { p->stage_bytes[slot] = 0; return rc; }
        p->stage_bytes[slot] = bytes;
    }
    int rc = launch_transpose(dev, p->stage[slot].p, p->S / p->orders[0], p->orders[0],
                              (int)(elem_bytes / 4), p->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(host, p->stage[slot].p, bytes, hipMemcpyDeviceToHost, p->stream));
    return SDP_OK;
}

static int gather_policy(sdp_problem *p);

// Rows [nb, ne) (device order) of a per-node array -> their place in the host array
// (reference C order), enqueued on `stream`.  Node layout: one contiguous copy.  Column
// layout: the rows are whole columns c0 .. c1-1, i.e. the host sub-block [:, c0:c1] of the
// [N0][P] array: transposed into the dense [N0][c1-c0] part of the conversion buffer, then
// one strided 2-D copy.
static int download_range_async(sdp_problem *p, void *host, const void *dev, size_t elem_bytes, int slot,
                                int64_t nb, int64_t ne, hipStream_t stream)
{
    if (ne <= nb) return SDP_OK;
    if (p->layout != SDP_LAYOUT_COLUMNS) {
        HIP_TRY(hipMemcpyAsync((char *)host + nb * elem_bytes, (const char *)dev + nb * elem_bytes,
                               (size_t)(ne - nb) * elem_bytes, hipMemcpyDeviceToHost, stream));
        return SDP_OK;
    }
    const size_t bytes = (size_t)p->S * elem_bytes;
    if (p->stage_bytes[slot] < bytes) {
        int rc = p->stage[slot].alloc(bytes);
        if (rc) { p->stage_bytes[slot] = 0; return rc; }
        p->stage_bytes[slot] = bytes;
    }
    const int64_t n0 = p->orders[0], P = p->S / n0, c0 = nb / n0, c1 = ne / n0;
    char *dense = (char *)p->stage[slot].p + (size_t)nb * elem_bytes;          // [N0][c1-c0]
    int rc = launch_transpose((const char *)dev + nb * elem_bytes, dense, c1 - c0, n0,
                              (int)(elem_bytes / 4), stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync((char *)host + c0 * elem_bytes, (size_t)P * elem_bytes, dense,
                             (size_t)(c1 - c0) * elem_bytes, (size_t)(c1 - c0) * elem_bytes, (size_t)n0,
                             hipMemcpyDeviceToHost, stream));
    return SDP_OK;
}

// One value_iteration call with host arrays in and out (stodynprog.py:466-534):
// upload of J_next (skipped when host_V is NULL: the device keeps its value buffer),
// the backup, the relative-DP shift and the downloads of J_k and the policy values
// are queued back to back on the problem stream; ONE synchronisation at the end.
extern "C" int sdp_problem_backup_host(sdp_problem *p, const void *host_V, double t_k, int rel_dp,
                                       int64_t ref_index, void *host_J, void *host_pol,
                                       int32_t *host_idx, double *J_ref_out)
{
    if (!p || !host_J) return fail(SDP_EINVAL, "NULL argument");
    p->peer_fence = true;                // first backup of a call (open_pushes)
    int rc;
    if ((rc = check_ref(p, rel_dp, ref_index))) return rc;
    if ((rc = ensure_refs(p, 1))) return rc;
    const size_t rs = real_size(p->dtype);
    if (host_V) {
        const size_t bytes = (size_t)p->S * rs;
        ++p->V_gen;
        if (p->layout != SDP_LAYOUT_COLUMNS) {
            HIP_TRY(hipMemcpyAsync(p->V.p, host_V, bytes, hipMemcpyHostToDevice, p->stream));
        } else {
            if ((rc = ensure_scratch(p, bytes))) return rc;
            HIP_TRY(hipMemcpyAsync(p->scratch.p, host_V, bytes, hipMemcpyHostToDevice, p->stream));
            if ((rc = launch_transpose(p->scratch.p, p->V.p, p->orders[0], p->S / p->orders[0],
                                       (int)(rs / 4), p->stream))) return rc;
        }
    }
    // Large single-GPU problems: the backup runs in a few phases of the node range and the
    // finished rows of a phase go to the host (copy stream) UNDER the kernel of the next
    // phase, so only the last phase's download is exposed.  J waits for the relative-DP
    // shift when there is one; the policy never does.
    const int64_t unit = p->layout == SDP_LAYOUT_COLUMNS ? p->orders[0] : 1;
    const int64_t units = (p->node_end - p->node_begin) / unit;
    const int n_ph = 4;
    if (p->host_overlap && !p->comm && p->S * (int64_t)rs >= ((int64_t)8 << 20) && units >= 64 * n_ph &&
        p->node_begin == 0 && p->node_end == p->S) {
        if (!p->copy_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
            for (auto &e : p->ev_host) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_copy, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(p->ev0, p->stream));
        for (int ph = 0; ph < n_ph; ++ph) {
            const int64_t nb = units * ph / n_ph * unit, ne = units * (ph + 1) / n_ph * unit;
            if ((rc = launch_sweep(p, t_k, nb, ne))) return rc;
            HIP_TRY(hipEventRecord(p->ev_host[ph], p->stream));
            HIP_TRY(hipStreamWaitEvent(p->copy_stream, p->ev_host[ph], 0));
            if (!rel_dp && (rc = download_range_async(p, host_J, p->J.p, rs, 0, nb, ne, p->copy_stream))) return rc;
            if (host_pol && (rc = download_range_async(p, host_pol, p->pol.p, (size_t)p->nu * rs, 1, nb, ne, p->copy_stream))) return rc;
            if (host_idx && (rc = download_range_async(p, host_idx, p->idx.p, 4, 2, nb, ne, p->copy_stream))) return rc;
        }
        HIP_TRY(hipEventRecord(p->ev1, p->stream));
        if (rel_dp) {
            if ((rc = rel_shift(p, ref_index, 0))) return rc;
            if ((rc = download_nodes_async(p, host_J, p->J.p, rs, 0))) return rc;    // (its own slot-0 buffer use:
        }                                                                             //  no phase copy of J is in flight)
        HIP_TRY(hipStreamSynchronize(p->copy_stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        p->last_kernel_ms = ms;
        if (rel_dp && J_ref_out) HIP_TRY(hipMemcpy(J_ref_out, p->refs.p, 8, hipMemcpyDeviceToHost));
        return SDP_OK;
    }
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    if ((rc = run_backup(p, false, t_k, -1, nullptr, true))) return rc;    // J goes to the host: all rows to all ranks
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    if ((rc = join_comm(p))) return rc;
    if (rel_dp && (rc = rel_shift(p, ref_index, 0))) return rc;
    if ((rc = download_nodes_async(p, host_J, p->J.p, rs, 0))) return rc;
    if (host_pol || host_idx) {
        if (p->comm && p->comm->nranks > 1 && (rc = gather_policy(p))) return rc;   // collective, synchronises
        if (host_pol && (rc = download_nodes_async(p, host_pol, p->pol.p, (size_t)p->nu * rs, 1))) return rc;
        if (host_idx && (rc = download_nodes_async(p, host_idx, p->idx.p, 4, 2))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->last_kernel_ms = ms;
    if (rel_dp && J_ref_out) HIP_TRY(hipMemcpy(J_ref_out, p->refs.p, 8, hipMemcpyDeviceToHost));
    return SDP_OK;
}

extern "C" int sdp_problem_set_host_overlap(sdp_problem *p, int on)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    p->host_overlap = on ? 1 : 0;
    return SDP_OK;
}

// Batched closed-loop simulation on the device (kernel sdp_simulate of the model's code
// object): see include/sdp_hip.h.
extern "C" int sdp_problem_simulate(sdp_problem *p, const void *host_pol, int64_t B, int64_t T,
                                    const void *host_x0, const void *host_w, double t0,
                                    void *host_x, void *host_u, void *host_g)
{
    if (!p || !host_pol || !host_x0 || !host_x || !host_u) return fail(SDP_EINVAL, "NULL argument");
    if (B < 0 || T < 0) return fail(SDP_EINVAL, "negative size");
    if (!p->f_simulate) return fail(SDP_EMODULE, "the model's code object has no sdp_simulate kernel");
    if (p->W > 0 && T > 0 && !host_w) return fail(SDP_EINVAL, "a stochastic system needs the perturbation sequences");
    if (B == 0) return SDP_OK;
    const size_t rs = real_size(p->dtype);
    DevBuf dpol, dx0, dw, dx, du, dg;
    int rc;
    if ((rc = upload(dpol, host_pol, (size_t)p->nu * p->S * rs))) return rc;
    if ((rc = upload(dx0, host_x0, (size_t)p->d * B * rs))) return rc;
    if (host_w && T > 0 && (rc = upload(dw, host_w, (size_t)T * B * rs))) return rc;
    if ((rc = dx.alloc((size_t)(T + 1) * p->d * B * rs))) return rc;
    if ((rc = du.alloc((size_t)T * p->nu * B * rs))) return rc;
    if (host_g && (rc = dg.alloc((size_t)T * B * rs))) return rc;
    SdpSimArgs a;
    memset(&a, 0, sizeof(a));
    a.pol = dpol.p; a.axes = p->axes.p; a.x0 = dx0.p; a.w = (host_w && T > 0) ? dw.p : nullptr;
    a.x = dx.p; a.u = du.p; a.g = host_g ? dg.p : nullptr;
    a.B = B; a.T = T; a.S = p->S; a.t0 = t0;
    for (int k = 0; k < SDP_MAXD; ++k) { a.orders[k] = p->orders[k]; a.axis_off[k] = p->axis_off[k]; }
    size_t size = sizeof(a);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size,
                     HIP_LAUNCH_PARAM_END};
    int64_t blocks = (B + 63) / 64;
    if (blocks > (int64_t)p->cus * 32) blocks = (int64_t)p->cus * 32;
    HIP_TRY(hipStreamSynchronize(p->stream));        // lifted constants set on the problem stream
    HIP_TRY(hipModuleLaunchKernel(p->f_simulate, (unsigned)blocks, 1, 1, 64, 1, 1, 0, p->stream, nullptr, extra));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(host_x, dx.p, (size_t)(T + 1) * p->d * B * rs, hipMemcpyDeviceToHost));
    if (T > 0) HIP_TRY(hipMemcpy(host_u, du.p, (size_t)T * p->nu * B * rs, hipMemcpyDeviceToHost));
    if (host_g && T > 0) HIP_TRY(hipMemcpy(host_g, dg.p, (size_t)T * B * rs, hipMemcpyDeviceToHost));
    return SDP_OK;
}

extern "C" int sdp_problem_get_value(sdp_problem *p, void *host_J)
{
    if (!p || !host_J) return fail(SDP_EINVAL, "NULL argument");
    int rc;
    if ((rc = complete_J(p))) return rc;                   // (sparse exchange: collective)
    return download_nodes(p, host_J, p->J.p, real_size(p->dtype));
}

// after a sharded sweep every rank holds the policy rows of its own parts only:
// collect the others (on request -- the sweep itself never pays for this)
static int gather_policy(sdp_problem *p)
{
    if (!p->comm || p->comm->nranks == 1) return SDP_OK;
    int rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int ph = 0; ph < p->n_phases; ++ph) {
        if ((rc = gather_phase_of(p, ph, p->pol.p, (size_t)p->nu * real_size(p->dtype)))) return rc;
        if ((rc = gather_phase_of(p, ph, p->idx.p, 4))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(p->comm->stream));
    return SDP_OK;
}

extern "C" int sdp_problem_get_policy(sdp_problem *p, void *host_pol, int32_t *host_idx)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    int rc;
    if ((rc = gather_policy(p))) return rc;
    if (host_pol && (rc = download_nodes(p, host_pol, p->pol.p, (size_t)p->nu * real_size(p->dtype)))) return rc;
    if (host_idx && (rc = download_nodes(p, host_idx, p->idx.p, 4))) return rc;
    return SDP_OK;
}

extern "C" int sdp_problem_last_kernel_ms(sdp_problem *p, double *ms)
{
    if (!p || !ms) return fail(SDP_EINVAL, "NULL argument");
    *ms = p->last_kernel_ms;
    return SDP_OK;
}

// diagnostic: clock stamps of SDP_STAMP code objects (see sdp_kernel_args.h)
extern "C" int sdp_problem_debug_stamps(sdp_problem *p, int enable, unsigned long long *host,
                                        int64_t n_words)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    if (enable && !p->stamps.p) {
        const int64_t words = (int64_t)65536 * 4;           // room for any grid the library launches
        int rc = p->stamps.alloc((size_t)words * 8);
        if (rc) return rc;
        HIP_TRY(hipMemset(p->stamps.p, 0, (size_t)words * 8));
        p->stamp_words = words;
    }
    if (host && n_words > 0) {
        if (!p->stamps.p) return fail(SDP_EINVAL, "stamps were never enabled");
        if (n_words > p->stamp_words) n_words = p->stamp_words;
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpy(host, p->stamps.p, (size_t)n_words * 8, hipMemcpyDeviceToHost));
    }
    if (!enable && p->stamps.p) { (void)hipFree(p->stamps.p); p->stamps.p = nullptr; p->stamp_words = 0; }
    return SDP_OK;
}

#ifdef SDP_TEST_HOOKS
// TEST BUILD ONLY: fill the device arrays a backup writes (J, policy, index; the layout-conversion
// buffers when they exist) with recognisable bytes, so that a row that reaches the host WITHOUT having
// been written by the kernel / the conversion shows which buffer it came from (tools/host_phase_stress.py)
extern "C" int sdp_problem_debug_poison(sdp_problem *p)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    const size_t rs = real_size(p->dtype);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(p->J.p, 0xF1, (size_t)p->S * rs));
    HIP_TRY(hipMemset(p->pol.p, 0xF2, (size_t)p->S * rs * p->nu));
    HIP_TRY(hipMemset(p->idx.p, 0xF3, (size_t)p->S * 4));
    for (int k = 0; k < 3; ++k)
        if (p->stage[k].p) HIP_TRY(hipMemset(p->stage[k].p, 0xE0 + k, p->stage_bytes[k]));
    HIP_TRY(hipDeviceSynchronize());
    return SDP_OK;
}
// TEST BUILD ONLY: leave `bytes` of freed device memory filled with `byte` behind (what a later
// hipMalloc may hand out again)
extern "C" int sdp_debug_pollute(size_t bytes, int byte)
{
    void *q = nullptr;
    HIP_TRY(hipMalloc(&q, bytes));
    HIP_TRY(hipMemset(q, byte, bytes));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(q));
    return SDP_OK;
}
#endif

extern "C" int sdp_problem_bench_sweeps(sdp_problem *p, int32_t reps, int rel_dp,
                                        int64_t ref_index, double *loop_ms, double *kernel_ms)
{
    if (!p || reps < 1) return fail(SDP_EINVAL, "bad arguments");
    p->peer_fence = true;                // first backup of a call (open_pushes)
    int rc;
    if ((rc = check_ref(p, rel_dp, ref_index))) return rc;
    if ((rc = ensure_refs(p, 1))) return rc;
    struct Events {                         // destroyed on every return path
        std::vector<hipEvent_t> v;
        ~Events() { for (auto e : v) if (e) (void)hipEventDestroy(e); }
    } events;
    events.v.assign(2 * (size_t)reps, nullptr);
    std::vector<hipEvent_t> &ev = events.v;
    for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(p->ev2, p->stream));
    for (int r = 0; r < reps; ++r) {
        if (r > 0) swap_buffers(p);
        HIP_TRY(hipEventRecord(ev[2 * r], p->stream));
        if ((rc = run_backup(p, false, 0.0))) return rc;
        HIP_TRY(hipEventRecord(ev[2 * r + 1], p->stream));
        if ((rc = join_comm(p))) return rc;
        if (rel_dp && (rc = rel_shift(p, ref_index, 0))) return rc;
    }
    HIP_TRY(hipEventRecord(p->ev3, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev2, p->ev3));
    if (loop_ms) *loop_ms = ms;
    double ksum = 0;
    for (int r = 0; r < reps; ++r) {
        HIP_TRY(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]));
        ksum += ms;
    }
    if (kernel_ms) *kernel_ms = ksum;
    p->last_kernel_ms = ksum / reps;
    return SDP_OK;
}

// Map every other rank's value / J buffer into this process (HIP IPC) and switch the exchange
// of the backups from RCCL all-gathers to peer writes (push_phase).  Collective: every rank of
// the communicator calls it, after sdp_problem_attach_comm.  The handles travel through the
// communicator itself (one all-gather of 2 x 64 bytes per rank).
extern "C" int sdp_problem_enable_peer_exchange(sdp_problem *p)
{
    if (!p || !p->comm) return fail(SDP_EINVAL, "no communicator attached");
    const int n = p->comm->nranks, me = p->comm->rank;
    if (n == 1) return SDP_OK;
    p->release_peers();
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
    std::vector<hipIpcMemHandle_t> all((size_t)2 * n);
    memset(all.data(), 0, all.size() * sizeof(hipIpcMemHandle_t));
    // From here to the agreement below every rank takes part in the same two collectives whatever
    // happens locally: a failure only raises `failed` (a rank that left early would leave the
    // others waiting in a collective).
    int failed = 0;
    char why[256] = "";
    {
        hipError_t e = hipIpcGetMemHandle(&all[2 * me], p->V.p);
        ipc_trace(e == hipSuccess ? "export V" : "export V FAILED", p->V.p, &all[2 * me], 0);
        if (e == hipSuccess) {
            p->V.ever_exported = true;
            e = hipIpcGetMemHandle(&all[2 * me + 1], p->J.p);
            ipc_trace(e == hipSuccess ? "export J" : "export J FAILED", p->J.p, &all[2 * me + 1], 0);
            if (e == hipSuccess) p->J.ever_exported = true;
        }
        if (e != hipSuccess) {
            failed = 1;
            snprintf(why, sizeof(why), "exporting this rank's buffers: %s", hipGetErrorString(e));
            (void)hipGetLastError();
        }
    }
    // Between the two agreements below nothing returns early: a local failure raises `failed`, and
    // a failure of the communicator's own calls (a broken communicator or device: the other ranks
    // would wait in a collective for ever) ends the process with a message.
#define COLLECTIVE_OR_DIE(expr)                                                                 \
    do {                                                                                        \
        const int r__ = (int)(expr);                                                            \
        if (r__ != 0) {                                                                         \
            fprintf(stderr, "sdp_problem_enable_peer_exchange: %s failed (%d) inside a collective "  \
                    "section; aborting this rank\n", #expr, r__);                               \
            abort();                                                                            \
        }                                                                                       \
    } while (0)
    hipStream_t cs = p->comm->stream;
    int *const d_agree = (int *)p->comm->d_scalar;          // 8 bytes owned by the communicator
    auto agree = [&](int mine) -> int {                     // max over the ranks of `mine`
        int any = 0;
        COLLECTIVE_OR_DIE(hipMemcpyAsync(d_agree, &mine, sizeof(int), hipMemcpyHostToDevice, cs));
        COLLECTIVE_OR_DIE(g_rccl.AllReduce(d_agree, d_agree, 1, NCCL_INT32, NCCL_MAX, p->comm->comm, cs));
        COLLECTIVE_OR_DIE(hipMemcpyAsync(&any, d_agree, sizeof(int), hipMemcpyDeviceToHost, cs));
        COLLECTIVE_OR_DIE(hipStreamSynchronize(cs));
        return any;
    };
    DevBuf stage;
    if (stage.alloc(sizeof(hipIpcMemHandle_t) * 2 * n) != SDP_OK && !failed) {
        failed = 1;
        snprintf(why, sizeof(why), "staging buffer for the handle exchange: out of device memory");
    }
    // first agreement: every rank can export its buffers and holds a staging buffer
    if (agree(failed)) {
        p->release_peers();
        return fail(SDP_ECOMM, "peer exchange not available (%s): the RCCL exchange stays in place",
                    failed ? why : "another rank could not export its buffers");
    }
    COLLECTIVE_OR_DIE(hipMemcpyAsync((char *)stage.p + (size_t)me * 128, &all[2 * me], 128, hipMemcpyHostToDevice, cs));
    COLLECTIVE_OR_DIE(g_rccl.AllGather((char *)stage.p + (size_t)me * 128, stage.p, 128, NCCL_INT8, p->comm->comm, cs));
    COLLECTIVE_OR_DIE(hipMemcpyAsync(all.data(), stage.p, (size_t)n * 128, hipMemcpyDeviceToHost, cs));
    COLLECTIVE_OR_DIE(hipStreamSynchronize(cs));
    p->peer_me = me;
    p->peer_V.assign(n, nullptr); p->peer_J.assign(n, nullptr);
    p->peer_stream.assign(n, nullptr); p->peer_done.assign(n, nullptr);
    p->peer_V[me] = p->V.p; p->peer_J[me] = p->J.p;
    for (int r = 0; r < n && !failed; ++r) {
        if (r == me) continue;
        hipError_t e = ipc_open(&p->peer_V[r], all[2 * r]);
        if (e != hipSuccess) p->peer_V[r] = nullptr;
        if (e == hipSuccess) {
            e = ipc_open(&p->peer_J[r], all[2 * r + 1]);
            if (e != hipSuccess) p->peer_J[r] = nullptr;
        }
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->peer_stream[r], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->peer_done[r], hipEventDisableTiming);
        if (e != hipSuccess) {
            failed = 1;
            snprintf(why, sizeof(why), "mapping rank %d's buffers: %s", r, hipGetErrorString(e));
            (void)hipGetLastError();
        }
    }
    if (!failed && hipMalloc((void **)&p->d_flag, 8) != hipSuccess) { failed = 1; snprintf(why, sizeof(why), "hipMalloc"); }
    if (!failed && (hipEventCreateWithFlags(&p->ev_enter, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&p->ev_fence, hipEventDisableTiming) != hipSuccess)) {
        failed = 1; snprintf(why, sizeof(why), "hipEventCreate");
    }
    if (!failed && hipMemset(p->d_flag, 0, 8) != hipSuccess) { failed = 1; snprintf(why, sizeof(why), "hipMemset"); }
    // second agreement, all ranks or none: the ranks agree on the outcome before anybody writes
    // into a peer (this also guarantees that every rank has mapped everything before the first write)
    if (agree(failed)) {
        p->release_peers();
        return fail(SDP_ECOMM, "peer exchange not available (%s): the RCCL exchange stays in place",
                    failed ? why : "another rank could not map its peers");
    }
#undef COLLECTIVE_OR_DIE
    p->peer_exchange = true;
    p->peer_fence = true;
    return SDP_OK;
}

extern "C" int sdp_problem_complete_value(sdp_problem *p)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    int rc;
    if ((rc = complete_J(p))) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return SDP_OK;
}

extern "C" int sdp_problem_disable_peer_exchange(sdp_problem *p)
{
    if (!p) return fail(SDP_EINVAL, "NULL handle");
    HIP_TRY(hipDeviceSynchronize());                       // (nothing of this rank is still writing into a peer)
    // (a value array that the sparse exchange left incomplete stays marked so: this call is for tearing a problem
    // down, not for switching exchanges in the middle of a chain)
    p->release_peers();
    p->mask_valid = false;
    return SDP_OK;
}

extern "C" int sdp_problem_set_peer_needs(sdp_problem *p, const int64_t *need_off, const int64_t *ranges)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    if (!need_off || !ranges) { p->sparse = false; p->need.clear(); p->mask_valid = false; return SDP_OK; }
    if (!p->comm || !(p->peer_exchange || p->sendrecv)) return fail(SDP_EINVAL, "sparse exchange needs the peer exchange (sdp_problem_enable_peer_exchange) or the send / receive one (sdp_problem_set_sendrecv_exchange)");
    if (p->J_partial || p->V_partial) return fail(SDP_EINVAL, "the value arrays are incomplete: fetch or set them first");
    const int n = p->comm->nranks;
    const int64_t unit = p->layout == SDP_LAYOUT_COLUMNS ? p->orders[0] : 1;
    std::vector<std::vector<std::pair<int64_t, int64_t>>> need((size_t)n);
    for (int q = 0; q < n; ++q) {
        int64_t at = 0;
        for (int64_t k = need_off[q]; k < need_off[q + 1]; ++k) {
            const int64_t b = ranges[2 * k], e = ranges[2 * k + 1];
            if (b < at || e <= b || e > p->S || b % unit || e % unit)
                return fail(SDP_EINVAL, "peer needs: ranges must be sorted, disjoint, inside the grid%s",
                            unit > 1 ? " and made of whole columns" : "");
            need[(size_t)q].emplace_back(b, e);
            at = e;
        }
    }
    p->need.swap(need);
    p->sparse = true;
    p->mask_valid = false;
    return SDP_OK;
}

extern "C" int sdp_problem_set_sendrecv_exchange(sdp_problem *p, int on)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    if (!on) { p->sendrecv = false; return SDP_OK; }
    if (!p->comm) return fail(SDP_EINVAL, "no communicator attached");
    if (p->peer_exchange) return fail(SDP_EINVAL, "the send / receive exchange is an alternative to the peer exchange, not an addition");
    p->sendrecv = true;
    return SDP_OK;
}

extern "C" int sdp_problem_set_direct_exchange(sdp_problem *p, int on)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    if (!on) { p->direct = false; return SDP_OK; }
    if (!p->comm || !p->peer_exchange) return fail(SDP_EINVAL, "the direct exchange needs the peers' buffers mapped (sdp_problem_enable_peer_exchange)");
    if (p->comm->nranks > SDP_MAX_PEERS) return fail(SDP_EINVAL, "the direct exchange serves at most %d ranks (one node), the communicator has %d", SDP_MAX_PEERS, p->comm->nranks);
    if (!(p->meta[SDP_META_FLAGS] & SDP_META_F_PEER_STORES)) return fail(SDP_EMODULE, "the code object's kernels do not store through sdp_store_J");
    p->direct = true;
    p->peer_fence = true;
    return SDP_OK;
}

extern "C" int sdp_problem_set_lead_halo(sdp_problem *p, int64_t rows)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    p->lead_halo = rows;
    p->red_V = nullptr;
    return SDP_OK;
}

extern "C" int sdp_problem_attach_comm(sdp_problem *p, sdp_comm *c, int32_t n_phases,
                                       const int64_t *part_bounds)
{
    if (!p) return fail(SDP_EINVAL, "NULL problem");
    if (!c) { p->release_peers(); p->comm = nullptr; p->parts.clear(); p->n_phases = 0; return SDP_OK; }
    if (!part_bounds || n_phases < 1) return fail(SDP_EINVAL, "phase partition missing");
    const int n = c->nranks;
    std::vector<int64_t> parts(part_bounds, part_bounds + (size_t)n_phases * (n + 1));
    int64_t at = 0;
    const int64_t unit = p->layout == SDP_LAYOUT_COLUMNS ? p->orders[0] : 1;
    for (int ph = 0; ph < n_phases; ++ph) {
        const int64_t *b = parts.data() + (size_t)ph * (n + 1);
        if (b[0] != at) return fail(SDP_EINVAL, "phases must be contiguous and start at node 0");
        for (int r = 0; r < n; ++r) {
            if (b[r] > b[r + 1]) return fail(SDP_EINVAL, "phase partition must be non-decreasing");
            if (b[r] % unit) return fail(SDP_EINVAL, "column layout: parts must consist of whole columns");
        }
        at = b[n];
    }
    if (at != p->S) return fail(SDP_EINVAL, "phases must cover all %lld nodes", (long long)p->S);
    {
        // gather buffer for the J exchange of phases whose parts are uneven (gather_phase_of): sized now, so
        // that no backup ever allocates with collectives in flight
        size_t need = 0;
        for (int ph = 0; ph < n_phases; ++ph) {
            const int64_t *b = parts.data() + (size_t)ph * (n + 1);
            int64_t longest = 0;
            bool even = true;
            for (int r = 0; r < n; ++r) { longest = std::max(longest, b[r + 1] - b[r]); even = even && (b[r + 1] - b[r] == b[1] - b[0]); }
            if (!even) need = std::max(need, (size_t)longest * (size_t)n * real_size(p->dtype));
        }
        if (need > p->gstage_bytes) {
            int rc = p->gstage.alloc(need);
            if (rc) { p->gstage_bytes = 0; return rc; }
            p->gstage_bytes = need;
        }
    }
    for (auto &e : p->ev_phase) (void)hipEventDestroy(e);
    p->ev_phase.assign((size_t)n_phases, nullptr);
    for (auto &e : p->ev_phase) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!p->ev_comm) HIP_TRY(hipEventCreateWithFlags(&p->ev_comm, hipEventDisableTiming));
    p->parts.swap(parts);
    p->n_phases = n_phases;
    p->comm = c;
    p->comm_pending = false;
    p->sparse = false;                     // (a new partition: the need lists no longer apply)
    p->need.clear();
    p->mask_valid = false;
    p->red_V = nullptr;
    return SDP_OK;
}
