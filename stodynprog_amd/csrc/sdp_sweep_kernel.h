// sdp_sweep_kernel.h -- hand-written Bellman-backup kernels for gfx950.
//
// Included LAST by a generated-model translation unit, after it has defined
//   SDP_REAL      double | float
//   SDP_D         state dimension (1..4)
//   SDP_NU        number of control variables (1..SDP_MAXU)
//   SDP_HAS_W     1 when the system has one perturbation, 0 when deterministic
//   SDP_LANES     lanes per state node (power of two, 1..64)
//   sdp_model_cell(x, u, w, t, xn, g)   the traced dyn/cost of the user's model
//
// Kernel `sdp_sweep` is one value-iteration backup (reference
// stodynprog.py:466-534 / 639-691):
//     J_k(x) = min_u  sum_w p_w * ( g(x,u,w) + J_next( f(x,u,w) ) )
// Mapping: a wavefront owns 64/SDP_LANES consecutive state nodes; the
// SDP_LANES lanes of a node stride over its control lattice, each lane runs
// the perturbation loop sequentially in registers (expectation accumulated in
// w order), then a DPP/shuffle butterfly reduces (J, flat control index) with
// first-occurrence tie-break (np.argmin semantics, stodynprog.py:686).
// Workgroups walk contiguous chunks of the node range per XCD so that the
// value-function neighbourhoods gathered by co-resident waves share one L2.
//
// Kernel `sdp_evalpol` is one fixed-policy backup (stodynprog.py:743-762): the
// same cell evaluation with the control read from a policy array, one lane
// per node.
#pragma once
#include "sdp_device.h"

typedef SDP_REAL sdp_real;

// Diagnostic builds (-DSDP_STAMP=1, never the production code object): thread 0
// of every workgroup records the shader clock and the 100 MHz reference clock at
// kernel entry and exit into a buffer nothing else reads.
#ifndef SDP_STAMP
#define SDP_STAMP 0
#endif
#if SDP_STAMP == 1
#define SDP_STAMP_BEGIN(a)                                                             \
    if ((a).stamps && threadIdx.x == 0) {                                              \
        (a).stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();                 \
        (a).stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();             \
    }
#define SDP_STAMP_END(a)                                                               \
    if ((a).stamps && threadIdx.x == 0) {                                              \
        (a).stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();                 \
        (a).stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();             \
    }
#else
#define SDP_STAMP_BEGIN(a)
#define SDP_STAMP_END(a)
#endif

// constant address space: wave-uniform reads through it become scalar loads
// (s_load_*) whose results feed VALU instructions as SGPR operands
typedef const __attribute__((address_space(4))) sdp_real sdp_cst_real;
typedef __attribute__((address_space(3))) sdp_real sdp_lds_real;

// Control lattice of one node: per control (lo, hi, n) and numpy.linspace
// point generation (stodynprog.py:458): u_i = i*step + lo, last point == hi.
struct SdpBox {
    sdp_real lo[SDP_NU], hi[SDP_NU], step[SDP_NU], delta[SDP_NU];
    int n[SDP_NU];
    int total;
};

SDP_DEV void sdp_load_box(const SdpSweepArgs &a, int64_t node, SdpBox &b)
{
    const sdp_real *lo = (const sdp_real *)a.box_lo;
    const sdp_real *hi = (const sdp_real *)a.box_hi;
    b.total = 1;
#pragma unroll
    for (int c = 0; c < SDP_NU; ++c) {
        const int64_t at = a.box_per_node ? (int64_t)c * a.S + node : (int64_t)c;
        b.lo[c] = lo[at];
        b.hi[c] = hi[at];
        b.n[c] = a.box_n[at];
        b.delta[c] = b.hi[c] - b.lo[c];
        // numpy.linspace: step = delta / div with div = num - 1
        b.step[c] = (b.n[c] > 1) ? b.delta[c] / (sdp_real)(b.n[c] - 1) : (sdp_real)0;
        b.total *= b.n[c];
    }
}

SDP_DEV sdp_real sdp_control_value(const SdpBox &b, int c, int k)
{
    if (b.n[c] == 1) return b.lo[c];                 // single point (host stores the centre)
    if (k == b.n[c] - 1) return b.hi[c];             // y[-1] = stop
    if (b.step[c] == (sdp_real)0)                    // linspace's denormal-step branch
        return ((sdp_real)k / (sdp_real)(b.n[c] - 1)) * b.delta[c] + b.lo[c];
    return (sdp_real)k * b.step[c] + b.lo[c];
}

// flat C-order lattice index (control 0 slowest) -> control values
SDP_DEV void sdp_controls_at(const SdpBox &b, int flat, sdp_real *u)
{
#if SDP_NU == 1
    u[0] = sdp_control_value(b, 0, flat);
    return;
#endif
#pragma unroll
    for (int c = SDP_NU - 1; c >= 0; --c) {
        const int k = flat % b.n[c];
        flat /= b.n[c];
        u[c] = sdp_control_value(b, c, k);
    }
}

SDP_DEV void sdp_node_coords(const SdpSweepArgs &a, int64_t node, sdp_real *x)
{
    const sdp_real *axes = (const sdp_real *)a.axes;
    int64_t r = node;
#pragma unroll
    for (int k = SDP_D - 1; k >= 0; --k) {
        const int i = (int)(r % a.orders[k]);
        r /= a.orders[k];
        x[k] = axes[a.axis_off[k] + i];
    }
}

SDP_DEV void sdp_grid_from_args(const SdpSweepArgs &a, SdpGrid<sdp_real, SDP_D> &g)
{
    const sdp_real *axes = (const sdp_real *)a.axes;
    sdp_real smin[SDP_D], smax[SDP_D];
#pragma unroll
    for (int k = 0; k < SDP_D; ++k) {
        smin[k] = axes[a.axis_off[k]];                       // x[0]   stodynprog.py:263
        smax[k] = axes[a.axis_off[k] + a.orders[k] - 1];     // x[-1]  stodynprog.py:264
    }
    sdp_make_grid<sdp_real, SDP_D>(g, a.orders, smin, smax);
}

// expected cost of one (node, control): sum_w p_w * (g + J_next(f))
template <bool SHIFT = false>
SDP_DEV sdp_real sdp_expected_cost(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_D> &grid,
                                   const sdp_real *__restrict__ V, const sdp_real *x,
                                   const sdp_real *u, sdp_real t)
{
    sdp_real xn[SDP_D], g;
#if SDP_HAS_W
    const sdp_real *__restrict__ wgrid = (const sdp_real *)a.wgrid;
    const sdp_real *__restrict__ proba = (const sdp_real *)a.proba;
    sdp_real acc = (sdp_real)0;
    for (int wi = 0; wi < a.W; ++wi) {
        sdp_model_cell(x, u, wgrid[wi], t, xn, g);
        const sdp_real jc = g + sdp_interp_point<sdp_real, SDP_D, sdp_real, SHIFT>(V, grid, xn);   // stodynprog.py:677
        acc = acc + jc * proba[wi];                                               // stodynprog.py:681
    }
    return acc;
#else
    sdp_model_cell(x, u, (sdp_real)0, t, xn, g);
    return g + sdp_interp_point<sdp_real, SDP_D, sdp_real, SHIFT>(V, grid, xn);                    // stodynprog.py:679-680
#endif
}

#if defined(SDP_LINE)
#include "sdp_line_kernel.h"    // sdp_sweep (+ sdp_lead_reduce) for ONE state variable with noise in its sums: the filter on the shifted lattice
#elif defined(SDP_LEAD_AXES)
#include "sdp_lead_kernel.h"    // sdp_sweep (+ sdp_lead_reduce) for several controlled state variables
#else
extern "C" __global__ void __launch_bounds__(256) sdp_sweep(SdpSweepArgs a)
{
    constexpr int L = SDP_LANES;
    constexpr int NPW = 64 / L;                       // nodes per wavefront
    const int lane = threadIdx.x & 63;
    const int sub = lane & (L - 1);
    const int slot = lane / L;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int64_t tile_nodes = (int64_t)NPW * waves;  // nodes per workgroup step

    const sdp_real *__restrict__ V = (const sdp_real *)a.V;
    SdpGrid<sdp_real, SDP_D> grid;
    sdp_grid_from_args(a, grid);
    const sdp_real t = (sdp_real)a.t_k;

    // XCD-aware walk: workgroups b and b+8 share an XCD (round-robin dispatch),
    // so XCD x takes the x-th contiguous eighth of the tiles.
    const int64_t n_nodes = a.node_end - a.node_begin;
    const int64_t n_tiles = (n_nodes + tile_nodes - 1) / tile_nodes;
    const int xcd = blockIdx.x & 7;
    const int64_t per_xcd = (n_tiles + 7) / 8;
    const int64_t t_end = min((int64_t)(xcd + 1) * per_xcd, n_tiles);
    const int64_t stride = gridDim.x >> 3;

    for (int64_t tile = (int64_t)xcd * per_xcd + (blockIdx.x >> 3); tile < t_end; tile += stride) {
        const int64_t node = a.node_begin + tile * tile_nodes + (int64_t)wave * NPW + slot;
        const bool live = node < a.node_end;
        sdp_real best = INFINITY;
        int ibest = INT_MAX;
        SdpBox box;
        sdp_real x[SDP_D];
        if (live) {
            sdp_node_coords(a, node, x);
            sdp_load_box(a, node, box);
            for (int ci = sub; ci < box.total; ci += L) {
                sdp_real u[SDP_NU];
                sdp_controls_at(box, ci, u);
                const sdp_real jc = sdp_expected_cost(a, grid, V, x, u, t);
                if (ibest == INT_MAX || sdp_better_seq(jc, best)) { best = jc; ibest = ci; }
            }
        }
        sdp_seg_argmin<sdp_real, L>(best, ibest);
        if (live && sub == 0) {
            sdp_store_J<sdp_real>(a, node, 0, best);
            if (a.idx) a.idx[node] = ibest;
            if (a.pol) {
                sdp_real u[SDP_NU];
                sdp_controls_at(box, ibest, u);
#pragma unroll
                for (int c = 0; c < SDP_NU; ++c) ((sdp_real *)a.pol)[node * SDP_NU + c] = u[c];
            }
        }
    }
}

#endif  // SDP_LEAD_AXES

extern "C" __global__ void __launch_bounds__(256) sdp_evalpol(SdpSweepArgs a)
{
    const sdp_real *__restrict__ V = (const sdp_real *)a.V;
    SdpGrid<sdp_real, SDP_D> grid;
    sdp_grid_from_args(a, grid);
    const sdp_real t = (sdp_real)a.t_k;
    // fused relative-DP shift of the previous step (see SdpLerp<.., SHIFT>)
    grid.shift = a.shift_index >= 0 ? V[a.shift_index] : (sdp_real)0;
    if (a.ref_out && blockIdx.x == 0 && threadIdx.x == 0) *a.ref_out = (double)grid.shift;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t node = a.node_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
         node < a.node_end; node += stride) {
        sdp_real x[SDP_D], u[SDP_NU];
        sdp_node_coords(a, node, x);
#pragma unroll
        for (int c = 0; c < SDP_NU; ++c) u[c] = ((const sdp_real *)a.pol_in)[node * SDP_NU + c];
        sdp_store_J<sdp_real>(a, node, 0, sdp_expected_cost<true>(a, grid, V, x, u, t));
    }
}

// B closed-loop trajectories of T steps, one lane each, without leaving the GPU:
//     u_k = policy(x_k)          multilinear interpolation of every control component
//                                (MlinInterpolator.__call__, stodynprog.py:269-289 ->
//                                multilinear_cython.pyx:51-300: lerp tree in double)
//     x_k+1 = dyn(x_k, u_k, w_k)  the traced model (stodynprog.py:674 evaluates the same
//                                callable); g_k = cost(x_k, u_k, w_k)
// -- the loop every closed-loop example of the reference writes by hand
// (examples/20 Searev storage control/storage_control.py:242-251).
extern "C" __global__ void __launch_bounds__(64) sdp_simulate(SdpSimArgs a)
{
    SdpGrid<sdp_real, SDP_D> grid;
    {
        const sdp_real *axes = (const sdp_real *)a.axes;
        sdp_real smin[SDP_D], smax[SDP_D];
#pragma unroll
        for (int k = 0; k < SDP_D; ++k) {
            smin[k] = axes[a.axis_off[k]];
            smax[k] = axes[a.axis_off[k] + a.orders[k] - 1];
        }
        sdp_make_grid<sdp_real, SDP_D>(grid, a.orders, smin, smax);
    }
    const sdp_real *__restrict__ pol = (const sdp_real *)a.pol;
    const sdp_real *__restrict__ wseq = (const sdp_real *)a.w;
    sdp_real *__restrict__ xo = (sdp_real *)a.x;
    sdp_real *__restrict__ uo = (sdp_real *)a.u;
    sdp_real *__restrict__ go = (sdp_real *)a.g;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += stride) {
        sdp_real x[SDP_D];
#pragma unroll
        for (int k = 0; k < SDP_D; ++k) {
            x[k] = ((const sdp_real *)a.x0)[k * a.B + b];
            xo[k * a.B + b] = x[k];
        }
        for (int64_t step = 0; step < a.T; ++step) {
            sdp_real u[SDP_NU], xn[SDP_D], g;
#pragma unroll
            for (int c = 0; c < SDP_NU; ++c)
                u[c] = sdp_interp_point<sdp_real, SDP_D, double>(pol + c * a.S, grid, x);
            const sdp_real w = wseq ? wseq[step * a.B + b] : (sdp_real)0;
            sdp_model_cell(x, u, w, (sdp_real)(a.t0 + (double)step), xn, g);
#pragma unroll
            for (int c = 0; c < SDP_NU; ++c) uo[(step * SDP_NU + c) * a.B + b] = u[c];
            if (go) go[step * a.B + b] = g;
#pragma unroll
            for (int k = 0; k < SDP_D; ++k) {
                x[k] = xn[k];
                xo[((step + 1) * SDP_D + k) * a.B + b] = xn[k];
            }
        }
    }
}

// what this code object was generated for (sdp_kernel_args.h, SDP_META_*); units that include
// sdp_column_kernel.h define it at the end of that file, where the column macros are complete
#if !defined(SDP_COL_N0)
extern "C" {
__constant__ int32_t sdp_meta[SDP_META_WORDS] = {
    SDP_META_MAGIC, (int32_t)sizeof(sdp_real), SDP_D, SDP_NU, SDP_HAS_W, 0, 0,
#if defined(SDP_LINE)
    SDP_LINE_W,
#else
    1,
#endif
#if defined(SDP_LINE)
    // (one state variable, the filter on the shifted lattice: sdp_lead_reduce fills aux_a with 2 S + 64 (A', B') pairs at most)
    SDP_META_F_LEAD | SDP_META_F_FILTER | SDP_META_F_SHIFT | SDP_META_F_PEER_STORES, 0, 0, 256, 0, 1, 0,
#elif defined(SDP_STG_THREADS)
    SDP_META_F_STAGED | SDP_META_F_PEER_STORES, 0, 0, SDP_STG_THREADS, 0, 0, 0,
#elif defined(SDP_LEAD_AXES)
    SDP_META_F_LEAD | SDP_META_F_FILTER | SDP_META_F_PEER_STORES, 0, 0, 256, 0, SDP_LEAD_AXES,
    // (state variable of logical axis j in nibble j: the host sizes the plane-major arrays with it)
    SDP_LP[0] | (SDP_LP[1] << 4) | (SDP_LP[2] << 8) | (SDP_LP[3] << 12),
#else
    SDP_META_F_PEER_STORES, 0, 0, 256, 0, 0, 0,
#endif
    0};
}
#endif
