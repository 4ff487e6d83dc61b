// sdp_staged_kernel.h -- Bellman backup for ANY traceable model with the value
// function's neighbourhood staged in LDS (the design BASELINE.json:north_star
// names: "stages the value-function stencil neighbourhood into LDS for the
// multilinear gather").  The reference gathers 2^d vertices per lattice cell
// from the whole value array (multilinear_cython.pyx:143-208, called per node
// from stodynprog.py:677); sdp_sweep (sdp_sweep_kernel.h) does the same with one
// global load per vertex and is bound by the issue of those scattered loads.
//
// Here a workgroup owns a TILE of the state grid (a box of SDP_STG_T0 x .. nodes,
// one lane per node, lanes consecutive along the last axis) and walks the
// (control x perturbation) lattice of its nodes in CHUNKS of SDP_STG_CU controls
// x SDP_STG_CW perturbation points.  Per chunk:
//   1. reach     every lane evaluates the dynamics at the corners of the chunk
//                and turns the next states into cell indices; a wave butterfly
//                and LDS atomics give the bounding box of the tile's next
//                states (plus one cell of margin);
//   2. stage     the value sub-block of that box is copied to LDS, rows along
//                the last axis (contiguous in memory) by consecutive lanes;
//   3. gather    every lane runs its chunk of lattice cells with the 2^d vertex
//                reads served by LDS: the same nested lerp, the same operands,
//                the same order as sdp_sweep -- results are bit-identical.
// The box of step 1 is only a PREDICTION (exact for dynamics monotone in u and
// w over a chunk, which is what storage models are): a control with a cell
// whose vertices are not all inside the staged box is evaluated again with every
// vertex read from global memory (same arithmetic), so the result never depends
// on the prediction; a box larger than the LDS budget is cut to fit and the
// rest falls back the same way.
// The expectation is accumulated in w order per control, the argmin runs in
// control order in-lane (first occurrence wins): stodynprog.py:681,686.
//
// Needs from the generated unit, besides what sdp_sweep_kernel.h needs:
//   SDP_STG_THREADS              workgroup size = nodes per tile (multiple of 64)
//   SDP_STG_T0 .. SDP_STG_T3     tile shape (product = SDP_STG_THREADS; unused axes 1)
//   SDP_STG_CU, SDP_STG_CW       chunk: controls x perturbation points per staged box
//   SDP_STG_CAP                  LDS budget of the box, in reals
#pragma once
#include "sdp_sweep_kernel.h"

#ifdef SDP_STG_THREADS

constexpr int SDP_STG_TILE[4] = {SDP_STG_T0, SDP_STG_T1, SDP_STG_T2, SDP_STG_T3};

SDP_DEV int sdp_stg_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// nested lerp of SdpLerp<real, D, real> (sdp_device.h) with the vertices read
// from the staged box: `ls[k]` = LDS stride of axis k, `base` = LDS index of the
// cell's lower corner
template <int K>
struct SdpStgLerp {
    static SDP_DEV sdp_real eval(const sdp_real *box, const int *ls, const sdp_real *lam,
                                 const sdp_real *oml, int base)
    {
        const sdp_real lo = SdpStgLerp<K + 1>::eval(box, ls, lam, oml, base);
        const sdp_real hi = SdpStgLerp<K + 1>::eval(box, ls, lam, oml, base + ls[K]);
        return oml[K] * lo + lam[K] * hi;                          // pyx:88,140,208,300
    }
};
template <>
struct SdpStgLerp<SDP_D - 1> {
    static SDP_DEV sdp_real eval(const sdp_real *box, const int *, const sdp_real *lam,
                                 const sdp_real *oml, int base)
    {
        const sdp_real lo = box[base];
        const sdp_real hi = box[base + 1];
        return oml[SDP_D - 1] * lo + lam[SDP_D - 1] * hi;
    }
};

struct SdpStgShared {
    int red[2][2 * SDP_D];      // per parity: min cell index, min of minus the max cell index
    int n_ctrl;                 // largest control count of the tile's nodes
};

// cell index of coordinate s along axis k for the PREDICTION only (a product
// with the reciprocal instead of the reference's division: a cell off by one is
// covered by the margin, and a wrong box costs time, never correctness)
SDP_DEV int sdp_stg_guess_cell(const SdpGrid<sdp_real, SDP_D> &g, const sdp_real *rspan, int k,
                               sdp_real s)
{
    const sdp_real p = (s - g.smin[k]) * rspan[k];
    return max(min(sdp_trunc_i32(p), g.ordm2[k]), 0);
}

// LDS row stride (in reals) of a box whose rows hold `len` vertices of the last
// axis.  A wavefront's lanes are TL = SDP_STG_TILE[last] consecutive nodes along
// the last axis times 64 / TL nodes along the axis before it: their vertices sit
// in neighbouring rows at neighbouring columns.  With the stride congruent to TL
// modulo 2 TL the rows of one 32-lane group land on disjoint sets of the 32
// 8-byte banks (TL = 8: rows at banks 0, 8, 16, 24 or 0, 24, 16, 8); an odd
// stride is used when a row of lanes already covers a whole group.
SDP_DEV int sdp_stg_row_stride(int len)
{
    constexpr int TL = SDP_STG_TILE[SDP_D - 1];
    if (SDP_D == 1 || TL >= 32 || sizeof(sdp_real) != 8) return len | 1;
    return (len + TL - 1) / (2 * TL) * (2 * TL) + TL;
}

// sdp_locate_axis (sdp_device.h; pyx:75-81) that also returns the cell index
SDP_DEV int sdp_stg_locate(const SdpGrid<sdp_real, SDP_D> &g, int k, sdp_real s,
                           SdpCell<sdp_real, SDP_D, sdp_real> &c)
{
    const sdp_real sn = sdp_div_span<sdp_real>(s - g.smin[k], g.span[k], g.rspan[k], (g.pow2 >> k) & 1);   // pyx:75
    const sdp_real p = sn * g.nm1[k];
    const int q = max(min(sdp_trunc_i32(p), g.ordm2[k]), 0);       // pyx:78
    c.lam[k] = p - (sdp_real)q;                                    // pyx:81 (unclamped)
    c.oml[k] = (sdp_real)1 - c.lam[k];
    c.off[k] = g.M[k] * q;
    return q;
}

#if defined(SDP_WAVES_CAP)      // (a rebuild with more registers: codegen.spill_hazards, _native.compile_model)
#define SDP_STG_MIN_WAVES ((SDP_STG_THREADS / 128) < SDP_WAVES_CAP ? (SDP_STG_THREADS / 128) : SDP_WAVES_CAP)
#else
#define SDP_STG_MIN_WAVES (SDP_STG_THREADS / 128)
#endif
extern "C" __global__ void __launch_bounds__(SDP_STG_THREADS, SDP_STG_MIN_WAVES) sdp_sweep_lds(SdpSweepArgs a)
{
    __shared__ sdp_real sdp_box[SDP_STG_CAP];
    __shared__ SdpStgShared sh;
    SDP_STAMP_BEGIN(a);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int WAVES = SDP_STG_THREADS / 64;
    const sdp_real *__restrict__ V = (const sdp_real *)a.V;
    SdpGrid<sdp_real, SDP_D> grid;
    sdp_grid_from_args(a, grid);
    sdp_real rspan[SDP_D];
#pragma unroll
    for (int k = 0; k < SDP_D; ++k) rspan[k] = grid.nm1[k] / grid.span[k];
    const sdp_real t = (sdp_real)a.t_k;
#if SDP_HAS_W
    // perturbation points and weights: wave-uniform, read with scalar loads
    const sdp_cst_real *wgrid = (const sdp_cst_real *)a.wgrid;
    const sdp_cst_real *proba = (const sdp_cst_real *)a.proba;
    const int Wn = a.W;
#else
    const int Wn = 1;
#endif

    // position of this lane's node inside the tile (last axis fastest)
    int in_tile[SDP_D];
    {
        int r = tid;
#pragma unroll
        for (int k = SDP_D - 1; k >= 0; --k) {
            in_tile[k] = r % SDP_STG_TILE[k];
            r /= SDP_STG_TILE[k];
        }
    }
    int tiles_per[SDP_D];
    int64_t n_tiles = 1;
#pragma unroll
    for (int k = 0; k < SDP_D; ++k) {
        tiles_per[k] = (a.orders[k] + SDP_STG_TILE[k] - 1) / SDP_STG_TILE[k];
        n_tiles *= tiles_per[k];
    }
    // XCD-aware walk (see sdp_sweep): XCD x takes the x-th contiguous eighth of the tiles
    const int xcd = blockIdx.x & 7;
    const int64_t per_xcd = (n_tiles + 7) / 8;
    const int64_t t_end = min((int64_t)(xcd + 1) * per_xcd, n_tiles);
    const int64_t stride = gridDim.x >> 3;
    if (tid < 2 * SDP_D) { sh.red[0][tid] = INT_MAX; sh.red[1][tid] = INT_MAX; }
    int parity = 0;

    for (int64_t tile = (int64_t)xcd * per_xcd + (blockIdx.x >> 3); tile < t_end; tile += stride) {
        // ---- this lane's node
        int64_t node = 0;
        bool live = true;
        sdp_real x[SDP_D];
        {
            int64_t r = tile;
            int torg[SDP_D];
#pragma unroll
            for (int k = SDP_D - 1; k >= 0; --k) {
                torg[k] = (int)(r % tiles_per[k]) * SDP_STG_TILE[k];
                r /= tiles_per[k];
            }
            const sdp_real *axes = (const sdp_real *)a.axes;
#pragma unroll
            for (int k = 0; k < SDP_D; ++k) {
                const int i = torg[k] + in_tile[k];
                live = live && i < a.orders[k];
                const int ic = min(i, a.orders[k] - 1);
                node = node * a.orders[k] + ic;
                x[k] = axes[a.axis_off[k] + ic];
            }
        }
        live = live && node >= a.node_begin && node < a.node_end;
        SdpBox box;
        sdp_load_box(a, node, box);                       // (clamped node: always a valid read)
        const int total = live ? box.total : 0;
        if (tid == 0) sh.n_ctrl = 0;
        __syncthreads();
        {
            const int m = sdp_wave_max(total);
            if (lane == 0 && m > 0) atomicMax(&sh.n_ctrl, m);
        }
        __syncthreads();
        const int n_ctrl = sdp_stg_uniform(sh.n_ctrl);

        sdp_real best = INFINITY;
        int ibest = INT_MAX;
        for (int c0 = 0; c0 < n_ctrl; c0 += SDP_STG_CU) {
            sdp_real acc[SDP_STG_CU];
            bool outside[SDP_STG_CU];       // a cell of this control missed the staged box
#pragma unroll
            for (int j = 0; j < SDP_STG_CU; ++j) { acc[j] = (sdp_real)0; outside[j] = false; }
            for (int w0 = 0; w0 < Wn; w0 += SDP_STG_CW) {
                const int w1 = min(w0 + SDP_STG_CW, Wn);
                // ---- 1. reach of the chunk: corners (first / last control) x (first / last w)
                int lo[SDP_D], nhi[SDP_D];
#pragma unroll
                for (int k = 0; k < SDP_D; ++k) { lo[k] = INT_MAX; nhi[k] = INT_MAX; }
                if (c0 < total) {
                    const int c_last = min(c0 + SDP_STG_CU, total) - 1;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        sdp_real u[SDP_NU];
                        sdp_controls_at(box, cc ? c_last : c0, u);
#pragma unroll
                        for (int ww = 0; ww < 2; ++ww) {
                            sdp_real xn[SDP_D], g;
#if SDP_HAS_W
                            sdp_model_cell(x, u, wgrid[ww ? w1 - 1 : w0], t, xn, g);
#else
                            sdp_model_cell(x, u, (sdp_real)0, t, xn, g);
#endif
#pragma unroll
                            for (int k = 0; k < SDP_D; ++k) {
                                const int q = sdp_stg_guess_cell(grid, rspan, k, xn[k]);
                                lo[k] = min(lo[k], q);
                                nhi[k] = min(nhi[k], -q);
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < SDP_D; ++k) {
                    lo[k] = sdp_wave_min(lo[k]);
                    nhi[k] = sdp_wave_min(nhi[k]);
                }
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < SDP_D; ++k) {
                        if (lo[k] != INT_MAX) {
                            atomicMin(&sh.red[parity][k], lo[k]);
                            atomicMin(&sh.red[parity][SDP_D + k], nhi[k]);
                        }
                    }
                }
                __syncthreads();       // (also: every lane is done reading the previous box)
                // box of the chunk: origin / extent in vertices per axis, one cell of margin,
                // cut to the LDS budget (axis 0 first); wave-uniform, kept in SGPRs
                int org[SDP_D], ext[SDP_D], ls[SDP_D];
                bool any = sdp_stg_uniform(sh.red[parity][0]) != INT_MAX;
#pragma unroll
                for (int k = 0; k < SDP_D; ++k) {
                    const int qlo = sdp_stg_uniform(sh.red[parity][k]);
                    const int qhi = -sdp_stg_uniform(sh.red[parity][SDP_D + k]);
                    org[k] = any ? max(qlo - 1, 0) : 0;
                    const int top = any ? min(qhi + 1, a.orders[k] - 2) : 0;
                    ext[k] = any ? top + 2 - org[k] : 0;         // vertices org .. top+1
                }
                if (tid < 2 * SDP_D) sh.red[parity ^ 1][tid] = INT_MAX;     // for the next chunk
                // LDS strides: the last axis is padded to an odd row length (bank spread).
                // A box beyond the budget is cut (axis 0 first, the last axis last); cells
                // that then fall outside read global memory.
                int rowlen = sdp_stg_row_stride(ext[SDP_D - 1]);
                {
                    int64_t vol = rowlen;
#pragma unroll
                    for (int k = 0; k < SDP_D - 1; ++k) vol *= ext[k];
#pragma unroll
                    for (int k = 0; k < SDP_D - 1; ++k) {
                        if (vol > SDP_STG_CAP) {
                            const int64_t other = vol / ext[k];
                            const int fit = max(2, (int)(SDP_STG_CAP / other));
                            if (fit < ext[k]) { ext[k] = fit; vol = other * fit; }
                        }
                    }
                    if (vol > SDP_STG_CAP) {
                        const int64_t other = vol / rowlen;
                        const int fit = (int)(SDP_STG_CAP / other);
                        if (fit < 3) any = false;                 // nothing useful fits
                        else { rowlen = fit; ext[SDP_D - 1] = min(ext[SDP_D - 1], fit); }
                    }
                }
                ls[SDP_D - 1] = 1;
#pragma unroll
                for (int k = SDP_D - 2; k >= 0; --k) ls[k] = (k == SDP_D - 2) ? rowlen : ls[k + 1] * ext[k + 1];
                // ---- 2. stage: rows of the box (all axes but the last), consecutive lanes along a row
                if (any) {
                    int n_rows = 1;
#pragma unroll
                    for (int k = 0; k < SDP_D - 1; ++k) n_rows *= ext[k];
                    const int len = ext[SDP_D - 1];
                    float inv_ext[SDP_D];
#pragma unroll
                    for (int k = 0; k < SDP_D; ++k) inv_ext[k] = 1.0f / (float)ext[k];
                    // P lanes per row (power of two >= len, at most 64), 64 / P rows per wave step
                    int P = 1;
                    while (P < len && P < 64) P <<= 1;
                    const int rows_per_wave = 64 / P;
                    const int sub = lane / P, col0 = lane & (P - 1);
                    for (int r0 = wave * rows_per_wave; r0 < n_rows; r0 += WAVES * rows_per_wave) {
                        const int row = r0 + sub;
                        if (row < n_rows) {
                            // row -> indices along axes 0 .. d-2 (small numbers: exact in float)
                            int rem = row, gofs = org[SDP_D - 1], lofs = 0;
#pragma unroll
                            for (int k = SDP_D - 2; k >= 0; --k) {
                                const int e = ext[k];
                                int qk = (int)(((float)rem + 0.5f) * inv_ext[k]);
                                int j = rem - qk * e;
                                if (j < 0) { --qk; j += e; } else if (j >= e) { ++qk; j -= e; }
                                rem = qk;
                                gofs += (org[k] + j) * grid.M[k];
                                lofs += j * ls[k];
                            }
                            for (int col = col0; col < len; col += P)
                                sdp_box[lofs + col] = V[gofs + col];
                        }
                    }
                }
                __syncthreads();
                // ---- 3. the chunk's lattice cells, vertices from the box.  The perturbation
                // loop is the outer one and the chunk's controls are unrolled inside it: what
                // depends on w alone (an exogenous axis' cell) is then computed once for all
                // of them, what depends on the control alone (a stock's cell) once per chunk
                // (the compiler sees identical expressions / loop invariants).  Lanes past
                // their node's control count compute a clamped control and drop the result.
                {
                    sdp_real uj[SDP_STG_CU][SDP_NU];
                    bool on[SDP_STG_CU];
#pragma unroll
                    for (int j = 0; j < SDP_STG_CU; ++j) {
                        on[j] = c0 + j < total;
                        sdp_controls_at(box, min(c0 + j, max(total, 1) - 1), uj[j]);
                    }
                    for (int wi = w0; wi < w1; ++wi) {
#if SDP_HAS_W
                        const sdp_real wv = wgrid[wi], pw = proba[wi];
#else
                        const sdp_real wv = (sdp_real)0;
#endif
#pragma unroll
                        for (int j = 0; j < SDP_STG_CU; ++j) {
                            sdp_real xn[SDP_D], g;
                            sdp_model_cell(x, uj[j], wv, t, xn, g);
                            SdpCell<sdp_real, SDP_D, sdp_real> c;
                            // vertex q .. q+1 inside org .. org+ext-1 on every axis?
                            bool inside = any;
                            int lbase = 0;
#pragma unroll
                            for (int k = 0; k < SDP_D; ++k) {
                                const int q = sdp_stg_locate(grid, k, xn[k], c);
                                const int rel = q - org[k];
                                inside = inside && (unsigned)rel < (unsigned)(ext[k] - 1);
                                lbase += rel * ls[k];
                            }
                            // a cell outside the box reads vertex 0 of it (in bounds, value unused)
                            // and marks its control, which is recomputed from global memory below
                            outside[j] = outside[j] || !inside;
                            const sdp_real val = SdpStgLerp<0>::eval(sdp_box, ls, c.lam, c.oml,
                                                                     inside ? lbase : 0);
                            const sdp_real jc = g + val;                         // stodynprog.py:677
#if SDP_HAS_W
                            if (on[j]) acc[j] = acc[j] + jc * pw;                // stodynprog.py:681
#else
                            if (on[j]) acc[j] = jc;                              // stodynprog.py:679-680
#endif
                        }
                    }
                }
                parity ^= 1;
            }
            // controls with a cell outside their staged boxes: the whole expectation again,
            // every vertex from global memory (sdp_expected_cost of sdp_sweep_kernel.h: the
            // same operations in the same order)
#pragma unroll
            for (int j = 0; j < SDP_STG_CU; ++j) {
                if (outside[j] && c0 + j < total) {
                    sdp_real u[SDP_NU];
                    sdp_controls_at(box, c0 + j, u);
                    acc[j] = sdp_expected_cost(a, grid, V, x, u, t);
                }
            }
            // ---- argmin in control order (first occurrence wins, stodynprog.py:686)
#pragma unroll
            for (int j = 0; j < SDP_STG_CU; ++j) {
                const int ci = c0 + j;
                if (ci < total && (ibest == INT_MAX || sdp_better_seq(acc[j], best))) { best = acc[j]; ibest = ci; }
            }
        }
        if (live) {
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
    SDP_STAMP_END(a);
}

#endif  // SDP_STG_THREADS
