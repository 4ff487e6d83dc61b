// sdp_colu_kernel.h -- the column kernels with a TABLE PER CONTROL (SDP_TRAIL_HAS_U: the trailing next states depend on
// the control but not on x0; docs/NOTEBOOK.md section 3.3b).  Included by sdp_column_kernel.h, which defines the building blocks.
#pragma once
// ---------------------------------------------------------------------------
// Trailing next states that depend on the control (but not on x0), e.g. a stock
// whose use also moves the exogenous-looking process:
//     x0' = f0(x, u[, w])      xk' = fk(x1.., u, w),  k >= 1
// inner(r) of the header comment then depends on (r, u, w).  The nodes of a column
// still share it control by control PROVIDED they share the control values (the
// admissible box does not depend on x0; checked on the host), so the workgroup
// loops over the controls and, for each one, rebuilds the W x N0 table (phases W
// and A: 2^(d-1) coalesced strip reads + the trailing lerps per entry) and runs
// phase B for its nodes: per lattice cell the table costs as much as it saves in
// a gather kernel's 2^d scattered reads and full lerp nest (about 15 operations
// and 2^(d-1) coalesced loads instead of about 100 operations), and nothing else
// changes: same operations on the same operands in the same order, argmin in
// control order in-lane.
// Shape: the workgroup has one thread per node of its unit (blockDim.x = nodes
// of a column, at most SDP_COL_THREADS; longer columns are split), so every lane
// works in phase B; the table holds SDP_COL_WCHUNK perturbation points at a time
// (about 32 KiB), which lets four workgroups share a CU and hide each other's
// barriers and load latencies.  Measured on the control-coupled benchmark
// (256^3 x 64 x 32 fp64): 65.6 ms against 92.4 ms for the staged tile kernel and 549 ms
// for the direct one; 24 vector instructions per cell, but 32 B per cell of strip reads:
// 2.15e9 vector loads and 315 GB of L2 misses per sweep (60 % L2 hit rate) bound it.
// (Taking the columns in 16 x 16 blocks of the (axis 1, axis 2) plane instead of
// row by row changed that by 1 % while the workgroups strode over the units; with the units
// claimed in order -- sdp_col_of_unit -- 8 x 8 tiles give 65.6 -> 59.5 ms.)

// phase A for the perturbation points w_lo .. w_lo+cnt-1: T[(w - w_lo)][r], all rows
SDP_DEV void sdp_colu_phase_a(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                              const SdpColShared &s, int w_lo, int cnt)
{
    constexpr int N0 = SDP_COL_N0;
    constexpr int G = SDP_COLU_A_GROUP;
    constexpr int NV = 1 << SDP_DT;
    const sdp_real *__restrict__ V = (const sdp_real *)a.V;
#if SDP_COLU_WIDE_LOADS
    // 16-byte vertex loads (cf. SDP_COL_A_WIDE_LOADS): a thread takes RPL adjacent rows, and the threads beyond
    // N0 / RPL take other perturbation points of the chunk: as many entries per thread, 1 / RPL of the
    // vector-memory instructions -- this kernel is bound by its strip reads
    {
        constexpr int RPL = 16 / (int)sizeof(sdp_real);
        static_assert(N0 % RPL == 0, "wide loads: whole groups of rows");
        typedef sdp_real sdp_rows __attribute__((ext_vector_type(RPL)));
        constexpr int ROWG = N0 / RPL;                           // row groups of a column
        const int lanes_r = min((int)blockDim.x, ROWG);
        const int wgroups = max((int)blockDim.x / lanes_r, 1);    // thread groups along w
        const int wg = threadIdx.x / lanes_r;
        for (int rg = threadIdx.x - wg * lanes_r; rg < ROWG && wg < wgroups; rg += lanes_r) {
            const int r = rg * RPL;
            for (int w0 = wg * G; w0 < cnt; w0 += wgroups * G) {
                sdp_rows vals2[G][NV];
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int w = w_lo + min(w0 + j, cnt - 1);           // clamp: result unused
                    int base[NV];
#pragma unroll
                    for (int q = 0; q < NV; ++q) {
                        int o = 0;
#pragma unroll
                        for (int k = 0; k < SDP_DT; ++k)
                            o += s.w_off[w * SDP_DT + k] + (((q >> (SDP_DT - 1 - k)) & 1) ? tg.M[k] : 0);
                        base[q] = o;
                    }
#pragma unroll
                    for (int q = 0; q < NV; ++q) vals2[j][q] = *(const sdp_rows *)(V + r + base[q]);
                }
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    if (w0 + j < cnt) {
                        const int w = w_lo + w0 + j;
                        sdp_real lam[SDP_DT], oml[SDP_DT];
#pragma unroll
                        for (int k = 0; k < SDP_DT; ++k) {
                            lam[k] = s.w_lam[w * SDP_DT + k];
                            oml[k] = s.w_oml[w * SDP_DT + k];
                        }
                        sdp_rows e;
#pragma unroll
                        for (int c = 0; c < RPL; ++c) {
                            sdp_real one[NV];
#pragma unroll
                            for (int q = 0; q < NV; ++q) one[q] = vals2[j][q][c];
                            e[c] = SdpColNest<0, false>::run(one, lam, oml, tg.shift);
                        }
                        *(sdp_rows *)(s.T + (w0 + j) * N0 + r) = e;
                    }
                }
            }
        }
        return;
    }
#endif
    // consecutive threads = consecutive rows (coalesced strips); a thread keeps its row and
    // takes G consecutive perturbation points per round: their strips overlap (L1 hits)
    for (int r = threadIdx.x; r < N0; r += blockDim.x) {
        for (int w0 = 0; w0 < cnt; w0 += G) {
            sdp_real vals[G][NV];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int w = w_lo + min(w0 + j, cnt - 1);               // clamp: result unused
                int off[SDP_DT];
#pragma unroll
                for (int k = 0; k < SDP_DT; ++k) off[k] = s.w_off[w * SDP_DT + k];
                SdpColGather<0>::run(V + r, tg, off, 0, vals[j]);
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (w0 + j < cnt) {
                    const int w = w_lo + w0 + j;
                    sdp_real lam[SDP_DT], oml[SDP_DT];
#pragma unroll
                    for (int k = 0; k < SDP_DT; ++k) {
                        lam[k] = s.w_lam[w * SDP_DT + k];
                        oml[k] = s.w_oml[w * SDP_DT + k];
                    }
                    s.T[(w0 + j) * N0 + r] = SdpColNest<0, false>::run(vals[j], lam, oml, tg.shift);
                }
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(SDP_COL_THREADS, SDP_COL_MIN_WAVES) sdp_sweep_col(SdpSweepArgs a)
{
    __shared__ SdpColLds sdp_lds;
    SDP_STAMP_BEGIN(a);
    constexpr int N0 = SDP_COL_N0;
    constexpr int Wn = SDP_COL_W;
    constexpr int WC = SDP_COL_WCHUNK;
    sdp_trap_unless(a.n_lead == N0 && (SDP_HAS_W ? a.W : 1) == Wn);
    const sdp_real t = (sdp_real)a.t_k;
    const sdp_real *__restrict__ axis0 = (const sdp_real *)a.axes + a.axis_off[0];
    SdpColShared s;
    sdp_col_carve(sdp_lds, s);
    SdpGrid<sdp_real, SDP_DT> tg;
    sdp_col_trailing_grid(a, tg);
    SdpLeadAxis l;
    sdp_col_lead_axis(a, l);
    SdpColWalk walk;
    sdp_col_walk(a, walk);
    SdpColWeights k;
    sdp_col_load_weights(a, k, sdp_lds.pw, sdp_lds.gw);
    const volatile sdp_lds_real *T = (const volatile sdp_lds_real *)s.T;

    // units of this XCD's share, claimed in order (see sdp_col_of_unit): neighbouring columns at
    // the same time, whatever the workgroups' speeds
    const int64_t u_base = walk.unit - (blockIdx.x >> 3), u_end = walk.end;
    unsigned int *claim = a.claim + 32 * (blockIdx.x & 7);
    for (;;) {
        __syncthreads();                                   // (everybody has read next_unit)
        if (threadIdx.x == 0) sdp_lds.next_unit = (int)atomicAdd(claim, 1u);
        __syncthreads();
        const int64_t unit = u_base + sdp_lds.next_unit;
        if (unit >= u_end) break;
        const int64_t col = sdp_col_of_unit(a, unit);
        const int part = (int)((unsigned)unit % (unsigned)a.col_splits);      // (32-bit: units < 2^31)
        const int i_lo = (int)((unsigned)(N0 * part) / (unsigned)a.col_splits);
        const int i_hi = (int)((unsigned)(N0 * (part + 1)) / (unsigned)a.col_splits);
        sdp_real x[SDP_D];
        sdp_col_coords(a, col, x);
        // the controls of the column (every node of it has this box)
        SdpBox box;
        sdp_load_box(a, col * N0 + i_lo, box);
        const int i = i_lo + (int)threadIdx.x;
        const bool mine = i < i_hi;
        if (mine) x[0] = axis0[i];
        sdp_real best = INFINITY;
        int ibest = INT_MAX;
        for (int ci = 0; ci < box.total; ++ci) {
            sdp_real u[SDP_NU];
            sdp_controls_at(box, ci, u);
            // (phase W overwrites what phase A of the previous control read: every thread is past
            // that phase's closing barrier; the table itself is protected by the chunk loop's)
            sdp_col_phase_w(a, tg, s, x, u, t);         // its inputs x[1..], u, w are workgroup-uniform
            // what the node's cells share: the cell of x0' and the cost, when they do not depend on w
            sdp_real lam0 = 0, oml0 = 0, g = 0, acc = (sdp_real)0;
            int q0 = 0;
#define SDP_COLU_LOCATE(wval)                                                           \
            {                                                                          \
                const sdp_real xn0_ = sdp_model_lead(x, u, (wval), t);                 \
                const sdp_real sn_ = sdp_div_span<sdp_real>(xn0_ - l.smin, l.span, l.rspan, l.pow2); /* pyx:75 */   \
                const sdp_real p_ = sn_ * l.nm1;                                       \
                q0 = max(min(sdp_trunc_i32(p_), l.ordm2), 0);           /* pyx:78 */   \
                lam0 = p_ - (sdp_real)q0;                               /* pyx:81 */   \
                oml0 = (sdp_real)1 - lam0;                                             \
            }
            if (mine) {
#if !SDP_LEAD_HAS_W || !SDP_HAS_W
                SDP_COLU_LOCATE((sdp_real)0)
#endif
#if !SDP_COST_HAS_W || !SDP_HAS_W
                g = sdp_model_cost(x, u, (sdp_real)0, t);
#endif
            }
            for (int w_lo = 0; w_lo < Wn; w_lo += WC) {
                const int cnt = min(WC, Wn - w_lo);
                __syncthreads();                        // phase W done / previous chunk's readers done
                sdp_colu_phase_a(a, tg, s, w_lo, cnt);
                __syncthreads();
                if (mine) {
#if SDP_HAS_W
#pragma unroll 4
                    for (int j = 0; j < cnt; ++j) {
                        const int w = w_lo + j;
#if SDP_LEAD_HAS_W
                        SDP_COLU_LOCATE(SDP_COL_GW(k, w))
#endif
                        const sdp_real lo = T[j * N0 + q0];
                        const sdp_real hi = T[j * N0 + q0 + 1];
                        const sdp_real val = oml0 * lo + lam0 * hi;               // pyx:88-300
#if SDP_COST_HAS_W
                        g = sdp_model_cost(x, u, SDP_COL_GW(k, w), t);
#endif
                        const sdp_real jc = g + val;                              // stodynprog.py:677
                        acc = acc + jc * SDP_COL_PW(k, w);                        // stodynprog.py:681
                    }
#else
                    const sdp_real lo = T[q0];
                    const sdp_real hi = T[q0 + 1];
                    acc = g + (oml0 * lo + lam0 * hi);                            // stodynprog.py:679-680
#endif
                }
            }
#undef SDP_COLU_LOCATE
            if (mine && (ibest == INT_MAX || sdp_better_seq(acc, best))) { best = acc; ibest = ci; }
        }
        if (mine) sdp_col_store(a, col * N0 + i, box, best, ibest);
    }
    if (threadIdx.x == 0) {                                // the last workgroup leaves the counters at zero
        __threadfence();
        if (atomicAdd(a.claim + 256, 1u) == gridDim.x - 1) {
            for (int q = 0; q < 8; ++q) atomicExch(a.claim + 32 * q, 0u);
            atomicExch(a.claim + 256, 0u);
        }
    }
    SDP_STAMP_END(a);
}

// fixed-policy backup: every node has its own control, so nothing is shared -- one
// lane per node gathers its 2^d vertices from the column-ordered value array
// (sdp_expected_cost of sdp_sweep_kernel.h with this layout's strides)
extern "C" __global__ void __launch_bounds__(SDP_COL_THREADS) sdp_evalpol_col(SdpSweepArgs a)
{
    constexpr int N0 = SDP_COL_N0;
    sdp_trap_unless(a.n_lead == N0);
    const sdp_real *__restrict__ V = (const sdp_real *)a.V;
    const sdp_real *__restrict__ axis0 = (const sdp_real *)a.axes + a.axis_off[0];
    SdpGrid<sdp_real, SDP_D> grid;
    sdp_grid_from_args(a, grid);
    {   // strides of the axis-0-fastest order
        int m = N0;
        grid.M[0] = 1;
#pragma unroll
        for (int k = SDP_D - 1; k >= 1; --k) { grid.M[k] = m; m *= a.orders[k]; }
    }
    const sdp_real t = (sdp_real)a.t_k;
    grid.shift = a.shift_index >= 0 ? V[a.shift_index] : (sdp_real)0;
    if (a.ref_out && blockIdx.x == 0 && threadIdx.x == 0) *a.ref_out = (double)grid.shift;
    const int64_t first = a.col_begin * N0, last = a.col_end * N0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t node = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; node < last; node += stride) {
        sdp_real x[SDP_D], u[SDP_NU];
        sdp_col_coords(a, node / N0, x);
        x[0] = axis0[node % N0];
#pragma unroll
        for (int c = 0; c < SDP_NU; ++c) u[c] = ((const sdp_real *)a.pol_in)[node * SDP_NU + c];
        sdp_store_J<sdp_real>(a, node, node / N0, sdp_expected_cost<true>(a, grid, V, x, u, t));
    }
}
