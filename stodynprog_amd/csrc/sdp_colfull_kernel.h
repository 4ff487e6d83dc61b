// sdp_colfull_kernel.h -- the FULL-TABLE form of the column kernels: sdp_sweep_col / sdp_evalpol_col with the whole W x rows
// table of a column (or a row window of it) in LDS.  Included by sdp_column_kernel.h, which defines every building block
// (table build, reduction, first and second pass, stores, unit claiming); the other two forms are sdp_colres_kernel.h
// (the table a chunk of perturbation points at a time) and sdp_colu_kernel.h (a table per control).
#pragma once
extern "C" __global__ void __launch_bounds__(SDP_COL_THREADS, SDP_COL_MIN_WAVES) sdp_sweep_col(SdpSweepArgs a)
{
    __shared__ SdpColLds sdp_lds;
    SDP_STAMP_BEGIN(a);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    constexpr int N0 = SDP_COL_N0;
    constexpr int Wn = SDP_COL_W;
    // the table dimensions are compiled in (the host checks `sdp_meta` against the problem
    // before the first launch: sdp_problem_create); a launch on any other grid is a bug
    sdp_trap_unless(a.n_lead == N0 && (SDP_HAS_W ? a.W : 1) == Wn);
    const sdp_real t = (sdp_real)a.t_k;
    const sdp_real *__restrict__ axis0 = (const sdp_real *)a.axes + a.axis_off[0];

    SdpColShared s;
    sdp_col_carve(sdp_lds, s);
    SdpGrid<sdp_real, SDP_DT> tg;
    sdp_col_trailing_grid(a, tg);
    SdpLeadAxis lead;
    sdp_col_lead_axis(a, lead);
    SdpColWalk walk;
    sdp_col_walk(a, walk);
    SdpColWeights wts;
    sdp_col_load_weights(a, wts, sdp_lds.pw, sdp_lds.gw);
#if SDP_COL_FILTER
    SdpColFilter filt;
    sdp_col_filter_setup(a, filt);
    const int axis_mode = __builtin_amdgcn_readfirstlane(sdp_col_axis_mode(lead));
    // (what is the same in every lane for the whole kernel lives in scalar registers: sdp_uniform)
    lead.smin = sdp_uniform(lead.smin); lead.span = sdp_uniform(lead.span); lead.nm1 = sdp_uniform(lead.nm1); lead.rspan = sdp_uniform(lead.rspan);
    filt.psum = sdp_uniform(filt.psum); filt.pcap = sdp_uniform(filt.pcap); filt.cu = sdp_uniform(filt.cu);
    filt.floor = sdp_uniform(filt.floor); filt.ratio = sdp_uniform(filt.ratio); filt.psum64 = sdp_uniform(filt.psum64);
    filt.gc = sdp_uniform(filt.gc); filt.glimit = sdp_uniform(filt.glimit);
    filt.k_rows = sdp_uniform(lead.nm1 / lead.span);
    filt.x_cap = sdp_uniform((sizeof(sdp_real) == 8 ? (sdp_real)0x1p30 : (sdp_real)0x1p13) / filt.k_rows - fabs(lead.smin));
    int guess = -1;                                        // branch and bound: this lane's best control at its previous node
    int tables_made = 1;                                   // control tables made so far (the first one before the loop): see SDP_LEAN2_A_FIXED
    (void)guess; (void)tables_made;
#endif
    if (SDP_COL_WINDOW && threadIdx.x < 4) sdp_lds.win[threadIdx.x >> 1][threadIdx.x & 1] = INT_MAX;
    if (threadIdx.x < 2) sdp_lds.dcol[threadIdx.x] = 0ull;
#if SDP_COL_SHIFT
    if (threadIdx.x < 2) sdp_col_shift_reset(sdp_lds, threadIdx.x);
#endif
    int parity = 0;
#if SDP_STAMP == 2     // diagnostic: shader clocks thread 0 spends in phases W, A, B (+ idle at barriers)
    unsigned long long tw = 0, ta = 0, tb = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0, tstart = __builtin_amdgcn_s_memtime();
    unsigned long long tr = 0, m0 = 0;   // filter: reduce
    (void)tr; (void)m0;
#endif
    SdpColDiag diag;                     // filter: first pass, second pass, survivor counts (SDP_STAMP 2 / 3)
    (void)diag;

#if SDP_COL_FILTER
    // The units of this XCD's share are handed out in order (one atomic per unit, claimed a
    // round ahead): whatever their speeds, the workgroups of an XCD work on neighbouring units.
    const int64_t u_base = walk.unit - (blockIdx.x >> 3), u_end = walk.end;
    unsigned int *claim = a.claim + 32 * (blockIdx.x & 7);
    if (threadIdx.x == 0) sdp_lds.next_unit = (int)atomicAdd(claim, 1u);
    __syncthreads();
    int64_t unit = u_base + sdp_lds.next_unit;
    int upar = 0;                                          // parity buffer of the control table
#if SDP_COL_UTAB
    sdp_trap_unless(!a.box_per_node);                      // (and SDP_COL_UTAB_N controls: sdp_meta, checked by the host)
#endif
    // what stays the same from unit to unit (SDP_COL_HOIST)
    SdpBox box_hold;
    const SdpBox *box_c = nullptr;
    const sdp_real *w_mine = nullptr;
    sdp_real w_hold = (sdp_real)0, x0_pre = (sdp_real)0;
    int i_pre = -1;
    if (SDP_COL_HOIST) {
        if (!a.box_per_node) { sdp_load_box(a, 0, box_hold); box_c = &box_hold; }
#if SDP_HAS_W
        if (lane < Wn) { w_hold = ((const sdp_real *)a.wgrid)[lane]; w_mine = &w_hold; }
#endif
        if (a.col_splits == 1 && N0 <= (int)blockDim.x && (int)threadIdx.x < N0) {     // one lane per node, whole columns:
            i_pre = (int)threadIdx.x;                                                   // this thread's node never changes
            x0_pre = axis0[i_pre];
        }
    }
    if (unit < u_end) {                                    // trailing cells of the first unit
        sdp_real xn[SDP_D];
        sdp_col_coords(a, sdp_col_of_unit(a, unit), xn);
        sdp_col_phase_w(a, tg, s, xn, nullptr, t);
#if SDP_COL_UTAB
        sdp_col_phase_u(a, sdp_lds.utab[0], xn, t, 0, SDP_COL_SHORT ? 64 : 0, box_c, filt.psum, filt.k_rows, filt.x_cap,
                        sizeof(sdp_real) == 4 ? filt.psum64 : (double)filt.psum);   // (one wave: it also reduces the table's statistics)
#endif
#if SDP_COL_SHIFT
        sdp_col_phase_shift(a, sdp_lds, lead, xn, t, 0);
#endif
    }
    while (unit < u_end) {
        int64_t next_unit;
#else
    for (int64_t unit = walk.unit; unit < walk.end; unit += walk.stride) {
#endif
        const int64_t col = sdp_col_of_unit(a, unit);
        const int part = (int)((unsigned)unit % (unsigned)a.col_splits);      // (32-bit: units < 2^31)
        const int i_lo = (int)((unsigned)(N0 * part) / (unsigned)a.col_splits);
        const int i_hi = (int)((unsigned)(N0 * (part + 1)) / (unsigned)a.col_splits);
        sdp_real x[SDP_D];
        sdp_col_coords(a, col, x);
#if SDP_COL_FILTER
        // The trailing cells of this unit were computed during the previous one (by its last
        // wave, next to the reduction of the table; before the loop for the first unit): phase W
        // is off the critical path.  This barrier publishes them and tells that every wave has
        // left phase B of the previous unit, i.e. that the table may be overwritten.
        SDP_COL_MARK(t0);
        __syncthreads();
        SDP_COL_MARK(t1);
        int nx = 0;                                        // the next unit: claimed here, the atomic's
        if (wave == waves - 1 && lane == 0) nx = (int)atomicAdd(claim, 1u);   // round trip hides under phase A
#if SDP_COL_SHIFT
        if (threadIdx.x == 0) sdp_col_shift_reset(sdp_lds, upar ^ 1);    // (its readers left at the barrier above)
        SdpColShiftCol shc;
        sdp_col_shift_col(sdp_lds, lead, upar, shc);
        sdp_col_shift_zero(sdp_lds, shc);
#endif
        sdp_col_phase_a<false>(a, tg, s);
        if (wave == waves - 1 && lane == 0) sdp_lds.next_unit = nx;      // (read after the barrier below, and after the next one)
        __syncthreads();
        SDP_COL_MARK(t2);
        {
            // the next unit's column-level tables (nothing reads the cells after phase A), a wave each where
            // the workgroup has that many: they are short but made of dependent latencies (divisions, loads
            // of the box and of the axes), and the barrier after the reduction waits for the last of them
            const int nxu = __builtin_amdgcn_readfirstlane(sdp_lds.next_unit);
            if (u_base + nxu < u_end) {
                sdp_real xn[SDP_D];
                sdp_col_coords(a, sdp_col_of_unit(a, u_base + nxu), xn);
                sdp_col_phase_w(a, tg, s, xn, nullptr, t, (waves - 1) * 64, 64, w_mine);
#if SDP_COL_UTAB
                sdp_col_phase_u(a, sdp_lds.utab[upar ^ 1], xn, t, max(waves - 2, 0) * 64, 64, box_c, filt.psum, filt.k_rows, filt.x_cap,
                                sizeof(sdp_real) == 4 ? filt.psum64 : (double)filt.psum, SDP_LEAN2_A_FIXED_ON && tables_made >= 2);
                ++tables_made;
#endif
#if SDP_COL_SHIFT
                sdp_col_phase_shift(a, sdp_lds, lead, xn, t, upar ^ 1, max(waves - 3, 0) * 64, 64);
#endif
            }
        }
#else
        // Phase W only writes the trailing cells, which nothing reads after phase A; the barrier
        // that follows it also tells that every wave has left phase B of the previous unit, i.e.
        // that the table may be overwritten.  (The row window publishes through win[] first.)
        if (SDP_COL_WINDOW) __syncthreads();
        SDP_COL_MARK(t0);
        sdp_col_window_predict<false>(a, lead, sdp_lds.win, parity, col, i_lo, i_hi, x, t);
        sdp_col_phase_w(a, tg, s, x, nullptr, t);
        __syncthreads();
        SDP_COL_MARK(t1);
        s.r0 = sdp_col_window_read(sdp_lds.win, parity);
        parity ^= 1;
        sdp_col_phase_a<false>(a, tg, s);
        __syncthreads();
        SDP_COL_MARK(t2);
#endif

#if SDP_COL_FILTER
        __builtin_amdgcn_s_setprio(SDP_COL_B_PRIO);        // (see phase B)
#if SDP_COL_SHIFT
        sdp_col_shift_reduce(a, sdp_lds, filt, shc, parity, upar);
#else
        sdp_col_filter_reduce(a, sdp_lds, filt, parity);
#endif
#if SDP_STAMP == 2
        { unsigned long long r1 = __builtin_amdgcn_s_memtime(); diag.n_slow += r1 - t2; }
#endif
        __syncthreads();
        const sdp_real dcol = sdp_col_filter_dcol(sdp_lds, parity);
        parity ^= 1;
        next_unit = u_base + sdp_lds.next_unit;
        SDP_COL_MARK(m0);
#if SDP_STAMP == 2
        tr += m0 - t2;
#endif
        // ---- phase B, filtered: sdp_col_filter_nodes
        sdp_col_filter_nodes(a, tg, s, wts, lead, filt, axis_mode, sdp_lds.ad, sdp_lds.utab[upar], dcol, col, i_lo, i_hi, wave, waves, x, t, diag,
                             box_c, i_pre, x0_pre
#if SDP_COL_SHIFT
                             , shc
#endif
                             , &guess
                             );
        upar ^= 1;
#else
        // ---- phase B.  One LANE per node (64 consecutive nodes of the column
        // per wavefront: their rows q0 are consecutive, so the LDS reads are
        // conflict-free), the control loop and the argmin run in-lane with no
        // cross-lane traffic.  With fewer 64-node groups than waves the
        // control lattice is cut into `chunks` consecutive ranges, one wave
        // each; the partial minima meet in LDS and are merged in chunk order
        // (first occurrence wins, stodynprog.py:686).
        const int n_nodes = i_hi - i_lo;
        const int groups = (n_nodes + 63) >> 6;
        const int chunks = groups < waves ? waves / groups : 1;
        __builtin_amdgcn_s_setprio(SDP_COL_B_PRIO);
        for (int item = wave; item < groups * chunks; item += waves) {
            const int grp = item / chunks;
            const int chunk = item - grp * chunks;
            const int i = i_lo + (grp << 6) + lane;
            if (i < i_hi) {
                const int64_t node = col * N0 + i;            // axis-0-fastest index
                SdpBox box;
                x[0] = axis0[i];
                sdp_load_box(a, node, box);
                const int c_lo = (int)((int64_t)box.total * chunk / chunks);
                const int c_hi = (int)((int64_t)box.total * (chunk + 1) / chunks);
                sdp_real best = INFINITY;
                int ibest = INT_MAX;
                constexpr int K = SDP_COL_UNROLL_U;
                int ci = c_lo;
                for (; ci + K <= c_hi; ci += K) {             // K controls at a time
                    sdp_real u[K][SDP_NU], jc[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) sdp_controls_at(box, ci + j, u[j]);
                    sdp_col_expected_cost<K>(a, tg, s, wts, lead, x, u, t, jc);
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        if (ibest == INT_MAX || sdp_better_seq(jc[j], best)) { best = jc[j]; ibest = ci + j; }
                }
                for (; ci < c_hi; ++ci) {                     // remainder
                    sdp_real u[1][SDP_NU], jc[1];
                    sdp_controls_at(box, ci, u[0]);
                    sdp_col_expected_cost<1>(a, tg, s, wts, lead, x, u, t, jc);
                    if (ibest == INT_MAX || sdp_better_seq(jc[0], best)) { best = jc[0]; ibest = ci; }
                }
                if (chunks == 1) {
                    sdp_col_store(a, node, box, best, ibest);
                } else {
                    s.part_J[chunk * n_nodes + (i - i_lo)] = best;
                    s.part_i[chunk * n_nodes + (i - i_lo)] = ibest;
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (chunks > 1) {
            __syncthreads();
            for (int n = threadIdx.x; n < n_nodes; n += blockDim.x) {
                sdp_real best = s.part_J[n];
                int ibest = s.part_i[n];
                for (int k = 1; k < chunks; ++k) {
                    const sdp_real cj = s.part_J[k * n_nodes + n];
                    const int cidx = s.part_i[k * n_nodes + n];
                    // an empty chunk (fewer controls than chunks) carries INT_MAX
                    if (cidx != INT_MAX && (ibest == INT_MAX || sdp_better_seq(cj, best))) {
                        best = cj;
                        ibest = cidx;
                    }
                }
                const int64_t node = col * N0 + i_lo + n;
                SdpBox box;
                sdp_load_box(a, node, box);
                sdp_col_store(a, node, box, best, ibest);
            }
        }
#endif  // SDP_COL_FILTER
#if SDP_STAMP == 2
        t3 = __builtin_amdgcn_s_memtime();
        tw += t1 - t0; ta += t2 - t1; tb += t3 - t2;
#endif
#if SDP_COL_FILTER
        unit = next_unit;
#endif
    }
#if SDP_COL_FILTER
    // the last workgroup to run out of units leaves the counters at zero for the next launch
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(a.claim + 256, 1u) == gridDim.x - 1) {
            for (int k = 0; k < 8; ++k) atomicExch(a.claim + 32 * k, 0u);
            atomicExch(a.claim + 256, 0u);
        }
    }
#endif
#if SDP_STAMP == 3 && SDP_COL_FILTER
    if (a.stamps) {
        atomicAdd((unsigned long long *)&a.stamps[0], diag.n_slow);
        atomicAdd((unsigned long long *)&a.stamps[1], diag.n_exact);
        atomicAdd((unsigned long long *)&a.stamps[2], diag.n_all);
    }
#elif SDP_STAMP == 2
    if (a.stamps && threadIdx.x == 0) {
        a.stamps[blockIdx.x * 4 + 0] = tw;
        a.stamps[blockIdx.x * 4 + 1] = ta;
        a.stamps[blockIdx.x * 4 + 2] = tb;
        a.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - tstart;
#if SDP_COL_FILTER
        a.stamps[(gridDim.x + blockIdx.x) * 4 + 0] = tr;
        a.stamps[(gridDim.x + blockIdx.x) * 4 + 1] = diag.tp1;
        a.stamps[(gridDim.x + blockIdx.x) * 4 + 2] = diag.tp2;
        a.stamps[(2 * gridDim.x + blockIdx.x) * 4 + 0] = diag.n_slow;
#endif
    }
#else
    SDP_STAMP_END(a);
#endif
}

extern "C" __global__ void __launch_bounds__(SDP_COL_THREADS) sdp_evalpol_col(SdpSweepArgs a)
{
    __shared__ SdpColLds sdp_lds;
    constexpr int N0 = SDP_COL_N0;
    constexpr int Wn = SDP_COL_W;
    sdp_trap_unless(a.n_lead == N0 && (SDP_HAS_W ? a.W : 1) == Wn);
    const sdp_real t = (sdp_real)a.t_k;
    const sdp_real *__restrict__ axis0 = (const sdp_real *)a.axes + a.axis_off[0];
    SdpColShared s;
    sdp_col_carve(sdp_lds, s);
    SdpGrid<sdp_real, SDP_DT> tg;
    sdp_col_trailing_grid(a, tg);
    // fused relative-DP shift of the previous step (see SdpLerp<.., SHIFT>)
    tg.shift = a.shift_index >= 0 ? ((const sdp_real *)a.V)[a.shift_index] : (sdp_real)0;
    if (a.ref_out && blockIdx.x == 0 && threadIdx.x == 0) *a.ref_out = (double)tg.shift;
    SdpLeadAxis lead;
    sdp_col_lead_axis(a, lead);
    SdpColWalk walk;
    sdp_col_walk(a, walk);
    SdpColWeights wts;
    sdp_col_load_weights(a, wts, sdp_lds.pw, sdp_lds.gw);
    if (SDP_COL_WINDOW && threadIdx.x < 4) sdp_lds.win[threadIdx.x >> 1][threadIdx.x & 1] = INT_MAX;
    int parity = 0;
    for (int64_t unit = walk.unit; unit < walk.end; unit += walk.stride) {
        const int64_t col = sdp_col_of_unit(a, unit);
        const int part = (int)((unsigned)unit % (unsigned)a.col_splits);      // (32-bit: units < 2^31)
        const int i_lo = (int)((unsigned)(N0 * part) / (unsigned)a.col_splits);
        const int i_hi = (int)((unsigned)(N0 * (part + 1)) / (unsigned)a.col_splits);
        sdp_real x[SDP_D];
        sdp_col_coords(a, col, x);
        __syncthreads();
        sdp_col_window_predict<true>(a, lead, sdp_lds.win, parity, col, i_lo, i_hi, x, t);
        sdp_col_phase_w(a, tg, s, x, nullptr, t);
        __syncthreads();
        s.r0 = sdp_col_window_read(sdp_lds.win, parity);
        parity ^= 1;
        sdp_col_phase_a<true>(a, tg, s);
        __syncthreads();
        for (int i = i_lo + threadIdx.x; i < i_hi; i += blockDim.x) {
            const int64_t node = col * N0 + i;
            sdp_real u[1][SDP_NU], jc[1];
            x[0] = axis0[i];
#pragma unroll
            for (int c = 0; c < SDP_NU; ++c) u[0][c] = ((const sdp_real *)a.pol_in)[node * SDP_NU + c];
            sdp_col_expected_cost<1, true>(a, tg, s, wts, lead, x, u, t, jc);
            sdp_store_J<sdp_real>(a, node, col, jc[0]);
        }
    }
}

