// sdp_column_pipe.h -- the filtered column sweep (sdp_column_kernel.h, SDP_COL_FILTER) as a
// producer / consumer pipeline inside ONE workgroup per CU.
//
// Why: the two halves of a column's work want opposite things.  The table build (phases W, A
// and the reduction over w) is a stream of strip reads from L2 -- latency, few instructions --
// while the two passes over the controls are pure vector issue.  With one workgroup doing
// both in turn and two such workgroups per CU (all that two 68 KiB tables allow), a SIMD holds
// two waves of which at most one is in the issue-bound half: a lone wave issues one vector
// instruction per 4 clocks whatever its type (MI355X_MICROARCH.md, cycle constants), so the
// kernel ran at 0.55 of the issue cycles (round 2, profiles/r02_filter_ab.txt) and every wave
// spent a quarter of its life waiting for the slowest one at the barrier before the build.
//
// Here the workgroup has SDP_PIPE_CONSUMERS waves that only run the passes (two lanes per node
// at N0 = 256: two issue-bound waves per SIMD, which interleave to the full rate) and
// SDP_PIPE_PRODUCERS waves that only build: while the consumers work on column k out of
// buffer k & 1, the producers build the table of column k + 1 -- and its reduction over w, in
// registers: a producer thread owns a row of axis 0 and walks the perturbation points, so
// A[r], D[r] never need a pass over the finished table -- into the other buffer.  ONE barrier
// per column swaps the roles of the buffers; units are claimed two steps ahead (the atomic's
// round trip hides under a whole step).  Same table entries, same passes
// (sdp_col_filter_nodes): J, policy and index are bit-identical to the plain kernel.
//
// Every producer wave computes the column's trailing cells itself, lane w the cell of
// perturbation point w (phase W costs two divisions per lane; repeating it per wave is cheaper
// than a barrier among the producers), and hands them to its threads through v_readlane.
#pragma once

#if SDP_COL_PIPE

#if !SDP_COL_FILTER || SDP_TRAIL_HAS_U
#error "SDP_COL_PIPE is a form of the filtered column kernel"
#endif
#ifndef SDP_PIPE_PRODUCERS
#define SDP_PIPE_PRODUCERS 4     // producer waves (one per SIMD)
#endif
#ifndef SDP_PIPE_G
// perturbation points whose 2^(d-1) vertex loads a producer thread keeps in flight
#define SDP_PIPE_G ((1 << (SDP_D - 1)) <= 4 ? 4 : 2)
#endif
#ifndef SDP_PIPE_PRIO_P
#define SDP_PIPE_PRIO_P 0        // wave priority of the producers (consumers: SDP_COL_B_PRIO)
#endif

struct __attribute__((aligned(16))) SdpPipeBuf {
    sdp_real T[SDP_COL_TW * SDP_COL_N0];                       // [w][r] (pairs of w for 4-byte reals)
    sdp_real ad[2 * SDP_COL_N0] __attribute__((aligned(16)));  // (A[r], D[r])
};
struct __attribute__((aligned(16))) SdpPipeLds {
    SdpPipeBuf buf[2];
    unsigned long long dcol[2][SDP_PIPE_PRODUCERS];   // lean filter: bits of max D[r] over the rows of a producer wave
    int unit[2];          // unit (relative to the XCD's share) whose table buf[b] holds
    int claimed[2];       // claimed[k & 1]: the unit the producers build during step k
};
static_assert(sizeof(SdpPipeLds) <= 160 * 1024, "two column tables exceed the 160 KiB LDS of a CU");

// lane `l`'s value of v as a wave-uniform value (SGPRs)
SDP_DEV double sdp_readlane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
SDP_DEV float sdp_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// The trailing cell of perturbation point w, held by LANE w of every producer wave
// (phase W in registers: no LDS, no barrier among the producers).
struct SdpPipeCell {
    int off[SDP_DT];
    sdp_real lam[SDP_DT], oml[SDP_DT], pw;
};
SDP_DEV void sdp_pipe_phase_w(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                              SdpPipeCell &c, const sdp_real *x, sdp_real t)
{
    const int w = min((int)(threadIdx.x & 63), SDP_COL_W - 1);
    sdp_real xn[SDP_D];
    sdp_model_trail(x, nullptr, ((const sdp_real *)a.wgrid)[w], t, xn);
    SdpCell<sdp_real, SDP_DT, sdp_real> cell;
#pragma unroll
    for (int k = 0; k < SDP_DT; ++k) {
        sdp_locate_axis<sdp_real, SDP_DT, sdp_real>(tg, k, xn[k + 1], cell);
        c.off[k] = cell.off[k];
        c.lam[k] = cell.lam[k];
        c.oml[k] = cell.oml[k];
    }
    c.pw = ((const sdp_real *)a.proba)[w];
}

// Table, A[r] and D[r] of one column by the producer threads: thread `ptid` of `pthreads` owns
// the rows ptid, ptid + pthreads, .. and walks the perturbation points in order.  The cell of
// a point is the same for every thread: it is read out of lane w into SGPRs (v_readlane), so
// the strip addresses are scalar and the lerp weights are scalar operands.  The vertex loads
// of SDP_PIPE_G points are issued together and TWO such groups are in flight: the loads of
// group g + 1 go out before the values of group g are used.
SDP_DEV void sdp_pipe_build(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg, const SdpColFilter &f,
                            SdpPipeBuf &out, unsigned long long *dcol_out, const SdpPipeCell &c, int ptid, int pthreads)
{
    sdp_real dmax = (sdp_real)0;
    constexpr int N0 = SDP_COL_N0;
    constexpr int Wn = SDP_COL_W;
    constexpr int G = SDP_PIPE_G;
    constexpr int NV = 1 << SDP_DT;
    static_assert(Wn <= 64, "one lane per perturbation point");
    const sdp_real *__restrict__ V = (const sdp_real *)a.V;
    for (int r = ptid; r < N0; r += pthreads) {
        sdp_real acc = (sdp_real)0, big = (sdp_real)0;
        auto issue = [&](int w0, sdp_real (*vals)[NV]) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int w = min(w0 + j, Wn - 1);                        // clamp: result unused
                int off[SDP_DT];
#pragma unroll
                for (int k = 0; k < SDP_DT; ++k) off[k] = __builtin_amdgcn_readlane(c.off[k], w);
                SdpColGather<0>::run(V + r, tg, off, 0, vals[j]);
            }
        };
        auto finish = [&](int w0, sdp_real (*vals)[NV]) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int w = w0 + j;
                if (w < Wn) {
                    sdp_real lam[SDP_DT], oml[SDP_DT];
#pragma unroll
                    for (int k = 0; k < SDP_DT; ++k) {
                        lam[k] = sdp_readlane(c.lam[k], w);
                        oml[k] = sdp_readlane(c.oml[k], w);
                    }
                    const sdp_real val = SdpColNest<0, false>::run(vals[j], lam, oml, tg.shift);
#if SDP_COL_WPAIR
                    out.T[((w >> 1) * N0 + r) * 2 + (w & 1)] = val;
#else
                    out.T[w * N0 + r] = val;
#endif
                    acc = acc + sdp_readlane(c.pw, w) * val;        // (what sdp_col_filter_reduce computes)
                    big = sdp_vmax(big, fabs(val));
                }
            }
        };
        sdp_real va[G][NV], vb[G][NV];
        issue(0, va);
#pragma unroll 1
        for (int w0 = 0; w0 < Wn; w0 += 2 * G) {
            if (w0 + G < Wn) issue(w0 + G, vb);
            finish(w0, va);
            if (w0 + 2 * G < Wn) issue(w0 + 2 * G, va);
            if (w0 + G < Wn) finish(w0 + G, vb);
        }
        const sdp_real d = acc == acc ? f.pcap * big + f.floor : (sdp_real)INFINITY;
        if (SDP_COL_LEAN_ON) {
            out.ad[r] = acc;
            dmax = sdp_vmax(dmax, d);
        } else {
            out.ad[2 * r] = acc;
            out.ad[2 * r + 1] = d;
        }
    }
    if (SDP_COL_LEAN_ON) {
        unsigned long long bits = (unsigned long long)__double_as_longlong((double)dmax);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = (unsigned long long)__shfl_xor((long long)bits, d, 64);
            bits = o > bits ? o : bits;
        }
        if ((threadIdx.x & 63) == 0) *dcol_out = bits;
    }
}

extern "C" __global__ void __launch_bounds__(SDP_COL_THREADS) sdp_sweep_col(SdpSweepArgs a)
{
    __shared__ SdpPipeLds L;
    SDP_STAMP_BEGIN(a);
    constexpr int N0 = SDP_COL_N0;
    constexpr int Wn = SDP_COL_W;
    constexpr int NP = SDP_PIPE_PRODUCERS;
    sdp_trap_unless(a.n_lead == N0 && a.W == Wn);          // the table dimensions are compiled in
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cwaves = (int)(blockDim.x >> 6) - NP;         // consumer waves: 0 .. cwaves-1
    const bool producer = wave >= cwaves;
    const int pwave = wave - cwaves;
    const sdp_real t = (sdp_real)a.t_k;

    SdpGrid<sdp_real, SDP_DT> tg;
    sdp_col_trailing_grid(a, tg);
    SdpLeadAxis lead;
    sdp_col_lead_axis(a, lead);
    SdpColWalk walk;
    sdp_col_walk(a, walk);
    SdpColWeights wts;
    sdp_col_load_weights(a, wts, nullptr, nullptr);
    SdpColFilter filt;
    sdp_col_filter_setup(a, filt);
    const int axis_mode = __builtin_amdgcn_readfirstlane(sdp_col_axis_mode(lead));
    SdpColShared s;
    s.T = nullptr; s.w_off = nullptr; s.w_lam = nullptr; s.w_oml = nullptr;
    s.part_J = nullptr; s.part_i = nullptr; s.r0 = 0;
    SdpColDiag diag;
    (void)diag; (void)lane;
#if SDP_STAMP == 2
    unsigned long long busy = 0, steps = 0, tstart = __builtin_amdgcn_s_memtime(), tm0 = 0;
#endif

    // units of this XCD's share, handed out in order (see sdp_column_kernel.h, sdp_col_of_unit)
    const int64_t u_base = walk.unit - (blockIdx.x >> 3), u_end = walk.end;
    unsigned int *claim = a.claim + 32 * (blockIdx.x & 7);
    if (threadIdx.x == 0) {
        L.unit[0] = (int)atomicAdd(claim, 1u);
        L.claimed[0] = (int)atomicAdd(claim, 1u);
    }
    __syncthreads();

    auto build = [&](int unit_rel, SdpPipeBuf &out, unsigned long long *dcol_out) {
        const int64_t unit = u_base + unit_rel;
        if (unit >= u_end) return;
#ifdef SDP_DIAG_NO_BUILD
        if (t != (sdp_real)123.456) return;
#endif
        sdp_real xn[SDP_D];
        sdp_col_coords(a, sdp_col_of_unit(a, unit), xn);
        SdpPipeCell cell;
        sdp_pipe_phase_w(a, tg, cell, xn, t);
        sdp_pipe_build(a, tg, filt, out, dcol_out, cell, (int)threadIdx.x - cwaves * 64, NP * 64);
    };

    if (producer) build(L.unit[0], L.buf[0], &L.dcol[0][pwave]);
    __syncthreads();
    for (int k = 0;; ++k) {
        const int cur = k & 1, nxt = cur ^ 1;
        const int u_cur = L.unit[cur];
        if (u_base + u_cur >= u_end) break;                 // (the same for every wave)
#if SDP_STAMP == 2
        tm0 = __builtin_amdgcn_s_memtime();
#endif
        if (producer) {
            __builtin_amdgcn_s_setprio(SDP_PIPE_PRIO_P);
            const int u_next = L.claimed[cur];              // claimed a step ago
            int nx = 0;
            if (pwave == 0 && lane == 0) nx = (int)atomicAdd(claim, 1u);   // for the step after this one
#ifdef SDP_DIAG_BUILD_ONCE
            if (k < 1)
#endif
            build(u_next, L.buf[nxt], &L.dcol[nxt][pwave]);
            if (pwave == 0 && lane == 0) {
                L.claimed[nxt] = nx;
                L.unit[nxt] = u_next;
            }
        } else {
            const int64_t unit = u_base + u_cur;
            const int64_t col = sdp_col_of_unit(a, unit);
            const int part = (int)(unit % a.col_splits);
            const int i_lo = (int)((int64_t)N0 * part / a.col_splits);
            const int i_hi = (int)((int64_t)N0 * (part + 1) / a.col_splits);
            sdp_real x[SDP_D];
            sdp_col_coords(a, col, x);
            s.T = L.buf[cur].T;
            unsigned long long dbits = 0;
            if (SDP_COL_LEAN_ON) {
#pragma unroll
                for (int q = 0; q < NP; ++q) dbits = L.dcol[cur][q] > dbits ? L.dcol[cur][q] : dbits;
            }
            const sdp_real dcol = (sdp_real)__longlong_as_double((long long)dbits);
#ifdef SDP_DIAG_NO_CONSUME
            if (t == (sdp_real)123.456)
#endif
            sdp_col_filter_nodes(a, tg, s, wts, lead, filt, axis_mode, L.buf[cur].ad, dcol, col, i_lo, i_hi,
                                 wave, cwaves, x, t, diag);
        }
#if SDP_STAMP == 2
        busy += __builtin_amdgcn_s_memtime() - tm0;
        ++steps;
#endif
        __syncthreads();
    }
    // the last workgroup to run out of units leaves the counters at zero for the next launch
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(a.claim + 256, 1u) == gridDim.x - 1) {
            for (int q = 0; q < 8; ++q) atomicExch(a.claim + 32 * q, 0u);
            atomicExch(a.claim + 256, 0u);
        }
    }
#if SDP_STAMP == 3
    if (a.stamps) {
        atomicAdd((unsigned long long *)&a.stamps[0], diag.n_slow);
        atomicAdd((unsigned long long *)&a.stamps[1], diag.n_exact);
        atomicAdd((unsigned long long *)&a.stamps[2], diag.n_all);
    }
#elif SDP_STAMP == 2
    // thread 0 (a consumer) and the first producer thread: clocks inside their half of a step
    // (the rest of a step is the wait at its barrier), steps, lifetime
    if (a.stamps && (threadIdx.x == 0 || (int)threadIdx.x == cwaves * 64)) {
        const int bank = threadIdx.x == 0 ? 0 : 1;
        unsigned long long *o = a.stamps + ((size_t)bank * gridDim.x + blockIdx.x) * 4;
        o[0] = busy;
        o[1] = bank == 0 ? diag.tp1 : steps;
        o[2] = bank == 0 ? diag.tp2 : 0;
        o[3] = __builtin_amdgcn_s_memtime() - tstart;
    }
#else
    SDP_STAMP_END(a);
#endif
}

#endif  // SDP_COL_PIPE
