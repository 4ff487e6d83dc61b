// sdp_colres_kernel.h -- RESIDENT-CHUNK form of the filtered column kernel (SDP_COL_WRES = C < W = SDP_COL_W).
// Included by sdp_column_kernel.h in place of its own sdp_sweep_col / sdp_evalpol_col; every building block
// (table build, first pass, bounds, stores, unit claiming) is the one defined there.
//
// Why.  The W x N0 table of the column kernel (64 KiB at 32 x 256 x 8 B) lets a CU hold two workgroups: two waves
// per SIMD, one of which is usually waiting (table loads, barriers, LDS round trips).  Measured on the benchmark
// problem: ONE workgroup per CU takes 3.39 ms, two take 1.67 -- the kernel is bound by how much there is to
// overlap, not by any unit (profiles/r04_column_ab.txt).  Holding only C of the W perturbation points at a time
// shrinks the table to C x N0 (32 KiB at C = 16: four workgroups per CU, four waves per SIMD).
//
// How.  The first pass needs A[r] = sum_w p_w T[w][r] over ALL points, the second pass needs T[w][q0], T[w][q0+1]
// of the surviving control for all points IN w ORDER (the reference accumulates its expectation in that order,
// stodynprog.py:681).  So per column:
//     build the tail  T[w], w = C .. W-1        partial sums of A (a register per row)
//     build the head  T[w], w = 0 .. C-1        A complete  ->  LDS;  first pass on A (all controls)
//     second pass, head:  the survivor's cells for w = 0 .. C-1 out of the resident table
//     build the tail again                      second pass, tail: w = C .. W-1, added to the same accumulator
// i.e. (W + (W - C)) / W times the table loads of the plain kernel (1.5 at C = W / 2) and seven barriers instead
// of three, for twice the workgroups per CU.  The survivor's cell, weights and cost are computed once (second pass,
// head) and kept in registers across the rebuild.  Every floating-point operation of the result is the reference's,
// on the same operands, in the same order: J, policy and index have the same bits as the plain kernel.
// A node with several survivors (near-ties, NaN / infinite values: none on ordinary 8-byte inputs) evaluates its
// candidates from global memory (sdp_col_cost_global: the same operations), which needs no table at all.
//
// One lane per node (SDP_COL_THREADS >= SDP_COL_N0); 8-byte lean first pass (with or without the control table);
// the perturbation does not reach the cost; plain table layout, exact arithmetic.
// The perturbation may reach the stock through final sums (SDP_COL_SHIFT, the filter on the shifted lattice of
// sdp_col_shift_reduce): the reduced table is then accumulated chunk by chunk into the same lattice (its LDS atomics
// do not care in how many pieces the points arrive), the second pass locates the cell along axis 0 per perturbation
// point as sdp_col_expected_cost does, and a node with exactly TWO survivors -- one in a hundred on the benchmark
// problem with noise in the stock, hence about every other wave -- carries both through the rebuild.
#pragma once

static_assert(SDP_COL_LEAN_ON && !SDP_COL_WIDE_ON, "resident chunks: the lean first pass of 8-byte reals");
static_assert(!SDP_COL_TOP2 || SDP_COL_SHIFT, "resident chunks: two carried survivors only on the shifted lattice");
static_assert(SDP_COL_WRES >= 1 && 2 * SDP_COL_WRES >= SDP_COL_W, "resident chunks: at least half of the points resident");
constexpr int SDP_COLRES_K = SDP_COL_TOP2 ? 2 : 1;        // survivors a lane can carry through the rebuild of the tail

// SDP_COL_TAIL_KEEP (round 6): the tail is built ONCE.  Its first build also stores the entries to a block of global memory
// private to the workgroup (SdpSweepArgs.tail: (W - C) x N0 reals, rewritten every column, so it lives in L2 / the Infinity
// Cache), and the second pass reads the survivor's T[w][q0], T[w][q0+1], w = C .. W-1, back from there: 2 (W - C) coalesced
// 8-byte loads per node (neighbouring lanes: neighbouring rows) in place of (W - C) N0 / threads rebuilt entries per thread
// (4 strip loads and 9 operations each) and the two barriers around them.  The entries are the reference's values either
// way -- the same SdpColNest on the same operands -- so nothing changes in J, policy or index.  A workgroup's waves run on
// one CU and share its vector L1 (write-through), and every build is followed by a workgroup barrier before anything reads:
// the block needs no fence beyond __syncthreads().
#ifndef SDP_COL_TAIL_KEEP
#define SDP_COL_TAIL_KEEP 0
#endif
#if SDP_COL_TAIL_KEEP && (SDP_COL_SHIFT || !SDP_COL_A_WIDE_LOADS || SDP_COL_A_ORDER != 2 || SDP_COL_WPAIR)
#error "SDP_COL_TAIL_KEEP: the plain resident-chunk kernel with 16-byte build loads"
#endif
// SDP_COL_TAIL_HOLD (round 6): the tail is built ONCE and its entries stay IN REGISTERS.  The thread that computes
// T[w][r .. r+1], w >= C, in the first build of the tail keeps them -- (W - C) N0 / threads entries, 32 registers at
// 16 x 256 / 256 -- through the first pass and the second pass over the head, and writes them back into the table where
// the rebuild of the tail used to be: no strip loads (a third of the kernel's vector-memory instructions), no
// interpolation (9 operations per entry), the same values by construction.  The registers are what it costs: the
// kernel no longer fits the 128 of four waves per SIMD, so such a build asks for three (168 registers; the LDS image
// still admits four workgroups per CU, the register file three).
#ifndef SDP_COL_TAIL_HOLD
#define SDP_COL_TAIL_HOLD 0
#endif
#if SDP_COL_TAIL_HOLD && (!SDP_COL_A_WIDE_LOADS || SDP_COL_A_ORDER != 2 || SDP_COL_WPAIR || SDP_COL_TAIL_KEEP || \
                          (SDP_COL_W - SDP_COL_WRES) % (SDP_COL_THREADS / SDP_COL_A_LW) != 0 || SDP_COL_ROWS % (2 * SDP_COL_A_LW) != 0)
#error "SDP_COL_TAIL_HOLD: the plain resident-chunk kernel with 16-byte build loads, whole rounds of points and rows"
#endif
#if SDP_COL_TAIL_HOLD
// the tail's build with the entries kept (BUILD) / the kept entries back into the table (!BUILD): the loops of
// sdp_col_phase_a's 16-byte form with compile-time trip counts, so that `held` lives in registers
typedef sdp_real sdp_held_rows __attribute__((ext_vector_type(2)));
constexpr int SDP_HOLD_WP = SDP_COL_THREADS / SDP_COL_A_LW;                    // perturbation points side by side
constexpr int SDP_HOLD_NW = (SDP_COL_W - SDP_COL_WRES) / SDP_HOLD_WP;            // points per thread
constexpr int SDP_HOLD_NJ = SDP_COL_ROWS / (2 * SDP_COL_A_LW);                   // row pairs per thread and point
template <bool BUILD>
SDP_DEV void sdp_colres_tail(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg, const SdpColShared &s,
                             sdp_held_rows (&held)[SDP_HOLD_NW][SDP_HOLD_NJ])
{
    constexpr int N0 = SDP_COL_ROWS, C = SDP_COL_WRES, LW = SDP_COL_A_LW, NV = 1 << SDP_DT;
    const sdp_real *__restrict__ V = (const sdp_real *)a.V + s.r0;
    const int rl = threadIdx.x % LW, wg = threadIdx.x / LW;
#pragma unroll
    for (int wi = 0; wi < SDP_HOLD_NW; ++wi) {
        const int tw = wg + wi * SDP_HOLD_WP, w = C + tw;
        if (BUILD) {
            int off[SDP_DT];
            sdp_real lam[SDP_DT], oml[SDP_DT];
#pragma unroll
            for (int k = 0; k < SDP_DT; ++k) {
                off[k] = s.w_off[w * SDP_DT + k];
                lam[k] = s.w_lam[w * SDP_DT + k];
                oml[k] = s.w_oml[w * SDP_DT + k];
            }
            int voff[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                int o = 0;
#pragma unroll
                for (int k = 0; k < SDP_DT; ++k) o += off[k] + (((q >> (SDP_DT - 1 - k)) & 1) ? tg.M[k] : 0);
                voff[q] = o;
            }
            sdp_held_rows vals2[SDP_HOLD_NJ][NV];
#pragma unroll
            for (int j = 0; j < SDP_HOLD_NJ; ++j) {
                const int r = (j * LW + rl) * 2;
#pragma unroll
                for (int q = 0; q < NV; ++q) vals2[j][q] = *(const sdp_held_rows *)(V + r + voff[q]);
            }
#pragma unroll
            for (int j = 0; j < SDP_HOLD_NJ; ++j) {
                sdp_held_rows e;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    sdp_real one[NV];
#pragma unroll
                    for (int q = 0; q < NV; ++q) one[q] = vals2[j][q][c];
                    e[c] = SdpColNest<0, false>::run(one, lam, oml, tg.shift);
                }
                held[wi][j] = e;
            }
        }
#pragma unroll
        for (int j = 0; j < SDP_HOLD_NJ; ++j) *(sdp_held_rows *)(s.T + tw * N0 + (j * LW + rl) * 2) = held[wi][j];
    }
}
#endif

// diagnostic builds only (results wrong): bit mask of the parts of a unit to leave out -- 1 first build of the tail and
// its reduction, 2 build of the head and its reduction, 4 first pass, 8 second pass over the head, 16 rebuild of the
// tail, 32 second pass over the tail, 64 the next unit's column-level tables, 128 the stores
#ifndef SDP_DIAG_SKIP
#define SDP_DIAG_SKIP 0
#endif
#define SDP_COLRES_TAIL_BYTES (SDP_COL_TAIL_KEEP ? (SDP_COL_W - SDP_COL_WRES) * SDP_COL_ROWS * (int)sizeof(sdp_real) : 0)

// cell of x0' along axis 0 of one control for the perturbation point value `wv`, as sdp_col_expected_cost computes it (pyx:75-81)
SDP_DEV void sdp_colres_cell(const SdpLeadAxis &l, const sdp_real *x, const sdp_real *u, sdp_real wv, sdp_real t,
                             int &q0, sdp_real &lam0, sdp_real &oml0)
{
    const sdp_real xn0 = sdp_model_lead(x, u, wv, t);
    const sdp_real sn = sdp_div_span<sdp_real>(xn0 - l.smin, l.span, l.rspan, l.pow2);   // pyx:75
    const sdp_real p = sn * l.nm1;
    q0 = max(min(sdp_trunc_i32(p), l.ordm2), 0);                                         // pyx:78
    lam0 = p - (sdp_real)q0;                                                             // pyx:81
    oml0 = (sdp_real)1 - lam0;
}

// What a lane carries of its surviving controls from the second pass over the head to the one over the tail
struct SdpColresCand {
    sdp_real u[SDP_COLRES_K][SDP_NU], g[SDP_COLRES_K], acc[SDP_COLRES_K];
    sdp_real lam0[SDP_COLRES_K], oml0[SDP_COLRES_K];       // (a stock the perturbation does not reach: located once)
    int q0[SDP_COLRES_K], idx[SDP_COLRES_K];
    int n;                                                 // 0 (decided already), 1 or 2
};

// The reference's operations for K controls over the perturbation points [w_lo, w_hi) (table rows w - t_base), the
// expectations `acc` carried from call to call: what sdp_col_expected_cost<K> does for those points, chains interleaved.
template <int K>
SDP_DEV void sdp_colres_partial(const sdp_real *T, const SdpColWeights &k, const SdpLeadAxis &l, const sdp_real *x,
                                sdp_real t, const int w_lo, const int w_hi, const int t_base, SdpColresCand &c)
{
    constexpr int N0 = SDP_COL_ROWS;
    constexpr int B = SDP_COL_BATCH;
    // (two separate 8-byte reads per cell, kept apart by `volatile`: see sdp_col_expected_cost)
    const volatile sdp_lds_real *tab = (const volatile sdp_lds_real *)T;
#pragma unroll SDP_COL_UNROLL_W
    for (int w0 = w_lo; w0 < w_hi; w0 += B) {
        sdp_real lo[B][K], hi[B][K];
#if SDP_LEAD_HAS_W
        sdp_real lam_b[B][K], oml_b[B][K];
#endif
#pragma unroll
        for (int b = 0; b < B; ++b)
            if (w0 + b < w_hi) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
#if SDP_LEAD_HAS_W
                    sdp_colres_cell(l, x, c.u[j], SDP_COL_GW(k, w0 + b), t, c.q0[j], lam_b[b][j], oml_b[b][j]);
#endif
                    lo[b][j] = tab[(w0 + b - t_base) * N0 + c.q0[j]];
                    hi[b][j] = tab[(w0 + b - t_base) * N0 + c.q0[j] + 1];
                }
            }
#pragma unroll
        for (int b = 0; b < B; ++b)
            if (w0 + b < w_hi) {
                const sdp_real pw = SDP_COL_PW(k, w0 + b);
#pragma unroll
                for (int j = 0; j < K; ++j) {
#if SDP_LEAD_HAS_W
                    const sdp_real val = oml_b[b][j] * lo[b][j] + lam_b[b][j] * hi[b][j];
#else
                    const sdp_real val = c.oml0[j] * lo[b][j] + c.lam0[j] * hi[b][j];     // pyx:88-300
#endif
                    const sdp_real jc = c.g[j] + val;                                  // stodynprog.py:677
                    c.acc[j] = c.acc[j] + jc * pw;                                     // stodynprog.py:681, w order
                }
            }
    }
}

#if SDP_COL_TAIL_KEEP
// sdp_colres_partial<1> for the points [w_lo, w_hi) with the table rows read from the workgroup's block of global
// memory (row w - w_lo of `G`): the loads of SDP_COL_KEEP_BATCH points are in flight together.
#ifndef SDP_COL_KEEP_BATCH
#define SDP_COL_KEEP_BATCH 8
#endif
#ifndef SDP_COL_KEEP_LOADS
#define SDP_COL_KEEP_LOADS 0
#endif
SDP_DEV void sdp_colres_partial_kept(const sdp_real *__restrict__ G, const SdpColWeights &k, const int w_lo, const int w_hi,
                                     SdpColresCand &c)
{
    constexpr int N0 = SDP_COL_ROWS;
    constexpr int B = SDP_COL_KEEP_BATCH;
    const sdp_real *__restrict__ g0 = G + c.q0[0];
#pragma unroll
    for (int w0 = w_lo; w0 < w_hi; w0 += B) {
        sdp_real lo[B], hi[B];
#pragma unroll
        for (int b = 0; b < B; ++b)
            if (w0 + b < w_hi) {
#if SDP_COL_KEEP_LOADS == 1          // plain loads
                lo[b] = g0[(w0 + b - w_lo) * N0];
                hi[b] = g0[(w0 + b - w_lo) * N0 + 1];
#elif SDP_COL_KEEP_LOADS == 2        // past the vector L1 (sc1)
                lo[b] = __hip_atomic_load(g0 + (w0 + b - w_lo) * N0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi[b] = __hip_atomic_load(g0 + (w0 + b - w_lo) * N0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                lo[b] = __builtin_nontemporal_load(g0 + (w0 + b - w_lo) * N0);
                hi[b] = __builtin_nontemporal_load(g0 + (w0 + b - w_lo) * N0 + 1);
#endif
            }
#pragma unroll
        for (int b = 0; b < B; ++b)
            if (w0 + b < w_hi) {
                const sdp_real pw = SDP_COL_PW(k, w0 + b);
                const sdp_real val = c.oml0[0] * lo[b] + c.lam0[0] * hi[b];            // pyx:88-300
                const sdp_real jc = c.g[0] + val;                                      // stodynprog.py:677
                c.acc[0] = c.acc[0] + jc * pw;                                         // stodynprog.py:681, w order
            }
    }
}
#endif

#if SDP_COL_LEAN2
// ---------------------------------------------------------------------------
// The short first pass (generated where x0' = X(x) +- a and cost = K(x) +- h, a and h entries of the column's control
// table: codegen.short_pass_source).  Vector issue binds this kernel and the first pass is half of its instructions
// (docs/NOTEBOOK.md section 3.1e), so it sheds what need not be per control:
//   * K does not change the argmin over the controls of a node: the pass orders  F' = fma(+-h, psum, lerp)  -- an
//     approximation of E - K P*, P* the exact sum of the weights -- instead of F (one addition less);
//   * |g| <= (|K| + max |h|)(1 + u), max |h| of the column from sdp_col_phase_u, replaces the running sum of the |F|
//     as the bound on the cost (one addition less), and S_node = Pcap (|K| + max |h|) + (1 + 2 L) D bounds
//     |g| Pcap + (1 + 2 |lam0|) D of section 3.1c directly;
//   * x0' is monotone in a, and so is every rounded step from it to the position p (a sum with X, a difference with
//     smin, a division by / product with positive numbers): the positions of all controls lie between those of the
//     column's smallest and largest a, so L = max(1, |lam0| at either end) >= every |lam0| of the node (one maximum less);
//   * the control's index rides in the low `bits` mantissa bits of F' (bits = ceil(log2(controls))), so the running
//     minimum carries it: no compare, no select.  That moves F' by < 2^bits ulp <= 2^(bits+1) u |F'| <= 2^(bits+1) u
//     S_node (1 + 3u)  (|F'| <= |h| |psum| + (1 + 2L) D (1 + 3u)).
// Error: |E - R| as before; R - K P* = (g - K) P* + lerp*, g = fl(K +- h) so |(g - K) -+ h| <= u (|K| + |h|);
// F' against +-h P* + lerp*: (W-1) u |h| P for psum, the lerp's terms as in 3.1c, one rounding of the outer fma.  In sum
//     |E - K P* - F'| <= (2W+8) u [ (|K| + |h|)(1 + u) P + (1 + 2 |lam0|) D ]  <=  (2W+8) u S_node (1 + 2u)
// and the radius (cu + 2^(bits+2) u) S_node, cu = 4 (W+8) u, covers it with the same factor 2 to spare, the packing
// with its own.  What is not finite: a or h (poisoned statistics: sdp_col_phase_u), K, the table (D), X (L) -- each
// makes S_node or L fail its test and the node `bad`; and S_node < 2^1000 keeps every F' finite (|lam0 (A1 - A0)| <= 2 L D),
// so the packing never meets an infinity.  tests/test_filter_bound_exact.py checks the inequality in exact arithmetic.
static_assert(!SDP_COST_HAS_W && (SDP_COL_SHIFT ? !SDP_COL_SHIFT_CHAIN : !SDP_COL_TOP2) && SDP_COL_UTAB && sizeof(sdp_real) == 8,
              "short first pass: 8-byte reals, control table, a perturbation that does not reach the cost and reaches the stock "
              "through final sums at most");
template <int AXIS>
SDP_DEV sdp_real sdp_lean2_value(const sdp_real *A, const sdp_real *utab, const SdpColFilter &f, const SdpLeadAxis &l,
                                 sdp_real X, int ci)
{
    int q0;
    sdp_real lam0;
    sdp_lean2_cell<AXIS>(l, SDP_LEAN2_LEAD(X, utab[ci * SDP_COL_UTAB + SDP_LEAN2_A_SLOT]), q0, lam0);
    const sdp_real a0 = A[q0], a1 = A[q0 + 1];
    const sdp_real h = fma(lam0, a1 - a0, a0);
    if (SDP_LEAN2_H_SLOT < 0) return h;
    const sdp_real hv = utab[ci * SDP_COL_UTAB + (SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT)];
    return fma(SDP_LEAN2_HNEG ? -hv : hv, f.psum, h);
}
// the two smallest F' over the controls [0, n) of one node, the index of each in its low bits.
// A group of SDP_LEAN2_GROUP controls at a time, in stages -- every cell first, then every read of the reduced table,
// then the values: the reads of a group are in flight together (written control by control the compiler waits for
// each read before it starts the next control's cell, and a wave spends its time in LDS latency: the pass was bound by
// that, not by its instruction count -- removing a fifth of its instructions bought 2 %).
#ifndef SDP_LEAN2_GROUP
#define SDP_LEAN2_GROUP 4
#endif
template <int AXIS>
SDP_DEV void sdp_lean2_pass1(const sdp_real *A, const sdp_real *utab, const SdpColFilter &f, const SdpLeadAxis &l,
                             sdp_real X, int n, int mask, sdp_real &f1, sdp_real &f2)
{
    auto insert = [&](sdp_real F, int ci) {
        const sdp_real Fp = __hiloint2double(__double2hiint(F), (__double2loint(F) & ~mask) | ci);
        f2 = sdp_vmin(f2, sdp_vmax(f1, Fp));
        f1 = sdp_vmin(f1, Fp);
    };
    constexpr int K = SDP_LEAN2_GROUP;
    constexpr int HS = SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT;
    int ci = 0;
    for (; ci + K <= n; ci += K) {
        int q0[K];
        sdp_real av[K], lam0[K], hv[K], a0[K], a1[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            av[j] = utab[(ci + j) * SDP_COL_UTAB + SDP_LEAN2_A_SLOT];
            hv[j] = SDP_LEAN2_H_SLOT < 0 ? (sdp_real)0 : utab[(ci + j) * SDP_COL_UTAB + HS];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) sdp_lean2_cell<AXIS>(l, SDP_LEAN2_LEAD(X, av[j]), q0[j], lam0[j]);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            a0[j] = A[q0[j]];
            a1[j] = A[q0[j] + 1];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const sdp_real h = fma(lam0[j], a1[j] - a0[j], a0[j]);
            insert(SDP_LEAN2_H_SLOT < 0 ? h : fma(SDP_LEAN2_HNEG ? -hv[j] : hv[j], f.psum, h), ci + j);
        }
    }
    for (; ci < n; ++ci) insert(sdp_lean2_value<AXIS>(A, utab, f, l, X, ci), ci);
}

#if SDP_COL_SHIFT
// The short first pass ON THE SHIFTED LATTICE (round 6): x0' = (X +- a) +- b_1(w) .. with final sums (docs/NOTEBOOK.md
// section 3.1d), a and the control's part h of the cost from the column's control table.  The lattice's reduced table A'
// takes the place of A, `l` is the lattice (koff = its first position, ordm2 = its rows - 2), and what the lerp between
// whole positions leaves out is bounded by B' of the control's own cell.  What the short pass sheds is the same as above:
//     F' = fma(+-h, psum, lerp(A', pa)),  the control's index in its low mantissa bits;
//     S_node = Pcap (|K| + max |h|) + H D,   H = (1 + 2 (L + Lc)) (3 + Es),   L = max(1, |lam0| at the column's smallest and largest a)
//     radius = (cu + 2^(bits+2) u) S_node + max B'[q0]
// -- sdp_col_shift_reduce's bound with |g| Pcap <= Pcap (|K| + max |h|)(1 + u) in the place of ratio (sum |F| + H D), the
// rounding of g = fl(K +- h) as in the short pass above, the packing as there (|F'| <= S_node).  The three smallest F'
// are kept: ~1 % of the nodes of the benchmark problem hold two controls inside the radius (B' is of the order of
// h^2 V'' / 16, not 1e-13), both are carried through the second pass.  tests/test_filter_bound_exact.py checks
// |E - K P* - F'| <= radius in exact arithmetic.
typedef sdp_real sdp_ab_pair __attribute__((ext_vector_type(2)));
template <int AXIS>
SDP_DEV sdp_real sdp_lean2s_value(const sdp_real *A, const sdp_real *utab, const SdpColFilter &f, const SdpLeadAxis &l,
                                  sdp_real X, int ci, sdp_real &bmax)
{
    int q0;
    sdp_real lam0;
    sdp_lean2s_cell<AXIS>(l, SDP_LEAN2_LEAD(X, utab[ci * SDP_COL_UTAB + SDP_LEAN2_A_SLOT]), q0, lam0);
    const sdp_ab_pair ab = *(const sdp_ab_pair *)(A + 2 * q0);
    const sdp_real a1 = A[2 * q0 + 2];
    bmax = sdp_vmax(bmax, ab.y);
    const sdp_real h = fma(lam0, a1 - ab.x, ab.x);
    if (SDP_LEAN2_H_SLOT < 0) return h;
    const sdp_real hv = utab[ci * SDP_COL_UTAB + (SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT)];
    return fma(SDP_LEAN2_HNEG ? -hv : hv, f.psum, h);
}
// the three smallest F' over the controls [0, n) of one node, the index of each in its low bits; the largest B' met
template <int AXIS>
SDP_DEV void sdp_lean2s_pass1(const sdp_real *A, const sdp_real *utab, const SdpColFilter &f, const SdpLeadAxis &l,
                              sdp_real X, int n, int mask, sdp_real &f1, sdp_real &f2, sdp_real &f3, sdp_real &bmax)
{
    auto insert = [&](sdp_real F, int ci) {
        const sdp_real Fp = __hiloint2double(__double2hiint(F), (__double2loint(F) & ~mask) | ci);
        f3 = sdp_vmin(f3, sdp_vmax(f2, Fp));
        f2 = sdp_vmin(f2, sdp_vmax(f1, Fp));
        f1 = sdp_vmin(f1, Fp);
    };
    constexpr int K = SDP_LEAN2_GROUP;
    constexpr int HS = SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT;
    int ci = 0;
    for (; ci + K <= n; ci += K) {
        int q0[K];
        sdp_real av[K], lam0[K], hv[K], a1[K];
        sdp_ab_pair ab[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            av[j] = utab[(ci + j) * SDP_COL_UTAB + SDP_LEAN2_A_SLOT];
            hv[j] = SDP_LEAN2_H_SLOT < 0 ? (sdp_real)0 : utab[(ci + j) * SDP_COL_UTAB + HS];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) sdp_lean2s_cell<AXIS>(l, SDP_LEAN2_LEAD(X, av[j]), q0[j], lam0[j]);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            ab[j] = *(const sdp_ab_pair *)(A + 2 * q0[j]);
            a1[j] = A[2 * q0[j] + 2];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            bmax = sdp_vmax(bmax, ab[j].y);
            const sdp_real h = fma(lam0[j], a1[j] - ab[j].x, ab[j].x);
            insert(SDP_LEAN2_H_SLOT < 0 ? h : fma(SDP_LEAN2_HNEG ? -hv[j] : hv[j], f.psum, h), ci + j);
        }
    }
    for (; ci < n; ++ci) insert(sdp_lean2s_value<AXIS>(A, utab, f, l, X, ci, bmax), ci);
}
#endif  // SDP_COL_SHIFT

#endif  // SDP_COL_LEAN2

extern "C" __global__ void __launch_bounds__(SDP_COL_THREADS, SDP_COL_MIN_WAVES) sdp_sweep_col(SdpSweepArgs a)
{
    __shared__ SdpColLds sdp_lds;
    SDP_STAMP_BEGIN(a);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    constexpr int N0 = SDP_COL_N0;
    constexpr int Wn = SDP_COL_W;
    constexpr int C = SDP_COL_WRES, R = Wn - C;            // resident head, rebuilt tail
    sdp_trap_unless(a.n_lead == N0 && a.W == Wn && (int)blockDim.x == SDP_COL_THREADS);
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
    SdpColFilter filt;
    sdp_col_filter_setup(a, filt);
#if SDP_COL_TAIL_KEEP
    sdp_real *__restrict__ kept = (sdp_real *)((char *)a.tail + (size_t)blockIdx.x * (size_t)SDP_COLRES_TAIL_BYTES);
#else
    sdp_real *kept = nullptr;
#endif
    const int axis_mode = __builtin_amdgcn_readfirstlane(sdp_col_axis_mode(lead));
    const sdp_cst_real *pw = (const sdp_cst_real *)a.proba;
    // (what is the same in every lane for the whole kernel lives in scalar registers: sdp_uniform)
    lead.smin = sdp_uniform(lead.smin); lead.span = sdp_uniform(lead.span); lead.nm1 = sdp_uniform(lead.nm1); lead.rspan = sdp_uniform(lead.rspan);
    filt.psum = sdp_uniform(filt.psum); filt.pcap = sdp_uniform(filt.pcap); filt.cu = sdp_uniform(filt.cu);
    filt.floor = sdp_uniform(filt.floor); filt.ratio = sdp_uniform(filt.ratio);
    const sdp_real k_rows = sdp_uniform(lead.nm1 / lead.span);        // rows of axis 0 per unit of x0 (branch and bound: positions relative to the node)
    // (on the shifted lattice the bounds' positions also lose the lattice's first position: half the range)
    const sdp_real x_cap = sdp_uniform((sdp_real)(SDP_COL_SHIFT ? 0x1p29 : 0x1p30) / k_rows - fabs(lead.smin));
    (void)k_rows; (void)x_cap;
    if (threadIdx.x < 2) sdp_lds.dcol[threadIdx.x] = 0ull;
    int parity = 0;

    // units of this XCD's share, claimed in order (see sdp_col_of_unit)
    const int64_t u_base = walk.unit - (blockIdx.x >> 3), u_end = walk.end;
    unsigned int *claim = a.claim + 32 * (blockIdx.x & 7);
    if (threadIdx.x == 0) sdp_lds.next_unit = (int)atomicAdd(claim, 1u);
    __syncthreads();
    int64_t unit = u_base + sdp_lds.next_unit;
    int upar = 0;                                          // parity buffer of the control table
#if SDP_STAMP == 2     // diagnostic: shader clocks of this thread in the table builds (+ their barriers), the first pass, the rest
    unsigned long long tm_build = 0, tm_p1 = 0, tm_all = 0, tm_top = 0, tm0 = 0, tm1 = 0, tm_start = __builtin_amdgcn_s_memtime();
#define SDP_RES_MARK(v) v = __builtin_amdgcn_s_memtime()
#else
#define SDP_RES_MARK(v)
#endif
    int guess = -1;                                        // branch and bound: this lane's best control in the previous unit of the workgroup
    int tables_made = 1;                                   // control tables made so far (the first one before the loop): see SDP_LEAN2_A_FIXED
    (void)tables_made;
    (void)guess;
#if SDP_COL_UTAB
    sdp_trap_unless(!a.box_per_node);
#endif
    // what stays the same from unit to unit (SDP_COL_HOIST)
    SdpBox box_hold;
    const SdpBox *box_c = nullptr;
    const sdp_real *w_mine = nullptr;
    sdp_real w_hold = (sdp_real)0;
    if (SDP_COL_HOIST) {
        if (!a.box_per_node) {
            sdp_load_box(a, 0, box_hold);
#pragma unroll
            for (int c = 0; c < SDP_NU; ++c) {
                box_hold.lo[c] = sdp_uniform(box_hold.lo[c]); box_hold.hi[c] = sdp_uniform(box_hold.hi[c]);
                box_hold.step[c] = sdp_uniform(box_hold.step[c]); box_hold.delta[c] = sdp_uniform(box_hold.delta[c]);
                box_hold.n[c] = sdp_uniform(box_hold.n[c]);
            }
            box_hold.total = sdp_uniform(box_hold.total);
            box_c = &box_hold;
        }
        if (lane < Wn) { w_hold = ((const sdp_real *)a.wgrid)[lane]; w_mine = &w_hold; }
    }
#if SDP_COL_SHIFT
    if (threadIdx.x < 2) sdp_col_shift_reset(sdp_lds, threadIdx.x);
    __syncthreads();
#endif
    if (unit < u_end) {                                    // trailing cells (and control table, shifts) of the first unit
        sdp_real xn[SDP_D];
        sdp_col_coords(a, sdp_col_of_unit(a, unit), xn);
        sdp_col_phase_w(a, tg, s, xn, nullptr, t);
#if SDP_COL_UTAB
        sdp_col_phase_u(a, sdp_lds.utab[0], xn, t, 0, 64, box_c, filt.psum, k_rows, x_cap, (double)filt.psum);      // (one wave: it also reduces the table's statistics)
#endif
#if SDP_COL_SHIFT
        sdp_col_phase_shift(a, sdp_lds, lead, xn, t, 0);
#endif
    }
    while (unit < u_end) {
        const int64_t col = sdp_col_of_unit(a, unit);
        const int part = (int)((unsigned)unit % (unsigned)a.col_splits);      // (32-bit: units < 2^31)
        const int i_lo = (int)((unsigned)(N0 * part) / (unsigned)a.col_splits);
        const int i_hi = (int)((unsigned)(N0 * (part + 1)) / (unsigned)a.col_splits);
        sdp_real x[SDP_D];
        sdp_col_coords(a, col, x);
        const int r = (int)threadIdx.x;                    // the table row this thread reduces
        // ---- tail of the table, partial sums of the reduced table
        SDP_RES_MARK(tm0);
        __syncthreads();                                   // the previous unit has left the table; this unit's cells are published
#if SDP_STAMP == 2
        { const unsigned long long tmb = __builtin_amdgcn_s_memtime(); tm_top += tmb - tm0; tm0 = tmb; }
#endif
        int nx = 0;
        if (wave == waves - 1 && lane == 0) nx = (int)atomicAdd(claim, 1u);      // (its round trip hides under the builds)
#if SDP_COL_SHIFT
        if (threadIdx.x == 0) sdp_col_shift_reset(sdp_lds, upar ^ 1);            // (its readers left at the barrier above)
        SdpColShiftCol shc;
        sdp_col_shift_col(sdp_lds, lead, upar, shc);
        sdp_col_shift_zero(sdp_lds, shc);
#endif
#if SDP_COL_TAIL_HOLD
        sdp_held_rows held[SDP_HOLD_NW][SDP_HOLD_NJ];
        sdp_colres_tail<true>(a, tg, s, held);
#else
        if (!(SDP_DIAG_SKIP & 1)) sdp_col_phase_a<false>(a, tg, s, C, R, kept);
#endif
        __syncthreads();
#if SDP_COL_SHIFT
        sdp_col_shift_reduce(a, sdp_lds, filt, shc, parity, upar, C, R);         // (adds into the cleared lattice)
#else
        sdp_real acc_t = (sdp_real)0, big = (sdp_real)0;
        if (r < N0 && !(SDP_DIAG_SKIP & 1)) {
#pragma unroll SDP_COL_FILTER_RUNROLL
            for (int w = 0; w < R; ++w) {
                const sdp_real v = sdp_lds.T[w * N0 + r];
                acc_t = acc_t + pw[C + w] * v;
                big = sdp_vmax_abs(big, v);
            }
        }
#endif
        __syncthreads();                                   // the tail has been read
        // ---- head of the table, the reduced table
        if (!(SDP_DIAG_SKIP & 2)) sdp_col_phase_a<false>(a, tg, s, 0, C);
        if (wave == waves - 1 && lane == 0) sdp_lds.next_unit = nx;
        __syncthreads();
        __builtin_amdgcn_s_setprio(SDP_COL_B_PRIO);
#if SDP_COL_SHIFT
        sdp_col_shift_reduce(a, sdp_lds, filt, shc, parity, upar, 0, C);
#else
        {
            sdp_real dmax = (sdp_real)0;
            if (r < N0) {
                sdp_real acc = (sdp_real)0;
#pragma unroll SDP_COL_FILTER_RUNROLL
                for (int w = 0; w < ((SDP_DIAG_SKIP & 2) ? 0 : C); ++w) {
                    const sdp_real v = sdp_lds.T[w * N0 + r];
                    acc = acc + pw[w] * v;
                    big = sdp_vmax_abs(big, v);
                }
                acc = acc + acc_t;
                // (>= tiny / cu: the radius never drops below `tiny`; a NaN entry, which the max skips, shows in acc)
                dmax = acc == acc ? filt.pcap * big + filt.floor : (sdp_real)INFINITY;
                sdp_lds.ad[r] = acc;
            }
            dmax = sdp_wave_max(dmax);
            if (lane == 0) atomicMax(&sdp_lds.dcol[parity], (unsigned long long)__double_as_longlong((double)dmax));
        }
#endif
        (void)r;
        __syncthreads();                                   // A[r] and the column's bound are complete
        SDP_RES_MARK(tm1);
#if SDP_STAMP == 2
        tm_build += tm1 - tm0;
#endif
        const sdp_real dcol = sdp_col_filter_dcol(sdp_lds, parity);
        parity ^= 1;
        const int64_t next_unit = u_base + sdp_lds.next_unit;
        // ---- first pass (every control of this lane's node), second pass over the head
        // the axis the FIRST pass locates its positions on: axis 0, or the shifted lattice of this column
        SdpLeadAxis lead1 = lead;
#if SDP_COL_SHIFT
        lead1.koff = (sdp_real)shc.kmin;
        lead1.ordm2 = shc.ok ? shc.rows - 2 : 0;
#endif
        const int i = i_lo + (int)threadIdx.x;
        const bool live = i < i_hi;
        const int64_t node = col * N0 + (live ? i : i_hi - 1);
        SdpBox box;
        if (box_c) box = *box_c;
        else sdp_load_box(a, node, box);
        sdp_real best = INFINITY;
        sdp_real diag_cnt = (sdp_real)-1;                  // (SDP_DIAG_BNB_COUNT builds)
        (void)diag_cnt;
        int ibest = INT_MAX;
        SdpColresCand cd;                                  // survivors this lane carries through the rebuild, in lattice order
        cd.n = 0;
#pragma unroll
        for (int j = 0; j < SDP_COLRES_K; ++j) {
            cd.acc[j] = cd.g[j] = cd.lam0[j] = cd.oml0[j] = (sdp_real)0;
            cd.q0[j] = 0;
            cd.idx[j] = INT_MAX;
#pragma unroll
            for (int c = 0; c < SDP_NU; ++c) cd.u[j][c] = (sdp_real)0;
        }
        if (live && (SDP_DIAG_SKIP & 4)) {
            x[0] = axis0[i];
            cd.n = 1; cd.idx[0] = 3; ibest = 3;
            sdp_controls_at(box, 3, cd.u[0]);
            cd.g[0] = sdp_model_cost(x, cd.u[0], (sdp_real)0, t);
            sdp_colres_cell(lead, x, cd.u[0], (sdp_real)0, t, cd.q0[0], cd.lam0[0], cd.oml0[0]);
        } else
        if (live) {
            x[0] = axis0[i];
#if SDP_COL_LEAN2 && SDP_COL_SHIFT
            // the short first pass on the shifted lattice (sdp_lean2s_pass1)
            const sdp_real *utab = sdp_lds.utab[upar];
            const sdp_real *ust = utab + SDP_COL_UTAB * SDP_COL_UTAB_N;          // a_lo, a_hi, max |h| (or NaN)
            const sdp_real X = sdp_model_lead_x(x, t), K = sdp_model_cost_x(x, t);
            const int bits = 32 - __clz(max(box.total - 1, 1));
            const int mask = (1 << bits) - 1;
            struct { sdp_real f1, f2, f3; int i1, i2; } bd;
            bd.f1 = bd.f2 = bd.f3 = INFINITY;
            int q_e;
            sdp_real lam_lo, lam_hi, b_max = (sdp_real)0;
            // (the node's bound needs nothing of the pass itself: it comes first, the branch and bound uses it)
            if (axis_mode == 2) {
                sdp_lean2s_cell<2>(lead1, SDP_LEAN2_LEAD(X, ust[0]), q_e, lam_lo);
                sdp_lean2s_cell<2>(lead1, SDP_LEAN2_LEAD(X, ust[1]), q_e, lam_hi);
            } else if (axis_mode == 1) {
                sdp_lean2s_cell<1>(lead1, SDP_LEAN2_LEAD(X, ust[0]), q_e, lam_lo);
                sdp_lean2s_cell<1>(lead1, SDP_LEAN2_LEAD(X, ust[1]), q_e, lam_hi);
            } else {
                sdp_lean2s_cell<0>(lead1, SDP_LEAN2_LEAD(X, ust[0]), q_e, lam_lo);
                sdp_lean2s_cell<0>(lead1, SDP_LEAN2_LEAD(X, ust[1]), q_e, lam_hi);
            }
            const sdp_real l_cap = sdp_vmax_abs(sdp_vmax_abs((sdp_real)1, lam_lo), lam_hi);
            const sdp_real h_cap = (((sdp_real)1 + (sdp_real)2 * (l_cap + shc.lc)) * ((sdp_real)3 + shc.es)) * dcol;
            const sdp_real s_node = fma(filt.pcap, fabs(K) + ust[2], h_cap);
            const bool bad = !filt.ok || !shc.ok || !(s_node < SDP_COL_FILTER_LIMIT) ||
                             !(fabs(lam_lo) + fabs(lam_hi) + shc.lc < (sdp_real)1073741824.0) || bits > 24 || box.total > SDP_COL_UTAB_N;
            const sdp_real radius0 = (filt.cu + (sdp_real)(SDP_COL_FILTER_SCALE) * ldexp(SDP_COL_FILTER_EPS, bits + 1)) * s_node;     // the roundings' part
#if SDP_COL_BNB
            // a block is skipped when its lower bound, less the largest B' of its rows, exceeds F' + B' of the guess by more
            // than twice the rounding radius + 24 u S_node (sdp_short_bnb).  Only where the column's blocks are in order and
            // |X| is small enough for the bounds' positions (ust[3]); a wave with a node that does not qualify, or a column
            // whose lattice is not usable, takes the full pass.
            const sdp_real slack = fma((sdp_real)2, radius0, ((sdp_real)12 * SDP_COL_FILTER_EPS) * s_node);
            const bool bnb = fabs(X) < ust[3] && shc.ok;
            if (__all(bnb)) {
                auto ins = [&](double Fq) {
                    bd.f3 = sdp_vmin(bd.f3, sdp_vmax(bd.f2, Fq));
                    bd.f2 = sdp_vmin(bd.f2, sdp_vmax(bd.f1, Fq));
                    bd.f1 = sdp_vmin(bd.f1, Fq);
                };
                if (axis_mode == 2) sdp_short_bnb<2, false>(sdp_lds.ad, utab, filt, lead1, X, k_rows, 0, box.total, mask, slack, guess, ins, diag_cnt, &b_max);
                else if (axis_mode == 1) sdp_short_bnb<1, false>(sdp_lds.ad, utab, filt, lead1, X, k_rows, 0, box.total, mask, slack, guess, ins, diag_cnt, &b_max);
                else sdp_short_bnb<0, false>(sdp_lds.ad, utab, filt, lead1, X, k_rows, 0, box.total, mask, slack, guess, ins, diag_cnt, &b_max);
            } else
#endif
            {
                if (axis_mode == 2) sdp_lean2s_pass1<2>(sdp_lds.ad, utab, filt, lead1, X, box.total, mask, bd.f1, bd.f2, bd.f3, b_max);
                else if (axis_mode == 1) sdp_lean2s_pass1<1>(sdp_lds.ad, utab, filt, lead1, X, box.total, mask, bd.f1, bd.f2, bd.f3, b_max);
                else sdp_lean2s_pass1<0>(sdp_lds.ad, utab, filt, lead1, X, box.total, mask, bd.f1, bd.f2, bd.f3, b_max);
            }
            // (b_max: the largest B' among the controls the pass evaluated -- the ones it skipped are ruled out with their own)
            const sdp_real radius = fma((sdp_real)(SDP_COL_FILTER_SCALE), b_max, radius0);
            bd.i1 = bd.f1 < (sdp_real)INFINITY ? (__double2loint(bd.f1) & mask) : INT_MAX;
            bd.i2 = bd.f2 < (sdp_real)INFINITY ? (__double2loint(bd.f2) & mask) : INT_MAX;
#elif SDP_COL_LEAN2
            const sdp_real *utab = sdp_lds.utab[upar];
            const sdp_real *ust = utab + SDP_COL_UTAB * SDP_COL_UTAB_N;          // a_lo, a_hi, max |h| (or NaN)
            const sdp_real X = sdp_model_lead_x(x, t), K = sdp_model_cost_x(x, t);
            const int bits = 32 - __clz(max(box.total - 1, 1));
            const int mask = (1 << bits) - 1;
            struct { sdp_real f1, f2; int i1, i2; } bd;
            bd.f1 = bd.f2 = INFINITY;
            bd.i2 = INT_MAX;
            int q_e;
            sdp_real lam_lo, lam_hi;
            // (the node's bound and radius need nothing of the pass itself: they come first, the branch and bound uses them)
            if (axis_mode == 2) {
                sdp_lean2_cell<2>(lead, SDP_LEAN2_LEAD(X, ust[0]), q_e, lam_lo);
                sdp_lean2_cell<2>(lead, SDP_LEAN2_LEAD(X, ust[1]), q_e, lam_hi);
            } else if (axis_mode == 1) {
                sdp_lean2_cell<1>(lead, SDP_LEAN2_LEAD(X, ust[0]), q_e, lam_lo);
                sdp_lean2_cell<1>(lead, SDP_LEAN2_LEAD(X, ust[1]), q_e, lam_hi);
            } else {
                sdp_lean2_cell<0>(lead, SDP_LEAN2_LEAD(X, ust[0]), q_e, lam_lo);
                sdp_lean2_cell<0>(lead, SDP_LEAN2_LEAD(X, ust[1]), q_e, lam_hi);
            }
            const sdp_real l_cap = sdp_vmax_abs(sdp_vmax_abs((sdp_real)1, lam_lo), lam_hi);
            const sdp_real s_node = fma(filt.pcap, fabs(K) + ust[2], ((sdp_real)1 + (sdp_real)2 * l_cap) * dcol);
            const bool bad = !filt.ok || !(s_node < SDP_COL_FILTER_LIMIT) || !(fabs(lam_lo) + fabs(lam_hi) < (sdp_real)1073741824.0) ||
                             bits > 24 || box.total > SDP_COL_UTAB_N;
            const sdp_real radius = (filt.cu + (sdp_real)(SDP_COL_FILTER_SCALE) * ldexp(SDP_COL_FILTER_EPS, bits + 1)) * s_node;
#if SDP_COL_BNB
            // a block is skipped when its lower bound exceeds f1 by more than 2 radius + 16 u S_node (8 x EPS = 16 u).
            // Only where the column's blocks are in order and |X| is small enough for the bounds' positions (ust[3]:
            // sdp_col_phase_u); a wave with a node that does not qualify takes the full pass.
            const sdp_real slack = fma((sdp_real)2, radius, ((sdp_real)8 * SDP_COL_FILTER_EPS) * s_node);
            const bool bnb = fabs(X) < ust[3];
            if (__all(bnb)) {                                  // (over the lanes that have a node)
                auto ins = [&](double Fq) {
                    bd.f2 = sdp_vmin(bd.f2, sdp_vmax(bd.f1, Fq));
                    bd.f1 = sdp_vmin(bd.f1, Fq);
                };
                if (axis_mode == 2) sdp_short_bnb<2, false>(sdp_lds.ad, utab, filt, lead, X, k_rows, 0, box.total, mask, slack, guess, ins, diag_cnt);
                else if (axis_mode == 1) sdp_short_bnb<1, false>(sdp_lds.ad, utab, filt, lead, X, k_rows, 0, box.total, mask, slack, guess, ins, diag_cnt);
                else sdp_short_bnb<0, false>(sdp_lds.ad, utab, filt, lead, X, k_rows, 0, box.total, mask, slack, guess, ins, diag_cnt);
            } else
#endif
            {
                if (axis_mode == 2) sdp_lean2_pass1<2>(sdp_lds.ad, utab, filt, lead, X, box.total, mask, bd.f1, bd.f2);
                else if (axis_mode == 1) sdp_lean2_pass1<1>(sdp_lds.ad, utab, filt, lead, X, box.total, mask, bd.f1, bd.f2);
                else sdp_lean2_pass1<0>(sdp_lds.ad, utab, filt, lead, X, box.total, mask, bd.f1, bd.f2);
            }
            bd.i1 = bd.f1 < (sdp_real)INFINITY ? (__double2loint(bd.f1) & mask) : INT_MAX;
#else
            SdpColBounds bd;
            bd.f1 = bd.f2 = bd.f3 = INFINITY;
            bd.s_max = bd.s_sum = bd.p_max = bd.b_max = (sdp_real)0;
            bd.i1 = bd.i2 = INT_MAX;
            const sdp_real *utab = sdp_lds.utab[upar];
            const bool plain = SDP_NU == 1 && box.n[0] > 1 && box.step[0] != (sdp_real)0;
            if (plain) {
                if (axis_mode == 2) sdp_col_filter_pass1<true, 2>(sdp_lds.ad, utab, filt, lead1, box, x, t, 0, box.total, bd);
                else if (axis_mode == 1) sdp_col_filter_pass1<true, 1>(sdp_lds.ad, utab, filt, lead1, box, x, t, 0, box.total, bd);
                else sdp_col_filter_pass1<true, 0>(sdp_lds.ad, utab, filt, lead1, box, x, t, 0, box.total, bd);
            } else {
                if (axis_mode == 2) sdp_col_filter_pass1<false, 2>(sdp_lds.ad, utab, filt, lead1, box, x, t, 0, box.total, bd);
                else if (axis_mode == 1) sdp_col_filter_pass1<false, 1>(sdp_lds.ad, utab, filt, lead1, box, x, t, 0, box.total, bd);
                else sdp_col_filter_pass1<false, 0>(sdp_lds.ad, utab, filt, lead1, box, x, t, 0, box.total, bd);
            }
            // the radius of the lean first pass (sdp_col_lean_core), on the shifted lattice with its H and B' (sdp_col_shift_reduce)
#if SDP_COL_SHIFT
            const sdp_real h_cap = (((sdp_real)1 + (sdp_real)2 * (bd.p_max + shc.lc)) * ((sdp_real)3 + shc.es)) * dcol;
            const sdp_real s_node = fma(filt.ratio, bd.s_sum + h_cap, h_cap);
            const bool bad = !filt.ok || !shc.ok || !(s_node < SDP_COL_FILTER_LIMIT) || !(bd.p_max + shc.lc < (sdp_real)1073741824.0);
            const sdp_real radius = fma(filt.cu, s_node, (sdp_real)(SDP_COL_FILTER_SCALE) * bd.b_max);
#else
            const sdp_real h_cap = ((sdp_real)1 + (sdp_real)2 * bd.p_max) * dcol;        // (1 + 2L) D
            const sdp_real s_node = fma(filt.ratio, bd.s_sum + h_cap, h_cap);
            const bool bad = !filt.ok || !(s_node < SDP_COL_FILTER_LIMIT) || !(bd.p_max < (sdp_real)1073741824.0);
            const sdp_real radius = filt.cu * s_node;
#endif
#endif  // SDP_COL_LEAN2
            const sdp_real m_hi = bd.f1 + radius;              // >= the minimum of E over the node
            const bool single = !bad && bd.i1 != INT_MAX && bd.f2 - radius > m_hi;
            // exactly two survivors (the lattice points either side of the continuous optimum): both carried
#if SDP_COL_LEAN2 && !SDP_COL_SHIFT
            const bool pair = false;
#else
            const bool pair = SDP_COL_TOP2 && !bad && !single && bd.i2 != INT_MAX && bd.f3 - radius > m_hi;
#endif
            if (single || pair) {
                // the survivors ARE the only candidates for the reference's argmin
                // (their controls, cost and -- a stock the perturbation does not reach -- cell: once)
                cd.n = single ? 1 : 2;
                cd.idx[0] = single ? bd.i1 : min(bd.i1, bd.i2);
                if (SDP_COLRES_K > 1) cd.idx[SDP_COLRES_K - 1] = single ? bd.i1 : max(bd.i1, bd.i2);
                else ibest = bd.i1;
#pragma unroll
                for (int j = 0; j < SDP_COLRES_K; ++j) {
                    sdp_controls_at(box, cd.idx[j], cd.u[j]);
                    cd.g[j] = sdp_model_cost(x, cd.u[j], (sdp_real)0, t);
#if !SDP_LEAD_HAS_W
                    sdp_colres_cell(lead, x, cd.u[j], (sdp_real)0, t, cd.q0[j], cd.lam0[j], cd.oml0[j]);
#endif
                }
            } else {
                // near-ties or special values: the candidates (all controls of a marked node) the long way, from
                // global memory -- the same operations, no table needed -- compared like the reference compares
                for (int ci = 0; ci < box.total; ++ci) {
                    sdp_real u[SDP_NU];
                    sdp_controls_at(box, ci, u);
                    bool cand = bad;
                    if (!cand) {
#if SDP_COL_LEAN2 && SDP_COL_SHIFT
                        sdp_real b_unused = (sdp_real)0;       // (the radius holds the largest B' of the node already)
                        const sdp_real F = lead.pow2 ? sdp_lean2s_value<1>(sdp_lds.ad, utab, filt, lead1, X, ci, b_unused)
                                                     : sdp_lean2s_value<0>(sdp_lds.ad, utab, filt, lead1, X, ci, b_unused);
#elif SDP_COL_LEAN2
                        const sdp_real F = lead.pow2 ? sdp_lean2_value<1>(sdp_lds.ad, utab, filt, lead, X, ci)
                                                     : sdp_lean2_value<0>(sdp_lds.ad, utab, filt, lead, X, ci);
#else
                        sdp_real F, pm = (sdp_real)0, gm = (sdp_real)0, bm = (sdp_real)0;
                        if (lead.pow2) sdp_col_lean_eval<1>(sdp_lds.ad, filt, lead1, x, u, t, F, pm, gm, bm);
                        else sdp_col_lean_eval<0>(sdp_lds.ad, filt, lead1, x, u, t, F, pm, gm, bm);
#endif
                        cand = !(F - radius > m_hi);
                    }
                    if (cand) {
                        const sdp_real jc = sdp_col_cost_global<false>(a, tg, s, wts, lead, x, u, t);
                        if (ibest == INT_MAX || sdp_better_seq(jc, best)) { best = jc; ibest = ci; }
                    }
                }
            }
        }
#if SDP_STAMP == 2
        tm_p1 += __builtin_amdgcn_s_memtime() - tm1;
#endif
        // (the lanes of a wave run the same code: two chains where any lane carries two survivors)
        const bool two = SDP_COLRES_K > 1 && __any(cd.n == 2);
        if (SDP_DIAG_SKIP & 8) {}
        else if (SDP_COLRES_K > 1 && two) sdp_colres_partial<SDP_COLRES_K>(sdp_lds.T, wts, lead, x, t, 0, C, 0, cd);
        else if (cd.n) sdp_colres_partial<1>(sdp_lds.T, wts, lead, x, t, 0, C, 0, cd);
#if SDP_COL_TAIL_KEEP
        // ---- second pass over the tail, from the copy its first build left in global memory (no barrier: the table
        // is not touched again before the top of the next unit)
        if (cd.n) sdp_colres_partial_kept(kept, wts, C, Wn, cd);
#ifdef SDP_DIAG_KEEP_BARRIER
        __syncthreads();
#endif
#else
        __syncthreads();                                   // the head has been read
        // ---- the tail again, second pass over it
        __builtin_amdgcn_s_setprio(0);
#if SDP_COL_TAIL_HOLD
        sdp_colres_tail<false>(a, tg, s, held);
#else
        if (!(SDP_DIAG_SKIP & 16)) sdp_col_phase_a<false>(a, tg, s, C, R);
#endif
        __syncthreads();
        __builtin_amdgcn_s_setprio(SDP_COL_B_PRIO);
        if (SDP_DIAG_SKIP & 32) {}
        else if (SDP_COLRES_K > 1 && two) sdp_colres_partial<SDP_COLRES_K>(sdp_lds.T, wts, lead, x, t, C, Wn, C, cd);
        else if (cd.n) sdp_colres_partial<1>(sdp_lds.T, wts, lead, x, t, C, Wn, C, cd);
#endif
        if (live) {
            // the carried survivors in lattice order, compared like the reference compares
#pragma unroll
            for (int j = 0; j < SDP_COLRES_K; ++j) {
                if (SDP_COLRES_K == 1) { if (cd.n) best = cd.acc[0]; }          // (its index is in ibest already)
                else if (j < cd.n && (ibest == INT_MAX || sdp_better_seq(cd.acc[j], best))) { best = cd.acc[j]; ibest = cd.idx[j]; }
            }
#ifdef SDP_DIAG_BNB_COUNT
            best = diag_cnt;
#endif
            if (!(SDP_DIAG_SKIP & 128)) sdp_col_store(a, node, box, best, ibest);
            if (ibest != INT_MAX) guess = ibest;
        }
        {
            // the next unit's column-level tables, a wave each (nothing reads the cells after the rebuild of the tail; here, after
            // the stores, a wave holds next to nothing in registers: between the two halves of the second pass these tables
            // cost the branch-and-bound build five reloads of spilled registers per unit, each a memory round trip)
            const int nxu = __builtin_amdgcn_readfirstlane(sdp_lds.next_unit);
            if (u_base + nxu < u_end && !(SDP_DIAG_SKIP & 64)) {
                sdp_real xn[SDP_D];
                sdp_col_coords(a, sdp_col_of_unit(a, u_base + nxu), xn);
                sdp_col_phase_w(a, tg, s, xn, nullptr, t, (waves - 1) * 64, 64, w_mine);
#if SDP_COL_UTAB
                sdp_col_phase_u(a, sdp_lds.utab[upar ^ 1], xn, t, max(waves - 2, 0) * 64, 64, box_c, filt.psum, k_rows, x_cap, (double)filt.psum,
                                SDP_LEAN2_A_FIXED_ON && tables_made >= 2);
                ++tables_made;
#endif
#if SDP_COL_SHIFT
                sdp_col_phase_shift(a, sdp_lds, lead, xn, t, upar ^ 1, max(waves - 3, 0) * 64, 64);
#endif
            }
        }
        __builtin_amdgcn_s_setprio(0);
        upar ^= 1;
        unit = next_unit;
    }
    // the last workgroup to run out of units leaves the counters at zero for the next launch
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(a.claim + 256, 1u) == gridDim.x - 1) {
            for (int k = 0; k < 8; ++k) atomicExch(a.claim + 32 * k, 0u);
            atomicExch(a.claim + 256, 0u);
        }
    }
#if SDP_STAMP == 2
    if (a.stamps && (threadIdx.x & 63) == 0) {
        tm_all = __builtin_amdgcn_s_memtime() - tm_start;
        unsigned long long *o = a.stamps + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
        o[0] = tm_build; o[1] = tm_p1; o[2] = tm_top; o[3] = tm_all;
    }
#else
    SDP_STAMP_END(a);
#endif
}

// fixed-policy backup: the table a chunk of C perturbation points at a time, the node's expectation carried in a register
extern "C" __global__ void __launch_bounds__(SDP_COL_THREADS) sdp_evalpol_col(SdpSweepArgs a)
{
    __shared__ SdpColLds sdp_lds;
    constexpr int N0 = SDP_COL_N0;
    constexpr int Wn = SDP_COL_W;
    constexpr int C = SDP_COL_WRES;
    sdp_trap_unless(a.n_lead == N0 && a.W == Wn);
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
    for (int64_t unit = walk.unit; unit < walk.end; unit += walk.stride) {
        const int64_t col = sdp_col_of_unit(a, unit);
        const int part = (int)((unsigned)unit % (unsigned)a.col_splits);      // (32-bit: units < 2^31)
        const int i_lo = (int)((unsigned)(N0 * part) / (unsigned)a.col_splits);
        const int i_hi = (int)((unsigned)(N0 * (part + 1)) / (unsigned)a.col_splits);
        sdp_real x[SDP_D];
        sdp_col_coords(a, col, x);
        __syncthreads();
        sdp_col_phase_w(a, tg, s, x, nullptr, t);
        // (a launch with fewer threads than nodes walks them in rounds: each round builds the chunks again)
        for (int i0 = i_lo; i0 < i_hi; i0 += (int)blockDim.x) {
            const int i = i0 + (int)threadIdx.x;
            const bool live = i < i_hi;
            const int64_t node = col * N0 + (live ? i : i_hi - 1);
            SdpColresCand cd;
            cd.n = 1;
            cd.acc[0] = (sdp_real)0;
            cd.idx[0] = 0;
            x[0] = axis0[live ? i : i_hi - 1];
#pragma unroll
            for (int c = 0; c < SDP_NU; ++c) cd.u[0][c] = ((const sdp_real *)a.pol_in)[node * SDP_NU + c];
            cd.g[0] = sdp_model_cost(x, cd.u[0], (sdp_real)0, t);
            cd.q0[0] = 0; cd.lam0[0] = cd.oml0[0] = (sdp_real)0;
#if !SDP_LEAD_HAS_W
            sdp_colres_cell(lead, x, cd.u[0], (sdp_real)0, t, cd.q0[0], cd.lam0[0], cd.oml0[0]);
#endif
            for (int w0 = 0; w0 < Wn; w0 += C) {
                const int cnt = min(C, Wn - w0);
                __syncthreads();                           // the cells are published / the previous chunk has been read
                sdp_col_phase_a<true>(a, tg, s, w0, cnt);
                __syncthreads();
                sdp_colres_partial<1>(sdp_lds.T, wts, lead, x, t, w0, w0 + cnt, w0, cd);
            }
            const sdp_real acc = cd.acc[0];
            if (live) sdp_store_J<sdp_real>(a, node, col, acc);
        }
    }
}
