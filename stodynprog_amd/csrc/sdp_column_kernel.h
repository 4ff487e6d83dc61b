// sdp_column_kernel.h -- Bellman backup for "storage-separable" models:
//
//     x0' = f0(x, u[, w])       leading axis: the controlled stock
//     xk' = fk(x1.., w), k>=1   trailing axes: an exogenous process driven by w
//
// (inventory-free storage problems: every storage-control example of the
// reference has this shape).  The reference interpolates J_next at every
// lattice cell with the nested lerp of multilinear_cython.pyx:88-300, axis 0
// OUTERMOST:
//     out = (1-lam0) * inner(q0) + lam0 * inner(q0+1)
//     inner(r) = lerp over axes 1..d-1 of V[r, ...] at the cell of (x1'..)
// For a separable model inner(r) depends on (r, w) only -- not on the control
// and not on the node's position along axis 0.  So one workgroup takes a whole
// COLUMN of nodes (fixed x1.., all x0) and
//   phase W  locates the trailing cell of each perturbation point      (W items)
//   phase A  tabulates T[w][r] = inner(r) for every row r of axis 0 in LDS
//            (W x N0 reals, e.g. 32 x 256 x 8 B = 64 KiB of the CU's 160 KiB);
//            with the value array stored axis-0-fastest the 2^(d-1) vertex
//            reads of a (w, r..r+255) strip are fully coalesced
//   phase B  runs the node x control x perturbation loop out of LDS: per cell
//            two ds_read_b64 (rows q0, q0+1; conflict-free: bank = row), one
//            lerp, the cost add and the weighted accumulation in w order;
//            then the (J, index) butterfly argmin over the node's lanes.
// Every floating-point operation of the reference is executed with the same
// operands in the same order (the table only removes repeats), so results are
// bit-identical to the generic kernel and to the oracle.
//
// Needs from the generated unit (besides what sdp_sweep_kernel.h needs):
//   sdp_model_lead(x, u, w, t)       -> x0'
//   SDP_LEAD_HAS_W                   0: x0' does not depend on w, its cell is
//                                    located once per control (the storage case)
//   sdp_model_trail(x, u, w, t, xn)  fills xn[1..SDP_D-1]  (x[0] is not read)
//   SDP_TRAIL_HAS_U                  0: the trailing next states do not depend on the
//                                    control (storage-separable).  1: they do -- the
//                                    table then differs from control to control and
//                                    is rebuilt for each one (kernels at the end of
//                                    this file); needs the nodes of a column to share
//                                    their control values (box independent of x0)
//   sdp_model_cost(x, u, w, t)       -> g
//   SDP_COST_HAS_W                   0: g is hoisted out of the w loop
//   SDP_COL_FILTER                   1: certified expectation-first filter in phase B (SdpColFilter below):
//                                    a first pass over every control on a table reduced over w -- lean
//                                    (SDP_COL_LEAN, 8-byte reals) or wide (SDP_COL_WIDE, 4-byte reals) --
//                                    and the reference's operations on the survivors only; same bits
//   SDP_COL_UTAB (+ _N)              K > 0: sdp_model_utab / sdp_model_lead_tab / sdp_model_cost_tab are
//                                    provided too: the K sub-expressions of x0' and of the cost that depend
//                                    on the control but not on x0, tabulated once per (column, control)
//   SDP_COL_N0, SDP_COL_W            points of axis 0 / perturbation points (1 if
//                                    deterministic): compile-time, they size the
//                                    statically allocated LDS table
//   SDP_COL_ROWS                     rows of axis 0 the table holds (default: all
//                                    SDP_COL_N0).  Fewer = ROW WINDOW, for grids
//                                    whose W x N0 table exceeds the LDS of a CU:
//                                    a unit (column, segment of nodes) tabulates
//                                    only the rows its next states reach -- a
//                                    prediction from the first / last control of
//                                    every node; a control whose rows fall outside
//                                    the window interpolates from global memory,
//                                    same operations, so the result never depends
//                                    on the prediction
#pragma once
#include "sdp_sweep_kernel.h"
#include <type_traits>

#if SDP_D >= 2

#ifndef SDP_COL_THREADS
#define SDP_COL_THREADS 512
#endif
#ifndef SDP_COL_UNROLL_W
#define SDP_COL_UNROLL_W 4       // unroll factor of the perturbation loop (in batches)
#endif
#ifndef SDP_COL_A_GROUP
#define SDP_COL_A_GROUP 4        // table entries per thread whose vertex loads are issued together
#endif
#ifndef SDP_COL_A_ORDER
#define SDP_COL_A_ORDER 0        // table build: 0 entries dealt round-robin; 1 a thread keeps its row and takes
#endif                           //   consecutive w; 2 a thread keeps its w (SDP_COL_A_LW lanes per w) and walks the rows
#ifndef SDP_COL_A_LW
#define SDP_COL_A_LW 16
#endif
#ifndef SDP_COLU_WIDE_LOADS
#define SDP_COLU_WIDE_LOADS 0    // table-per-control kernel: 16-byte vertex loads in its table build (see sdp_colu_phase_a)
#endif
#ifndef SDP_COLU_A_GROUP
// perturbation points per round of that build.  Same-box A/B on the control-coupled benchmark (256^3 x 64 x 32
// fp64): 8-byte loads 59.3 ms (groups of 4; 2: 60.4); 16-byte loads, groups of 1 / 2 / 3 / 4 / 8: 38.4 / 37.4 /
// 41.0 / 43.7 / 73.6 ms
#define SDP_COLU_A_GROUP (SDP_COLU_WIDE_LOADS ? 2 : SDP_COL_A_GROUP)
#endif
#ifndef SDP_COL_A_WIDE_LOADS
#define SDP_COL_A_WIDE_LOADS 0   // table build, order 2: 16-byte vertex loads, two adjacent rows per lane (see there)
#endif
#ifndef SDP_COL_MIN_WAVES
#define SDP_COL_MIN_WAVES 4      // waves per SIMD the register allocation must allow (2 workgroups per CU)
#endif
#if defined(SDP_WAVES_CAP) && SDP_COL_MIN_WAVES > SDP_WAVES_CAP
// (a rebuild with more registers: the first build's spill code was unsafe -- codegen.spill_hazards, _native.compile_model)
#undef SDP_COL_MIN_WAVES
#define SDP_COL_MIN_WAVES SDP_WAVES_CAP
#endif
#ifndef SDP_COL_WPAIR
#define SDP_COL_WPAIR 0          // 1 (4-byte reals): the table interleaves perturbation points 2k and
#endif                           //    2k+1, T2[k][r] = (inner_2k(r), inner_2k+1(r)); one 8-byte LDS read
                                 //    serves two cells and the cell arithmetic runs as packed fp32
#ifndef SDP_COL_BATCH
// perturbation points whose LDS reads are issued together (x SDP_COL_UNROLL_U controls).
// Same-box A/B on MI355X, 256^3 x 64 x 32: plain fp64 table 4 -> 8.96 ms, 2 -> 9.23 ms (Searev
// 128^3: 0.460 vs 0.481 ms); the pair layout of 4-byte reals spills at 4 (10.3 vs 5.5 ms).
#if SDP_COL_WPAIR || SDP_LEAD_HAS_W
#define SDP_COL_BATCH 2
#else
#define SDP_COL_BATCH 4
#endif
#endif
#ifndef SDP_COL_UNROLL_U
// controls evaluated together per lane (independent chains).  Same-box A/B, 256^3 x 64 x 32:
// fp64 plain table 2 -> 8.96 ms, 4 -> 9.08 ms; fp32 pair layout 2 -> 5.35 ms, 4 -> 5.19 ms
#if SDP_COL_WPAIR
#define SDP_COL_UNROLL_U 4
#else
#define SDP_COL_UNROLL_U 2
#endif
#endif
#if SDP_COL_WPAIR && (!SDP_HAS_W || SDP_LEAD_HAS_W)
#error "SDP_COL_WPAIR needs a perturbation and an x0' that does not depend on it"
#endif
#ifndef SDP_TRAIL_HAS_U
#define SDP_TRAIL_HAS_U 0
#endif
#ifndef SDP_COL_ROWS
#define SDP_COL_ROWS SDP_COL_N0
#endif
#ifndef SDP_COL_FILTER
#define SDP_COL_FILTER 0         // 1: certified expectation-first filter in phase B (see sdp_col_filter_*)
#endif
#ifndef SDP_COL_FILTER_UNROLL
// controls per round of the filter's first pass.  Same-box A/B, 256^3 x 64 x 32 fp64 with the lean
// pass and the control table: 2 -> 2.05 ms, 4 -> 1.94, 8 -> 1.89, 16 -> 1.90; 4-byte reals keep 4
#define SDP_COL_FILTER_UNROLL (sizeof(SDP_REAL) == 8 ? 8 : 4)
#endif
#ifndef SDP_COL_B_PRIO
#define SDP_COL_B_PRIO 3         // wave priority (s_setprio) while in phase B
#endif
#ifndef SDP_COL_FILTER_RUNROLL
#define SDP_COL_FILTER_RUNROLL 16   // table entries in flight per thread in the reduction over w
#endif
#ifndef SDP_COL_LEAN
// Lean first pass of the filter (8-byte reals by default): ONE error bound per node -- from the
// largest |lam0| and the sum of the |F| of its controls and a bound D of the whole column -- instead
// of one per control, F as two fused multiply-adds on a table of A[r] alone.  See SdpColFilter.
#define SDP_COL_LEAN -1          // -1: for 8-byte reals; 0 / 1 force it (A/B runs)
#endif
#ifndef SDP_COL_WIDE
// Wide first pass for 4-byte reals: F in 8-byte arithmetic on a table of A[r] accumulated in 8-byte
// reals -- its own error is negligible, so the radius only has to cover the reference's W x 6
// roundings -- and a bound per control that follows those roundings term by term (position-weighted:
// a term added at step w passes through W - w + 4 roundings, not W + 4) with the actual |T[w][r]|
// instead of their maximum.  In 4-byte reals the radius is what decides how many nodes keep a
// second control (5 % at first, more and more as the cost-to-go grows over a chain of sweeps), i.e.
// how often the long way runs twice: see SdpColWide.
#define SDP_COL_WIDE -1          // -1: for 4-byte reals; 0 / 1 force it off / on (A/B runs; needs 4-byte reals)
#endif
#ifndef SDP_COL_UTAB
// K > 0: the generated unit provides sdp_model_utab / sdp_model_lead_tab / sdp_model_cost_tab
// (codegen.control_table_source): the K sub-expressions of x0' and of the cost that depend on the
// control but not on x0 are tabulated once per (column, control) -- at most SDP_COL_UTAB_N controls, the
// same lattice at every node -- and the first pass of the filter reads them from LDS instead of
// recomputing them at every node: the same operations on the same operands, the same bits.
#define SDP_COL_UTAB 0
#endif
#ifndef SDP_COL_UTAB_N
#define SDP_COL_UTAB_N 1
#endif
#ifndef SDP_COL_FILTER_SCALE
#define SDP_COL_FILTER_SCALE 1   // test knob: multiplies the error radius (any value >= 1 gives the same bits)
#endif
#ifndef SDP_COL_SHIFT
// 1: the perturbation reaches x0' through final sums, x0' = a(x, u) +- b_1(x_1.., w) +- b_2 .. (the generated
// unit provides sdp_model_lead_a / sdp_model_lead_b, at most four terms): the first pass of the filter reads
// the table reduced over w on a lattice SHIFTED by the perturbation points -- see sdp_col_shift_reduce.
#define SDP_COL_SHIFT 0
#endif
#ifndef SDP_COL_SHIFT_CHAIN
// m > 0: the sums are a chain in another nesting, x + (w - u), with m <= 3 additions, regrouped by the tracer: a is the
// sum of the chain's w-free leaves (sdp_model_lead_aabs: the sum of their magnitudes) -- see sdp_col_lean_eval
#define SDP_COL_SHIFT_CHAIN 0
#endif
#ifndef SDP_COL_SHIFT_ROWS
#define SDP_COL_SHIFT_ROWS (2 * SDP_COL_N0)     // rows of the shifted lattice held in LDS
#endif
#ifndef SDP_COL_WRES
// Perturbation points the LDS table holds at a time (default: all SDP_COL_W).  Fewer = RESIDENT CHUNK form of
// the filtered kernel (sdp_colres_kernel.h): the W x N0 table is what limits a CU to two workgroups (64 KiB each
// at 256 x 32 x 8 B) -- two waves per SIMD, and the kernel's time halves when a second workgroup joins the first
// (measured: 3.39 -> 1.67 ms), i.e. it is bound by how much there is to overlap, not by a unit.  With C < W
// points resident the table shrinks to C x N0 (32 KiB at C = 16: four workgroups per CU), at the price of
// building the tail W - C points twice per column: once for the reduced table, once more for the second pass
// (whose accumulation runs in w order: head first, then the tail).  Same operations on the same operands in
// the same order: same bits.
#define SDP_COL_WRES SDP_COL_W
#endif
#if SDP_COL_WRES < SDP_COL_W && (!SDP_COL_FILTER || SDP_COL_WPAIR || SDP_COST_HAS_W || \
                                 SDP_COL_ROWS < SDP_COL_N0 || SDP_COL_THREADS < SDP_COL_N0 || 2 * SDP_COL_WRES < SDP_COL_W)
#error "SDP_COL_WRES: lean filtered kernel, plain full-column table, one lane per node, at least half of the points resident"
#endif
#ifndef SDP_COL_LEAN2
#define SDP_COL_LEAN2 0          // generated: x0' = X(x) +- a(u), cost = K(x) +- h(u) -- the short first pass of sdp_colres_kernel.h
#endif
#ifndef SDP_COL_WIDE2
#define SDP_COL_WIDE2 0          // generated, 4-byte reals: the same shape -- the short wide first pass of sdp_col_filter_nodes
#endif
#define SDP_COL_SHORT (SDP_COL_LEAN2 || SDP_COL_WIDE2)
#ifndef SDP_COL_BNB
// 1 (generated with SDP_COL_LEAN2): the short first pass as a certified branch and bound over BLOCKS of controls
// (sdp_lean2_bnb of sdp_colres_kernel.h): a block whose lower bound lies above the smallest F' seen so far by more than
// twice the radius holds no survivor, and its controls are never evaluated.  0: every control (round 4; A/B runs)
#define SDP_COL_BNB 0
#endif
// controls per block of the branch and bound, and the block statistics kept beside the control table (four reals per
// block: smallest a, largest a, smallest +-h psum, unused): at most 64 blocks -- one lane of the table's wave each
#ifndef SDP_LEAN2_A_FIXED
#define SDP_LEAN2_A_FIXED 0      // generated: the part a of x0' = X +- a depends on the control alone (not on the column)
#endif
constexpr bool SDP_LEAN2_A_FIXED_ON = SDP_LEAN2_A_FIXED && SDP_COL_BNB;
constexpr int sdp_bnb_block(int n) { int b = 8; while ((n + b - 1) / b > 64) b *= 2; return b; }
#ifndef SDP_COL_LDS_PAD
#define SDP_COL_LDS_PAD 0        // diagnostic builds: unused bytes in the LDS image (fewer workgroups per CU: occupancy A/B)
#endif
#ifndef SDP_COL_HOIST
// 1: what does not change from unit to unit is fetched ONCE per workgroup instead of once per unit -- the control
// box of a constant-box problem (its loads and the division of numpy.linspace's step sat at the head of every
// first pass and of every control table), the perturbation point of a helper thread, the axis-0 coordinate of a
// thread's node: global-memory round trips behind the co-resident workgroup's table build, each followed by a
// dependent chain, several times per unit.  0: as in round 3 (A/B runs)
#define SDP_COL_HOIST 1
#endif
#if SDP_COL_FILTER && (!SDP_HAS_W || (SDP_LEAD_HAS_W && !SDP_COL_SHIFT) || SDP_TRAIL_HAS_U || SDP_COL_ROWS < SDP_COL_N0)
#error "SDP_COL_FILTER needs a perturbation that reaches x0' through a final sum at most, and the plain full-column table"
#endif
#if SDP_COL_SHIFT && (!SDP_COL_FILTER || !SDP_LEAD_HAS_W || SDP_COL_WPAIR)
#error "SDP_COL_SHIFT is a form of the certified filter for a perturbation that reaches x0'"
#endif
#if SDP_COL_FILTER && SDP_COST_HAS_W && SDP_COL_UTAB
#error "the control table holds sub-expressions without the perturbation: not with a cost that depends on it"
#endif
#if SDP_TRAIL_HAS_U && (SDP_COL_WPAIR || SDP_COL_ROWS < SDP_COL_N0)
#error "control-dependent trailing dynamics: plain full table, exact arithmetic only"
#endif
#if SDP_COL_ROWS < SDP_COL_N0 && SDP_COL_WPAIR
#error "the row window is built for the plain table layout with exact arithmetic"
#endif
constexpr bool SDP_COL_WINDOW = SDP_COL_ROWS < SDP_COL_N0;
constexpr int SDP_DT = SDP_D - 1;
// rows of the LDS table: perturbation points, rounded up to whole pairs for the pair layout
constexpr int SDP_COL_TW = SDP_COL_WPAIR ? (SDP_COL_W + 1) / 2 * 2 : SDP_COL_W;

#ifndef SDP_COL_WMODE
#define SDP_COL_WMODE 1          // where the inner loop takes the perturbation weights from: see SdpColWeights
#endif
struct SdpColShared {
    sdp_real *T;        // [Wn][N0]
    int *w_off;         // [Wn][SDP_DT]   M[k]*q[k] of the trailing cell (x N0)
    sdp_real *w_lam;    // [Wn][SDP_DT]
    sdp_real *w_oml;    // [Wn][SDP_DT]
    sdp_real *part_J;   // [chunks][nodes of the unit], chunks*nodes <= SDP_COL_THREADS
    int *part_i;
    int r0;             // first row of axis 0 held by the table (0 without a row window)
};

// statically sized LDS image (a single workgroup may use up to 160 KiB)
#ifndef SDP_COL_WCHUNK
#define SDP_COL_WCHUNK SDP_COL_W     // per-control table only: perturbation points tabulated at a time
#endif
// (members a build does not use shrink to one element: the image decides how many workgroups share a CU)
constexpr int SDP_COL_LDS_WCOPY = SDP_COL_WMODE == 2 ? SDP_COL_W : 1;
constexpr int SDP_COL_LDS_PART = (SDP_COL_FILTER || SDP_TRAIL_HAS_U) ? 1 : SDP_COL_THREADS;
// reals per row of the reduced table: A[r] alone in the lean form, (A[r], D[r]) otherwise, 16 bytes for 4-byte reals
constexpr int SDP_BNB_BLOCK = sdp_bnb_block(SDP_COL_UTAB_N);
constexpr int SDP_BNB_BLOCKS = (SDP_COL_UTAB_N + SDP_BNB_BLOCK - 1) / SDP_BNB_BLOCK;
constexpr int SDP_BNB_WORDS = SDP_COL_UTAB ? 4 * (SDP_BNB_BLOCKS + 1) : 0;      // (a record of 16 bytes per block and one for the end of the lattice: 4 x 4-byte or 2 x 8-byte reals)
// (the short wide first pass reads A[r] alone: there the 8-byte sums lie side by side in the first half of `ad` -- a lane
// per row then reads conflict-free; at 16 bytes per row the same read hits every bank four times: 39 % of the LDS cycles
// of config 5 were bank conflicts, profiles/r05_synth512f32_summary.txt)
#define SDP_AD_A(ad, r) (SDP_COL_WIDE2 ? ((double *)(ad))[(r)] : *(double *)((ad) + 4 * (r)))
#define SDP_AD_A_CONST(ad, r) (SDP_COL_WIDE2 ? ((const double *)(ad))[(r)] : *(const double *)((ad) + 4 * (r)))
constexpr int SDP_COL_LDS_AD = sizeof(SDP_REAL) == 4 ? 4 : ((SDP_COL_LEAN != 0 && !SDP_COL_SHIFT) ? 1 : 2);
struct __attribute__((aligned(16))) SdpColLds {
    sdp_real T[(SDP_TRAIL_HAS_U ? SDP_COL_WCHUNK : (SDP_COL_WRES < SDP_COL_W ? SDP_COL_WRES : SDP_COL_TW)) * SDP_COL_ROWS];
    sdp_real w_lam[SDP_COL_W * SDP_DT];
    sdp_real w_oml[SDP_COL_W * SDP_DT];
    sdp_real pw[SDP_COL_LDS_WCOPY];        // weight / point copies (SDP_COL_WMODE 2)
    sdp_real gw[SDP_COL_LDS_WCOPY];
    sdp_real part_J[SDP_COL_LDS_PART];     // partial minima of the control chunks (unfiltered sweep)
    int part_i[SDP_COL_LDS_PART];
    int w_off[SDP_COL_W * SDP_DT];
    int win[2][2];                         // row window: per parity (min row, minus max row) of the unit
    int next_unit;                         // filtered kernel: the unit claimed for the next round
    unsigned long long dcol[2];            // lean filter: per parity of the unit, bits of max_r D[r] (>= 0: ordered as integers)
    // per parity of the unit: the tabulated values of every control of the column (SDP_COL_UTAB)
    // (+ 4: statistics of the column's table for the short first pass, SDP_COL_LEAN2 -- sdp_col_phase_u)
    // (+ 4 per block of controls: statistics of the blocks for the branch and bound of the short first pass)
    sdp_real utab[2][SDP_COL_UTAB ? SDP_COL_UTAB * SDP_COL_UTAB_N + 4 + SDP_BNB_WORDS : 2] __attribute__((aligned(16)));
#if SDP_COL_LDS_PAD
    char pad_[SDP_COL_LDS_PAD];
#endif
#if SDP_COL_FILTER
    // filter: per row r of axis 0 the pair (A[r], D[r]) = (sum_w p_w T[w][r], Pcap max_w |T[w][r]|)
    // (wide first pass of 4-byte reals: 16 bytes per row -- A[r] as a double, then the bound B[r])
    sdp_real ad[SDP_COL_LDS_AD * (SDP_COL_SHIFT ? SDP_COL_SHIFT_ROWS : SDP_COL_ROWS)] __attribute__((aligned(16)));
#endif
#if SDP_COL_SHIFT
    // per parity of the unit: the shift of every perturbation point in rows of axis 0 -- whole part, fraction
    // in [0, 1), |p_w| f (1 - f) -- and (largest whole part, minus the smallest, "not usable")
    int sh_q[2][SDP_COL_W];
    sdp_real sh_f[2][SDP_COL_W];
    sdp_real sh_c[2][SDP_COL_W];
    int sh_k[2][4];
#endif
};
static_assert(sizeof(SdpColLds) <= 160 * 1024, "column table exceeds the 160 KiB LDS of a CU");

// diagnostic builds only (SDP_STAMP 2 / 3): clocks of the two passes, survivor counts
struct SdpColDiag {
    unsigned long long m1 = 0, m2 = 0, tp1 = 0, tp2 = 0, n_slow = 0, n_exact = 0, n_all = 0;
};
#if SDP_STAMP == 2
#define SDP_COL_MARK(v) v = __builtin_amdgcn_s_memtime()
#else
#define SDP_COL_MARK(v)
#endif

SDP_DEV void sdp_col_carve(SdpColLds &m, SdpColShared &s)
{
    s.T = m.T;
    s.w_lam = m.w_lam;
    s.w_oml = m.w_oml;
    s.w_off = m.w_off;
    s.part_J = m.part_J;
    s.part_i = m.part_i;
    s.r0 = 0;
}

// grid of the trailing axes over the axis-0-fastest array: strides in elements
SDP_DEV double sdp_uniform_real(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
SDP_DEV float sdp_uniform_real(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
SDP_DEV void sdp_col_trailing_grid(const SdpSweepArgs &a, SdpGrid<sdp_real, SDP_DT> &g)
{
    const sdp_real *axes = (const sdp_real *)a.axes;
    int m = SDP_COL_N0;
    g.pow2 = 0;
#pragma unroll
    for (int k = SDP_DT - 1; k >= 0; --k) {
        const int ax = k + 1;
        g.smin[k] = axes[a.axis_off[ax]];
        g.span[k] = axes[a.axis_off[ax] + a.orders[ax] - 1] - g.smin[k];
        g.rspan[k] = (sdp_real)1 / g.span[k];
        if (sdp_is_pow2(g.span[k])) g.pow2 |= 1 << k;
        g.nm1[k] = (sdp_real)(a.orders[ax] - 1);
        g.ordm2[k] = a.orders[ax] - 2;
        g.M[k] = m;
        m *= a.orders[ax];
    }
    g.pow2 = __builtin_amdgcn_readfirstlane(g.pow2);
    g.shift = (sdp_real)0;
    // (the same in every lane for the whole kernel: scalar registers -- sdp_uniform, defined below, is a template here)
#pragma unroll
    for (int k = 0; k < SDP_DT; ++k) {
        g.smin[k] = sdp_uniform_real(g.smin[k]); g.span[k] = sdp_uniform_real(g.span[k]);
        g.rspan[k] = sdp_uniform_real(g.rspan[k]); g.nm1[k] = sdp_uniform_real(g.nm1[k]);
    }
}

// The nested lerp of SdpLerp<real, SDP_DT, real> (sdp_device.h; reference
// multilinear_cython.pyx:88-300, last axis innermost) split in two: the vertex
// loads, in the nest's order, and the arithmetic on the loaded values -- the
// same operations on the same operands, so the result is bit-identical.
template <int K>
struct SdpColGather {
    static SDP_DEV void run(const sdp_real *__restrict__ V, const SdpGrid<sdp_real, SDP_DT> &g,
                            const int *off, int base, sdp_real *vals)
    {
        SdpColGather<K + 1>::run(V, g, off, base + off[K], vals);
        SdpColGather<K + 1>::run(V, g, off, base + off[K] + g.M[K], vals + (1 << (SDP_DT - K - 1)));
    }
};
template <>
struct SdpColGather<SDP_DT> {
    static SDP_DEV void run(const sdp_real *__restrict__ V, const SdpGrid<sdp_real, SDP_DT> &,
                            const int *, int base, sdp_real *vals)
    {
        vals[0] = V[base];
    }
};
template <int K, bool SHIFT>
struct SdpColNest {
    static SDP_DEV sdp_real run(const sdp_real *vals, const sdp_real *lam, const sdp_real *oml,
                                sdp_real shift)
    {
        const sdp_real lo = SdpColNest<K + 1, SHIFT>::run(vals, lam, oml, shift);
        const sdp_real hi = SdpColNest<K + 1, SHIFT>::run(vals + (1 << (SDP_DT - K - 1)), lam, oml, shift);
        return oml[K] * lo + lam[K] * hi;
    }
};
template <bool SHIFT>
struct SdpColNest<SDP_DT - 1, SHIFT> {
    static SDP_DEV sdp_real run(const sdp_real *vals, const sdp_real *lam, const sdp_real *oml,
                                sdp_real shift)
    {
        sdp_real lo = vals[0], hi = vals[1];
        if (SHIFT) { lo = lo - shift; hi = hi - shift; }
        return oml[SDP_DT - 1] * lo + lam[SDP_DT - 1] * hi;
    }
};

// phase W for column `c`: trailing cell of every perturbation point -> s.w_*
// (`first`: the threads from `first` on do it, the others pass)
// (`w_mine`: the perturbation point of this thread's first item, fetched by the caller -- or null)
SDP_DEV void sdp_col_phase_w(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                             const SdpColShared &s, const sdp_real *x, const sdp_real *u, sdp_real t,
                             int first = 0, int count = 0, const sdp_real *w_mine = nullptr)
{
    // (threads first .. first + count - 1 do the work; count = 0: all threads from `first` on)
    constexpr int Wn = SDP_COL_W;
#if SDP_HAS_W
    const sdp_real *__restrict__ wgrid = (const sdp_real *)a.wgrid;
#endif
    if (count == 0) count = (int)blockDim.x - first;
    if ((int)threadIdx.x < first || (int)threadIdx.x >= first + count) return;
    for (int w = (int)threadIdx.x - first; w < Wn; w += count) {
        sdp_real xn[SDP_D];
#if SDP_HAS_W
        sdp_model_trail(x, u, (w_mine && w == (int)threadIdx.x - first) ? *w_mine : wgrid[w], t, xn);
#else
        sdp_model_trail(x, u, (sdp_real)0, t, xn);
#endif
        SdpCell<sdp_real, SDP_DT, sdp_real> c;
#pragma unroll
        for (int k = 0; k < SDP_DT; ++k) {
            sdp_locate_axis<sdp_real, SDP_DT, sdp_real>(tg, k, xn[k + 1], c);
#ifdef SDP_DIAG_SAME_CELL      // timing diagnostic (WRONG results): every perturbation point reads the strips of one cell
            s.w_off[w * SDP_DT + k] = 0;
#else
            s.w_off[w * SDP_DT + k] = c.off[k];
#endif
            s.w_lam[w * SDP_DT + k] = c.lam[k];
            s.w_oml[w * SDP_DT + k] = c.oml[k];
        }
    }
}

// phase A: T[w][r - s.r0] = lerp over the trailing axes of V[r, .] for the
// SDP_COL_ROWS rows from s.r0 on.  The Wn*rows entries are dealt to all threads
// (consecutive threads = consecutive rows r: coalesced strip reads); each thread
// handles SDP_COL_A_GROUP entries at a time.  The 2^(d-1) vertex loads of the
// whole group are issued first -- only the loaded values live in registers
// meanwhile; the interpolation weights are re-read from LDS afterwards -- so one
// memory round trip serves SDP_COL_A_GROUP entries.
extern "C" __device__ double __ockl_wfred_max_f64(double);
extern "C" __device__ double __ockl_wfred_min_f64(double);
extern "C" __device__ double __ockl_wfred_add_f64(double);
extern "C" __device__ float __ockl_wfred_max_f32(float);
extern "C" __device__ float __ockl_wfred_min_f32(float);
extern "C" __device__ float __ockl_wfred_add_f32(float);
SDP_DEV double sdp_wave_max(double v) { return __ockl_wfred_max_f64(v); }      // DPP row operations, no LDS traffic
// a value that is the same in every lane, moved to scalar registers (what the compiler will not conclude by itself for
// numbers that came out of vector arithmetic or vector loads: kept in vector registers they crowd out the passes'
// working set -- and what does not fit there goes to scratch memory, a memory round trip per use)
SDP_DEV double sdp_uniform(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
SDP_DEV float sdp_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
SDP_DEV int sdp_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the value of another lane of the row of 16 (DPP control CTRL: 0xB1 the neighbour, 0x4E the pair's partner, 0x141 the
// mirrored lane of the half row): no LDS traffic
template <int CTRL>
SDP_DEV double sdp_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
SDP_DEV float sdp_dpp_f64(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
SDP_DEV float sdp_wave_max(float v) { return __ockl_wfred_max_f32(v); }
SDP_DEV double sdp_wave_min(double v) { return __ockl_wfred_min_f64(v); }
SDP_DEV float sdp_wave_min(float v) { return __ockl_wfred_min_f32(v); }
SDP_DEV double sdp_wave_sum(double v) { return __ockl_wfred_add_f64(v); }
SDP_DEV float sdp_wave_sum(float v) { return __ockl_wfred_add_f32(v); }

// (w_begin, w_count: the perturbation points to tabulate, into table rows 0 .. w_count-1 -- all of them by
// default; the resident-chunk kernel builds the table a part at a time; `keep`: a copy of the entries in global
// memory, same layout -- SDP_COL_TAIL_KEEP of sdp_colres_kernel.h, 16-byte build loads only)
template <bool SHIFT = false>
SDP_DEV void sdp_col_phase_a(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                             const SdpColShared &s, const int w_begin = 0, const int w_count = SDP_COL_W,
                             sdp_real *__restrict__ keep = nullptr)
{
    constexpr int N0 = SDP_COL_ROWS;        // rows held by the table (the whole axis without a window)
    const int Wn = w_begin + w_count;       // one past the last point
    const sdp_real *__restrict__ V = (const sdp_real *)a.V + s.r0;
    constexpr int G = SDP_COL_A_GROUP;
    constexpr int NV = 1 << SDP_DT;
#if SDP_COL_A_ORDER == 2
    // a thread keeps its perturbation point w and takes rows r, r + LW, r + 2 LW, ..: the
    // trailing cell (offsets and weights) is read from LDS ONCE per thread instead of once
    // per entry, and the addresses of consecutive entries differ by a constant (immediate
    // offsets of the loads and of the table stores: no address arithmetic per entry).  LW
    // consecutive lanes read LW consecutive rows (LW x sizeof(real) contiguous bytes).
    constexpr int LW = SDP_COL_A_LW;                     // lanes (= consecutive rows) per perturbation point
    const int WPASS = blockDim.x / LW;                   // perturbation points handled side by side (the
    const int rl = threadIdx.x % LW;                     //  policy-evaluation launch has fewer threads)
    for (int w = w_begin + threadIdx.x / LW; w < Wn; w += WPASS) {
        const int tw = w - w_begin;                      // row of the table
        int off[SDP_DT];
        sdp_real lam[SDP_DT], oml[SDP_DT];
#pragma unroll
        for (int k = 0; k < SDP_DT; ++k) {
            off[k] = s.w_off[w * SDP_DT + k];
            lam[k] = s.w_lam[w * SDP_DT + k];
            oml[k] = s.w_oml[w * SDP_DT + k];
        }
#if SDP_COL_A_WIDE_LOADS
        // 16-byte vertex loads: a lane takes RPL = 16 / sizeof(real) ADJACENT rows (RPL rl .. RPL rl + RPL - 1),
        // then the rows RPL LW further on, ..: 1 / RPL of the vector-memory instructions for the same bytes
        // (the build is bound by the rate of its load instructions through the texture path, not only by
        // bytes: 256^3 x 64 x 32 fp64, same box: 1.86 -> 1.74 ms, with 32 lanes per perturbation point
        // 1.65 ms), and with the plain layout the RPL entries go out as one 16-byte LDS store.  Needs N0 to be
        // a multiple of RPL and no row window (the strips then start on 16-byte boundaries).
        {
            constexpr int RPL = 16 / (int)sizeof(sdp_real);
            static_assert(N0 % RPL == 0 && SDP_COL_ROWS == SDP_COL_N0, "wide loads: whole groups of rows, no row window");
            typedef sdp_real sdp_rows __attribute__((ext_vector_type(RPL)));
            constexpr int GP = G / RPL > 0 ? G / RPL : 1;        // row groups per round
            int voff[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                int o = 0;
#pragma unroll
                for (int k = 0; k < SDP_DT; ++k) o += off[k] + (((q >> (SDP_DT - 1 - k)) & 1) ? tg.M[k] : 0);
                voff[q] = o;
            }
            for (int j0 = 0; j0 * RPL * LW < N0; j0 += GP) {
                sdp_rows vals2[GP][NV];
#pragma unroll
                for (int j = 0; j < GP; ++j) {
                    const int r = min(((j0 + j) * LW + rl) * RPL, N0 - RPL);  // clamp: result unused
#pragma unroll
                    for (int q = 0; q < NV; ++q) vals2[j][q] = *(const sdp_rows *)(V + r + voff[q]);
                }
#pragma unroll
                for (int j = 0; j < GP; ++j) {
                    const int r = ((j0 + j) * LW + rl) * RPL;
                    if (r < N0) {
                        sdp_rows e;
#pragma unroll
                        for (int c = 0; c < RPL; ++c) {
                            sdp_real one[NV];
#pragma unroll
                            for (int q = 0; q < NV; ++q) one[q] = vals2[j][q][c];
                            e[c] = SdpColNest<0, SHIFT>::run(one, lam, oml, tg.shift);
                        }
#if SDP_COL_WPAIR
#pragma unroll
                        for (int c = 0; c < RPL; ++c) s.T[((tw >> 1) * N0 + r + c) * 2 + (tw & 1)] = e[c];
#else
                        *(sdp_rows *)(s.T + tw * N0 + r) = e;
                        if (keep) *(sdp_rows *)(keep + tw * N0 + r) = e;
#endif
                    }
                }
            }
            continue;
        }
#endif
        for (int j0 = 0; j0 * LW < N0; j0 += G) {
            sdp_real vals[G][NV];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int r = min((j0 + j) * LW + rl, N0 - 1);           // clamp: result unused
#ifdef SDP_DIAG_NO_A_LOADS
                for (int q = 0; q < NV; ++q) vals[j][q] = (sdp_real)(r + q);
#else
                SdpColGather<0>::run(V + r, tg, off, 0, vals[j]);
#endif
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int r = (j0 + j) * LW + rl;
                if (r < N0) {
                    const sdp_real val = SdpColNest<0, SHIFT>::run(vals[j], lam, oml, tg.shift);
                    const sdp_real entry = val;
#if SDP_COL_WPAIR
                    s.T[((tw >> 1) * N0 + r) * 2 + (tw & 1)] = entry;
#else
                    s.T[tw * N0 + r] = entry;
#endif
                }
            }
        }
    }
#elif SDP_COL_A_ORDER == 1
    // a thread keeps its row r and takes G CONSECUTIVE perturbation points per round:
    // neighbouring points share part of their 2^(d-1) vertex strips (the cell of an
    // exogenous process moves by about one grid step per point), so the second read of
    // a strip comes from the CU's L1 instead of L2
    constexpr int LANES_R = SDP_COL_THREADS < N0 ? SDP_COL_THREADS : N0;   // threads along the rows
    constexpr int GROUPS = SDP_COL_THREADS / LANES_R;                        // thread groups along w
    const int W_PER = (w_count + GROUPS - 1) / GROUPS;
    const int grp = threadIdx.x / LANES_R;
    const int w_lo = w_begin + grp * W_PER, w_hi = min(Wn, w_lo + W_PER);
    for (int r = threadIdx.x - grp * LANES_R; r < N0 && grp < GROUPS; r += LANES_R) {
        for (int w0 = w_lo; w0 < w_hi; w0 += G) {
            sdp_real vals[G][NV];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int w = min(w0 + j, w_hi - 1);                 // clamp: result unused
                int off[SDP_DT];
#pragma unroll
                for (int k = 0; k < SDP_DT; ++k) off[k] = s.w_off[w * SDP_DT + k];
                SdpColGather<0>::run(V + r, tg, off, 0, vals[j]);
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int w = w0 + j;
                if (w < w_hi) {
                    sdp_real lam[SDP_DT], oml[SDP_DT];
#pragma unroll
                    for (int k = 0; k < SDP_DT; ++k) {
                        lam[k] = s.w_lam[w * SDP_DT + k];
                        oml[k] = s.w_oml[w * SDP_DT + k];
                    }
                    const sdp_real val = SdpColNest<0, SHIFT>::run(vals[j], lam, oml, tg.shift);
                    const sdp_real entry = val;
                    const int tw = w - w_begin;
#if SDP_COL_WPAIR
                    s.T[((tw >> 1) * N0 + r) * 2 + (tw & 1)] = entry;
#else
                    s.T[tw * N0 + r] = entry;
#endif
                }
            }
        }
    }
#else
    const int total = w_count * N0;
    for (int item0 = threadIdx.x; item0 < total; item0 += G * blockDim.x) {
        sdp_real vals[G][NV];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int item = min(item0 + j * (int)blockDim.x, total - 1);   // clamp: result unused
            const int tw = item / N0;
            const int r = item - tw * N0;
            const int w = w_begin + tw;
            int off[SDP_DT];
#pragma unroll
            for (int k = 0; k < SDP_DT; ++k) off[k] = s.w_off[w * SDP_DT + k];
            SdpColGather<0>::run(V + r, tg, off, 0, vals[j]);
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int item = item0 + j * (int)blockDim.x;
            if (item < total) {
                const int tw = item / N0;
                const int r = item - tw * N0;
                const int w = w_begin + tw;
                sdp_real lam[SDP_DT], oml[SDP_DT];
#pragma unroll
                for (int k = 0; k < SDP_DT; ++k) {
                    lam[k] = s.w_lam[w * SDP_DT + k];
                    oml[k] = s.w_oml[w * SDP_DT + k];
                }
                const sdp_real val = SdpColNest<0, SHIFT>::run(vals[j], lam, oml, tg.shift);
                const sdp_real entry = val;
#if SDP_COL_WPAIR
                s.T[((tw >> 1) * N0 + r) * 2 + (tw & 1)] = entry;
#else
                s.T[tw * N0 + r] = entry;
#endif
            }
        }
    }
#endif
}

// inner(r) = lerp over the trailing axes of V[r, .] at the cell of perturbation
// point w, straight from global memory: what phase A tabulates, for a row the
// table does not hold (row window only).  Same loads, same operations.
template <bool SHIFT>
SDP_DEV sdp_real sdp_col_inner_global(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                                      const SdpColShared &s, int w, int r)
{
    sdp_real vals[1 << SDP_DT], lam[SDP_DT], oml[SDP_DT];
    int off[SDP_DT];
#pragma unroll
    for (int k = 0; k < SDP_DT; ++k) {
        off[k] = s.w_off[w * SDP_DT + k];
        lam[k] = s.w_lam[w * SDP_DT + k];
        oml[k] = s.w_oml[w * SDP_DT + k];
    }
    SdpColGather<0>::run((const sdp_real *)a.V + r, tg, off, 0, vals);
    return SdpColNest<0, SHIFT>::run(vals, lam, oml, tg.shift);
}

struct SdpLeadAxis {
    sdp_real smin, span, nm1, rspan;
    sdp_real koff;      // shifted lattice (SDP_COL_SHIFT, first pass only): position of its first row
    int ordm2;
    bool pow2;          // span is a power of two: (x - smin) / span == (x - smin) * rspan, bit for bit
};

SDP_DEV void sdp_col_lead_axis(const SdpSweepArgs &a, SdpLeadAxis &l)
{
    const sdp_real *axes = (const sdp_real *)a.axes;
    l.smin = axes[a.axis_off[0]];
    l.span = axes[a.axis_off[0] + a.orders[0] - 1] - l.smin;
    l.rspan = (sdp_real)1 / l.span;
    l.pow2 = __builtin_amdgcn_readfirstlane((int)sdp_is_pow2(l.span)) != 0;
    l.nm1 = (sdp_real)(a.orders[0] - 1);
    l.ordm2 = a.orders[0] - 2;
    l.koff = (sdp_real)0;
}

// Where the inner loop takes the perturbation weights p_w (and points w, when
// the cost depends on w) from -- SDP_COL_WMODE:
//   0  VGPRs: every lane holds the same W values for the whole kernel (the w
//      loop is fully unrolled, so the arrays are never indexed dynamically);
//      2*W registers, no memory instruction besides the two LDS reads per cell
//   1  scalar loads (s_load_*) from the constant address space into SGPRs,
//      used directly as VALU operands: no VGPR cost
//   2  LDS broadcast reads of a copy made at kernel start
// Default: scalar loads (A/B on MI355X: as fast as VGPRs, and no register cost).

struct SdpColWeights {
#if SDP_COL_WMODE == 0 && SDP_HAS_W
    sdp_real p[SDP_COL_W];
#if SDP_COST_HAS_W || SDP_LEAD_HAS_W
    sdp_real w[SDP_COL_W];
#endif
#endif
    const volatile sdp_cst_real *cp, *cw;  // mode 1 (volatile: stay inside the loop)
    const volatile sdp_lds_real *lp, *lw;  // mode 2
};

#if SDP_COL_WMODE == 0
#define SDP_COL_PW(k, w) ((k).p[w])
#define SDP_COL_GW(k, w) ((k).w[w])
#elif SDP_COL_WMODE == 1
#define SDP_COL_PW(k, w) ((k).cp[w])
#define SDP_COL_GW(k, w) ((k).cw[w])
#else
#define SDP_COL_PW(k, w) ((k).lp[w])
#define SDP_COL_GW(k, w) ((k).lw[w])
#endif

SDP_DEV void sdp_col_load_weights(const SdpSweepArgs &a, SdpColWeights &k, sdp_real *lds_p,
                                  sdp_real *lds_w)
{
    const sdp_real *gp = (const sdp_real *)a.proba;
    const sdp_real *gw = (const sdp_real *)a.wgrid;
    k.cp = (const volatile sdp_cst_real *)a.proba;
    k.cw = (const volatile sdp_cst_real *)a.wgrid;
    k.lp = (const volatile sdp_lds_real *)lds_p;
    k.lw = (const volatile sdp_lds_real *)lds_w;
    (void)gp; (void)gw;
#if SDP_HAS_W
#if SDP_COL_WMODE == 0
#pragma unroll
    for (int w = 0; w < SDP_COL_W; ++w) {
        // the empty asm pins each (wave-uniform) value in a VGPR: left to
        // itself the compiler keeps them in SGPRs, runs out and spills
        sdp_real v = gp[w];
        asm volatile("" : "+v"(v));
        k.p[w] = v;
#if SDP_COST_HAS_W || SDP_LEAD_HAS_W
        v = gw[w];
        asm volatile("" : "+v"(v));
        k.w[w] = v;
#endif
    }
#elif SDP_COL_WMODE == 2
    for (int w = threadIdx.x; w < SDP_COL_W; w += blockDim.x) {
        lds_p[w] = gp[w];
        lds_w[w] = gw[w];
    }
    __syncthreads();
#endif
#endif
}

#if SDP_COL_WPAIR
// Pair layout (4-byte reals): perturbation points 2k, 2k+1 of one control are
// processed together.  Each lane of the packed operations below is the
// reference's operation on the same operands (v_pk_mul_f32 / v_pk_add_f32 round
// each half like the scalar instruction), and the expectation is accumulated
// in w order (first .x, then .y), so the result is bit-identical to the plain
// layout.
typedef sdp_real sdp_v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) sdp_v2 sdp_lds_v2;

template <int K, bool SHIFT = false>
SDP_DEV void sdp_col_expected_cost(const SdpSweepArgs &, const SdpGrid<sdp_real, SDP_DT> &,
                                   const SdpColShared &s, const SdpColWeights &k, const SdpLeadAxis &l,
                                   const sdp_real *x,
                                   const sdp_real (*u)[SDP_NU], sdp_real t, sdp_real *out)
{
    const sdp_real *T = s.T;
    constexpr int N0 = SDP_COL_N0;
    constexpr int Wn = SDP_COL_W;
    constexpr int WP = Wn / 2;                      // whole pairs; an odd last point is the tail
    sdp_real lam0[K], oml0[K], acc[K], g[K];
    const volatile sdp_lds_v2 *row[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const sdp_real xn0 = sdp_model_lead(x, u[j], (sdp_real)0, t);
        const sdp_real sn = sdp_div_span<sdp_real>(xn0 - l.smin, l.span, l.rspan, l.pow2);   /* pyx:75 */
        const sdp_real p = sn * l.nm1;
        const int q0 = max(min(sdp_trunc_i32(p), l.ordm2), 0);          /* pyx:78 */
        lam0[j] = p - (sdp_real)q0;                                     /* pyx:81 */
        oml0[j] = (sdp_real)1 - lam0[j];
        row[j] = (const volatile sdp_lds_v2 *)T + q0;
        acc[j] = (sdp_real)0;
#if !SDP_COST_HAS_W
        g[j] = sdp_model_cost(x, u[j], (sdp_real)0, t);
#endif
    }
    constexpr int B = SDP_COL_BATCH;
#pragma unroll SDP_COL_UNROLL_W
    for (int p0 = 0; p0 < WP; p0 += B) {
        sdp_v2 lo[B][K], hi[B][K];
#pragma unroll
        for (int b = 0; b < B; ++b)
            if (p0 + b < WP) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    lo[b][j] = row[j][(p0 + b) * N0];
                    hi[b][j] = row[j][(p0 + b) * N0 + 1];
                }
            }
#pragma unroll
        for (int b = 0; b < B; ++b)
            if (p0 + b < WP) {
                const int w = 2 * (p0 + b);
                sdp_v2 pw2;
                pw2.x = SDP_COL_PW(k, w);
                pw2.y = SDP_COL_PW(k, w + 1);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const sdp_v2 val = (sdp_v2)(oml0[j]) * lo[b][j] + (sdp_v2)(lam0[j]) * hi[b][j];  // pyx:88-300
                    sdp_v2 g2;
#if SDP_COST_HAS_W
                    g2.x = sdp_model_cost(x, u[j], SDP_COL_GW(k, w), t);
                    g2.y = sdp_model_cost(x, u[j], SDP_COL_GW(k, w + 1), t);
#else
                    g2 = (sdp_v2)(g[j]);
#endif
                    const sdp_v2 jc = g2 + val;                       // stodynprog.py:677
                    const sdp_v2 tt = jc * pw2;
                    acc[j] = acc[j] + tt.x;                           // stodynprog.py:681, w order
                    acc[j] = acc[j] + tt.y;
                }
            }
    }
    if (Wn & 1) {                                   // odd W: the last point sits alone in its pair
        constexpr int w = Wn - 1;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const sdp_real lo = row[j][WP * N0].x;
            const sdp_real hi = row[j][WP * N0 + 1].x;
            const sdp_real val = oml0[j] * lo + lam0[j] * hi;
#if SDP_COST_HAS_W
            g[j] = sdp_model_cost(x, u[j], SDP_COL_GW(k, w), t);
#endif
            const sdp_real jc = g[j] + val;
            acc[j] = acc[j] + jc * SDP_COL_PW(k, w);
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        out[j] = acc[j];
    }
}
#else   // plain layout
// Row window only: expected cost of ONE control whose rows of axis 0 are not in
// the table -- the operations of sdp_col_expected_cost on values interpolated
// straight from global memory (sdp_col_inner_global), so the same bits.
template <bool SHIFT>
SDP_DEV sdp_real sdp_col_cost_global(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                                     const SdpColShared &s, const SdpColWeights &k,
                                     const SdpLeadAxis &l, const sdp_real *x, const sdp_real *u,
                                     sdp_real t)
{
    sdp_real lam0, oml0, g;
    int q0;
#define SDP_COL_LOCATE1(wval)                                                           \
    {                                                                                  \
        const sdp_real xn0_ = sdp_model_lead(x, u, (wval), t);                         \
        const sdp_real sn_ = sdp_div_span<sdp_real>(xn0_ - l.smin, l.span, l.rspan, l.pow2); /* pyx:75 */   \
        const sdp_real p_ = sn_ * l.nm1;                                               \
        q0 = max(min(sdp_trunc_i32(p_), l.ordm2), 0);                   /* pyx:78 */   \
        lam0 = p_ - (sdp_real)q0;                                       /* pyx:81 */   \
        oml0 = (sdp_real)1 - lam0;                                                     \
    }
#if !SDP_LEAD_HAS_W || !SDP_HAS_W
    SDP_COL_LOCATE1((sdp_real)0)
#endif
#if !SDP_COST_HAS_W || !SDP_HAS_W
    g = sdp_model_cost(x, u, (sdp_real)0, t);
#endif
#if SDP_HAS_W
    sdp_real acc = (sdp_real)0;
    for (int w = 0; w < SDP_COL_W; ++w) {
#if SDP_LEAD_HAS_W
        SDP_COL_LOCATE1(SDP_COL_GW(k, w))
#endif
        const sdp_real lo = sdp_col_inner_global<SHIFT>(a, tg, s, w, q0);
        const sdp_real hi = sdp_col_inner_global<SHIFT>(a, tg, s, w, q0 + 1);
        const sdp_real val = oml0 * lo + lam0 * hi;                   // pyx:88-300
#if SDP_COST_HAS_W
        g = sdp_model_cost(x, u, SDP_COL_GW(k, w), t);
#endif
        const sdp_real jc = g + val;                                  // stodynprog.py:677
        acc = acc + jc * SDP_COL_PW(k, w);                            // stodynprog.py:681
    }
    return acc;
#else
    const sdp_real lo = sdp_col_inner_global<SHIFT>(a, tg, s, 0, q0);
    const sdp_real hi = sdp_col_inner_global<SHIFT>(a, tg, s, 0, q0 + 1);
    return g + (oml0 * lo + lam0 * hi);
#endif
#undef SDP_COL_LOCATE1
}

// Expected cost of K controls of one node out of the table.  The K cost
// chains are independent, so interleaving them gives the in-order wave K times
// the instruction-level parallelism per LDS round trip (and one scalar load of
// p_w serves K cells); each chain is evaluated exactly as for K = 1.
// With a row window (SDP_COL_ROWS < SDP_COL_N0) a control whose rows q0, q0+1
// are not both in the table reads row 0 instead (in bounds, value unused) and
// is recomputed from global memory at the end.
template <int K, bool SHIFT = false>
SDP_DEV void sdp_col_expected_cost(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                                   const SdpColShared &s, const SdpColWeights &k, const SdpLeadAxis &l,
                                   const sdp_real *x,
                                   const sdp_real (*u)[SDP_NU], sdp_real t, sdp_real *out)
{
    const sdp_real *T = s.T;
    constexpr int N0 = SDP_COL_ROWS;        // row stride of the table
    constexpr int Wn = SDP_COL_W;
    sdp_real lam0[K], oml0[K], acc[K], g[K];
    bool outside[K];
    // two separate 8-byte LDS reads per cell (rows q0 and q0+1 of T[w]): as
    // ds_read_b64 they cost 2 LDS cycles each, conflict-free (bank = row);
    // `volatile` keeps the compiler from fusing them into ds_read2_b64, which
    // runs at half the LDS rate (MI355X_MICROARCH.md, LDS table)
    const volatile sdp_lds_real *row[K];
#define SDP_COL_LOCATE(j, wval)                                                         \
    {                                                                                  \
        const sdp_real xn0_ = sdp_model_lead(x, u[j], (wval), t);                      \
        const sdp_real sn_ = sdp_div_span<sdp_real>(xn0_ - l.smin, l.span, l.rspan, l.pow2); /* pyx:75 */   \
        const sdp_real p_ = sn_ * l.nm1;                                               \
        const int q0_ = max(min(sdp_trunc_i32(p_), l.ordm2), 0);        /* pyx:78 */   \
        lam0[j] = p_ - (sdp_real)q0_;                                   /* pyx:81 */   \
        oml0[j] = (sdp_real)1 - lam0[j];                                               \
        int rel_ = q0_;                                                                \
        if (SDP_COL_WINDOW) {                                                          \
            rel_ = q0_ - s.r0;                                                         \
            const bool in_ = (unsigned)rel_ < (unsigned)(SDP_COL_ROWS - 1);            \
            outside[j] = outside[j] || !in_;                                           \
            rel_ = in_ ? rel_ : 0;                                                     \
        }                                                                              \
        row[j] = (const volatile sdp_lds_real *)(T + rel_);                            \
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        outside[j] = false;
#if !SDP_LEAD_HAS_W || !SDP_HAS_W
        SDP_COL_LOCATE(j, (sdp_real)0)
#endif
        acc[j] = (sdp_real)0;
#if !SDP_COST_HAS_W || !SDP_HAS_W
        g[j] = sdp_model_cost(x, u[j], (sdp_real)0, t);
#endif
    }
#if SDP_HAS_W
    // The LDS reads are issued in batches of SDP_COL_BATCH perturbation points
    // (volatile keeps their order); the expectation is accumulated strictly in
    // w order.
    constexpr int B = SDP_COL_BATCH;
    // a partially unrolled loop bounds the region the instruction scheduler sees
    // (fully unrolled it tends to hoist every read and spill); mode-0 weights
    // live in registers and need static indices, hence the full unroll there
#if SDP_COL_WMODE == 0
#pragma unroll
#else
#pragma unroll SDP_COL_UNROLL_W
#endif
    for (int w0 = 0; w0 < Wn; w0 += B) {
        sdp_real lo[B][K], hi[B][K];
#if SDP_LEAD_HAS_W
        sdp_real lam_b[B][K], oml_b[B][K];
#endif
#pragma unroll
        for (int b = 0; b < B; ++b) {
            if (w0 + b < Wn) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
#if SDP_LEAD_HAS_W
                    // x0' depends on w: its cell is located per lattice cell; the
                    // weights of batch entry b are kept for the compute step
                    SDP_COL_LOCATE(j, SDP_COL_GW(k, w0 + b))
                    lam_b[b][j] = lam0[j];
                    oml_b[b][j] = oml0[j];
#endif
                    lo[b][j] = row[j][(w0 + b) * N0];
                    hi[b][j] = row[j][(w0 + b) * N0 + 1];
                }
            }
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int w = w0 + b;
            if (w < Wn) {
                const sdp_real pw = SDP_COL_PW(k, w);
#if SDP_COST_HAS_W
                const sdp_real gw = SDP_COL_GW(k, w);
#endif
#pragma unroll
                for (int j = 0; j < K; ++j) {
#if SDP_LEAD_HAS_W
                    const sdp_real val = oml_b[b][j] * lo[b][j] + lam_b[b][j] * hi[b][j];
#else
                    const sdp_real val = oml0[j] * lo[b][j] + lam0[j] * hi[b][j];   // pyx:88-300
#endif
#if SDP_COST_HAS_W
                    g[j] = sdp_model_cost(x, u[j], gw, t);
#endif
                    const sdp_real jc = g[j] + val;                   // stodynprog.py:677
                    acc[j] = acc[j] + jc * pw;                        // stodynprog.py:681
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) out[j] = acc[j];
#else
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const sdp_real lo = row[j][0];
        const sdp_real hi = row[j][1];
        out[j] = g[j] + (oml0[j] * lo + lam0[j] * hi);
    }
#endif
#undef SDP_COL_LOCATE
    if (SDP_COL_WINDOW) {
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (outside[j]) out[j] = sdp_col_cost_global<SHIFT>(a, tg, s, k, l, x, u[j], t);
    }
}
#endif  // SDP_COL_WPAIR

SDP_DEV void sdp_col_store(const SdpSweepArgs &a, int64_t node, const SdpBox &box,
                           sdp_real best, int ibest)
{
    sdp_store_J<sdp_real>(a, node, node / SDP_COL_N0, best);        // (N0 is a compile-time constant)
    if (a.idx) a.idx[node] = ibest;
    if (a.pol) {
        sdp_real u[SDP_NU];
        sdp_controls_at(box, ibest, u);
#pragma unroll
        for (int c = 0; c < SDP_NU; ++c) ((sdp_real *)a.pol)[node * SDP_NU + c] = u[c];
    }
}

// unit = (column, split); XCD-aware walk like sdp_sweep
struct SdpColWalk {
    int64_t unit, end, stride;
};
SDP_DEV void sdp_col_walk(const SdpSweepArgs &a, SdpColWalk &w)
{
    const int64_t n_units = (a.col_end - a.col_begin) * a.col_splits;
    const int xcd = blockIdx.x & 7;
    const int64_t per_xcd = (n_units + 7) / 8;
    w.unit = (int64_t)xcd * per_xcd + (blockIdx.x >> 3);
    w.end = min((int64_t)(xcd + 1) * per_xcd, n_units);
    w.stride = gridDim.x >> 3;
}

// Column of a unit.  The filtered kernel hands its units out IN ORDER (SdpSweepArgs.claim), so the
// workgroups resident on an XCD are always at ~64 consecutive units; with two trailing axes
// those are taken as an 8 x 8 tile of (x1, x2) instead of 64 columns along x2: the cells of
// 8 x 8 neighbouring columns overlap almost entirely, and the strips their tables are built
// from (a ~40 x 25 patch of 2 KiB strips for the benchmark dynamics) stay in the XCD's
// 4 MiB L2, where 64 columns in a row reach over three times as many.  A bijection on every
// aligned block of 8 rows of columns inside the launch's range (identity elsewhere); the
// results do not depend on the order.  (With the static striding of the other kernels the
// workgroups drift apart and the order made no difference: measured.)
#ifndef SDP_COL_TILE
#define SDP_COL_TILE 1
#endif
#ifndef SDP_COL_SHARE_X2
// 1 (three state variables, a launch over the whole grid, x2 a multiple of 64 and x1 of 8 points): the eight
// XCDs' shares of the columns are cut along x2 -- every XCD walks ALL rows of x1, an eighth of x2 each -- instead of
// along x1.  The strips a column's table is built from lie around its next trailing state; with the shares cut along x1
// the chip works on eight bands of x1 at once and their strips together (~70 MB per band at 512^3 x 4 bytes) do not fit
// the cache, so every strip comes from HBM once per band that touches it (measured: 4.0 GB fetched per sweep for a
// 0.54 GB array).  Cut along x2, the eight shares move through the SAME band of x1 together and share its strips:
// 512^3 x 4 bytes 7.90 -> 7.54 ms, 256^3 x 8 bytes (which fits the cache) 1.336 -> 1.320 ms, same box.  0: along x1 (A/B runs)
#define SDP_COL_SHARE_X2 1
#endif
SDP_DEV int64_t sdp_col_of_unit(const SdpSweepArgs &a, int64_t unit)
{
#if SDP_D == 3 && SDP_COL_SHARE_X2 && (SDP_COL_FILTER || SDP_TRAIL_HAS_U)
    {
        const int n1 = a.orders[1], n2 = a.orders[2];
        const int64_t cols = (int64_t)n1 * n2;
        if (a.col_splits == 1 && a.col_begin == 0 && a.col_end == cols && (n2 & 63) == 0 && (n1 & 7) == 0 && cols < ((int64_t)1 << 31)) {
            const unsigned per = (unsigned)(cols >> 3), u = (unsigned)unit;      // units of an XCD's share (sdp_col_walk)
            const unsigned k = u / per, v = u - k * per;
            const unsigned w2 = (unsigned)n2 >> 3;                                // columns of x2 in a share
            const unsigned band = v / (8u * w2), r = v - band * 8u * w2;
            const unsigned tile = r >> 6, within = r & 63u;
            return (int64_t)(band * 8u + (within >> 3)) * n2 + (k * w2 + tile * 8u + (within & 7u));
        }
    }
#endif
    const int64_t col = a.col_begin + (int64_t)((unsigned)unit / (unsigned)a.col_splits);     // (units < 2^31: see sdp_col_coords)
#if SDP_D == 3 && (SDP_COL_FILTER || SDP_TRAIL_HAS_U) && SDP_COL_TILE
    const int64_t n2 = a.orders[2];
    if ((n2 & 7) == 0) {
        const int64_t blk = 8 * n2;
        const int64_t b0 = (int64_t)((unsigned)col / (unsigned)blk) * blk;
        if (b0 >= a.col_begin && b0 + blk <= a.col_end) {
            const int64_t local = col - b0, tile = local >> 6, within = local & 63;
            return b0 + (within >> 3) * n2 + (tile << 3) + (within & 7);
        }
    }
#endif
    return col;
}

SDP_DEV void sdp_col_coords(const SdpSweepArgs &a, int64_t col, sdp_real *x)
{
    const sdp_real *axes = (const sdp_real *)a.axes;
    // (32-bit unsigned arithmetic: the grid has fewer than 2^31 nodes -- sdp_problem_create -- and a 64-bit division of
    // uniform values is a hundred scalar instructions, several times per wave and unit)
    unsigned r = (unsigned)col;
#pragma unroll
    for (int k = SDP_D - 1; k >= 1; --k) {
        const unsigned n = (unsigned)a.orders[k], q = r / n;
        const int i = (int)(r - q * n);
        r = q;
        x[k] = axes[a.axis_off[k] + i];
    }
    x[0] = (sdp_real)0;
}

// Row window of a unit (nodes i_lo .. i_hi-1 of column `col`): predicted from
// the rows the first and the last control of every node lead to (and the first
// / last perturbation point when x0' depends on it) -- exact when x0' is monotone
// in u and w, which is what a stock is; anything else costs time, not
// correctness (sdp_col_expected_cost).  `pol`: fixed-policy evaluation, the
// one control of a node is its policy value.  Every thread of the workgroup
// calls it; the result is published by sdp_col_window_read after a barrier.
SDP_DEV int sdp_col_guess_row(const SdpLeadAxis &l, sdp_real rspan, sdp_real xn0)
{
    const sdp_real p = (xn0 - l.smin) * rspan;         // (prediction only: reciprocal, not pyx:75)
    return max(min(sdp_trunc_i32(p), l.ordm2), 0);
}

template <bool POL>
SDP_DEV void sdp_col_window_predict(const SdpSweepArgs &a, const SdpLeadAxis &l, int (*win)[2],
                                    int parity, int64_t col, int i_lo, int i_hi, sdp_real *x,
                                    sdp_real t)
{
    if (!SDP_COL_WINDOW) return;
    const sdp_real *__restrict__ axis0 = (const sdp_real *)a.axes + a.axis_off[0];
    const sdp_real rspan = l.nm1 / l.span;
    int qmin = INT_MAX, nqmax = INT_MAX;
    for (int i = i_lo + (int)threadIdx.x; i < i_hi; i += blockDim.x) {
        const int64_t node = col * SDP_COL_N0 + i;
        x[0] = axis0[i];
        SdpBox box;
        if (!POL) sdp_load_box(a, node, box);
#pragma unroll
        for (int cc = 0; cc < (POL ? 1 : 2); ++cc) {
            sdp_real u[SDP_NU];
            if (POL) {
#pragma unroll
                for (int c = 0; c < SDP_NU; ++c) u[c] = ((const sdp_real *)a.pol_in)[node * SDP_NU + c];
            } else {
                sdp_controls_at(box, cc ? box.total - 1 : 0, u);
            }
#if SDP_LEAD_HAS_W && SDP_HAS_W
#pragma unroll
            for (int ww = 0; ww < 2; ++ww) {
                const int q = sdp_col_guess_row(l, rspan, sdp_model_lead(x, u, ((const sdp_real *)a.wgrid)[ww ? SDP_COL_W - 1 : 0], t));
                qmin = min(qmin, q);
                nqmax = min(nqmax, -q);
            }
#else
            const int q = sdp_col_guess_row(l, rspan, sdp_model_lead(x, u, (sdp_real)0, t));
            qmin = min(qmin, q);
            nqmax = min(nqmax, -q);
#endif
        }
    }
    qmin = sdp_wave_min(qmin);
    nqmax = sdp_wave_min(nqmax);
    if ((threadIdx.x & 63) == 0 && qmin != INT_MAX) {
        atomicMin(&win[parity][0], qmin);
        atomicMin(&win[parity][1], nqmax);
    }
}

// after the barrier that follows sdp_col_window_predict: first row of the table.
// The window is centred on the predicted rows when they fit (one row of margin
// below), and starts at the lowest predicted row otherwise.
SDP_DEV int sdp_col_window_read(int (*win)[2], int parity)
{
    if (!SDP_COL_WINDOW) return 0;
    const int qmin = __builtin_amdgcn_readfirstlane(win[parity][0]);
    const int qmax = -__builtin_amdgcn_readfirstlane(win[parity][1]);
    if (threadIdx.x < 2) win[parity ^ 1][threadIdx.x] = INT_MAX;     // for the next unit
    if (qmin == INT_MAX) return 0;
    const int need = qmax + 2 - qmin;                                 // rows qmin .. qmax+1
    int r0 = need < SDP_COL_ROWS ? qmin - (SDP_COL_ROWS - need) / 2 : qmin;
    return max(min(r0, SDP_COL_N0 - SDP_COL_ROWS), 0);
}

#if SDP_COL_FILTER
#include "sdp_colfilter_kernel.h"     // the certified expectation-first filter: reduced table, shifted lattice, first passes, branch and bound
#endif  // SDP_COL_FILTER

#if !SDP_TRAIL_HAS_U && SDP_COL_WRES < SDP_COL_W
#include "sdp_colres_kernel.h"      // sdp_sweep_col / sdp_evalpol_col with the table a chunk of perturbation points at a time
#elif !SDP_TRAIL_HAS_U
#include "sdp_colfull_kernel.h"     // sdp_sweep_col / sdp_evalpol_col with the whole table (or a row window of it) in LDS
#else   // SDP_TRAIL_HAS_U
#include "sdp_colu_kernel.h"        // sdp_sweep_col / sdp_evalpol_col with a table per control
#endif  // SDP_TRAIL_HAS_U

extern "C" {
__constant__ int32_t sdp_meta[SDP_META_WORDS] = {
    SDP_META_MAGIC, (int32_t)sizeof(sdp_real), SDP_D, SDP_NU, SDP_HAS_W, 1, SDP_COL_N0, SDP_COL_W,
    (SDP_COL_FILTER ? SDP_META_F_FILTER : 0) | (SDP_COL_ROWS < SDP_COL_N0 ? SDP_META_F_WINDOW : 0) |
        (SDP_TRAIL_HAS_U ? SDP_META_F_TRAIL_HAS_U : 0) | (SDP_COL_WPAIR ? SDP_META_F_WPAIR : 0) |
#if SDP_COL_FILTER
        (SDP_COL_LEAN_ON ? SDP_META_F_LEAN : 0) | (SDP_COL_SHIFT ? SDP_META_F_SHIFT : 0) |
#endif
        ((SDP_COL_FILTER || SDP_TRAIL_HAS_U) ? SDP_META_F_CLAIMS : 0) | SDP_META_F_PEER_STORES,
    SDP_COL_FILTER ? SDP_COL_UTAB : 0, SDP_COL_FILTER ? SDP_COL_UTAB_N : 0, SDP_COL_THREADS, SDP_COL_ROWS,
    0, 0,
#ifdef SDP_COLRES_TAIL_BYTES
    SDP_COLRES_TAIL_BYTES
#else
    0
#endif
    };
}

#else   // SDP_D < 2: no column kernels; the unit is a node-order one after all
extern "C" {
__constant__ int32_t sdp_meta[SDP_META_WORDS] = {
    SDP_META_MAGIC, (int32_t)sizeof(sdp_real), SDP_D, SDP_NU, SDP_HAS_W, 0, 0, 1, SDP_META_F_PEER_STORES, 0, 0, 256, 0, 0, 0, 0};
}
#endif  // SDP_D >= 2
