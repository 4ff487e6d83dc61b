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
#ifndef SDP_COL_FUSED
#define SDP_COL_FUSED 0          // 1: opt-in fused arithmetic (not the reference's rounding sequence)
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
#if SDP_COL_A_WIDE_LOADS && SDP_COL_FUSED
#error "SDP_COL_A_WIDE_LOADS: exact arithmetic only"
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
#if SDP_COL_WRES < SDP_COL_W && (!SDP_COL_FILTER || SDP_COL_WPAIR || SDP_COST_HAS_W || SDP_COL_FUSED || \
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
#if SDP_COL_FILTER && (!SDP_HAS_W || (SDP_LEAD_HAS_W && !SDP_COL_SHIFT) || SDP_TRAIL_HAS_U || SDP_COL_FUSED || SDP_COL_ROWS < SDP_COL_N0)
#error "SDP_COL_FILTER needs a perturbation that reaches x0' through a final sum at most, and the plain full-column table"
#endif
#if SDP_COL_SHIFT && (!SDP_COL_FILTER || !SDP_LEAD_HAS_W || SDP_COL_WPAIR)
#error "SDP_COL_SHIFT is a form of the certified filter for a perturbation that reaches x0'"
#endif
#if SDP_COL_FILTER && SDP_COST_HAS_W && SDP_COL_UTAB
#error "the control table holds sub-expressions without the perturbation: not with a cost that depends on it"
#endif
#if SDP_TRAIL_HAS_U && (SDP_COL_WPAIR || SDP_COL_FUSED || SDP_COL_ROWS < SDP_COL_N0)
#error "control-dependent trailing dynamics: plain full table, exact arithmetic only"
#endif
#if SDP_COL_ROWS < SDP_COL_N0 && (SDP_COL_WPAIR || SDP_COL_FUSED)
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
// default; the resident-chunk kernel builds the table a part at a time)
template <bool SHIFT = false>
SDP_DEV void sdp_col_phase_a(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                             const SdpColShared &s, const int w_begin = 0, const int w_count = SDP_COL_W)
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
#if SDP_COL_FUSED && SDP_HAS_W
        const sdp_real pw_ = ((const sdp_real *)a.proba)[w];
#endif
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
#if SDP_COL_FUSED && SDP_HAS_W
                    const sdp_real entry = val * pw_;
#else
                    const sdp_real entry = val;
#endif
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
#if SDP_COL_FUSED && SDP_HAS_W
                    const sdp_real entry = val * ((const sdp_real *)a.proba)[w];
#else
                    const sdp_real entry = val;
#endif
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
#if SDP_COL_FUSED && SDP_HAS_W
                // fused arithmetic: the table holds p_w * inner(r), so phase B is
                // two fused multiply-adds per cell
                const sdp_real entry = val * ((const sdp_real *)a.proba)[w];
#else
                const sdp_real entry = val;
#endif
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
    sdp_real psum;                         // sum of the weights (fused arithmetic only)
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
    k.psum = (sdp_real)0;
#if SDP_HAS_W && SDP_COL_FUSED
    for (int w = 0; w < SDP_COL_W; ++w) k.psum = k.psum + gp[w];
#endif
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
// layout; the fused variant keeps two partial sums (even / odd points).
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
#if SDP_COL_FUSED
    sdp_v2 acc2[K];
    sdp_real gacc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { acc2[j] = (sdp_v2)(0); gacc[j] = (sdp_real)0; }
#endif
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
#if SDP_COL_FUSED
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    acc2[j] = __builtin_elementwise_fma((sdp_v2)(lam0[j]), hi[b][j],
                              __builtin_elementwise_fma((sdp_v2)(oml0[j]), lo[b][j], acc2[j]));
#if SDP_COST_HAS_W
                    gacc[j] = fma(sdp_model_cost(x, u[j], SDP_COL_GW(k, w), t), SDP_COL_PW(k, w), gacc[j]);
                    gacc[j] = fma(sdp_model_cost(x, u[j], SDP_COL_GW(k, w + 1), t), SDP_COL_PW(k, w + 1), gacc[j]);
#endif
                }
#else
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
#endif
            }
    }
#if SDP_COL_FUSED
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = acc2[j].x + acc2[j].y;
#endif
    if (Wn & 1) {                                   // odd W: the last point sits alone in its pair
        constexpr int w = Wn - 1;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const sdp_real lo = row[j][WP * N0].x;
            const sdp_real hi = row[j][WP * N0 + 1].x;
#if SDP_COL_FUSED
            acc[j] = fma(lam0[j], hi, fma(oml0[j], lo, acc[j]));
#if SDP_COST_HAS_W
            gacc[j] = fma(sdp_model_cost(x, u[j], SDP_COL_GW(k, w), t), SDP_COL_PW(k, w), gacc[j]);
#endif
#else
            const sdp_real val = oml0[j] * lo + lam0[j] * hi;
#if SDP_COST_HAS_W
            g[j] = sdp_model_cost(x, u[j], SDP_COL_GW(k, w), t);
#endif
            const sdp_real jc = g[j] + val;
            acc[j] = acc[j] + jc * SDP_COL_PW(k, w);
#endif
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
#if SDP_COL_FUSED && SDP_COST_HAS_W
        out[j] = gacc[j] + acc[j];
#elif SDP_COL_FUSED
        out[j] = fma(g[j], k.psum, acc[j]);
#else
        out[j] = acc[j];
#endif
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
#if SDP_HAS_W && SDP_COL_FUSED
    // OPT-IN fused arithmetic (DPSolver.arithmetic = 'fused'): mathematically
    //     sum_w p_w (g + val_w) = sum_w p_w g_w + sum_w [ (1-lam0) p_w inner_w(q0) + lam0 p_w inner_w(q0+1) ]
    // with the table pre-scaled by p_w the cell costs two FMAs instead of the
    // reference's six separately rounded operations.  NOT the reference's
    // rounding sequence: J differs by a few ulp (tests bound it at 1e-12
    // relative, the north-star tolerance is 1e-10).
    sdp_real gacc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) gacc[j] = (sdp_real)0;
#pragma unroll SDP_COL_UNROLL_W
    for (int w = 0; w < Wn; ++w) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
#if SDP_LEAD_HAS_W
            SDP_COL_LOCATE(j, SDP_COL_GW(k, w))
#endif
            const sdp_real lo = row[j][w * N0];
            const sdp_real hi = row[j][w * N0 + 1];
            acc[j] = fma(lam0[j], hi, fma(oml0[j], lo, acc[j]));
#if SDP_COST_HAS_W
            gacc[j] = fma(sdp_model_cost(x, u[j], SDP_COL_GW(k, w), t), SDP_COL_PW(k, w), gacc[j]);
#endif
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
#if SDP_COST_HAS_W
        out[j] = gacc[j] + acc[j];
#else
        out[j] = fma(g[j], k.psum, acc[j]);
#endif
    }
#elif SDP_HAS_W
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
// ---------------------------------------------------------------------------
// Certified expectation-first filter for phase B.
//
// When the perturbation reaches neither x0' nor the cost, a control's expected
// cost is, in real arithmetic,
//     R(u) = sum_w p_w (g + oml0 T[w][q0] + lam0 T[w][q0+1])
//          = g sum_w p_w + oml0 A[q0] + lam0 A[q0+1],      A[r] = sum_w p_w T[w][r],
// i.e. ONE lerp on a table reduced over w instead of W of them.  The reference's
// value E(u) (the W x 6 separately rounded operations of sdp_col_expected_cost)
// and the short form F(u) (sdp_col_filter_eval) are both roundings of R(u) --
// same q0, lam0, oml0, g and table entries, which are computed once -- so with
// u = the unit roundoff and S(u) = sum_w |p_w| (|g| + |oml0 T[w][q0]| + |lam0 T[w][q0+1]|):
//     |E - R| <= gamma_{W+4} S,   |F - R| <= gamma_{W+3} S,   gamma_n = n u / (1 - n u)
// (a vertex value passes through at most W+4 roundings on the long path: two of
// the lerp, the add of g, the weight, and at most W accumulations; W+3 on the
// short one).  With the computable bound
//     S^(u) = Pcap |g| + (|oml0| + |lam0|) max(D[q0], D[q0+1]),   D[r] = Pcap max_w |T[w][r]|,
//     Pcap = max(1, sum_w |p_w|)  >=  S(u),
// the radius  delta(u) = 4 (W + 8) u S^(u) (+ tiny)  covers |E - F| with a factor
// ~1.8 to spare (the spare absorbs the roundings of S^, of F +- delta and of the
// sums of the weights; `tiny` = the smallest normal number covers the absolute
// errors of operations that underflow -- it enters through D[r], which carries 2 tiny / (4 (W+8) u)
// on top, and |oml0| + |lam0| >= 1 - u).  Nothing overflows on the long path as
// long as |F| + S^ < 2^1000 (2^100 for 4-byte reals) and sum |p_w| <= 1024:
// anything else -- infinities, NaNs, huge values -- marks the node `bad`.
//
// Per node, with ONE radius delta = the largest delta(u) of its controls (it covers each
// of them):  m = min_u F + delta  bounds the minimum of E from above, so a
// control with F - delta > m is strictly worse than the best one and can be
// neither the argmin nor tied with it.  If exactly one control survives it IS
// the first-occurrence argmin of the reference and J = E of that control,
// evaluated with the reference's operations: same bits.  If several survive
// (near-ties), or the node is bad, the survivors (all controls of a bad node)
// are evaluated with the reference's operations in lattice order and compared
// like the reference does.  The outcome never depends on F or delta beyond
// "which controls were skipped", and skipped controls are provably not minimal:
// J, policy and index are bit-identical to the plain kernel for every input.
// ---------------------------------------------------------------------------
// v_min / v_max as single instructions (the compiler's fmin/fmax may add canonicalising
// operations); only used on values that are not NaN, or on nodes the NaN sends the long way
SDP_DEV double sdp_vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
SDP_DEV double sdp_vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
SDP_DEV float sdp_vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
SDP_DEV float sdp_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// max(a, |b|) with the absolute value as an operand modifier (no separate instruction)
SDP_DEV double sdp_vmax_abs(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
SDP_DEV float sdp_vmax_abs(float a, float b) { float r; asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }

struct SdpColFilter {
    sdp_real psum;      // fl(sum_w p_w), accumulated in w order
    sdp_real pcap;      // max(1, sum_w |p_w|)
    sdp_real cu;        // 4 (W + 8) u  (x SDP_COL_FILTER_SCALE)
    sdp_real floor;     // 2 tiny / cu, added to every D[r]: cu S^ >= tiny whatever the values
    sdp_real ratio;     // pcap / |psum| (lean first pass: |g| pcap <= ratio (|F| + |h|))
    const sdp_cst_real *p, *wg;   // weights and points (scalar loads): the cost's expectation when the cost depends on w
    double psum64;      // wide first pass: sum_w p_w in 8-byte arithmetic
    sdp_real gc;        // wide first pass: sum_w n_w |p_w| (rounded up), n_w = roundings the term of w passes through
    sdp_real glimit;    // min(1, sum_w |p_w|): a bound below LIMIT x glimit keeps |g| itself below LIMIT
    bool ok;            // weights are finite and of ordinary size
    // branch and bound of the short first passes: rows of axis 0 per unit of x0, and what |X| + max |a| may be at most
    sdp_real k_rows, x_cap;
};
constexpr bool SDP_COL_LEAN_ON = SDP_COL_LEAN < 0 ? sizeof(sdp_real) == 8 : SDP_COL_LEAN != 0;
constexpr bool SDP_COL_WIDE_ON = sizeof(sdp_real) == 4 && !SDP_COL_LEAN_ON && SDP_COL_WIDE != 0;
static_assert(!SDP_COL_SHIFT || SDP_COL_LEAN_ON, "the shifted lattice is a form of the lean first pass (8-byte reals)");
static_assert(!SDP_COST_HAS_W || SDP_COL_LEAN_ON || SDP_COL_WIDE_ON,
              "a cost that depends on the perturbation needs the lean / wide first pass (sdp_col_cost_expect)");
// type of the filter values F and of the radius
typedef std::conditional<SDP_COL_WIDE_ON, double, sdp_real>::type sdp_fkey;
// roundings the term of perturbation point w passes through on the reference's path: the two products
// and the sum of the lerp, the cost, the weight, and the additions from step w on (the first one, 0 + t, is exact)
SDP_DEV sdp_real sdp_col_wide_nw(int w) { return (sdp_real)(w == 0 ? SDP_COL_W + 3 : SDP_COL_W - w + 4); }
template <typename R> struct SdpFilterConst;
template <> struct SdpFilterConst<double> {
    static constexpr double tiny = 2.2250738585072014e-308, limit = 0x1p1000, eps = 0x1p-52;
};
template <> struct SdpFilterConst<float> {
    static constexpr float tiny = 1.17549435e-38f, limit = 0x1p100f, eps = 0x1p-23f;
};
constexpr sdp_real SDP_COL_FILTER_TINY = SdpFilterConst<sdp_real>::tiny;      // smallest normal number
constexpr sdp_real SDP_COL_FILTER_LIMIT = SdpFilterConst<sdp_real>::limit;
constexpr sdp_real SDP_COL_FILTER_EPS = SdpFilterConst<sdp_real>::eps;        // 2 u

SDP_DEV void sdp_col_filter_setup(const SdpSweepArgs &a, SdpColFilter &f)
{
    const sdp_cst_real *p = (const sdp_cst_real *)a.proba;
    sdp_real ps = (sdp_real)0, pa = (sdp_real)0;
    for (int w = 0; w < SDP_COL_W; ++w) {
        ps = ps + p[w];
        pa = pa + (p[w] < (sdp_real)0 ? -p[w] : p[w]);
    }
    f.psum = ps;
    f.pcap = pa > (sdp_real)1 ? pa : (sdp_real)1;
    f.cu = (sdp_real)SDP_COL_FILTER_SCALE * (sdp_real)(2 * (SDP_COL_W + 8)) * SDP_COL_FILTER_EPS;   // u = eps / 2
    f.floor = (sdp_real)2 * SDP_COL_FILTER_TINY / f.cu;
    f.ratio = f.pcap / fabs(ps);                           // (psum = 0: infinite -> every node takes the long way)
    double ps64 = 0.0;
    sdp_real gc = (sdp_real)0;
    for (int w = 0; w < SDP_COL_W; ++w) {
        ps64 += (double)p[w];
        gc = gc + sdp_col_wide_nw(w) * (p[w] < (sdp_real)0 ? -p[w] : p[w]);
    }
    f.p = p;
    f.wg = (const sdp_cst_real *)a.wgrid;
    f.psum64 = ps64;
    f.gc = gc * (sdp_real)1.0001;                          // (the roundings of this sum itself)
    f.glimit = pa < (sdp_real)1 ? pa : (sdp_real)1;        // (bound >= n_w |p_w| |g| summed >= |g| sum |p_w|)
    f.ok = pa <= (sdp_real)1024;                           // false for NaN
}

// after phase A (and a barrier): the reduced table, one thread per row.
// Lean form: ad[r] = A[r] alone, and ONE bound for the column, dcol[parity] = max_r D[r] (an
// integer maximum of the bit patterns: D >= 0), which the caller reads after the next barrier.
SDP_DEV void sdp_col_filter_reduce(const SdpSweepArgs &a, SdpColLds &m, const SdpColFilter &f, int parity)
{
    constexpr int N0 = SDP_COL_ROWS;
    const sdp_cst_real *p = (const sdp_cst_real *)a.proba;
    sdp_real dmax = (sdp_real)0;
    if (SDP_COL_WIDE_ON) {
        // wide layout, 16 bytes per row: A[r] = sum_w p_w T[w][r] accumulated in 8-byte reals (the
        // products are exact there), then B[r] = sum_w n_w |p_w| |T[w][r]| (sdp_col_wide_nw)
        // The bounds weigh |T| with the weights, which may be tiny: the RAW magnitude of the column's
        // entries is published beside them (dcol), for the check that nothing can overflow on the
        // reference's 4-byte path ((1 + 2 max |p|) max |T| < 2^100, sdp_col_filter_nodes).
        sdp_real tmax = (sdp_real)0;
        for (int r = threadIdx.x; r < N0; r += blockDim.x) {
            double acc = 0.0;
            sdp_real bsum = (sdp_real)0;
#pragma unroll SDP_COL_FILTER_RUNROLL
            for (int w = 0; w < SDP_COL_W; ++w) {
#if SDP_COL_WPAIR
                const sdp_real v = m.T[((w >> 1) * N0 + r) * 2 + (w & 1)];
#else
                const sdp_real v = m.T[w * N0 + r];
#endif
                acc = fma((double)p[w], (double)v, acc);
#if !SDP_COL_WIDE2                                            // (the short wide pass bounds B[r] by Gc max |T|)
                bsum = fma(sdp_col_wide_nw(w) * fabs(p[w]), fabs(v), bsum);
#endif
                tmax = sdp_vmax_abs(tmax, v);
            }
            SDP_AD_A(m.ad, r) = acc;
            if (!SDP_COL_WIDE2) m.ad[4 * r + 2] = bsum;
            if (!(acc == acc)) tmax = (sdp_real)INFINITY;        // (a NaN entry, which the maximum skips)
        }
        tmax = sdp_wave_max(tmax);
        if ((threadIdx.x & 63) == 0)
            atomicMax(&m.dcol[parity], (unsigned long long)__double_as_longlong((double)tmax));
        return;
    }
    for (int r = threadIdx.x; r < N0; r += blockDim.x) {
        sdp_real acc = (sdp_real)0, big = (sdp_real)0;
#pragma unroll SDP_COL_FILTER_RUNROLL
        for (int w = 0; w < SDP_COL_W; ++w) {
#if SDP_COL_WPAIR
            const sdp_real v = m.T[((w >> 1) * N0 + r) * 2 + (w & 1)];
#else
            const sdp_real v = m.T[w * N0 + r];
#endif
            acc = acc + p[w] * v;
            big = sdp_vmax_abs(big, v);
        }
        // (>= tiny / cu: the radius never drops below `tiny`; a NaN entry, which the max skips,
        // shows in acc and makes the row's bound infinite)
        const sdp_real d = acc == acc ? f.pcap * big + f.floor : (sdp_real)INFINITY;
        if (SDP_COL_LEAN_ON) {
            m.ad[r] = acc;
            dmax = sdp_vmax(dmax, d);
        } else {
            m.ad[2 * r] = acc;
            m.ad[2 * r + 1] = d;
        }
    }
    if (SDP_COL_LEAN_ON) {
        dmax = sdp_wave_max(dmax);                          // (>= 0, or +inf; never a NaN)
        if ((threadIdx.x & 63) == 0)
            atomicMax(&m.dcol[parity], (unsigned long long)__double_as_longlong((double)dmax));
    }
}
// after the barrier that follows the reduction: the column's bound; the slot of the next unit is cleared
SDP_DEV sdp_real sdp_col_filter_dcol(SdpColLds &m, int parity)
{
    if (!SDP_COL_LEAN_ON && !SDP_COL_WIDE_ON) return (sdp_real)0;
    const unsigned long long bits = m.dcol[parity];
    if (threadIdx.x == 0) m.dcol[parity ^ 1] = 0ull;
    return (sdp_real)__longlong_as_double((long long)bits);
}

#if SDP_COL_SHIFT
// ---------------------------------------------------------------------------
// Shifted lattice (SDP_COL_SHIFT): the certified filter when the perturbation reaches x0' through final
// sums, x0' = fl(.. fl(fl(a(x, u) +- b_1) +- b_2) ..), b_i = b_i(x_1.., w), k <= 4 terms -- the stock of the
// inventory example, `x + u - w` (reference doc/example_inventory.py:31-33; expectation at
// stodynprog.py:679-683), a reservoir `x + u - 0.5 w - 0.1 y`.
// With Tw(s) the reference's interpolation of row w of the table at axis-0 position s (continuous,
// piecewise linear, linear beyond both ends: pyx:75-88 clamps the cell, not lam), c = (N0-1)/span,
//     R(u) = sum_w p_w (g + Tw(s_w)),      s_w = the reference's position of its x0'
// and in real arithmetic s_w = pa + pb_w, pa = (a - smin) c, pb_w = B_w c, B_w = sum_i +- b_i: every control
// of the column sees the perturbation points as the SAME shifts pb_w of its own position pa.  So
//     G(s) = sum_w p_w Tw(s + pb_w)
// is ONE function per column, R(u) = g P + G(pa(u)).  G is piecewise linear with W kinks per row; it is
// tabulated at the whole positions k = kmin .. kmax (A'[k] = G(k), 32 x 3 LDS reads per entry -- once per
// column, not per control) and the first pass is the lean one on that table: F = fma(g, psum, lerp(A', pa)).
// What the lerp between whole positions leaves out is bounded cell by cell: on [k, k+1] the function
// Tw(. + pb_w) has one kink, at distance f_w = frac(pb_w) from the cell's upper end, where its slope changes
// by the second difference d2 = T[w][j+2] - 2 T[w][j+1] + T[w][j] (j = k + floor(pb_w); no kink beyond the
// ends of the axis); a function with one kink of size d2 leaves its chord by at most f (1 - f) |d2|.  Hence
//     |G(s) - chord_k(s)| <= B'[k] = sum_w |p_w| f_w (1 - f_w) |d2_w,k|        for s in [k, k+1]
// and G is LINEAR below 1 - max pb and above N0 - 2 - min pb: with kmin = -(max floor pb + 1), kmax =
// N0 - 1 - min floor pb the first and the last cell of the lattice lie in those ranges, the clamped cell
// with an unclamped lam extrapolates G exactly, and B' is zero there by construction.
// Roundings (u the unit roundoff, D = Pcap max |T| of the column as in the lean pass, P = sum |p_w|):
//  * positions: the reference rounds its k sums (each partial sum is at most |a| + sum |b_i|), then
//    (. - smin) / span * (N0-1); this pass rounds B (k-1 sums), pa, pb_w, pa - kmin.  a and the b_i are the
//    reference's own values (same operations on the same inputs), so with PB >= c sum_i |b_i|
//    |s_w - (pa + pb_w)| <= u ((k + 6.2) |pa| + (2k + 5.2) PB + k |smin| c) <= 14 u (|pa| + PB + |smin| c),
//    and |Tw(s) - Tw(s')| <= 2 max|T| |s - s'|;
//  * |E - R| <= (W+4) u [ |g| P + (1 + 2 Lam) D ],  Lam = max_w |lam_w| <= |s_w| + N0;
//  * A'[k] carries (W+4) u (1 + 2 (spread + 2)) D (lam of an entry reaches spread + 2 at the ends of the
//    lattice, spread = max - min floor pb), the lerp and the fma 4 u (1 + 2 |lam0|) max |A'|.
// With L = max |lam0| of the node (|pa - kmin| <= rows + L), Lc = rows + |kmin| + max PB + |smin| c + N0 + 1
// (column-uniform) and Es = 1 + 2 (spread + 2) all of it is below
//     (2W+8) u |g| P  +  (W+8) u H D,      H = (1 + 2 L + 2 Lc) (3 + Es)
// and with |g| Pcap <= ratio (|F| (1 + u) + |h|), |h| <= (1 + 2 L) Es D <= H D, as in the lean pass:
//     radius = cu S_node + max_u B'[q0(u)],    S_node = ratio (sum |F| + H D) + H D,   cu = 4 (W+8) u
// (a factor 2 on the |g| term, 4 on the D term to spare; the roundings of B' itself -- (W+8) u relative and
// 3 u max |T| per d2 -- are far inside that slack since B' <= D).  |s_w| < 2^31 for every w (the x86
// truncation of the reference, sdp_trunc_i32) follows from L + Lc < 2^30.  A column whose shifts are not
// finite, exceed 2^29 rows or need more than SDP_COL_SHIFT_ROWS rows marks all its nodes: they evaluate
// every control the long way, like a node with a NaN.
// The second pass is unchanged (sdp_col_expected_cost with the position located per perturbation point).
// In 8-byte reals the radius is now B' -- of the order of h^2 V'' / 16 for a smooth cost-to-go -- and no
// longer 1e-13: ~1 % of the nodes of the benchmark problem keep two controls, hence SDP_COL_FILTER_TOP2.

// the shifts of the perturbation points for the column at x[1..] into the tables of parity `par`
// (threads `first` ..; sh_k[par] was reset a barrier ago)
SDP_DEV void sdp_col_phase_shift(const SdpSweepArgs &a, SdpColLds &m, const SdpLeadAxis &l, const sdp_real *x,
                                 sdp_real t, int par, int first = 0, int count = 0)
{
    if (count == 0) count = (int)blockDim.x - first;
    if ((int)threadIdx.x < first || (int)threadIdx.x >= first + count) return;
    const sdp_real *__restrict__ wgrid = (const sdp_real *)a.wgrid;
    const sdp_real *__restrict__ p = (const sdp_real *)a.proba;
    for (int w = (int)threadIdx.x - first; w < SDP_COL_W; w += count) {
        sdp_real b, babs;                                            // the signed sum of the b_i, and the sum of the |b_i|
        sdp_model_lead_b(x, wgrid[w], t, b, babs);
        const sdp_real pb = sdp_div_span<sdp_real>(b, l.span, l.rspan, l.pow2) * l.nm1;
        const sdp_real pbabs = fabs(sdp_div_span<sdp_real>(babs, l.span, l.rspan, l.pow2) * l.nm1);
        const bool ok = fabs(pb) < (sdp_real)536870912.0 && pbabs < (sdp_real)536870912.0;     // (false for a NaN)
        const sdp_real fl = ok ? floor(pb) : (sdp_real)0;
        const sdp_real f = ok ? pb - fl : (sdp_real)0;                // exact, in [0, 1)
        const int q = (int)fl;
        m.sh_q[par][w] = q;
        m.sh_f[par][w] = f;
        m.sh_c[par][w] = fabs(p[w]) * (f * ((sdp_real)1 - f));
        atomicMax(&m.sh_k[par][0], q);
        atomicMax(&m.sh_k[par][1], -q);
        if (!ok) atomicMax(&m.sh_k[par][2], 1);
        atomicMax(&m.sh_k[par][3], ok ? (int)pbabs + 1 : 0);         // >= sum_i |b_i| c, in rows
    }
}
SDP_DEV void sdp_col_shift_reset(SdpColLds &m, int par)
{
    m.sh_k[par][0] = INT_MIN;
    m.sh_k[par][1] = INT_MIN;
    m.sh_k[par][2] = 0;
    m.sh_k[par][3] = 0;
}
// what the first pass needs of the lattice of parity `par` (after the barrier that follows sdp_col_phase_shift)
struct SdpColShiftCol {
    int kmin, rows;     // first whole position, number of positions
    int flmin, flmax;   // smallest / largest whole part of a shift
    bool ok;            // usable (else every node of the unit takes the long way)
    sdp_real lc, es;    // Lc and Es of the bound
};
SDP_DEV void sdp_col_shift_col(const SdpColLds &m, const SdpLeadAxis &l, int par, SdpColShiftCol &c)
{
    const int flmax = __builtin_amdgcn_readfirstlane(m.sh_k[par][0]);
    const int nflmin = __builtin_amdgcn_readfirstlane(m.sh_k[par][1]);
    const int flag = __builtin_amdgcn_readfirstlane(m.sh_k[par][2]);
    c.kmin = -(flmax + 1);
    c.rows = SDP_COL_N0 + flmax + nflmin + 1;
    c.flmax = flmax;
    c.flmin = -nflmin;
    c.ok = flag == 0 && c.rows <= SDP_COL_SHIFT_ROWS && c.rows >= 2 && SDP_COL_N0 >= 3;
    const sdp_real pbmax = (sdp_real)(max(max(abs(flmax), abs(nflmin)), __builtin_amdgcn_readfirstlane(m.sh_k[par][3])) + 1);
    const sdp_real p0 = fabs(l.smin) * (l.nm1 / fabs(l.span));
    c.lc = (sdp_real)c.rows + (sdp_real)abs(c.kmin) + pbmax + p0 * (sdp_real)1.001 + (sdp_real)(SDP_COL_N0 + 1);
    c.es = (sdp_real)(1 + 2 * (flmax + nflmin + 2));
    if (!(c.lc < (sdp_real)1073741824.0)) c.ok = false;              // (a NaN or an infinity of p0)
    c.lc = sdp_uniform(c.lc);                               // (the same in every lane: scalar registers)
    c.es = sdp_uniform(c.es);
}

// before phase A (the readers of the previous unit's table left at the barrier): clear the lattice
SDP_DEV void sdp_col_shift_zero(SdpColLds &m, const SdpColShiftCol &c)
{
    if (!c.ok) return;
    for (int i = threadIdx.x; i < 2 * c.rows; i += blockDim.x) m.ad[i] = (sdp_real)0;
}
// after phase A (and a barrier): the table reduced over w on the shifted lattice: ad[2 i] = A'[kmin + i],
// ad[2 i + 1] = B'[kmin + i] (the cell above it); dcol as in the lean pass.  A wave takes every waves-th
// perturbation point and walks the lattice in blocks of 64 positions (a lane per position: consecutive rows
// of one table row, no bank conflict), adding its partial sums into the cleared lattice (LDS atomics, two per
// lane and block): the positions rarely fill a whole number of thread-per-position rounds, the (block, w)
// items do.  Blocks whose positions stay inside the axis for every perturbation point -- all but the first
// and the last ones -- skip the clamps.
// (w_lo, w_cnt: the perturbation points held by table rows 0 .. w_cnt-1 -- all of them by default; the resident-chunk
// kernel adds the points to the lattice a part of the table at a time)
SDP_DEV void sdp_col_shift_reduce(const SdpSweepArgs &a, SdpColLds &m, const SdpColFilter &f,
                                  const SdpColShiftCol &c, int parity, int par, const int w_lo = 0,
                                  const int w_cnt = SDP_COL_W)
{
    constexpr int N0 = SDP_COL_N0;
    if (!c.ok) return;
    const sdp_cst_real *p = (const sdp_cst_real *)a.proba;
    const int lane = threadIdx.x & 63;
    // (compile-time trip counts; the shifts of a wave's perturbation points are fetched once, so that the table
    // reads of a block do not wait for them one after the other: the loop is bound by LDS latency, not by issue)
    constexpr int waves = SDP_COL_THREADS / 64, rounds_all = (SDP_COL_W + waves - 1) / waves, CH = rounds_all < 8 ? rounds_all : 8;
    const int rounds = (w_cnt + waves - 1) / waves;
    sdp_trap_unless(blockDim.x == SDP_COL_THREADS);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave >= w_cnt) return;
    const int blocks = (c.rows + 63) >> 6;
    sdp_real dmax = (sdp_real)0;
    for (int i0 = 0; i0 < rounds; i0 += CH) {
        int tq[CH];
        sdp_real tf[CH], tc[CH], tp[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int w = w_lo + wave + (i0 + i) * waves;
            const bool valid = w < w_lo + w_cnt;                       // (past the end: this wave's first point
            const int wv = valid ? w : w_lo + wave;                    //  again, with weights zero)
            tq[i] = (wv - w_lo) * N0 + __builtin_amdgcn_readfirstlane(m.sh_q[par][wv]);
            tf[i] = m.sh_f[par][wv];
            tc[i] = valid ? m.sh_c[par][wv] : (sdp_real)0;
            tp[i] = valid ? (sdp_real)p[wv] : (sdp_real)0;
        }
        for (int b = 0; b < blocks; ++b) {
            const int ki = b * 64 + lane;
            const int kb = c.kmin + b * 64;
            const int k = min(kb + lane, c.kmin + c.rows - 1);
            sdp_real acc = (sdp_real)0, bnd = (sdp_real)0, big = (sdp_real)0;
            if (kb + c.flmin >= 0 && kb + 63 + c.flmax <= N0 - 3) {    // (wave-uniform)
                sdp_real tv[CH][3];                                    // (all reads of the block first)
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const sdp_real *row = m.T + (tq[i] + k);
                    tv[i][0] = row[0];
                    tv[i][1] = row[1];
                    tv[i][2] = row[2];
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const sdp_real t0 = tv[i][0], t1 = tv[i][1], t2 = tv[i][2];
                    acc = fma(tp[i], fma(tf[i], t1 - t0, t0), acc);
                    bnd = fma(tc[i], fabs((t2 - t1) - (t1 - t0)), bnd);
                    big = sdp_vmax_abs(big, t1);                       // (every entry is the t1 of some position,
                }                                                      //  row 0 the t0 of a clamped one)
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int wq = w_lo + wave + (i0 + i) * waves;
                    const int trow = (wq < w_lo + w_cnt ? wq : w_lo + wave) - w_lo;        // table row of the point
                    const int j = k + (tq[i] - trow * N0);
                    const int q = max(min(j, N0 - 2), 0);
                    const sdp_real lam = (sdp_real)(j - q) + tf[i];
                    const sdp_real *row = m.T + trow * N0 + q;
                    const sdp_real t0 = row[0], t1 = row[1], t2 = row[q + 2 < N0 ? 2 : 1];
                    acc = fma(tp[i], fma(lam, t1 - t0, t0), acc);
                    const sdp_real d2 = (t2 - t1) - (t1 - t0);
                    const bool kink = (unsigned)j <= (unsigned)(N0 - 3);   // a kink inside the cell: row j+1 is an inner row
                    bnd = fma(tc[i], kink ? fabs(d2) : (sdp_real)0, bnd);
                    big = sdp_vmax_abs(sdp_vmax_abs(big, t0), t1);
                }
            }
            if (ki < c.rows) {
                __hip_atomic_fetch_add(&m.ad[2 * ki], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&m.ad[2 * ki + 1], bnd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // (a NaN entry, which the max skips, shows in the partial sum and makes the bound infinite)
            dmax = sdp_vmax(dmax, acc == acc ? f.pcap * big + f.floor : (sdp_real)INFINITY);
        }
    }
    dmax = sdp_wave_max(dmax);
    if (lane == 0)
        atomicMax(&m.dcol[parity], (unsigned long long)__double_as_longlong((double)dmax));
}
#endif  // SDP_COL_SHIFT

// F(u) and S^(u) of one control (x0' cell and cost exactly as sdp_col_expected_cost computes them).
// AXIS, a template argument so that the loop of the first pass carries no branch: 0 the true
// division of pyx:75; 1 a power-of-two span (product with the reciprocal: sdp_div_span); 2 the
// axis [0, 1] (x - 0.0 and x * 1.0 are x, bit for bit).  `pmax` collects |p|: the truncation
// to an int has x86 semantics beyond 2^31 (sdp_trunc_i32) -- a node that gets there takes the
// long way instead of paying for the check on every control.
template <int AXIS>
SDP_DEV void sdp_col_filter_eval(const sdp_real *ad_tab, const SdpColFilter &f, const SdpLeadAxis &l,
                                 const sdp_real *x, const sdp_real *u, sdp_real t, sdp_real &F, sdp_real &S,
                                 sdp_real &pmax)
{
    const sdp_real xn0 = sdp_model_lead(x, u, (sdp_real)0, t);
    const sdp_real sn = AXIS == 2 ? xn0 : (AXIS == 1 ? (xn0 - l.smin) * l.rspan : (xn0 - l.smin) / l.span);
    const sdp_real p = sn * l.nm1;
    pmax = sdp_vmax(pmax, fabs(p));
    const int q0 = max(min((int)p, l.ordm2), 0);            // (saturating conversion; NaN -> 0)
    const sdp_real lam0 = p - (sdp_real)q0;
    const sdp_real oml0 = (sdp_real)1 - lam0;
    const sdp_real g = sdp_model_cost(x, u, (sdp_real)0, t);
    const sdp_real *ad = ad_tab + 2 * q0;
    const sdp_real a0 = ad[0], d0 = ad[1], a1 = ad[2], d1 = ad[3];
    F = g * f.psum + (oml0 * a0 + lam0 * a1);
    S = fma(fabs(g), f.pcap, (fabs(oml0) + fabs(lam0)) * sdp_vmax(d0, d1));
}
SDP_DEV int sdp_col_axis_mode(const SdpLeadAxis &l)
{
#ifdef SDP_COL_AXIS_MODE                                   // generated where the host knows axis 0 of the grid: one form of the passes instead of three
    (void)l;
    return SDP_COL_AXIS_MODE;
#endif
    if (!l.pow2) return 0;
    return (l.smin == (sdp_real)0 && l.span == (sdp_real)1) ? 2 : 1;
}

// ---------------------------------------------------------------------------
// Lean first pass (SDP_COL_LEAN).  Vector issue is what binds the first pass, and on gfx950
// every vector instruction of a mixed stream costs ~4.2-4.4 clocks of its SIMD whatever its
// type (profiles/r03_ubench_valu_rate.txt), so the pass is as fast as it is short.  Per control
// it keeps only what must be per control:
//     q0, lam0, g       exactly as the reference computes them (they are inputs of E too)
//     F = fma(g, psum, fma(lam0, A[q0+1] - A[q0], A[q0]))      (3 operations; free to fuse: F is a
//                        filter value, not a result)
//     L = max |lam0|,   Fs = sum |F|                             (2 operations)
// and bounds the error ONCE per node.  With R(u) = g P + oml0 A0* + lam0 A1* the real number both
// E (the reference's W x 6 roundings) and F approximate, D >= sum_w |p_w T[w][r]| for every row r
// of the column (dcol), P = sum |p_w|, u the unit roundoff, first order in u:
//     |E - R| <= (W+4) u [ |g| P + (|oml0| + |lam0|) D ]                        (as before)
//     |F - R| <= u [ (W+1) |g| P + D ((W+1) |1 - lam0| + (W+6) |lam0| + 2) ]
//       (g (psum - P*): W-1 additions;  A0, A1: W roundings per term;  oml0 = fl(1 - lam0) of the
//        reference against the exact 1 - lam0 inside the fma: u |1 - lam0| |A0*|;  the difference
//        A1 - A0: u (|A0| + |A1|) |lam0|;  the two fma roundings: u (|h| + |g psum + h|))
//     |E - F| <= (2W+8) u [ |g| P + (1 + 2 |lam0|) D ]        since |oml0|, |1 - lam0| <= 1 + |lam0| (1 + u)
// |g| is not tracked: F = (g psum + h)(1 + d) gives |g| Pcap <= ratio (|F| (1 + u) + |h|), ratio =
// Pcap / |psum|, |h| <= (1 + 2 |lam0|) D (1 + 3u).  So with
//     S_node = ratio (Fs + (1 + 2L) D) + (1 + 2L) D   >=   |g| Pcap + (1 + 2 |lam0|) D   for every control
// the radius cu S_node, cu = 4 (W+8) u, covers |E - F| with a factor 2 to spare for the second-order
// terms and the roundings of S_node itself.  A NaN anywhere (g, lam0, the table) makes F a NaN, which
// sticks in Fs; an infinity makes Fs or D infinite; |p| >= 2^31 (where the truncation of the
// reference has x86 semantics, sdp_trunc_i32) makes |lam0| >= 2^30: all of them mark the node
// `bad`, and a bad node evaluates every control the long way.
#if SDP_COST_HAS_W
// A cost that depends on the perturbation (x0' still does not): the expectation no longer reduces g to
// one value, but it still commutes with the lerp along axis 0 -- R(u) = sum_w p_w g_w + oml0 A0* + lam0 A1*.
// The first pass accumulates G = sum_w p_w g_w with the reference's own g_w (W cost evaluations per control:
// the lerp, the add, the weight and the accumulation of the long way -- 6 of its c + 6 operations per
// perturbation point -- are what is saved) and, for the bound, Gabs = sum_w c_w |g_w| with c_w = |p_w| (lean:
// |fl(G) - sum p_w g_w| <= gamma_W sum |p_w g_w|, and the reference's own path adds gamma_{W+4} of the same
// sum) or n_w |p_w| (wide: the roundings of term w, sdp_col_wide_nw).  Gabs replaces |g| Pcap / Gc |g| in
// the bounds of sdp_col_lean_core / sdp_col_wide_core; everything else is unchanged.
template <bool WIDE, typename ACC>
SDP_DEV void sdp_col_cost_expect(const SdpColFilter &f, const sdp_real *x, const sdp_real *u, sdp_real t,
                                 ACC &G, sdp_real &Gabs)
{
    G = (ACC)0;
    Gabs = (sdp_real)0;
    sdp_real graw = (sdp_real)0;
#pragma unroll 4
    for (int w = 0; w < SDP_COL_W; ++w) {
        const sdp_real pw = f.p[w];
        const sdp_real gw = sdp_model_cost(x, u, f.wg[w], t);
        G = fma((ACC)pw, (ACC)gw, G);
        Gabs = fma((WIDE ? sdp_col_wide_nw(w) : (sdp_real)1) * fabs(pw), fabs(gw), Gabs);
        graw = sdp_vmax_abs(graw, gw);
    }
    // (the RAW magnitude too: a tiny or zero weight must not hide a g_w that overflows g_w + val on the
    // reference's path -- the bound is compared with the overflow limit)
    Gabs = sdp_vmax(Gabs, graw);
}
#endif

template <int AXIS>
SDP_DEV void sdp_col_lean_core(const sdp_real *A, const SdpColFilter &f, const SdpLeadAxis &l,
                               sdp_real xn0, sdp_real g, sdp_real &F, sdp_real &lmax, sdp_real &bmax)
{
    const sdp_real sn = AXIS == 2 ? xn0 : (AXIS == 1 ? (xn0 - l.smin) * l.rspan : (xn0 - l.smin) / l.span);
    const sdp_real p = sn * l.nm1;
#if SDP_COL_SHIFT
    // shifted lattice: xn0 = a(x, u), `l` = the lattice (koff = its first position, ordm2 = its rows - 2),
    // A = (A', B') pairs; pk >= 0 inside the lattice, so the truncation is the floor there, and below it
    // the clamp takes the first cell, which extrapolates G exactly (see sdp_col_shift_reduce)
    const sdp_real pk = p - l.koff;
    int q0 = (int)pk;
    asm("v_med3_i32 %0, %0, 0, %1" : "+v"(q0) : "s"(l.ordm2));
    const sdp_real lam0 = pk - (sdp_real)q0;
    lmax = sdp_vmax_abs(lmax, lam0);
    const sdp_real a0 = A[2 * q0], a1 = A[2 * q0 + 2];
    bmax = sdp_vmax(bmax, A[2 * q0 + 1]);
#else
    (void)bmax;
    int q0 = (int)p;                                        // (saturating conversion; NaN -> 0)
    asm("v_med3_i32 %0, %0, 0, %1" : "+v"(q0) : "s"(l.ordm2));   // max(min(q0, ordm2), 0): ordm2 >= 0
    const sdp_real lam0 = p - (sdp_real)q0;
    lmax = sdp_vmax_abs(lmax, lam0);
    const sdp_real a0 = A[q0], a1 = A[q0 + 1];
#endif
#if SDP_COST_HAS_W
    F = g + fma(lam0, a1 - a0, a0);                          // (g: the expectation G of the cost)
#else
    F = fma(g, f.psum, fma(lam0, a1 - a0, a0));
#endif
}
// gmax: with a cost that depends on w, the largest Gabs of the node's controls (else untouched)
template <int AXIS>
SDP_DEV void sdp_col_lean_eval(const sdp_real *A, const SdpColFilter &f, const SdpLeadAxis &l,
                               const sdp_real *x, const sdp_real *u, sdp_real t, sdp_real &F, sdp_real &lmax,
                               sdp_real &gmax, sdp_real &bmax)
{
#if SDP_COL_SHIFT
    const sdp_real xn0 = sdp_model_lead_a(x, u, t);
#if SDP_COL_SHIFT_CHAIN
    // A chain of sums that was regrouped (x + (w - u): a = x - u is not a value the reference computes).  With SA, SB the
    // sums of the magnitudes of the chain's w-free and other leaves and m <= 3 its additions, the reference's sum and
    // this pass's a + B each lie within gamma_m (SA + SB) of the real sum: 2 m u (1 + eps) (PA + PB) rows, PA = SA c,
    // instead of the k u (|pa| + PB + |smin| c) .. of the final-sum form.  With the 6.2 |pa| + 5.2 PB of the position's own
    // roundings and |pa| <= PA + |smin| c that is below 14 u (PA + PB + |smin| c) -- the bound of sdp_col_shift_col's
    // comment with PA in the place of |pa| --, and PA enters where |pa| does: through L.
    lmax = sdp_vmax_abs(lmax, sdp_model_lead_aabs(x, u, t) * (fabs(l.nm1 * l.rspan) * (sdp_real)1.002));
#endif
#else
    const sdp_real xn0 = sdp_model_lead(x, u, (sdp_real)0, t);
#endif
#if SDP_COST_HAS_W
    sdp_real g, gabs;
    sdp_col_cost_expect<false>(f, x, u, t, g, gabs);
    gmax = sdp_vmax(gmax, gabs);
#else
    (void)gmax;
    const sdp_real g = sdp_model_cost(x, u, (sdp_real)0, t);
#endif
    sdp_col_lean_core<AXIS>(A, f, l, xn0, g, F, lmax, bmax);
}
#if SDP_COL_UTAB
// the same with the column-uniform parts of x0' and of the cost read from the control table
template <int AXIS>
SDP_DEV void sdp_col_lean_eval_tab(const sdp_real *A, const sdp_real *utab, const SdpColFilter &f,
                                   const SdpLeadAxis &l, const sdp_real *x, int ci, sdp_real t,
                                   sdp_real &F, sdp_real &lmax, sdp_real &bmax)
{
    sdp_real tab[SDP_COL_UTAB];
#pragma unroll
    for (int k = 0; k < SDP_COL_UTAB; ++k) tab[k] = utab[ci * SDP_COL_UTAB + k];
    const sdp_real xn0 = sdp_model_lead_tab(x, tab, t);
    const sdp_real g = sdp_model_cost_tab(x, tab, t);
#if SDP_COL_SHIFT && SDP_COL_SHIFT_CHAIN
    lmax = sdp_vmax_abs(lmax, sdp_model_lead_aabs_tab(x, tab, t) * (fabs(l.nm1 * l.rspan) * (sdp_real)1.002));      // (see sdp_col_lean_eval)
#endif
    sdp_col_lean_core<AXIS>(A, f, l, xn0, g, F, lmax, bmax);
}
// the table of one column (its coordinates in x[1..]; x[0] is not read): threads `first` ..
SDP_DEV void sdp_col_phase_u(const SdpSweepArgs &a, sdp_real *utab, const sdp_real *x, sdp_real t, int first = 0,
                             int count = 0, const SdpBox *box_c = nullptr, sdp_real psum = (sdp_real)0,
                             sdp_real k_rows = (sdp_real)0, sdp_real x_cap = (sdp_real)0, double psum_d = 0.0,
                             bool a_known = false)
{
    // a_known (SDP_LEAN2_A_FIXED: the part a of x0' = X +- a depends on the control alone -- the same lattice of positions
    // in every column): this parity buffer already holds what follows from a -- its smallest and largest value, the blocks'
    // starts, their order, the row count -- from the table made here two units ago; only what follows from h is redone.
    (void)psum; (void)k_rows; (void)x_cap; (void)psum_d; (void)a_known;
    if (count == 0) count = (int)blockDim.x - first;
    if ((int)threadIdx.x < first || (int)threadIdx.x >= first + count) return;
    SdpBox box;
    if (box_c) box = *box_c;
    else sdp_load_box(a, 0, box);                           // (one box for every node: checked at launch)
    const int n_tab = min(box.total, SDP_COL_UTAB_N);        // (SDP_COL_UTAB_N is a capacity: the host checks total <= it)
#if SDP_COL_SHORT
    // the short first pass bounds what it no longer tracks per control by the column's smallest and largest a and
    // largest |h| (x0' = X +- a, cost = K +- h: sdp_colres_kernel.h); a value that is not finite poisons them.
    // ONE wave builds the table there (count == 64), so a wave reduction completes them.
    sdp_real a_lo = INFINITY, a_hi = -INFINITY, h_abs = (sdp_real)0, fin = (sdp_real)0;
#endif
#if SDP_COL_BNB
    // Records of the BLOCKS of controls for the branch and bound of the short first pass (sdp_lean2_bnb), made as the
    // table is: in ROWS of axis 0 relative to the node -- a control's position is p = pX + pa, pX = +-(X -+ smin) k of
    // the node, pa = +-a k of the control, k = (N0 - 1) / span (as real numbers; the kernel's own positions differ from
    // that sum by roundings far below the margin DELTA the ends are moved out by).  Record b: (where block b starts:
    // its smallest pa - DELTA;  its smallest +-h psum, as the first pass forms it); record n_blocks: (where the last
    // block ends: its largest pa + DELTA;  how many rows beyond two can lie between the starts of two neighbouring
    // blocks, as an integer).  The pass uses them only where the blocks are in order, each ending before the next
    // starts (an ordinary lattice, a monotone in the control) and everything is finite: st[3] says so.
    constexpr sdp_real BNB_DELTA = sizeof(sdp_real) == 8 ? (sdp_real)0x1p-20 : (sdp_real)0x1p-8;      // (4-byte reals: the kernel's own positions are off by ~2^-15 rows)
    constexpr bool BNB_WIDE = sizeof(sdp_real) == 4;       // records: (start as a 4-byte real, -, smallest +-h psum as an 8-byte real)
    sdp_real *rec = utab + SDP_COL_UTAB * SDP_COL_UTAB_N + 4;
    const int lane_u = (int)threadIdx.x - first;
    bool bnb_fine = count == 64 && SDP_BNB_BLOCK <= 64;
    sdp_real bnb_prev_hi = -INFINITY, bnb_prev_lo = (sdp_real)NAN, bnb_between = (sdp_real)0, bnb_amax = (sdp_real)0;
    for (int c0 = 0; c0 < n_tab; c0 += 64) {               // (every lane of the wave takes part in every round: shuffles)
        int ci = c0 + lane_u;
        // (opaque to the optimiser: with a constant box the control of a lane and everything computed from it alone are
        // the same in every unit, and the compiler hoists them out of the kernel's unit loop into registers it does not
        // have -- measured: three reloads from scratch memory per unit in this helper wave, which the whole workgroup
        // then waits for at the next barrier.  Recomputing them is a handful of instructions.)
        asm volatile("" : "+v"(ci));
        sdp_real u[SDP_NU], tab[SDP_COL_UTAB];
        sdp_controls_at(box, min(ci, n_tab - 1), u);
        sdp_model_utab(x, u, t, tab);
        if (ci < n_tab) {
#pragma unroll
            for (int k = 0; k < SDP_COL_UTAB; ++k) utab[ci * SDP_COL_UTAB + k] = tab[k];
            if (!a_known) {
                a_lo = sdp_vmin(a_lo, tab[SDP_LEAN2_A_SLOT]);
                a_hi = sdp_vmax(a_hi, tab[SDP_LEAN2_A_SLOT]);
                fin = fin + fabs(tab[SDP_LEAN2_A_SLOT]);
            }
            if (SDP_LEAN2_H_SLOT >= 0) {
                h_abs = sdp_vmax_abs(h_abs, tab[SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT]);
                fin = fin + fabs(tab[SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT]);
            }
        }
        const sdp_real av = tab[SDP_LEAN2_A_SLOT];
        const sdp_real hv = SDP_LEAN2_H_SLOT < 0 ? (sdp_real)0 : tab[SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT];
        const double hpv = (double)(SDP_LEAN2_HNEG ? -hv : hv) * psum_d;
        const bool have = ci < n_tab;
        double hp = have ? hpv : (double)INFINITY;
        constexpr int SEG = SDP_BNB_BLOCK < 64 ? SDP_BNB_BLOCK : 64;
        const bool head = have && (lane_u % SEG) == 0;
        if (a_known) {
            // (the smallest +-h psum of every block, nothing else)
            if (SEG == 8) {
                hp = sdp_vmin(hp, sdp_dpp_f64<0xB1>(hp)); hp = sdp_vmin(hp, sdp_dpp_f64<0x4E>(hp)); hp = sdp_vmin(hp, sdp_dpp_f64<0x141>(hp));
            } else {
#pragma unroll
                for (int d = 1; d < SEG; d <<= 1) hp = sdp_vmin(hp, __shfl_xor(hp, d, 64));
            }
            if (head) {
                const int b = ci / SDP_BNB_BLOCK;
                if (BNB_WIDE) *(double *)(rec + 4 * b + 2) = hp;
                else rec[2 * b + 1] = (sdp_real)hp;
            }
            continue;
        }
        const sdp_real pa = (SDP_LEAN2_FORM == 1 ? -av : av) * k_rows;
        bnb_fine = bnb_fine && (!have || (pa == pa && hpv == hpv && fabs(pa) < (sdp_real)INFINITY));
        bnb_amax = sdp_vmax_abs(bnb_amax, have ? av : (sdp_real)0);
        sdp_real lo = have ? pa : (sdp_real)INFINITY, hi = have ? pa : -(sdp_real)INFINITY;
        // minima / maxima over the aligned groups of SEG lanes: the blocks
        if (SEG == 8) {
            // (data-parallel primitives inside a row of 16 lanes: neighbours, pairs, the mirrored half -- no LDS round trips)
            lo = sdp_vmin(lo, sdp_dpp_f64<0xB1>(lo)); hi = sdp_vmax(hi, sdp_dpp_f64<0xB1>(hi)); hp = sdp_vmin(hp, sdp_dpp_f64<0xB1>(hp));
            lo = sdp_vmin(lo, sdp_dpp_f64<0x4E>(lo)); hi = sdp_vmax(hi, sdp_dpp_f64<0x4E>(hi)); hp = sdp_vmin(hp, sdp_dpp_f64<0x4E>(hp));
            lo = sdp_vmin(lo, sdp_dpp_f64<0x141>(lo)); hi = sdp_vmax(hi, sdp_dpp_f64<0x141>(hi)); hp = sdp_vmin(hp, sdp_dpp_f64<0x141>(hp));
        } else {
#pragma unroll
            for (int d = 1; d < SEG; d <<= 1) {
                lo = sdp_vmin(lo, __shfl_xor(lo, d, 64));
                hi = sdp_vmax(hi, __shfl_xor(hi, d, 64));
                hp = sdp_vmin(hp, __shfl_xor(hp, d, 64));
            }
        }
        // the block that follows in this round, or nothing (the last one of the round meets its successor next round)
        const sdp_real next_lo = __shfl_down(lo, SEG, 64);
        const bool has_next = lane_u + SEG < 64 && ci + SEG < n_tab;
        if (head) {
            const int b = ci / SDP_BNB_BLOCK;
            if (BNB_WIDE) { rec[4 * b] = lo - BNB_DELTA; *(double *)(rec + 4 * b + 2) = hp; }
            else { rec[2 * b] = lo - BNB_DELTA; rec[2 * b + 1] = (sdp_real)hp; }
            if (has_next) {
                bnb_fine = bnb_fine && hi + 2 * BNB_DELTA <= next_lo;      // (a block's controls, each within DELTA / 2 of its
                                                                           //  pa, stay below the start of the next block)
                bnb_between = sdp_vmax(bnb_between, next_lo - lo);
            }
            if (lane_u == 0 && c0 > 0) {                   // (against the last block of the previous round)
                bnb_fine = bnb_fine && bnb_prev_hi + 2 * BNB_DELTA <= lo;
                bnb_between = sdp_vmax(bnb_between, lo - bnb_prev_lo);
            }
            if (ci + SDP_BNB_BLOCK >= n_tab) {             // the last block: where the lattice ends
                rec[(BNB_WIDE ? 4 : 2) * (b + 1)] = hi + BNB_DELTA;
                bnb_between = sdp_vmax(bnb_between, hi - lo);
            }
        }
        if (c0 + 64 < n_tab) {                             // (another round follows)
            bnb_prev_hi = __shfl(hi, 64 - SEG, 64);
            bnb_prev_lo = __shfl(lo, 64 - SEG, 64);
        }
    }
    if (!a_known)
    {
        const bool sorted = __all(bnb_fine);
        bnb_between = sdp_wave_max(bnb_between);
        bnb_amax = sdp_wave_max(bnb_amax);
        if (lane_u == 0) {
            const int n_blocks = (n_tab + SDP_BNB_BLOCK - 1) / SDP_BNB_BLOCK;
            // rows strictly between the starts of two neighbouring blocks: at most floor(between) + 1, of which the ends'
            // own cells bring two along (A[q+1] of the lower end, A[q] of the upper one): how many more the pass reads
            const int extra = (bnb_between == bnb_between && bnb_between < (sdp_real)SDP_COL_N0)
                                  ? max((int)(bnb_between + 4 * BNB_DELTA) - 1, 0) : SDP_COL_N0;
            if (BNB_WIDE) rec[4 * n_blocks + 1] = (sdp_real)__int_as_float(extra);
            else rec[2 * n_blocks + 1] = (sdp_real)__hiloint2double(0, extra);
            // st[3]: what |X| may be at most for the pass's positions to stay within DELTA / 2 of the kernel's own
            // (8 u (|X| + |smin| + max |a|) k < DELTA / 2), or -1: no branch and bound in this column
            sdp_real *st = utab + SDP_COL_UTAB * SDP_COL_UTAB_N;
            const sdp_real cap = x_cap - bnb_amax;         // (x_cap = 2^30 / k - |smin| (2^13 / k for 4-byte reals): once per workgroup, by the caller)
            st[3] = (sorted && cap == cap && extra < SDP_COL_N0) ? cap : (sdp_real)-1;
        }
    }
#else
    for (int ci_ = (int)threadIdx.x - first; ci_ < n_tab; ci_ += count) {
        int ci = ci_;
        asm volatile("" : "+v"(ci));                        // (opaque to the optimiser: see the loop above)
        sdp_real u[SDP_NU], tab[SDP_COL_UTAB];
        sdp_controls_at(box, ci, u);
        sdp_model_utab(x, u, t, tab);
#pragma unroll
        for (int k = 0; k < SDP_COL_UTAB; ++k) utab[ci * SDP_COL_UTAB + k] = tab[k];
#if SDP_COL_SHORT
        a_lo = sdp_vmin(a_lo, tab[SDP_LEAN2_A_SLOT]);
        a_hi = sdp_vmax(a_hi, tab[SDP_LEAN2_A_SLOT]);
        fin = fin + fabs(tab[SDP_LEAN2_A_SLOT]);
        if (SDP_LEAN2_H_SLOT >= 0) {
            h_abs = sdp_vmax_abs(h_abs, tab[SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT]);
            fin = fin + fabs(tab[SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT]);
        }
#endif
    }
#endif  // SDP_COL_BNB
#if SDP_COL_SHORT
    h_abs = sdp_wave_max(h_abs);
    fin = sdp_wave_sum(fin);
    if (a_known) {
        // (what this buffer said two units ago about a stands; a poisoned entry stays poisoned)
        if ((int)threadIdx.x == first) {
            sdp_real *st = utab + SDP_COL_UTAB * SDP_COL_UTAB_N;
            const sdp_real before = st[2];
            st[2] = (count == 64 && fin < SDP_COL_FILTER_LIMIT && before == before) ? h_abs : (sdp_real)NAN;
        }
        return;
    }
    a_lo = sdp_wave_min(a_lo);
    a_hi = sdp_wave_max(a_hi);
    if ((int)threadIdx.x == first) {
        sdp_real *st = utab + SDP_COL_UTAB * SDP_COL_UTAB_N;
        st[0] = a_lo;
        st[1] = a_hi;
        st[2] = (count == 64 && fin < SDP_COL_FILTER_LIMIT) ? h_abs : (sdp_real)NAN;     // (NaN: every node the long way)
    }
#endif
}
#endif

// ---------------------------------------------------------------------------
// Wide first pass (SDP_COL_WIDE, 4-byte reals).  q0, lam0, oml0 = fl(1 - lam0) and g are the
// reference's 4-byte values (inputs of E); everything after them runs in 8-byte arithmetic:
//     F = g P + oml0 A[q0] + lam0 A[q0+1]        P, A accumulated in 8-byte reals
// so |F - R| is of the order of the 8-byte roundoff (~1e-15 of the terms) and the radius has to
// cover |E - R| alone, which it follows rounding by rounding:
//     |E - R| <= sum_w gamma_{n_w} |p_w| (|g| + |oml0 T[w][q0]| + |lam0 T[w][q0+1]|)
//             <= u' ( Gc |g| + |oml0| B[q0] + |lam0| B[q0+1] ),     B[r] = sum_w n_w |p_w| |T[w][r]|,  Gc = sum_w n_w |p_w|
// n_w = the roundings the term of point w passes through (sdp_col_wide_nw); u' = u (1 + 1e-3) absorbs
// gamma_n / (n u) <= 1 + 5e-6, the 4-byte roundings of B, Gc and of the bound itself (a few (W+8) u
// relative), and |F - R|; the bound carries `floor` so that the radius never drops below the smallest
// normal number (operations that underflow).  One radius per node: u' x the largest bound of its controls.
// (with a cost that depends on w: g64 = the expectation G accumulated in 8-byte reals, gabs = Gabs)
template <int AXIS>
SDP_DEV void sdp_col_wide_core(const sdp_real *ad, const SdpColFilter &f, const SdpLeadAxis &l,
                               sdp_real xn0, double g64, sdp_real gabs, double &F, sdp_real &bound, sdp_real &pmax)
{
    const sdp_real sn = AXIS == 2 ? xn0 : (AXIS == 1 ? (xn0 - l.smin) * l.rspan : (xn0 - l.smin) / l.span);
    const sdp_real p = sn * l.nm1;
    pmax = sdp_vmax_abs(pmax, p);
    int q0 = (int)p;                                        // (saturating conversion; NaN -> 0)
    asm("v_med3_i32 %0, %0, 0, %1" : "+v"(q0) : "s"(l.ordm2));
    const sdp_real lam0 = p - (sdp_real)q0;
    const sdp_real oml0 = (sdp_real)1 - lam0;
    const sdp_real *row = ad + 4 * q0;
    const double a0 = *(const double *)row, a1 = *(const double *)(row + 4);
    const sdp_real b0 = row[2], b1 = row[6];
#if SDP_COST_HAS_W
    F = g64 + fma((double)oml0, a0, (double)lam0 * a1);
    bound = gabs + fma(fabs(oml0), b0, fabs(lam0) * b1);
#else
    F = fma(g64, f.psum64, fma((double)oml0, a0, (double)lam0 * a1));
    bound = fma(gabs, f.gc, fma(fabs(oml0), b0, fabs(lam0) * b1));
#endif
}
template <int AXIS>
SDP_DEV void sdp_col_wide_eval(const sdp_real *ad, const SdpColFilter &f, const SdpLeadAxis &l,
                               const sdp_real *x, const sdp_real *u, sdp_real t, double &F, sdp_real &bound, sdp_real &pmax)
{
    const sdp_real xn0 = sdp_model_lead(x, u, (sdp_real)0, t);
#if SDP_COST_HAS_W
    double g64;
    sdp_real gabs;
    sdp_col_cost_expect<true>(f, x, u, t, g64, gabs);
    sdp_col_wide_core<AXIS>(ad, f, l, xn0, g64, gabs, F, bound, pmax);
#else
    const sdp_real g = sdp_model_cost(x, u, (sdp_real)0, t);
    sdp_col_wide_core<AXIS>(ad, f, l, xn0, (double)g, fabs(g), F, bound, pmax);
#endif
}
#if SDP_COL_UTAB
template <int AXIS>
SDP_DEV void sdp_col_wide_eval_tab(const sdp_real *ad, const sdp_real *utab, const SdpColFilter &f,
                                   const SdpLeadAxis &l, const sdp_real *x, int ci, sdp_real t,
                                   double &F, sdp_real &bound, sdp_real &pmax)
{
    sdp_real tab[SDP_COL_UTAB];
#pragma unroll
    for (int k = 0; k < SDP_COL_UTAB; ++k) tab[k] = utab[ci * SDP_COL_UTAB + k];
    const sdp_real g = sdp_model_cost_tab(x, tab, t);
    sdp_col_wide_core<AXIS>(ad, f, l, sdp_model_lead_tab(x, tab, t), (double)g, fabs(g), F, bound, pmax);
}
#endif

// What the first pass keeps of a node's controls: the two smallest F (and whose the smallest
// is), the largest S^ -- one radius cu * s_max then covers every control of the node -- and the
// sum of the S^, in which a NaN or an infinity of any control sticks (|F| <~ S^, and a NaN of
// F comes with a NaN or an infinity of S^: through D[r] -- sdp_col_filter_reduce --, |g|, |lam0|).
// With TOP2 also the third smallest F and whose the second is: when exactly two controls
// survive -- the usual near-tie, the lattice points either side of the continuous optimum --
// the second pass takes those two without looking at the lattice again.  Worth its five
// instructions per control where near-ties are common (4-byte reals: ~5 % of the nodes of the
// benchmark problem; 8-byte reals: none).
#ifndef SDP_COL_FILTER_TOP2
#define SDP_COL_FILTER_TOP2 -1   // -1: for 4-byte reals only
#endif
constexpr bool SDP_COL_TOP2 = SDP_COL_FILTER_TOP2 < 0 ? (sizeof(sdp_real) == 4 || SDP_COL_SHIFT) : SDP_COL_FILTER_TOP2 != 0;
struct SdpColBounds {
    sdp_fkey f1, f2, f3, s_sum;      // (wide: s_sum = the sum of the |F|, which catches NaN / infinite values)
    sdp_real s_max, p_max;           // (wide: s_max = the largest bound of a control)
    sdp_real b_max;                  // shifted lattice: the largest B' of the cells the controls fall in
    int i1, i2;
};
// 8-byte reals: the SUM of the S^ serves as the node's bound (no running maximum; a radius
// U times the necessary one, ~1e-12 relative, still leaves one survivor); 4-byte reals keep the maximum
constexpr bool SDP_COL_RADIUS_FROM_SUM = sizeof(sdp_real) == 8;
// one more value (of control ci) into the running two / three smallest
SDP_DEV void sdp_col_bounds_insert(SdpColBounds &b, sdp_fkey F, int ci)
{
    if (SDP_COL_TOP2) {
        b.f3 = sdp_vmin(b.f3, sdp_vmax(b.f2, F));
        const bool c1 = F < b.f1, c2 = F < b.f2;
        b.i2 = c1 ? b.i1 : (c2 ? ci : b.i2);
    }
    b.f2 = sdp_vmin(b.f2, sdp_vmax(b.f1, F));
    b.i1 = F < b.f1 ? ci : b.i1;
    b.f1 = sdp_vmin(b.f1, F);
}

// first pass over the controls [c_lo, c_hi) of one node.  PLAIN: a one-dimensional lattice that
// is an ordinary linspace (n > 1, step != 0): sdp_control_value without its rare branches, the
// last point (`stop`, numpy.linspace) taken out of the loop.
template <bool PLAIN, int AXIS>
SDP_DEV void sdp_col_filter_pass1(const sdp_real *ad_tab, const sdp_real *utab, const SdpColFilter &f,
                                  const SdpLeadAxis &l, const SdpBox &box, sdp_real *x, sdp_real t,
                                  int c_lo, int c_hi, SdpColBounds &b)
{
#if SDP_COL_UTAB
    if (SDP_COL_LEAN_ON || SDP_COL_WIDE_ON) {
        (void)box;
        auto one = [&](int ci) {
            sdp_fkey F;
            if (SDP_COL_WIDE_ON) {
                double Fw;
                sdp_real bound;
                sdp_col_wide_eval_tab<AXIS>(ad_tab, utab, f, l, x, ci, t, Fw, bound, b.p_max);
                b.s_max = sdp_vmax(b.s_max, bound);
                F = (sdp_fkey)Fw;
            } else {
                sdp_real Fl;
                sdp_col_lean_eval_tab<AXIS>(ad_tab, utab, f, l, x, ci, t, Fl, b.p_max, b.b_max);
                F = (sdp_fkey)Fl;
            }
            b.s_sum = b.s_sum + fabs(F);
            sdp_col_bounds_insert(b, F, ci);
        };
        constexpr int K = SDP_COL_FILTER_UNROLL;
        int ci = c_lo;
        for (; ci + K <= c_hi; ci += K) {
#pragma unroll
            for (int j = 0; j < K; ++j) one(ci + j);
        }
        for (; ci < c_hi; ++ci) one(ci);
        return;
    }
#endif
    (void)utab;
    auto eval = [&](int ci, const sdp_real *u) {
        sdp_real F, S;
        if (SDP_COL_WIDE_ON) {
            double Fw;
            sdp_real bound;
            sdp_col_wide_eval<AXIS>(ad_tab, f, l, x, u, t, Fw, bound, b.p_max);
            b.s_max = sdp_vmax(b.s_max, bound);
            b.s_sum = b.s_sum + (sdp_fkey)fabs(Fw);
            sdp_col_bounds_insert(b, (sdp_fkey)Fw, ci);
            return;
        }
        if (SDP_COL_LEAN_ON) {
            // (p_max holds the largest |lam0|, s_sum the sum of the |F|: see sdp_col_lean_eval)
            sdp_col_lean_eval<AXIS>(ad_tab, f, l, x, u, t, F, b.p_max, b.s_max, b.b_max);
            b.s_sum = b.s_sum + fabs(F);
        } else {
            sdp_col_filter_eval<AXIS>(ad_tab, f, l, x, u, t, F, S, b.p_max);
            b.s_sum = b.s_sum + S;
            if (!SDP_COL_RADIUS_FROM_SUM) b.s_max = sdp_vmax(b.s_max, S);
        }
        sdp_col_bounds_insert(b, F, ci);
    };
    auto one = [&](int ci) {
        sdp_real u[SDP_NU];
        if (PLAIN) u[0] = (sdp_real)ci * box.step[0] + box.lo[0];
        else sdp_controls_at(box, ci, u);
        eval(ci, u);
    };
    const int last = PLAIN ? box.n[0] - 1 : INT_MAX;
    const int c_main = min(c_hi, last);
    constexpr int K = SDP_COL_FILTER_UNROLL;
    int ci = c_lo;
    for (; ci + K <= c_main; ci += K) {
#pragma unroll
        for (int j = 0; j < K; ++j) one(ci + j);
    }
    for (; ci < c_main; ++ci) one(ci);
    if (PLAIN && c_hi > last && c_lo <= last) eval(last, box.hi);
}

#if SDP_COL_SHORT
// ---------------------------------------------------------------------------
// Short first passes (generated where x0' = X(x) +- a(u) and cost = K(x) +- h(u), a and h entries of the column's
// control table: codegen.short_pass_source).  The cell of a control as the reference computes it:
template <int AXIS>
SDP_DEV void sdp_lean2_cell(const SdpLeadAxis &l, sdp_real xn0, int &q0, sdp_real &lam0)
{
    const sdp_real sn = AXIS == 2 ? xn0 : (AXIS == 1 ? (xn0 - l.smin) * l.rspan : (xn0 - l.smin) / l.span);   // pyx:75
    const sdp_real p = sn * l.nm1;
    q0 = (int)p;                                            // (saturating conversion; NaN -> 0)
    asm("v_med3_i32 %0, %0, 0, %1" : "+v"(q0) : "s"(l.ordm2));   // pyx:78
    lam0 = p - (sdp_real)q0;                                // pyx:81
}
#endif
#if SDP_COL_WIDE2
// Short WIDE first pass (4-byte reals; the 8-byte one is in sdp_colres_kernel.h, where the reasoning is spelled out).
// Per control: the reference's cell (q0, lam0), then in 8-byte arithmetic on A[r] accumulated in 8-byte reals
//     F' = fma(+-h, P, fma(lam0, A[q0+1] - A[q0], A[q0]))          ~  R - K P*,   R = g P* + oml0 A0* + lam0 A1*
// with the index of the control in the low mantissa bits.  Nothing else per control: one bound for the node,
//     S = (Gc + Pcap) [ |K| + max |h| + (1 + 2 L) max |T| ]
// covers  Gc |g| + |oml0| B[q0] + |lam0| B[q0+1]  of sdp_col_wide_core (|g| <= (|K| + |h|)(1 + u), B[r] <= Gc max |T|,
// |oml0| + |lam0| <= (1 + 2 L)(1 + u)) and the two liberties F' takes with the reference's inputs: g = fl(K +- h)
// differs from K +- h by u (|K| + |h|), times P; and F' uses the exact 1 - lam0 where R has oml0 = fl(1 - lam0):
// u |1 - lam0| |A0*| <= u (1 + L) P max |T|.  L = max(1, |lam0| at the column's smallest and largest a) as in the
// 8-byte pass.  Radius u' (S + floor) + 2^(bits+1) 2^-52 S (the packing).  Values that are not finite: the same net.
template <int AXIS>
SDP_DEV double sdp_wide2_value(const sdp_real *ad, const sdp_real *utab, const SdpColFilter &f, const SdpLeadAxis &l,
                               sdp_real X, int ci)
{
    int q0;
    sdp_real lam0;
    sdp_lean2_cell<AXIS>(l, SDP_LEAN2_LEAD(X, utab[ci * SDP_COL_UTAB + SDP_LEAN2_A_SLOT]), q0, lam0);
    const double a0 = SDP_AD_A_CONST(ad, q0), a1 = SDP_AD_A_CONST(ad, q0 + 1);
    const double h = fma((double)lam0, a1 - a0, a0);
    if (SDP_LEAN2_H_SLOT < 0) return h;
    const sdp_real hv = utab[ci * SDP_COL_UTAB + (SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT)];
    return fma((double)(SDP_LEAN2_HNEG ? -hv : hv), f.psum64, h);
}
struct SdpShortBounds { double f1, f2, f3; };
SDP_DEV void sdp_short_insert(SdpShortBounds &b, double F)
{
    b.f3 = sdp_vmin(b.f3, sdp_vmax(b.f2, F));
    b.f2 = sdp_vmin(b.f2, sdp_vmax(b.f1, F));
    b.f1 = sdp_vmin(b.f1, F);
}
#ifndef SDP_WIDE2_GROUP
#define SDP_WIDE2_GROUP 4
#endif
// the three smallest F' over the controls [c_lo, c_hi) of one node, groups of controls in stages (all cells, all reads
// of the reduced table, all values: the reads of a group are in flight together)
template <int AXIS>
SDP_DEV void sdp_wide2_pass1(const sdp_real *ad, const sdp_real *utab, const SdpColFilter &f, const SdpLeadAxis &l,
                             sdp_real X, int c_lo, int c_hi, int mask, SdpShortBounds &b)
{
    auto insert = [&](double F, int ci) {
        sdp_short_insert(b, __hiloint2double(__double2hiint(F), (__double2loint(F) & ~mask) | ci));
    };
    constexpr int K = SDP_WIDE2_GROUP;
    constexpr int HS = SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT;
    int ci = c_lo;
    for (; ci + K <= c_hi; ci += K) {
        int q0[K];
        sdp_real av[K], hv[K], lam0[K];
        double a0[K], a1[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            av[j] = utab[(ci + j) * SDP_COL_UTAB + SDP_LEAN2_A_SLOT];
            hv[j] = SDP_LEAN2_H_SLOT < 0 ? (sdp_real)0 : utab[(ci + j) * SDP_COL_UTAB + HS];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) sdp_lean2_cell<AXIS>(l, SDP_LEAN2_LEAD(X, av[j]), q0[j], lam0[j]);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            a0[j] = SDP_AD_A_CONST(ad, q0[j]);
            a1[j] = SDP_AD_A_CONST(ad, q0[j] + 1);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double h = fma((double)lam0[j], a1[j] - a0[j], a0[j]);
            insert(SDP_LEAN2_H_SLOT < 0 ? h : fma((double)(SDP_LEAN2_HNEG ? -hv[j] : hv[j]), f.psum64, h), ci + j);
        }
    }
    for (; ci < c_hi; ++ci) insert(sdp_wide2_value<AXIS>(ad, utab, f, l, X, ci), ci);
}
#endif

#if SDP_COL_BNB
#ifndef SDP_SHORT_GROUP
#define SDP_SHORT_GROUP 4
#endif
#ifndef SDP_BNB_CHUNK
#define SDP_BNB_CHUNK (sizeof(SDP_REAL) == 8 ? 4 : 2)      // blocks whose bounds are evaluated together (registers: 8-byte values throughout)
#endif
// ---------------------------------------------------------------------------
// The short first pass as a certified BRANCH AND BOUND over blocks of controls (round 5).
// A node's filter value is F'(c) = hp_c + L(p_c): hp_c = +-h_c psum from the control table, L the piecewise-linear
// interpolant of the reduced table A along axis 0 (linear beyond the first and the last cell: the clamped cell with an
// unclamped lam0), p_c the position of control c.  Over a BLOCK B of controls the positions lie between those of the
// block's smallest and largest a (x0' = X +- a and every rounded step from it to the position are monotone in a:
// see the short first pass above), and a piecewise-linear function takes its minimum over an interval at an end of
// the interval or at a breakpoint inside it -- the breakpoints are the grid rows, where L = A[r].  So
//     LB(B) = min_B hp  +  min( L(p_lo), L(p_hi), A[r] for the rows r strictly between the two positions )
// bounds every F'(c), c in B, from below in real arithmetic; as evaluated, both sides are off by a few roundings of
// numbers bounded by S_node (|hp| <= Pcap max |h|, |L| <= (1 + 2 L_cap) D), together < 16 u S_node.
// A block with   LB(B) > f1 + 2 radius + 16 u S_node   (f1: the smallest F' seen so far, which only decreases) holds
// only controls with F' - radius > f1 + radius >= m_hi: by the radius' own theorem none of them is the reference's
// argmin or ties with it, exactly what the full pass concludes from F' itself (`single`, and the candidate test of the
// multi-survivor path, which re-evaluates F' of EVERY control and does not depend on this pass).  Its controls are
// never evaluated.  J, policy and index keep their bits: which controls are skipped is all that changes.
// Order: the block of the lane's GUESS first (the node's best control in the previous unit of this workgroup: the
// neighbouring column -- any guess is valid, a good one makes f1 tight at once), then the bounds of all blocks against
// that f1, then the blocks that survive, lane by lane (a lane reads ITS blocks' entries of the control table; lanes
// with nothing left idle through the trip).  On the benchmark problem 1.2 blocks of 8 per wave instead of 8.
// Values that are not finite: a NaN bound fails its comparison (the block is evaluated); the node-level tests of the
// caller (S_node, L) are unchanged.
// WIDE: the short wide first pass of 4-byte reals (positions and the control table in 4-byte reals, the reduced table and
// F' in 8-byte reals; the block records hold the start as a 4-byte real and the smallest +-h psum as an 8-byte one; the
// ends are moved out by 2^-8 rows there, which covers the 4-byte roundings of the kernel's own positions).
// c_lo: the controls [c_lo, n) of the lattice are this lane's (0: all of them; the branch and bound runs with one lane
// per node).  `insert` receives the packed F' of every control that is evaluated.
template <int AXIS, bool WIDE, typename INSERT>
SDP_DEV void sdp_short_bnb(const sdp_real *A, const sdp_real *utab, const SdpColFilter &f, const SdpLeadAxis &l,
                           sdp_real X, sdp_real k_rows, int c_lo, int n, int mask, double slack, int guess, INSERT &insert,
                           sdp_real &sdp_diag_cnt)
{
    (void)sdp_diag_cnt; (void)c_lo;
    constexpr int BS = SDP_BNB_BLOCK, NB = SDP_BNB_BLOCKS;
    constexpr int HS = SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT;
    static_assert(NB <= 64, "branch and bound: at most 64 blocks");
    const sdp_real *rec = utab + SDP_COL_UTAB * SDP_COL_UTAB_N + 4;
    const double psum = WIDE ? f.psum64 : (double)f.psum;
    auto pack = [&](double F, int ci) { return __hiloint2double(__double2hiint(F), (__double2loint(F) & ~mask) | ci); };
    auto row = [&](int q) -> double { return WIDE ? SDP_AD_A_CONST(A, q) : (double)A[q]; };
    auto start = [&](int b) -> sdp_real { return WIDE ? rec[4 * b] : rec[2 * b]; };
    auto least = [&](int b) -> double { return WIDE ? *(const double *)(rec + 4 * b + 2) : (double)rec[2 * b + 1]; };
    const int n_blocks = (n + BS - 1) / BS;
    const int g = guess < 0 ? (n >> 1) : min(guess, n - 1);              // (no guess yet: the middle of the lattice)
    // ---- stage 1: what the bounds need from the LDS -- the guess's entry of the control table, the blocks' records
    // ---- stage 2: the cells, the reads of the reduced table        (all of a stage's reads are in flight together)
    // ---- stage 3: F' of the guess (an upper bound of the node's smallest F'), the bounds, the blocks to evaluate
    const sdp_real ga = utab[g * SDP_COL_UTAB + SDP_LEAN2_A_SLOT];
    const sdp_real gh = SDP_LEAN2_H_SLOT < 0 ? (sdp_real)0 : utab[g * SDP_COL_UTAB + HS];
    const sdp_real pX = SDP_LEAN2_FORM == 2 ? -((X + l.smin) * k_rows) : (X - l.smin) * k_rows;
    const int extra = __builtin_amdgcn_readfirstlane(WIDE ? __float_as_int((float)rec[4 * n_blocks + 1]) : __double2loint((double)rec[2 * n_blocks + 1]));
    int gq;
    sdp_real glam;
    sdp_lean2_cell<AXIS>(l, SDP_LEAN2_LEAD(X, ga), gq, glam);
    const double gA0 = row(gq), gA1 = row(gq + 1);
    unsigned long long need = 0ull;
    double thresh = 0.0;
    constexpr int CB = NB < SDP_BNB_CHUNK ? NB : SDP_BNB_CHUNK;
    for (int b0 = 0; b0 < n_blocks; b0 += CB) {            // (uniform; one chunk on the benchmark lattice)
        int q[CB + 1];
        double P[CB + 1], hp[CB], Aq[CB + 1], Aq1[CB + 1], m[CB];
#pragma unroll
        for (int j = 0; j <= CB; ++j) {
            const int b = min(b0 + j, n_blocks);           // end j of the chunk: where block b0 + j starts, or the lattice ends
            P[j] = (double)(pX + start(b));
            if (j < CB) hp[j] = least(min(b, n_blocks - 1));
        }
#pragma unroll
        for (int j = 0; j <= CB; ++j) {
            q[j] = (int)P[j];                              // (saturating conversion; NaN -> 0)
            asm("v_med3_i32 %0, %0, 0, %1" : "+v"(q[j]) : "s"(l.ordm2));
        }
#pragma unroll
        for (int j = 0; j <= CB; ++j) {
            Aq[j] = row(q[j]);
            Aq1[j] = row(q[j] + 1);
        }
        if (b0 == 0) {
            // F' of the guess: its packed value is a first f1 (the block of the guess is evaluated like any other below)
            const double h = fma((double)glam, gA1 - gA0, gA0);
            const double Fg = pack(SDP_LEAN2_H_SLOT < 0 ? h : fma((double)(SDP_LEAN2_HNEG ? -gh : gh), psum, h), g);
            thresh = Fg + slack;
        }
#pragma unroll
        for (int j = 0; j <= CB; ++j) {
            const double lam = P[j] - (double)q[j];
            P[j] = fma(lam, Aq1[j] - Aq[j], Aq[j]);        // (P: now L at the end)
        }
#pragma unroll
        for (int j = 0; j < CB; ++j) m[j] = sdp_vmin(sdp_vmin(P[j], P[j + 1]), sdp_vmin(Aq1[j], Aq[j + 1]));
        for (int k = 0; k < extra; ++k) {                  // (one more row per block on the benchmark lattice)
            double more[CB];
#pragma unroll
            for (int j = 0; j < CB; ++j) more[j] = row(max(min(q[j] + 2 + k, q[j + 1] - 1), 0));
#pragma unroll
            for (int j = 0; j < CB; ++j) m[j] = sdp_vmin(m[j], more[j]);
        }
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            if (q[j + 1] - q[j] - 2 > extra) m[j] = -(double)INFINITY;      // (never seen; a count too small must not cost a row)
            const double lbv = hp[j] + m[j];
            // pruned only on a comparison that HOLDS (a NaN anywhere keeps the block); the guess's own block always stays
            if ((!(lbv > thresh) || b0 + j == g / BS) && b0 + j < n_blocks) need |= 1ull << (b0 + j);
        }
    }
#ifdef SDP_DIAG_BNB_COUNT                                  // diagnostic: J := blocks asked for (+ 100 x the guess's block)
    sdp_diag_cnt = (sdp_real)(__popcll(need) + 100 * (g / BS));
#endif
    // ---- the blocks that stay, lane by lane, groups of controls in stages (a lane reads ITS block's entries of the
    // control table; lanes with nothing left idle through the trip)
    while (__any(need != 0ull)) {
        const bool on = need != 0ull;
        const int b = on ? __ffsll((long long)need) - 1 : 0;
        need &= need - 1ull;
        constexpr int K = SDP_SHORT_GROUP;
        static_assert(BS % K == 0, "branch and bound: whole groups per block");
        for (int j0 = 0; j0 < BS; j0 += K) {
            int q0[K], ci[K];
            sdp_real av[K], lam0[K], hv[K];
            double a0[K], a1[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                ci[j] = min(b * BS + j0 + j, n - 1);       // (past the end: the last control again, not inserted)
                av[j] = utab[ci[j] * SDP_COL_UTAB + SDP_LEAN2_A_SLOT];
                hv[j] = SDP_LEAN2_H_SLOT < 0 ? (sdp_real)0 : utab[ci[j] * SDP_COL_UTAB + HS];
            }
#pragma unroll
            for (int j = 0; j < K; ++j) sdp_lean2_cell<AXIS>(l, SDP_LEAN2_LEAD(X, av[j]), q0[j], lam0[j]);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                a0[j] = row(q0[j]);
                a1[j] = row(q0[j] + 1);
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const double h = fma((double)lam0[j], a1[j] - a0[j], a0[j]);
                const double Fp = pack(SDP_LEAN2_H_SLOT < 0 ? h : fma((double)(SDP_LEAN2_HNEG ? -hv[j] : hv[j]), psum, h), ci[j]);
                // (a lane with nothing to evaluate in this trip, a control past the end: the largest finite number never wins)
                insert(on && b * BS + j0 + j < n ? Fp : 0x1.fffffffffffffp+1023);
            }
        }
    }
}
#endif  // SDP_COL_BNB

SDP_DEV void sdp_col_bounds_merge(SdpColBounds &b, int d)
{
    const sdp_fkey o_f1 = sdp_shfl_xor(b.f1, d), o_f2 = sdp_shfl_xor(b.f2, d);
    const sdp_real o_max = sdp_shfl_xor(b.s_max, d);
    const sdp_fkey o_sum = sdp_shfl_xor(b.s_sum, d);
    const int o_i1 = __shfl_xor(b.i1, d, 64);
    if (SDP_COL_TOP2) {
        // the other lane's three smallest, one after the other (its third cannot end up among
        // the two smallest of the union unless it ties with them -- and then f3 says so)
        const sdp_fkey o_f3 = sdp_shfl_xor(b.f3, d);
        const int o_i2 = __shfl_xor(b.i2, d, 64);
        sdp_col_bounds_insert(b, o_f1, o_i1);
        sdp_col_bounds_insert(b, o_f2, o_i2);
        sdp_col_bounds_insert(b, o_f3, INT_MAX);
    } else {
        const sdp_fkey mx = o_f1 > b.f1 ? o_f1 : b.f1;
        b.f2 = o_f2 < b.f2 ? o_f2 : b.f2;
        b.f2 = mx < b.f2 ? mx : b.f2;
        if (o_f1 < b.f1) { b.f1 = o_f1; b.i1 = o_i1; }     // (equal: f2 = f1, the node keeps both)
    }
    b.s_max = o_max > b.s_max ? o_max : b.s_max;
    b.s_sum = b.s_sum + o_sum;
    const sdp_real o_p = sdp_shfl_xor(b.p_max, d);
    b.p_max = o_p > b.p_max ? o_p : b.p_max;
    if (SDP_COL_SHIFT) {
        const sdp_real o_b = sdp_shfl_xor(b.b_max, d);
        b.b_max = o_b > b.b_max ? o_b : b.b_max;
    }
}


// Phase B with the filter for the nodes i_lo .. i_hi-1 of column `col`, by the `waves` waves
// that call it (this one is number `wave`); `ad_tab` = the (A[r], D[r]) pairs, s.T the table.
// A wave takes 64 / chunks consecutive nodes; the lanes l, l + npw, l + 2 npw, .. of a node
// share its control lattice in `chunks` consecutive ranges and meet through lane shuffles.
// Lanes past the end of the unit repeat its last node (they must stay active for the
// shuffles) and store nothing.
SDP_DEV void sdp_col_filter_nodes(const SdpSweepArgs &a, const SdpGrid<sdp_real, SDP_DT> &tg,
                                  const SdpColShared &s, const SdpColWeights &wts, const SdpLeadAxis &lead,
                                  const SdpColFilter &filt, int axis_mode, const sdp_real *ad_tab,
                                  const sdp_real *utab, sdp_real dcol,
                                  int64_t col, int i_lo, int i_hi, int wave, int waves,
                                  sdp_real *x, sdp_real t, SdpColDiag &diag,
                                  const SdpBox *box_c, int i_pre, sdp_real x0_pre
#if SDP_COL_SHIFT
                                  , const SdpColShiftCol &shc
#endif
                                  , int *guess_p = nullptr
                                  )
{
    (void)guess_p;
    constexpr int N0 = SDP_COL_N0;
    const int lane = threadIdx.x & 63;
    // the axis the FIRST pass locates its positions on: axis 0, or the shifted lattice of this column
    SdpLeadAxis lead1 = lead;
#if SDP_COL_SHIFT
    lead1.koff = (sdp_real)shc.kmin;
    lead1.ordm2 = shc.ok ? shc.rows - 2 : 0;
#endif
    const sdp_real *__restrict__ axis0 = (const sdp_real *)a.axes + a.axis_off[0];
    (void)diag;
    const int n_nodes = i_hi - i_lo;
    const int groups = (n_nodes + 63) >> 6;
    int chunks = groups < waves ? waves / groups : 1;
    chunks = 1 << (31 - __builtin_clz(chunks < 64 ? chunks : 64));     // power of two <= 64
    const int npw = 64 / chunks;                                        // nodes per wave
    const int items = (n_nodes + npw - 1) / npw;
    // the issue-bound phase goes first: the co-resident workgroup's table build mostly
    // waits for memory and fills the gaps (measured: 2.79 -> 2.57 ms)
    __builtin_amdgcn_s_setprio(SDP_COL_B_PRIO);
    for (int item = wave; item < items; item += waves) {
        const int chunk = lane / npw;
        const int i_raw = i_lo + item * npw + (lane - chunk * npw);
        const bool live = i_raw < i_hi;
        const int i = live ? i_raw : i_hi - 1;
        const int64_t node = col * N0 + i;
        SdpBox box;
        x[0] = i == i_pre ? x0_pre : axis0[i];              // (i_pre: the node whose coordinate the caller holds)
        if (box_c) box = *box_c;                            // (constant box, fetched once per workgroup)
        else sdp_load_box(a, node, box);
        const int c_lo = (int)((int64_t)box.total * chunk / chunks);
        const int c_hi = (int)((int64_t)box.total * (chunk + 1) / chunks);
        // pass 1: bounds of every control of this lane's range
        SDP_COL_MARK(diag.m1);
#if SDP_COL_WIDE2
        static_assert(SDP_COL_WIDE_ON && SDP_COL_TOP2 && !SDP_COST_HAS_W && !SDP_COL_SHIFT && SDP_COL_UTAB,
                      "short first pass of the full-table kernel: the wide form of 4-byte reals");
        struct { double f1, f2, f3; int i1, i2; } bd;
        bool bad;
        sdp_fkey radius;
        const sdp_real X = sdp_model_lead_x(x, t);
        {
            const sdp_real *ust = utab + SDP_COL_UTAB * SDP_COL_UTAB_N;          // a_lo, a_hi, max |h| (or NaN)
            const sdp_real K = sdp_model_cost_x(x, t);
            const int bits = 32 - __clz(max(box.total - 1, 1));
            const int mask = (1 << bits) - 1;
            SdpShortBounds sb;
            sb.f1 = sb.f2 = sb.f3 = INFINITY;
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
            // (dcol: the largest |T| of the column, raw -- sdp_col_filter_reduce)
            const sdp_real s_node = (filt.gc + filt.pcap) * ((fabs(K) + ust[2]) + ((sdp_real)1 + (sdp_real)2 * l_cap) * dcol);
            bad = !filt.ok || !(s_node < SDP_COL_FILTER_LIMIT) || !(fabs(lam_lo) + fabs(lam_hi) < (sdp_real)1073741824.0) ||
                  bits > 24 || box.total > SDP_COL_UTAB_N;
            radius = (sdp_fkey)(SDP_COL_FILTER_SCALE) * (sdp_fkey)(1.001 * 0x1p-24) * ((sdp_fkey)s_node + (sdp_fkey)filt.floor) +
                     ldexp((sdp_fkey)s_node, bits - 51);
            bool full_pass = true;
#if SDP_COL_BNB
            // one lane per node, the column's blocks in order, |X| small enough for the bounds' positions (ust[3]:
            // sdp_col_phase_u): a block is skipped when its lower bound exceeds f1 by more than 2 radius (+ 2^-40 S for the
            // bound's own 8-byte roundings); a wave with a node that does not qualify takes the full pass
            if (__all(chunks == 1 && fabs(X) < ust[3] && !bad)) {
                full_pass = false;
                sdp_real cnt_unused = (sdp_real)0;
                const double slack = fma(2.0, (double)radius, 0x1p-40 * (double)s_node);
                auto ins = [&](double Fq) { sdp_short_insert(sb, Fq); };
                if (axis_mode == 2) sdp_short_bnb<2, true>(ad_tab, utab, filt, lead, X, filt.k_rows, c_lo, c_hi, mask, slack, *guess_p, ins, cnt_unused);
                else if (axis_mode == 1) sdp_short_bnb<1, true>(ad_tab, utab, filt, lead, X, filt.k_rows, c_lo, c_hi, mask, slack, *guess_p, ins, cnt_unused);
                else sdp_short_bnb<0, true>(ad_tab, utab, filt, lead, X, filt.k_rows, c_lo, c_hi, mask, slack, *guess_p, ins, cnt_unused);
            }
#endif
            if (full_pass) {
                if (axis_mode == 2) sdp_wide2_pass1<2>(ad_tab, utab, filt, lead, X, c_lo, c_hi, mask, sb);
                else if (axis_mode == 1) sdp_wide2_pass1<1>(ad_tab, utab, filt, lead, X, c_lo, c_hi, mask, sb);
                else sdp_wide2_pass1<0>(ad_tab, utab, filt, lead, X, c_lo, c_hi, mask, sb);
            }
            for (int d = npw; d < 64; d <<= 1) {           // the lanes that share the node (ranges of its lattice)
                const double o1 = sdp_shfl_xor(sb.f1, d), o2 = sdp_shfl_xor(sb.f2, d), o3 = sdp_shfl_xor(sb.f3, d);
                sdp_short_insert(sb, o1);
                sdp_short_insert(sb, o2);
                sdp_short_insert(sb, o3);
            }
            bd.f1 = sb.f1; bd.f2 = sb.f2; bd.f3 = sb.f3;
            bd.i1 = sb.f1 < (double)INFINITY ? (__double2loint(sb.f1) & mask) : INT_MAX;
            bd.i2 = sb.f2 < (double)INFINITY ? (__double2loint(sb.f2) & mask) : INT_MAX;
        }
        SDP_COL_MARK(diag.m2);
#else
        SdpColBounds bd;
        bd.f1 = bd.f2 = bd.f3 = INFINITY;
        bd.s_max = bd.s_sum = bd.p_max = bd.b_max = (sdp_real)0;
        bd.i1 = bd.i2 = INT_MAX;
        const bool plain = SDP_NU == 1 && box.n[0] > 1 && box.step[0] != (sdp_real)0;
#ifdef SDP_DIAG_NO_PASS1
        if (t == (sdp_real)123.456)
#endif
        if (__all(plain)) {
            if (axis_mode == 2) sdp_col_filter_pass1<true, 2>(ad_tab, utab, filt, lead1, box, x, t, c_lo, c_hi, bd);
            else if (axis_mode == 1) sdp_col_filter_pass1<true, 1>(ad_tab, utab, filt, lead1, box, x, t, c_lo, c_hi, bd);
            else sdp_col_filter_pass1<true, 0>(ad_tab, utab, filt, lead1, box, x, t, c_lo, c_hi, bd);
        } else {
            if (axis_mode == 2) sdp_col_filter_pass1<false, 2>(ad_tab, utab, filt, lead1, box, x, t, c_lo, c_hi, bd);
            else if (axis_mode == 1) sdp_col_filter_pass1<false, 1>(ad_tab, utab, filt, lead1, box, x, t, c_lo, c_hi, bd);
            else sdp_col_filter_pass1<false, 0>(ad_tab, utab, filt, lead1, box, x, t, c_lo, c_hi, bd);
        }
        for (int d = npw; d < 64; d <<= 1) sdp_col_bounds_merge(bd, d);
        // pass 2: the reference's operations on the survivors
        SDP_COL_MARK(diag.m2);
#if SDP_STAMP == 2
        diag.tp1 += diag.m2 - diag.m1;
#endif
        bool bad;
        sdp_fkey radius;
        if (SDP_COL_WIDE_ON) {
            // (s_sum: a NaN or an infinity of any F sticks in it; s_max: the largest bound -- infinite when a
            // value is, and < 2^100 means that nothing overflows on the reference's 4-byte path)
            // (dcol: the largest |T| of the column, raw -- the bounds weigh |T| and |g| with the weights)
            bad = !filt.ok || !(bd.s_sum == bd.s_sum) || !(bd.s_max < SDP_COL_FILTER_LIMIT * filt.glimit) ||
                  !(bd.p_max < (sdp_real)2147483648.0) ||
                  !(((sdp_real)1 + (sdp_real)2 * bd.p_max) * dcol < SDP_COL_FILTER_LIMIT);
            radius = (sdp_fkey)(SDP_COL_FILTER_SCALE) * (sdp_fkey)(1.001 * 0x1p-24) * ((sdp_fkey)bd.s_max + (sdp_fkey)filt.floor);
        } else if (SDP_COL_LEAN_ON) {
#if SDP_COL_SHIFT
            // H D, H = (1 + 2 L + 2 Lc) (3 + Es): see sdp_col_shift_reduce
            const sdp_real h_cap = (((sdp_real)1 + (sdp_real)2 * (bd.p_max + shc.lc)) * ((sdp_real)3 + shc.es)) * dcol;
#else
            const sdp_real h_cap = ((sdp_real)1 + (sdp_real)2 * bd.p_max) * dcol;        // (1 + 2L) D
#endif
#if SDP_COST_HAS_W
            // (the largest Gabs of the node's controls stands where |g| Pcap stood; a NaN shows in the sum of the |F|)
            const sdp_real s_node = bd.s_sum == bd.s_sum ? (sdp_real)bd.s_max + h_cap : (sdp_real)NAN;
#else
            const sdp_real s_node = fma(filt.ratio, bd.s_sum + h_cap, h_cap);
#endif
#if SDP_COL_SHIFT
            bad = !filt.ok || !shc.ok || !(s_node < SDP_COL_FILTER_LIMIT) || !(bd.p_max + shc.lc < (sdp_real)1073741824.0);
            radius = fma(filt.cu, s_node, (sdp_real)(SDP_COL_FILTER_SCALE) * bd.b_max);
#else
            bad = !filt.ok || !(s_node < SDP_COL_FILTER_LIMIT) || !(bd.p_max < (sdp_real)1073741824.0);
            radius = filt.cu * s_node;
#endif
        } else {
            bad = !filt.ok || !(bd.s_sum < SDP_COL_FILTER_LIMIT) || !(bd.p_max < (sdp_real)2147483648.0);
            radius = filt.cu * (SDP_COL_RADIUS_FROM_SUM ? (sdp_real)bd.s_sum : bd.s_max);
        }
#endif  // SDP_COL_WIDE2
        const sdp_fkey m_hi = bd.f1 + radius;                  // >= the minimum of E over the node
        const bool single = !bad && bd.i1 != INT_MAX && bd.f2 - radius > m_hi;
        // exactly two survivors (TOP2): the lanes of the node take one each (a lane alone takes both)
        const bool pair = SDP_COL_TOP2 && !bad && !single && bd.i2 != INT_MAX && bd.f3 - radius > m_hi;
        const int p_lo = min(bd.i1, bd.i2), p_hi = max(bd.i1, bd.i2);
        int first = c_lo, last = c_hi, stride = 1;
        if (single) { first = bd.i1; last = bd.i1 + 1; }
        if (pair) {
            if (chunks == 1) { first = p_lo; last = p_hi + 1; stride = max(p_hi - p_lo, 1); }
            else { first = (chunk & 1) ? p_hi : p_lo; last = chunk < 2 ? first + 1 : first; }
        }
        sdp_real best = INFINITY;
        int ibest = INT_MAX;
#ifdef SDP_DIAG_NO_PASS2
        if (bd.f1 == (sdp_real)123.456)
#endif
        for (int ci = first; ci < last; ci += stride) {
            sdp_real u[1][SDP_NU], jc[1];
            sdp_controls_at(box, ci, u[0]);
            bool cand = single || pair || bad;
#if SDP_COL_WIDE2
            if (!cand) {
                const double Fw = lead.pow2 ? sdp_wide2_value<1>(ad_tab, utab, filt, lead, X, ci)
                                            : sdp_wide2_value<0>(ad_tab, utab, filt, lead, X, ci);
                cand = !((sdp_fkey)Fw - radius > m_hi);
            }
#else
            if (!cand && SDP_COL_WIDE_ON) {
                double Fw;
                sdp_real bnd, pm = (sdp_real)0;
                if (lead.pow2) sdp_col_wide_eval<1>(ad_tab, filt, lead, x, u[0], t, Fw, bnd, pm);
                else sdp_col_wide_eval<0>(ad_tab, filt, lead, x, u[0], t, Fw, bnd, pm);
                cand = !((sdp_fkey)Fw - radius > m_hi);
            } else if (!cand) {
                sdp_real F, S;
                sdp_real pm = (sdp_real)0;
                if (SDP_COL_LEAN_ON) {
                    sdp_real gm = (sdp_real)0, bm = (sdp_real)0;
                    if (lead.pow2) sdp_col_lean_eval<1>(ad_tab, filt, lead1, x, u[0], t, F, pm, gm, bm);
                    else sdp_col_lean_eval<0>(ad_tab, filt, lead1, x, u[0], t, F, pm, gm, bm);
                } else if (lead.pow2) sdp_col_filter_eval<1>(ad_tab, filt, lead, x, u[0], t, F, S, pm);
                else sdp_col_filter_eval<0>(ad_tab, filt, lead, x, u[0], t, F, S, pm);
                cand = !(F - radius > m_hi);
            }
#endif
            if (cand) {
#if SDP_STAMP == 3
                if (live) ++diag.n_exact;
#endif
                sdp_col_expected_cost<1>(a, tg, s, wts, lead, x, u, t, jc);
                if (ibest == INT_MAX || sdp_better_seq(jc[0], best)) { best = jc[0]; ibest = ci; }
            }
        }
        for (int d = npw; d < 64; d <<= 1) {       // ranges are in lattice order: lower index wins ties
            const sdp_real ov = sdp_shfl_xor(best, d);
            const int oi = __shfl_xor(ibest, d, 64);
            if (oi != INT_MAX && (ibest == INT_MAX || sdp_better_idx(ov, oi, best, ibest))) { best = ov; ibest = oi; }
        }
#if SDP_STAMP == 3
        if (live && chunk == 0) { ++diag.n_all; if (!single && !pair) ++diag.n_slow; }
#endif
        if (live && chunk == 0) sdp_col_store(a, node, box, best, ibest);
        if (guess_p && live && ibest != INT_MAX) *guess_p = ibest;      // (branch and bound: where the next node of this lane starts)
#if SDP_STAMP == 2
        diag.tp2 += __builtin_amdgcn_s_memtime() - diag.m2;
#endif
    }
    __builtin_amdgcn_s_setprio(0);
}
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
    0, 0, 0};
}

#else   // SDP_D < 2: no column kernels; the unit is a node-order one after all
extern "C" {
__constant__ int32_t sdp_meta[SDP_META_WORDS] = {
    SDP_META_MAGIC, (int32_t)sizeof(sdp_real), SDP_D, SDP_NU, SDP_HAS_W, 0, 0, 1, SDP_META_F_PEER_STORES, 0, 0, 256, 0, 0, 0, 0};
}
#endif  // SDP_D >= 2
