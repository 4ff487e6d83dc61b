// sdp_colfilter_kernel.h -- the certified expectation-first filter of the column kernels (included by
// sdp_colfilter_kernel.h where SDP_COL_FILTER is set, after the table machinery it builds on): the reduced table and the
// column's bound (sdp_col_filter_reduce), the shifted lattice for a perturbation that reaches the stock (SDP_COL_SHIFT),
// the lean / tabulated / wide first passes, the short first passes of the additive shape and their certified branch and
// bound (SDP_COL_BNB), and phase B with the filter for the whole-table forms (sdp_col_filter_nodes).  The three forms of
// sdp_sweep_col (sdp_colfull_kernel.h, sdp_colres_kernel.h, sdp_colu_kernel.h) call into it.  docs/NOTEBOOK.md 3.1b - 3.1f.
#pragma once

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
#if SDP_COL_SHIFT
// .. and on the shifted lattice (`l`: the lattice -- koff its first position, ordm2 its rows - 2), as sdp_col_lean_core
// locates it: inside the lattice pk >= 0 and the truncation is the floor
template <int AXIS>
SDP_DEV void sdp_lean2s_cell(const SdpLeadAxis &l, sdp_real xn0, int &q0, sdp_real &lam0)
{
    const sdp_real sn = AXIS == 2 ? xn0 : (AXIS == 1 ? (xn0 - l.smin) * l.rspan : (xn0 - l.smin) / l.span);
    const sdp_real pk = sn * l.nm1 - l.koff;
    q0 = (int)pk;                                           // (saturating conversion; NaN -> 0)
    asm("v_med3_i32 %0, %0, 0, %1" : "+v"(q0) : "s"(l.ordm2));
    lam0 = pk - (sdp_real)q0;
}
#endif
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
                           sdp_real &sdp_diag_cnt, sdp_real *b_seen = nullptr)
{
    // SDP_COL_SHIFT (round 6): `A` holds the (A', B') pairs of the shifted lattice, `l` is the lattice.  A control's F' is off
    // from the real number it stands for by its OWN cell's chord bound B'[q0] on top of the rounding radius, so a block is
    // ruled out against the guess g only when   LB - max B'[the rows the block's positions can fall in]  >  F'_g + B'[q0(g)] + slack
    // (slack: twice the ROUNDING radius + the bound's own roundings): then E_c >= F'_c - rho_c > F'_g + rho_g >= E_g >= the
    // node's minimum for every control c of the block.  The rows are those the bound reads anyway (q_b, q_b + 1, the rows
    // between, q_b+1): B' rides along in the same 16-byte reads.  *b_seen: the largest B' among the controls evaluated.
    (void)sdp_diag_cnt; (void)c_lo; (void)b_seen;
    constexpr bool SHIFT = SDP_COL_SHIFT != 0;
    static_assert(!(SHIFT && WIDE), "branch and bound on the shifted lattice: 8-byte reals");
    constexpr double BSCALE = (double)(SDP_COL_FILTER_SCALE);
    constexpr int BS = SDP_BNB_BLOCK, NB = SDP_BNB_BLOCKS;
    constexpr int HS = SDP_LEAN2_H_SLOT < 0 ? 0 : SDP_LEAN2_H_SLOT;
    static_assert(NB <= 64, "branch and bound: at most 64 blocks");
    const sdp_real *rec = utab + SDP_COL_UTAB * SDP_COL_UTAB_N + 4;
    const double psum = WIDE ? f.psum64 : (double)f.psum;
    auto pack = [&](double F, int ci) { return __hiloint2double(__double2hiint(F), (__double2loint(F) & ~mask) | ci); };
    auto row = [&](int q) -> double { return WIDE ? SDP_AD_A_CONST(A, q) : (double)A[SHIFT ? 2 * q : q]; };
    auto brow = [&](int q) -> double { return SHIFT ? (double)A[2 * q + 1] : 0.0; };
    auto cell = [&](sdp_real xn0, int &q0, sdp_real &lam0) {
#if SDP_COL_SHIFT
        sdp_lean2s_cell<AXIS>(l, xn0, q0, lam0);
#else
        sdp_lean2_cell<AXIS>(l, xn0, q0, lam0);
#endif
    };
    auto start = [&](int b) -> sdp_real { return WIDE ? rec[4 * b] : rec[2 * b]; };
    auto least = [&](int b) -> double { return WIDE ? *(const double *)(rec + 4 * b + 2) : (double)rec[2 * b + 1]; };
    const int n_blocks = (n + BS - 1) / BS;
    const int g = guess < 0 ? (n >> 1) : min(guess, n - 1);              // (no guess yet: the middle of the lattice)
    // ---- stage 1: what the bounds need from the LDS -- the guess's entry of the control table, the blocks' records
    // ---- stage 2: the cells, the reads of the reduced table        (all of a stage's reads are in flight together)
    // ---- stage 3: F' of the guess (an upper bound of the node's smallest F'), the bounds, the blocks to evaluate
    const sdp_real ga = utab[g * SDP_COL_UTAB + SDP_LEAN2_A_SLOT];
    const sdp_real gh = SDP_LEAN2_H_SLOT < 0 ? (sdp_real)0 : utab[g * SDP_COL_UTAB + HS];
    const sdp_real pX = (SDP_LEAN2_FORM == 2 ? -((X + l.smin) * k_rows) : (X - l.smin) * k_rows) - (SHIFT ? l.koff : (sdp_real)0);
    const int extra = __builtin_amdgcn_readfirstlane(WIDE ? __float_as_int((float)rec[4 * n_blocks + 1]) : __double2loint((double)rec[2 * n_blocks + 1]));
    int gq;
    sdp_real glam;
    cell(SDP_LEAN2_LEAD(X, ga), gq, glam);
    const double gA0 = row(gq), gA1 = row(gq + 1), gB = brow(gq);
    unsigned long long need = 0ull;
    double thresh = 0.0;
    constexpr int CB = NB < SDP_BNB_CHUNK ? NB : SDP_BNB_CHUNK;
    for (int b0 = 0; b0 < n_blocks; b0 += CB) {            // (uniform; one chunk on the benchmark lattice)
        int q[CB + 1];
        double P[CB + 1], hp[CB], Aq[CB + 1], Aq1[CB + 1], m[CB], Bq[SHIFT ? CB + 1 : 1], bm[SHIFT ? CB : 1];
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
            if (SHIFT) Bq[j] = brow(q[j]);
        }
        if (SHIFT) {
#pragma unroll
            for (int j = 0; j < CB; ++j) bm[j] = sdp_vmax(sdp_vmax(Bq[j], brow(q[j] + 1)), Bq[j + 1]);
        }
        if (b0 == 0) {
            // F' of the guess: its packed value is a first f1 (the block of the guess is evaluated like any other below)
            const double h = fma((double)glam, gA1 - gA0, gA0);
            const double Fg = pack(SDP_LEAN2_H_SLOT < 0 ? h : fma((double)(SDP_LEAN2_HNEG ? -gh : gh), psum, h), g);
            thresh = (SHIFT ? fma(BSCALE, gB, Fg) : Fg) + slack;
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
            if (SHIFT) {
#pragma unroll
                for (int j = 0; j < CB; ++j) bm[j] = sdp_vmax(bm[j], brow(max(min(q[j] + 2 + k, q[j + 1] - 1), 0)));
            }
        }
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            if (q[j + 1] - q[j] - 2 > extra) m[j] = -(double)INFINITY;      // (never seen; a count too small must not cost a row)
            const double lbv = SHIFT ? fma(-BSCALE, bm[j], hp[j] + m[j]) : hp[j] + m[j];
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
            double a0[K], a1[K], bq[SHIFT ? K : 1];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                ci[j] = min(b * BS + j0 + j, n - 1);       // (past the end: the last control again, not inserted)
                av[j] = utab[ci[j] * SDP_COL_UTAB + SDP_LEAN2_A_SLOT];
                hv[j] = SDP_LEAN2_H_SLOT < 0 ? (sdp_real)0 : utab[ci[j] * SDP_COL_UTAB + HS];
            }
#pragma unroll
            for (int j = 0; j < K; ++j) cell(SDP_LEAN2_LEAD(X, av[j]), q0[j], lam0[j]);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                a0[j] = row(q0[j]);
                a1[j] = row(q0[j] + 1);
                if (SHIFT) bq[j] = brow(q0[j]);
            }
            if (SHIFT) {
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (on && b * BS + j0 + j < n) *b_seen = sdp_vmax(*b_seen, (sdp_real)bq[j]);
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
