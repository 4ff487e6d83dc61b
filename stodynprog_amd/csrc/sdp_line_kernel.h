// sdp_line_kernel.h -- ONE state variable whose perturbation enters the next state through final sums,
//     x' = fl(.. fl(fl(a(x, u) +- b_1(w)) +- b_2(w)) ..)          (the shop inventory `x + u - w`,
// reference doc/example_inventory.py:31-33; the stocks of examples/01 Deterministic storage control), with a cost that
// does not see the perturbation: the CERTIFIED FILTER ON THE SHIFTED LATTICE of sdp_colfilter_kernel.h (SDP_COL_SHIFT;
// docs/NOTEBOOK.md section 3.1d) for d = 1, where the "table" T[w][r] of the column kernel IS the value array -- no trailing
// axes, nothing to interpolate before axis 0 -- so nothing has to sit in LDS at all.
// Included by sdp_sweep_kernel.h in place of its own sdp_sweep when the generated unit defines SDP_LINE; the unit brings
//     sdp_model_lead_a(x, u, t)              a, with the reference's own operations
//     sdp_model_lead_b(x, w, t, b, babs)     the signed sum B of the b_i in the reference's order, and sum |b_i|
//     sdp_model_cost(x, u, w, t)             the cost (w unused)
//     SDP_LINE_CHAIN m, sdp_model_lead_aabs  (a chain of sums in another nesting, regrouped by the tracer: see
//                                             SDP_COL_SHIFT_CHAIN in sdp_colfilter_kernel.h)
//     SDP_LINE_W                             perturbation points (compile time: the second pass deals them to lanes)
//
// In real arithmetic the reference's value of a control is  R(u) = g P + G(pa(u)),  G(s) = sum_w p_w V~(s + pb_w),
// V~ the reference's interpolant of the value array along the axis (piecewise linear, linear beyond both ends:
// pyx:75-88 clamps the cell, not lam), pa = (a - smin) c, pb_w = B_w c, c = (N-1)/span: ONE function G for the whole
// problem.  Two kernels per sweep:
//   sdp_lead_reduce   (the name the host library launches before a sweep whose code object has SDP_META_F_LEAD)
//       per whole position k = kmin .. kmin + rows - 1 of the lattice the perturbation points have shifted:
//       A'[k] = G(k); the chord bound B'[k] = sum_w |p_w| f_w (1 - f_w) |d2_{w,k}| of the cell above k; C'[k], what the
//       roundings around k are proportional to (local magnitude x local extrapolation) -- triplets in SdpSweepArgs.aux_a;
//       and the steepest step of the value array between neighbouring rows into aux_vmax.
//   sdp_sweep
//       first level: F = fma(g, psum, lerp(A', pa - kmin)) per control with the half-width e of the interval that holds
//       the reference's value -- PER CONTROL: the chord bound of its own cell plus its own roundings (sdp_line_value);
//       a control with F - e > min (F + e) is not the reference's argmin nor tied with it.  Second level, for the nodes
//       that keep more than two controls: G itself in fused arithmetic on those (no chord).  The survivors (one, or two)
//       are evaluated with the reference's operations: the workgroup's threads take a (node, perturbation point) each
//       (dyn, cell, lerp of V, cost, weight: stodynprog.py:674-677), the expectation is then summed in w order by one
//       chain of additions (stodynprog.py:681).  Anything else -- true near-ties, NaN / infinite values, a lattice that
//       does not fit -- evaluates its candidates the long way in lattice order (sdp_expected_cost).
// J, policy and index carry the reference's bits: only WHICH controls are skipped depends on the filter.
// The bounds follow the header comment of sdp_col_phase_shift (same operations, T = V) with its column-wide factors made
// local (sdp_line_value); checked in exact rational arithmetic by tests/test_filter_bound_exact.py (test_the_line_*).
#pragma once

static_assert(SDP_D == 1 && SDP_NU >= 1 && SDP_HAS_W && sizeof(sdp_real) == 8, "sdp_line_kernel.h: one state variable, 8-byte reals");
#ifndef SDP_LINE_CHAIN
#define SDP_LINE_CHAIN 0
#endif
#ifndef SDP_LINE_FILTER_SCALE
#define SDP_LINE_FILTER_SCALE 1.0      // test knob: multiplies the radius (any value below 1 voids the guarantee)
#endif
#ifndef SDP_LINE_TOP2
#define SDP_LINE_TOP2 1
#endif
// diagnostic builds (SDP_LINE_DIAG; SdpSweepArgs.stamps enabled): how the nodes were decided -- words 0 .. 8: one survivor /
// two at the first level, undecided there, one / two at the second level, candidates the long way (not bad), bad nodes,
// second-level evaluations, long-way evaluations
#ifdef SDP_LINE_DIAG
#define SDP_LN_COUNT(k, v) do { if (a.stamps) atomicAdd(a.stamps + (k), (unsigned long long)(v)); } while (0)
#else
#define SDP_LN_COUNT(k, v) do { } while (0)
#endif

SDP_DEV double sdp_ln_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
SDP_DEV double sdp_ln_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
SDP_DEV double sdp_ln_max_abs(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
extern "C" __device__ double __ockl_wfred_max_f64(double);

constexpr double SDP_LN_TINY = 2.2250738585072014e-308, SDP_LN_LIMIT = 0x1p1000, SDP_LN_EPS = 0x1p-52;     // EPS = 2 u
constexpr int SDP_LN_W = SDP_LINE_W;
static_assert(SDP_LN_W >= 1 && SDP_LN_W <= 1024, "sdp_line_kernel.h: 1 .. 1024 perturbation points");

// what every workgroup of both kernels works out for itself from the W perturbation points (LDS)
struct SdpLineLds {
    int sh_q[SDP_LN_W];          // whole part of the shift of point w, in rows
    double sh_f[SDP_LN_W];       // its fraction in [0, 1)
    double sh_c[SDP_LN_W];       // |p_w| f (1 - f)
    double sh_p[SDP_LN_W];       // p_w
    int sh_k[4];                 // max q, max -q, "not usable", max rows of sum |b_i|
    double red[8];               // weights: psum, sum |p|  (one wave)
};
struct SdpLineAxis {
    double smin, span, rspan, nm1, koff, lc;
    int ordm2, n;
    bool pow2;
};
struct SdpLineFilter { double psum, pcap, cu, floor, ratio; bool ok; };
struct SdpLineCol {
    int kmin, rows, flmin, flmax;
    bool ok;
    double lc, es;
};

SDP_DEV void sdp_line_setup(const SdpSweepArgs &a, SdpLineLds &m, SdpLineAxis &l, SdpLineFilter &f, SdpLineCol &c)
{
    const double *__restrict__ axes = (const double *)a.axes;
    const double *__restrict__ wgrid = (const double *)a.wgrid;
    const double *__restrict__ p = (const double *)a.proba;
    sdp_trap_unless(a.W == SDP_LN_W);
    l.n = a.orders[0];
    l.smin = axes[a.axis_off[0]];
    l.span = axes[a.axis_off[0] + l.n - 1] - l.smin;                 // pyx:75, denominator
    l.rspan = 1.0 / l.span;
    l.pow2 = __builtin_amdgcn_readfirstlane((int)sdp_is_pow2(l.span)) != 0;
    l.nm1 = (double)(l.n - 1);
    l.ordm2 = l.n - 2;
    const double t = a.t_k;
    if (threadIdx.x < 4) m.sh_k[threadIdx.x] = threadIdx.x < 2 ? INT_MIN : 0;
    __syncthreads();
    // the shifts of the perturbation points (sdp_col_phase_shift: the same operations)
    for (int w = (int)threadIdx.x; w < SDP_LN_W; w += (int)blockDim.x) {
        double b, babs;
        const double x0[1] = {0.0};
        sdp_model_lead_b(x0, wgrid[w], t, b, babs);
        const double pb = sdp_div_span<double>(b, l.span, l.rspan, l.pow2) * l.nm1;
        const double pbabs = fabs(sdp_div_span<double>(babs, l.span, l.rspan, l.pow2) * l.nm1);
        const bool ok = fabs(pb) < 536870912.0 && pbabs < 536870912.0;      // (false for a NaN)
        const double fl = ok ? floor(pb) : 0.0;
        const double fr = ok ? pb - fl : 0.0;                                // exact, in [0, 1)
        const int q = (int)fl;
        m.sh_q[w] = q;
        m.sh_f[w] = fr;
        m.sh_c[w] = fabs(p[w]) * (fr * (1.0 - fr));
        m.sh_p[w] = p[w];
        atomicMax(&m.sh_k[0], q);
        atomicMax(&m.sh_k[1], -q);
        if (!ok) atomicMax(&m.sh_k[2], 1);
        atomicMax(&m.sh_k[3], ok ? (int)pbabs + 1 : 0);
    }
    // the weights' sums (sdp_col_filter_setup): in w order, by one thread
    if (threadIdx.x == 0) {
        double ps = 0.0, pa = 0.0;
        for (int w = 0; w < SDP_LN_W; ++w) {
            ps = ps + p[w];
            pa = pa + (p[w] < 0.0 ? -p[w] : p[w]);
        }
        m.red[0] = ps;
        m.red[1] = pa;
    }
    __syncthreads();
    f.psum = m.red[0];
    const double pa = m.red[1];
    f.pcap = pa > 1.0 ? pa : 1.0;
    f.cu = (double)(SDP_LINE_FILTER_SCALE) * (double)(2 * (SDP_LN_W + 8)) * SDP_LN_EPS;
    f.floor = 2.0 * SDP_LN_TINY / f.cu;
    f.ratio = f.pcap / fabs(f.psum);
    f.ok = pa <= 1024.0;
    // the lattice (sdp_col_shift_col)
    const int flmax = __builtin_amdgcn_readfirstlane(m.sh_k[0]);
    const int nflmin = __builtin_amdgcn_readfirstlane(m.sh_k[1]);
    const int flag = __builtin_amdgcn_readfirstlane(m.sh_k[2]);
    c.kmin = -(flmax + 1);
    c.rows = l.n + flmax + nflmin + 1;
    c.flmax = flmax;
    c.flmin = -nflmin;
    // (the host sizes aux_a for 2 S + 64 positions)
    c.ok = flag == 0 && (int64_t)c.rows <= 2 * a.S + 64 && c.rows >= 2 && l.n >= 3 && abs(flmax) < (1 << 28) && abs(nflmin) < (1 << 28);
    const double pbmax = (double)(max(max(abs(flmax), abs(nflmin)), __builtin_amdgcn_readfirstlane(m.sh_k[3])) + 1);
    const double p0 = fabs(l.smin) * (l.nm1 / fabs(l.span));
    c.lc = (double)c.rows + (double)abs(c.kmin) + pbmax + p0 * 1.001 + (double)(l.n + 1);
    c.es = (double)(1 + 2 * (flmax + nflmin + 2));
    if (!(c.lc < 1073741824.0)) c.ok = false;
    l.koff = (double)c.kmin;
    l.lc = c.lc;
}

// ---- the reduced table on the shifted lattice: one thread per position (sdp_col_shift_reduce with T[w][.] = V).
// Per position k three numbers:  A'[k] = G(k);  B'[k], the chord bound of the cell above k;  C'[k] = (3 + 2 Lam_k) D_k with
// D_k = Pcap max |V| over the rows the entry reads (+ floor) and Lam_k the largest |lam| of its W lerps (1 inside the
// axis, more where a shifted point leaves it): what the ROUNDINGS around position k are proportional to.
extern "C" __global__ void __launch_bounds__(256) sdp_lead_reduce(SdpSweepArgs a)
{
    __shared__ SdpLineLds m;
    SdpLineAxis l;
    SdpLineFilter f;
    SdpLineCol c;
    sdp_line_setup(a, m, l, f, c);
    // (the slot the NEXT reduction will use -- SdpSweepArgs.aux_e here -- is cleared now: the sweep that read it is over)
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.aux_e) *(unsigned long long *)a.aux_e = 0ull;
    if (!c.ok) return;
    const double *__restrict__ V = (const double *)a.V;
    double *__restrict__ ad = (double *)a.aux_a;
    const int N0 = l.n;
    double smax = 0.0;
    bool any = false;
    for (int64_t ki = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ki < c.rows; ki += (int64_t)gridDim.x * blockDim.x) {
        const int k = c.kmin + (int)ki;
        double acc = 0.0, bnd = 0.0, big = 0.0, dv = 0.0, lmax = 1.0;
        const bool inner = k + c.flmin >= 0 && k + c.flmax <= N0 - 3;
        if (inner) {
#pragma unroll 4
            for (int w = 0; w < SDP_LN_W; ++w) {
                const double *row = V + (k + m.sh_q[w]);
                const double t0 = row[0], t1 = row[1], t2 = row[2];
                acc = fma(m.sh_p[w], fma(m.sh_f[w], t1 - t0, t0), acc);
                bnd = fma(m.sh_c[w], fabs((t2 - t1) - (t1 - t0)), bnd);
                // (the row below too: a position within a rounding of a whole number may fall into the cell below on the reference's path)
                big = sdp_ln_max_abs(sdp_ln_max_abs(sdp_ln_max_abs(sdp_ln_max_abs(big, t0), t1), t2), row[k + m.sh_q[w] > 0 ? -1 : 0]);
                dv = sdp_ln_max_abs(sdp_ln_max_abs(dv, t1 - t0), t2 - t1);
            }
        } else {
            for (int w = 0; w < SDP_LN_W; ++w) {
                const int j = k + m.sh_q[w];
                const int q = max(min(j, N0 - 2), 0);
                const double lam = (double)(j - q) + m.sh_f[w];
                const double *row = V + q;
                const double t0 = row[0], t1 = row[1], t2 = row[q + 2 < N0 ? 2 : 1];
                acc = fma(m.sh_p[w], fma(lam, t1 - t0, t0), acc);
                const double d2 = (t2 - t1) - (t1 - t0);
                const bool kink = (unsigned)j <= (unsigned)(N0 - 3);       // a kink inside the cell: row j+1 is an inner row
                bnd = fma(m.sh_c[w], kink ? fabs(d2) : 0.0, bnd);
                big = sdp_ln_max_abs(sdp_ln_max_abs(sdp_ln_max_abs(sdp_ln_max_abs(big, t0), t1), t2), row[q > 0 ? -1 : 0]);
                dv = sdp_ln_max_abs(sdp_ln_max_abs(dv, t1 - t0), t2 - t1);
                lmax = sdp_ln_max_abs(lmax, lam);
            }
        }
        // (a NaN entry, which the maxima skip, shows in the sum and makes the bounds infinite)
        const bool fin = acc == acc && bnd == bnd;
        ad[3 * ki] = acc;
        ad[3 * ki + 1] = fin ? bnd : (double)INFINITY;
        ad[3 * ki + 2] = fin ? (3.0 + 2.0 * (lmax + 1.0)) * (f.pcap * big + f.floor) : (double)INFINITY;
        // the steepest step of the value array between neighbouring rows (x 1.01: its own roundings), weights included:
        // what a position that is off by delta rows costs the interpolant at most -- delta x this
        smax = sdp_ln_max(smax, fin ? f.pcap * (dv * 1.01) + f.floor : (double)INFINITY);
        any = true;
    }
    // one atomic per workgroup that had positions (8192 waves adding to ONE word took 93 of this kernel's first 98 microseconds)
    smax = __ockl_wfred_max_f64(smax);
    __shared__ unsigned long long wg_max;
    if (threadIdx.x == 0) wg_max = 0ull;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && __any(any)) atomicMax(&wg_max, (unsigned long long)__double_as_longlong(smax));
    __syncthreads();
    if (threadIdx.x == 0 && wg_max != 0ull) atomicMax(a.aux_vmax, wg_max);
}

// What a node's first level keeps.  Per control c the reference's value E_c lies in [F_c - e_c, F_c + e_c] (e_c: below).
// f1 <= f2 <= f3: the three smallest lower ends, i1 / i2 the controls of the two smallest; ub: the smallest upper end;
// [c_lo, c_hi]: the lattice indices of the controls whose lower end lay at or below the smallest upper end seen so far --
// a superset of { c : F_c - e_c <= min_c' (F_c' + e_c') }, the only controls that can be the reference's argmin or tie
// with it; s_sum: the sum of the |F| and the e (a NaN or an infinity anywhere shows in it); p_max: the largest |lam0|.
struct SdpLineBounds {
    double f1, f2, f3, ub, s_sum, p_max;
    int i1, i2, c_lo, c_hi;
};
SDP_DEV void sdp_line_clear(SdpLineBounds &b)
{
    b.f1 = b.f2 = b.f3 = b.ub = INFINITY;
    b.s_sum = b.p_max = 0.0;
    b.i1 = b.i2 = b.c_lo = INT_MAX;
    b.c_hi = -1;
}
SDP_DEV void sdp_line_insert(SdpLineBounds &b, double F, int ci)
{
    if (SDP_LINE_TOP2) {
        b.f3 = sdp_ln_min(b.f3, sdp_ln_max(b.f2, F));
        const bool c1 = F < b.f1, c2 = F < b.f2;
        b.i2 = c1 ? b.i1 : (c2 ? ci : b.i2);
    }
    b.f2 = sdp_ln_min(b.f2, sdp_ln_max(b.f1, F));
    b.i1 = F < b.f1 ? ci : b.i1;
    b.f1 = sdp_ln_min(b.f1, F);
}
// One control on the lattice: F = fma(g, psum, lerp(A', pa - kmin)) (sdp_col_lean_core, shifted form) and the half-width
// e of the interval that holds the reference's value,
//     e = B'[q0] + cu ( ratio (|F| + lf Cq) + lf Cq + (Lc + |lam0|) Dv ),   lf = 2 + 3 |lam0|,  Cq = max(C'[q0], C'[q0 + 1]),  cu = 4 (W + 8) u:
//  * B'[q0]: what the chord of G over the control's cell leaves out (header of sdp_col_phase_shift);
//  * the reference's W terms (two products and a sum for the lerp, the cost, the weight, the running sum: W + 4
//    roundings at most on any of them): (W + 4) u [ |g| P + (1 + 2 Lam) D ] with Lam the largest |lam| of ITS cells --
//    each of which is a cell of the entries q0 or q0 + 1 moved by this control's own lam0, so Lam <= Lam_q + 1 + |lam0|
//    and (1 + 2 Lam) D <= (1 + |lam0|) Cq;
//  * this pass: the entries' own roundings (W + 4) u (1 + 2 Lam_k) D_k <= (W + 4) u C'[k], carried through the lerp
//    with weights |1 - lam0| + |lam0|, three roundings of the lerp on values below Cq, one of the fma: (W + 8) u (1 + 2 |lam0|) Cq + u |F|;
//  * |g| P <= ratio (|F| + |lerp|) (1 + 2u), |lerp| <= (1 + 2 |lam0|) Cq;
//  * positions: the reference rounds its sums, (. - smin) / span (N - 1) and the cell; this pass rounds B, pa, pb_w and
//    pa - kmin: together below 14 u (Lc + |lam0|) rows per point on either path (header of sdp_col_phase_shift: |pa| <= rows + |kmin| + |lam0|), and a position that
//    is off by delta moves the interpolant by at most delta x its steepest step between neighbouring rows: 28 u (Lc + |lam0|) Dv
//    with the weights in Dv (sdp_lead_reduce) -- where the column kernel takes delta x 2 max |T|, which on an axis of
//    65 536 rows kept a dozen neighbours of every optimum inside the radius.
// Sum: (2W + 12) u [ ratio (|F| + lf Cq) + lf Cq ] + 28 u Lc Dv  <  half of cu ( .. ) -- the factor 2 the column kernels keep too.
template <int POW2 = -1>      // 1 / 0: the axis' span is / is not a power of two (known to the caller); -1: look it up
SDP_DEV double sdp_line_value(const double *__restrict__ ad, const SdpLineFilter &f, const SdpLineAxis &l, int rows_m2, double lcdv,
                              const double *x, const double *u, double t, double &lmax, double &e, double &rnd)
{
    const double xn0 = sdp_model_lead_a(x, u, t);
#if SDP_LINE_CHAIN
    // a chain of sums that was regrouped (x + (w - u): a = x - u is not a value the reference computes): the reference's
    // nesting and a + B each lie within 2 m u (PA + PB) rows of the real position, PA the sum of the magnitudes of the
    // chain's w-free leaves in rows (SDP_COL_SHIFT_CHAIN of sdp_colfilter_kernel.h) -- PA joins Lc in the position term
    const double pa_abs = sdp_model_lead_aabs(x, u, t) * (fabs(l.nm1 * l.rspan) * 1.002);
    lmax = sdp_ln_max_abs(lmax, pa_abs);
#else
    const double pa_abs = 0.0;
#endif
    const double g = sdp_model_cost(x, u, 0.0, t);
    const bool pw = POW2 < 0 ? l.pow2 : POW2 != 0;
    const double sn = pw ? (xn0 - l.smin) * l.rspan : (xn0 - l.smin) / l.span;
    const double p = sn * l.nm1;
    const double pk = p - l.koff;
    int q0 = (int)pk;                                        // (saturating conversion; NaN -> 0)
    q0 = max(min(q0, rows_m2), 0);
    const double lam0 = pk - (double)q0;
    lmax = sdp_ln_max_abs(lmax, lam0);
    const double *row = ad + 3 * q0;
    const double a0 = row[0], bq = row[1], c0 = row[2], a1 = row[3], c1 = row[5];
    const double F = fma(g, f.psum, fma(lam0, a1 - a0, a0));
    const double lc_q = fma(3.0, fabs(lam0), 2.0) * sdp_ln_max(c0, c1);
    // (x cu: the rounding part; the second level doubles it.  lcdv = (Lc, Dv): |pa| <= Lc + |lam0|)
    rnd = fma(f.ratio, fabs(F) + lc_q, lc_q) + ((l.lc + fabs(lam0)) + pa_abs) * lcdv;
    const double w = fma(f.cu, rnd, bq);
    e = (double)(SDP_LINE_FILTER_SCALE) == 1.0 ? w : (double)(SDP_LINE_FILTER_SCALE) * w;
    return F;
}
SDP_DEV void sdp_line_take(SdpLineBounds &b, double F, double e, int ci)
{
    const double lo = F - e, hi = F + e;
    b.s_sum = b.s_sum + (fabs(F) + e);
    b.ub = sdp_ln_min(b.ub, hi);
    sdp_line_insert(b, lo, ci);
    if (lo <= b.ub) { b.c_lo = min(b.c_lo, ci); b.c_hi = max(b.c_hi, ci); }
}

// first level over the controls gs, gs + P, .. of one node: four controls per round, their reads of the table in flight together
template <bool PLAIN, int POW2>
SDP_DEV void sdp_line_pass1(const double *__restrict__ ad, const SdpLineFilter &f, const SdpLineAxis &l, int rows_m2, double lcdv,
                            const double *x, double t, const SdpBox &box, int gs, SdpLineBounds &b)
{
    constexpr int K = 4, P = 4 * SDP_LANES;
    const int last = PLAIN ? box.n[0] - 1 : INT_MAX;            // (the last point of a linspace is its `stop`)
    const int n_main = PLAIN ? box.total - 1 : box.total;
    auto point = [&](int ci, double *u) {
        if (PLAIN) u[0] = (double)ci * box.step[0] + box.lo[0];
        else sdp_controls_at(box, ci, u);
    };
    int ci = gs;
    for (; ci + (K - 1) * P < n_main; ci += K * P) {
        double F[K], e[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double u[SDP_NU], rnd;
            point(ci + k * P, u);
            F[k] = sdp_line_value<POW2>(ad, f, l, rows_m2, lcdv, x, u, t, b.p_max, e[k], rnd);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) sdp_line_take(b, F[k], e[k], ci + k * P);
    }
    for (; ci < n_main; ci += P) {
        double u[SDP_NU], e, rnd;
        point(ci, u);
        const double F = sdp_line_value<POW2>(ad, f, l, rows_m2, lcdv, x, u, t, b.p_max, e, rnd);
        sdp_line_take(b, F, e, ci);
    }
    if (PLAIN && ci == last) {
        double e, rnd;
        const double F = sdp_line_value<POW2>(ad, f, l, rows_m2, lcdv, x, box.hi, t, b.p_max, e, rnd);
        sdp_line_take(b, F, e, last);
    }
}

// The second level of the filter: G(pa) itself, in this pass's own (fused) arithmetic -- every perturbation point's
// whole shift and fraction from the LDS tables, the two rows of the value array, one fma for the lerp, one for the weight.
// No chord: what separates F2 = fma(g, psum, G2) from the reference's value is rounding alone -- the same terms as above
// with G2's W fused steps in the place of the table's entries --, so its half-width is 2 cu ( .. ) (`rnd` of
// sdp_line_value) without B': ~1e-13 of the values, where the chord bound of a fine grid keeps tens of neighbours of the
// optimum.  W x ~10 instructions per control: only for the controls the first level could not decide.
SDP_DEV double sdp_line_value2(const double *__restrict__ V, const SdpLineLds &m, const SdpLineFilter &f, const SdpLineAxis &l,
                               const double *x, const double *u, double t)
{
    const double xn0 = sdp_model_lead_a(x, u, t);
    const double g = sdp_model_cost(x, u, 0.0, t);
    const double sn = l.pow2 ? (xn0 - l.smin) * l.rspan : (xn0 - l.smin) / l.span;
    const double p = sn * l.nm1;
    double pf = floor(p);
    pf = pf < -1073741824.0 ? -1073741824.0 : (pf > 1073741824.0 ? 1073741824.0 : pf);       // (a NaN passes: the node is bad then)
    const double fr = p - pf;                                  // in [0, 1) (exact)
    const int k = (int)pf;
    const int N0 = l.n;
    double acc = 0.0;
    constexpr int B = 8;
#pragma unroll 1
    for (int w0 = 0; w0 < SDP_LN_W; w0 += B) {
        double t0[B], t1[B], lam[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int w = min(w0 + b, SDP_LN_W - 1);
            // position k + fr + q_w + f_w: whole part j, fraction lam (the carry of the two fractions into j)
            const double fs = fr + m.sh_f[w];
            const int carry = fs >= 1.0 ? 1 : 0;
            const int jw = k + m.sh_q[w] + carry;
            const int q = max(min(jw, N0 - 2), 0);
            lam[b] = (double)(jw - q) + (fs - (double)carry);
            t0[b] = V[q];
            t1[b] = V[q + 1];
        }
#pragma unroll
        for (int b = 0; b < B; ++b)
            if (w0 + b < SDP_LN_W) acc = fma(m.sh_p[w0 + b], fma(lam[b], t1[b] - t0[b], t0[b]), acc);
    }
    return fma(g, f.psum, acc);
}

// ---- the sweep.  A workgroup of four waves takes a TILE of NPW = 64 / SDP_LANES consecutive nodes; lane s NPW + j of
// wave v stands for node j of the tile and for slice v SDP_LANES + s of its control lattice (controls slice, slice + P,
// slice + 2 P, .., P = 4 SDP_LANES slices in all): lanes with consecutive j read consecutive rows of the reduced table,
// slices next to each other read rows a control step apart -- a wave's reads fall on a few cache lines.
// The slices' bounds meet through shuffles (within a wave) and LDS (the four waves), in a fixed order, so that every
// lane of a node holds the same values.  A node whose first level leaves more than two controls (a fine grid: the
// chord bound exceeds the differences between neighbouring controls) runs the second level on those.
// The survivors of a tile -- one or two per node -- are then evaluated the reference's way by ALL the workgroup's
// threads, an item per (node, perturbation point): dyn, the cell of pyx:75-81, the lerp of V, the cost, the weight
// (stodynprog.py:674-677) into an LDS table of terms, which one lane per node adds up in w order (stodynprog.py:681):
// the operations of sdp_expected_cost on the same operands, the same bits.  What is left -- true near-ties, NaN /
// infinite values -- evaluates its candidates with sdp_expected_cost itself, a slice's share per lane.
constexpr int SDP_LN_L = SDP_LANES, SDP_LN_NPW = 64 / SDP_LANES, SDP_LN_WAVES = 4, SDP_LN_P = SDP_LN_WAVES * SDP_LANES;
static_assert(SDP_LN_NPW * SDP_LN_W <= 2048, "sdp_line_kernel.h: nodes per tile x perturbation points <= 2048 (the planner picks SDP_LANES)");
struct SdpLineNode { double f1, f2, f3, ub, s_sum, p_max; int i1, i2, c_lo, c_hi; };
struct SdpLineSweepLds {
    SdpLineLds m;
    SdpLineNode bnd[SDP_LN_WAVES][SDP_LN_NPW];
    double terms[2][SDP_LN_W][SDP_LN_NPW];
    int cand[2][SDP_LN_NPW];                  // the survivors' lattice indices (INT_MAX: none)
    double fb_v[SDP_LN_WAVES][SDP_LN_NPW];    // candidates the long way: a wave's best of a node
    int fb_i[SDP_LN_WAVES][SDP_LN_NPW];
};

// what a node's slices have seen, merged: the slices of a wave through shuffles (lanes j, j + NPW, ..), the four waves
// through LDS, every lane reading the same four records in the same order (barriers inside: the whole workgroup calls)
SDP_DEV void sdp_line_merge(SdpLineSweepLds &lds, SdpLineBounds &b, int wave, int sl, int j)
{
    constexpr int NPW = SDP_LN_NPW;
#pragma unroll
    for (int s = NPW; s < 64; s <<= 1) {
        const double o1 = __shfl_xor(b.f1, s, 64), o2 = __shfl_xor(b.f2, s, 64), o3 = __shfl_xor(b.f3, s, 64);
        const int j1 = __shfl_xor(b.i1, s, 64), j2 = __shfl_xor(b.i2, s, 64);
        b.s_sum = b.s_sum + __shfl_xor(b.s_sum, s, 64);
        b.p_max = sdp_ln_max(b.p_max, __shfl_xor(b.p_max, s, 64));
        b.ub = sdp_ln_min(b.ub, __shfl_xor(b.ub, s, 64));
        b.c_lo = min(b.c_lo, __shfl_xor(b.c_lo, s, 64));
        b.c_hi = max(b.c_hi, __shfl_xor(b.c_hi, s, 64));
        if (j1 != INT_MAX) sdp_line_insert(b, o1, j1);
        if (SDP_LINE_TOP2) {
            if (j2 != INT_MAX) sdp_line_insert(b, o2, j2);
            b.f3 = sdp_ln_min(b.f3, o3);
        } else b.f2 = sdp_ln_min(b.f2, o2);
    }
    __syncthreads();                                       // (the records of the previous use have been read)
    if (sl == 0) {
        SdpLineNode &o = lds.bnd[wave][j];
        o.f1 = b.f1; o.f2 = b.f2; o.f3 = b.f3; o.ub = b.ub; o.s_sum = b.s_sum; o.p_max = b.p_max;
        o.i1 = b.i1; o.i2 = b.i2; o.c_lo = b.c_lo; o.c_hi = b.c_hi;
    }
    __syncthreads();
    sdp_line_clear(b);
#pragma unroll
    for (int v = 0; v < SDP_LN_WAVES; ++v) {
        const SdpLineNode &o = lds.bnd[v][j];
        b.s_sum = b.s_sum + o.s_sum;
        b.p_max = sdp_ln_max(b.p_max, o.p_max);
        b.ub = sdp_ln_min(b.ub, o.ub);
        b.c_lo = min(b.c_lo, o.c_lo);
        b.c_hi = max(b.c_hi, o.c_hi);
        if (o.i1 != INT_MAX) sdp_line_insert(b, o.f1, o.i1);
        if (SDP_LINE_TOP2) {
            if (o.i2 != INT_MAX) sdp_line_insert(b, o.f2, o.i2);
            b.f3 = sdp_ln_min(b.f3, o.f3);
        } else b.f2 = sdp_ln_min(b.f2, o.f2);
    }
}

extern "C" __global__ void __launch_bounds__(256) sdp_sweep(SdpSweepArgs a)
{
    __shared__ SdpLineSweepLds lds;
    constexpr int L = SDP_LN_L, NPW = SDP_LN_NPW, P = SDP_LN_P, W = SDP_LN_W;
    const int lane = threadIdx.x & 63;
    const int j = lane % NPW;                         // node of the tile
    const int sl = lane / NPW;                        // slice within the wave
    const int wave = threadIdx.x >> 6;
    const int gs = wave * L + sl;                     // slice of the control lattice
    sdp_trap_unless(blockDim.x == 64 * SDP_LN_WAVES);
    SdpLineAxis l;
    SdpLineFilter f;
    SdpLineCol c;
    sdp_line_setup(a, lds.m, l, f, c);
    const double *__restrict__ V = (const double *)a.V;
    const double *__restrict__ ad = (const double *)a.aux_a;
    const double *__restrict__ axis = (const double *)a.axes + a.axis_off[0];
    const double *__restrict__ wgrid = (const double *)a.wgrid;
    const double *__restrict__ proba = (const double *)a.proba;
    SdpGrid<double, 1> grid;
    sdp_grid_from_args(a, grid);
    const double t = a.t_k;
    // Lc x the steepest step of the value array (sdp_lead_reduce; a NaN in the value array makes it infinite)
    const double dvcol = c.ok ? __longlong_as_double((long long)__builtin_nontemporal_load(a.aux_vmax)) : (double)INFINITY;
    const double lcdv = dvcol;                        // (the position term's factor: sdp_line_value multiplies by Lc + |lam0|)
    const int rows_m2 = c.ok ? c.rows - 2 : 0;
    const bool usable = f.ok && c.ok;

    const int64_t n_nodes = a.node_end - a.node_begin;
    const int64_t n_tiles = (n_nodes + NPW - 1) / NPW;
    const int xcd = blockIdx.x & 7;
    const int64_t per_xcd = (n_tiles + 7) / 8;
    const int64_t t_end = min((int64_t)(xcd + 1) * per_xcd, n_tiles);
    const int64_t stride = gridDim.x >> 3;
    for (int64_t tile = (int64_t)xcd * per_xcd + (blockIdx.x >> 3); tile < t_end; tile += stride) {
        const int64_t node0 = a.node_begin + tile * NPW;
        const int64_t node = node0 + j;
        const bool live = node < a.node_end;
        const int64_t nd = live ? node : a.node_end - 1;               // (a lane without a node repeats the last one)
        double x[1];
        x[0] = axis[nd];
        SdpBox box;
        sdp_load_box(a, nd, box);
        // ---- first level: the chord of the reduced table, every control of the slice
        SdpLineBounds b;
        sdp_line_clear(b);
        if (usable) {
            // (an ordinary lattice -- one control, numpy.linspace with more than one point and a step that is not zero -- takes
            // its points as i step + lo without sdp_control_value's rare branches, the last one (`stop`) apart)
            const bool plain = SDP_NU == 1 && box.n[0] > 1 && box.step[0] != 0.0;
            if (__all(plain)) {
                if (l.pow2) sdp_line_pass1<true, 1>(ad, f, l, rows_m2, lcdv, x, t, box, gs, b);
                else sdp_line_pass1<true, 0>(ad, f, l, rows_m2, lcdv, x, t, box, gs, b);
            } else {
                if (l.pow2) sdp_line_pass1<false, 1>(ad, f, l, rows_m2, lcdv, x, t, box, gs, b);
                else sdp_line_pass1<false, 0>(ad, f, l, rows_m2, lcdv, x, t, box, gs, b);
            }
        }
        sdp_line_merge(lds, b, wave, sl, j);
        // a control with  F - e > min (F + e)  is not the reference's argmin, nor tied with it
        const bool bad = !usable || !(b.s_sum < SDP_LN_LIMIT) || !(b.p_max + c.lc < 1073741824.0);
        const double m_hi = b.ub;
        bool single = !bad && b.i1 != INT_MAX && b.f2 > m_hi;
        bool pair = SDP_LINE_TOP2 && !bad && !single && b.i1 != INT_MAX && b.i2 != INT_MAX && b.f3 > m_hi;
        // the controls that can still be candidates lie in [c_lo, c_hi]; the first one of this lane's slice in it
        const int r_lo = b.c_hi >= 0 ? b.c_lo : 0, r_hi = b.c_hi >= 0 ? b.c_hi + 1 : 0;
        const int r_first = r_lo + ((gs - r_lo) % P + P) % P;
        // ---- second level, for the nodes the first one leaves with more than two controls
        const bool undecided = !bad && !single && !pair;
        if (live && wave == 0 && sl == 0) { SDP_LN_COUNT(0, single); SDP_LN_COUNT(1, pair); SDP_LN_COUNT(2, undecided); SDP_LN_COUNT(6, bad); }
        double m_hi2 = INFINITY;
        if (__syncthreads_or((int)(undecided && live))) {
            SdpLineBounds b2;
            sdp_line_clear(b2);
            if (undecided) {
                for (int ci = r_first; ci < r_hi; ci += P) {
                    double u[SDP_NU], e, rnd, pm = 0.0;
                    sdp_controls_at(box, ci, u);
                    const double F = sdp_line_value(ad, f, l, rows_m2, lcdv, x, u, t, pm, e, rnd);
                    if (!(F - e > m_hi)) {
                        const double F2 = sdp_line_value2(V, lds.m, f, l, x, u, t);
                        // (|F2| against |F|: both are within e of the reference's value; the bound takes the larger)
                        const double e2 = (double)(SDP_LINE_FILTER_SCALE) * (2.0 * f.cu) * (rnd + f.ratio * fabs(F2 - F));
                        sdp_line_take(b2, F2, e2, ci);
                        if (live) SDP_LN_COUNT(7, 1);
                    }
                }
            }
            sdp_line_merge(lds, b2, wave, sl, j);
            if (undecided) {
                m_hi2 = b2.ub;
                single = b2.i1 != INT_MAX && b2.f2 > m_hi2;
                pair = SDP_LINE_TOP2 && !single && b2.i1 != INT_MAX && b2.i2 != INT_MAX && b2.f3 > m_hi2;
                b.i1 = b2.i1;
                b.i2 = b2.i2;
                if (live && wave == 0 && sl == 0) { SDP_LN_COUNT(3, single); SDP_LN_COUNT(4, pair); SDP_LN_COUNT(5, !single && !pair); }
            }
        }
        if (wave == 0 && sl == 0) {
            lds.cand[0][j] = (single || pair) ? (pair ? min(b.i1, b.i2) : b.i1) : INT_MAX;
            lds.cand[1][j] = pair ? max(b.i1, b.i2) : INT_MAX;
        }
        const int any_pair = __syncthreads_or((int)(pair && live));
        // ---- the survivors' terms: an item per (slot, perturbation point, node of the tile)
        const int slots = any_pair ? 2 : 1;
        for (int item = (int)threadIdx.x; item < slots * W * NPW; item += (int)blockDim.x) {
            const int jj = item % NPW, w = (item / NPW) % W, slot = item / (NPW * W);
            const int ci = lds.cand[slot][jj];
            const int64_t nn = min(node0 + jj, a.node_end - 1);
            if (ci != INT_MAX) {
                double xx[1], u[SDP_NU], xn[1], g;
                xx[0] = axis[nn];
                SdpBox bx;
                sdp_load_box(a, nn, bx);
                sdp_controls_at(bx, ci, u);
                sdp_model_cell(xx, u, wgrid[w], t, xn, g);
                const double jc = g + sdp_interp_point<double, 1, double, false>(V, grid, xn);   // stodynprog.py:677
                lds.terms[slot][w][jj] = jc * proba[w];
            }
        }
        // ---- what is left: the candidates (all controls of a bad node) the long way, a slice's share in lattice order,
        // compared like the reference compares
        double best = INFINITY;
        int ibest = INT_MAX;
        const bool fallback = !(single || pair);
        if (fallback) {
            const int f_first = bad ? gs : r_first, f_end = bad ? box.total : r_hi;
            for (int ci = f_first; ci < f_end; ci += P) {
                double u[SDP_NU];
                sdp_controls_at(box, ci, u);
                bool cnd = bad;
                if (!cnd) {
                    double e, rnd, pm = 0.0;
                    const double F = sdp_line_value(ad, f, l, rows_m2, lcdv, x, u, t, pm, e, rnd);
                    cnd = !(F - e > m_hi);
                    if (cnd) {
                        const double F2 = sdp_line_value2(V, lds.m, f, l, x, u, t);
                        const double e2 = (double)(SDP_LINE_FILTER_SCALE) * (2.0 * f.cu) * (rnd + f.ratio * fabs(F2 - F));
                        cnd = !(F2 - e2 > m_hi2);
                    }
                }
                if (cnd) {
                    if (live) SDP_LN_COUNT(8, 1);
                    const double jc = sdp_expected_cost(a, grid, V, x, u, t);
                    if (ibest == INT_MAX || sdp_better_seq(jc, best)) { best = jc; ibest = ci; }
                }
            }
        }
        if (__any(fallback)) {
#pragma unroll
            for (int s = NPW; s < 64; s <<= 1) {
                const double ov = __shfl_xor(best, s, 64);
                const int oi = __shfl_xor(ibest, s, 64);
                if (oi != INT_MAX && (ibest == INT_MAX || sdp_better_idx(ov, oi, best, ibest))) { best = ov; ibest = oi; }
            }
        }
        if (sl == 0) { lds.fb_v[wave][j] = best; lds.fb_i[wave][j] = ibest; }
        __syncthreads();
        if (wave == 0 && sl == 0 && live) {
            if (fallback) {
                best = INFINITY; ibest = INT_MAX;
#pragma unroll
                for (int v = 0; v < SDP_LN_WAVES; ++v) {
                    const double ov = lds.fb_v[v][j];
                    const int oi = lds.fb_i[v][j];
                    if (oi != INT_MAX && (ibest == INT_MAX || sdp_better_idx(ov, oi, best, ibest))) { best = ov; ibest = oi; }
                }
            } else {
                double acc = 0.0;
#pragma unroll 8
                for (int w = 0; w < W; ++w) acc = acc + lds.terms[0][w][j];                     // stodynprog.py:681, w order
                best = acc;
                ibest = lds.cand[0][j];
                if (pair) {
                    double acc1 = 0.0;
#pragma unroll 8
                    for (int w = 0; w < W; ++w) acc1 = acc1 + lds.terms[1][w][j];
                    if (sdp_better_seq(acc1, best)) { best = acc1; ibest = lds.cand[1][j]; }
                }
            }
            sdp_store_J<double>(a, node, 0, best);
            if (a.idx) a.idx[node] = ibest;
            if (a.pol) {
                double u[SDP_NU];
                sdp_controls_at(box, ibest, u);
#pragma unroll
                for (int cc = 0; cc < SDP_NU; ++cc) ((double *)a.pol)[node * SDP_NU + cc] = u[cc];
            }
        }
    }
}
