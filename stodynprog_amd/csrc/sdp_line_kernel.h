// sdp_line_kernel.h -- ONE state variable whose perturbation enters the next state through final sums,
//     x' = fl(.. fl(fl(a(x, u) +- b_1(w)) +- b_2(w)) ..)          (the shop inventory `x + u - w`,
// reference doc/example_inventory.py:31-33; the stocks of examples/01 Deterministic storage control), with a cost that
// does not see the perturbation: the CERTIFIED FILTER ON THE SHIFTED LATTICE of sdp_colfilter_kernel.h (SDP_COL_SHIFT;
// DESIGN.md section 3.1d) for d = 1, where the "table" T[w][r] of the column kernel IS the value array -- no trailing
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
//       A'[k] = G(k) at the whole positions k = kmin .. kmin + rows - 1 of the lattice the perturbation points have
//       shifted, and the chord bound B'[k] = sum_w |p_w| f_w (1 - f_w) |d2_{w,k}| of the cell above k, as (A', B') pairs
//       in SdpSweepArgs.aux_a; max_k D[k] (D = Pcap max |V| over the rows an entry reads, + floor) into aux_vmax.
//   sdp_sweep
//       SDP_LANES lanes per node stride its control lattice: F = fma(g, psum, lerp(A', pa - kmin)) per control, the
//       three smallest F, sum |F|, max |lam0|, max B' -- merged over the node's lanes -- give the radius
//       cu S_node + max B' exactly as in sdp_colres_kernel.h; a control with F - radius > min F + radius is not the
//       reference's argmin nor tied with it.  The survivors (one, or the two lattice points either side of the
//       continuous optimum) are evaluated with the reference's operations: the node's lanes take a perturbation
//       point each (dyn, cell, lerp of V, cost, weight: stodynprog.py:674-677), the expectation is then summed in w
//       order by one chain of additions (stodynprog.py:681).  Anything else -- near-ties, NaN / infinite values,
//       a lattice that does not fit -- evaluates its candidates the long way in lattice order (sdp_expected_cost).
// J, policy and index carry the reference's bits: only WHICH controls are skipped depends on the filter.
// The radius is derived in the header comment of sdp_col_phase_shift (same roundings, T = V); checked in exact
// rational arithmetic by tests/test_filter_bound_exact.py (the shifted-lattice cases with a one-row table).
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
    double smin, span, rspan, nm1, koff;
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
}

// ---- the reduced table on the shifted lattice: one thread per position (sdp_col_shift_reduce with T[w][.] = V)
extern "C" __global__ void __launch_bounds__(256) sdp_lead_reduce(SdpSweepArgs a)
{
    __shared__ SdpLineLds m;
    SdpLineAxis l;
    SdpLineFilter f;
    SdpLineCol c;
    sdp_line_setup(a, m, l, f, c);
    if (!c.ok) return;
    const double *__restrict__ V = (const double *)a.V;
    double *__restrict__ ad = (double *)a.aux_a;
    const int N0 = l.n;
    double dmax = 0.0;
    for (int64_t ki = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ki < c.rows; ki += (int64_t)gridDim.x * blockDim.x) {
        const int k = c.kmin + (int)ki;
        double acc = 0.0, bnd = 0.0, big = 0.0;
        const bool inner = k + c.flmin >= 0 && k + c.flmax <= N0 - 3;
        if (inner) {
#pragma unroll 4
            for (int w = 0; w < SDP_LN_W; ++w) {
                const double *row = V + (k + m.sh_q[w]);
                const double t0 = row[0], t1 = row[1], t2 = row[2];
                acc = fma(m.sh_p[w], fma(m.sh_f[w], t1 - t0, t0), acc);
                bnd = fma(m.sh_c[w], fabs((t2 - t1) - (t1 - t0)), bnd);
                big = sdp_ln_max_abs(big, t1);
                big = sdp_ln_max_abs(big, t0);
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
                big = sdp_ln_max_abs(sdp_ln_max_abs(big, t0), t1);
            }
        }
        ad[2 * ki] = acc;
        ad[2 * ki + 1] = bnd;
        // (a NaN entry, which the max skips, shows in the sum and makes the bound infinite)
        dmax = sdp_ln_max(dmax, acc == acc ? f.pcap * big + f.floor : (double)INFINITY);
    }
    dmax = __ockl_wfred_max_f64(dmax);
    if ((threadIdx.x & 63) == 0) atomicMax(a.aux_vmax, (unsigned long long)__double_as_longlong(dmax));
}

// the running three smallest F of a node with the indices of the two smallest (sdp_col_bounds_insert)
struct SdpLineBounds {
    double f1, f2, f3, s_sum, p_max, b_max;
    int i1, i2;
};
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
// F of one control on the lattice (sdp_col_lean_eval + sdp_col_lean_core, shifted form)
SDP_DEV double sdp_line_value(const double *__restrict__ ad, const SdpLineFilter &f, const SdpLineAxis &l, int rows_m2,
                              const double *x, const double *u, double t, double &lmax, double &bmax)
{
    const double xn0 = sdp_model_lead_a(x, u, t);
#if SDP_LINE_CHAIN
    lmax = sdp_ln_max_abs(lmax, sdp_model_lead_aabs(x, u, t) * (fabs(l.nm1 * l.rspan) * 1.002));
#endif
    const double g = sdp_model_cost(x, u, 0.0, t);
    const double sn = l.pow2 ? (xn0 - l.smin) * l.rspan : (xn0 - l.smin) / l.span;
    const double p = sn * l.nm1;
    const double pk = p - l.koff;
    int q0 = (int)pk;                                        // (saturating conversion; NaN -> 0)
    q0 = max(min(q0, rows_m2), 0);
    const double lam0 = pk - (double)q0;
    lmax = sdp_ln_max_abs(lmax, lam0);
    const double a0 = ad[2 * q0], bq = ad[2 * q0 + 1], a1 = ad[2 * q0 + 2];
    bmax = sdp_ln_max(bmax, bq);
    return fma(g, f.psum, fma(lam0, a1 - a0, a0));
}

// ---- the sweep.  A workgroup of four waves takes a TILE of NPW = 64 / SDP_LANES consecutive nodes; lane s NPW + j of
// wave v stands for node j of the tile and for slice v SDP_LANES + s of its control lattice (controls slice, slice + P,
// slice + 2 P, .., P = 4 SDP_LANES slices in all): lanes with consecutive j read consecutive rows of the reduced table,
// slices next to each other read rows a control step apart -- a wave's reads fall on a few cache lines (a lane per
// control of ONE node, as in the direct kernel, spreads them over 64 x the control step in rows: the first version of
// this kernel spent its time in the vector memory path, 2.0 ms where this form takes 0.3 at 65 536 x 4097 x 16).
// The slices' bounds meet through shuffles (within a wave) and LDS (the four waves), in a fixed order, so that every
// lane of a node holds the same values.  The survivors of a tile -- one or two per node -- are then evaluated the
// reference's way by ALL the workgroup's threads, an item per (node, perturbation point): dyn, the cell of
// pyx:75-81, the lerp of V, the cost, the weight (stodynprog.py:674-677) into an LDS table of terms, which one lane per
// node adds up in w order (stodynprog.py:681): the operations of sdp_expected_cost on the same operands, the same bits.
constexpr int SDP_LN_L = SDP_LANES, SDP_LN_NPW = 64 / SDP_LANES, SDP_LN_WAVES = 4, SDP_LN_P = SDP_LN_WAVES * SDP_LANES;
static_assert(SDP_LN_NPW * SDP_LN_W <= 2048, "sdp_line_kernel.h: nodes per tile x perturbation points <= 2048 (the planner picks SDP_LANES)");
struct SdpLineNode { double f1, f2, f3, s_sum, p_max, b_max; int i1, i2; };
struct SdpLineSweepLds {
    SdpLineLds m;
    SdpLineNode bnd[SDP_LN_WAVES][SDP_LN_NPW];
    double terms[2][SDP_LN_W][SDP_LN_NPW];
    int cand[2][SDP_LN_NPW];                  // the survivors' lattice indices (INT_MAX: none)
    double fb_v[SDP_LN_WAVES][SDP_LN_NPW];    // candidates the long way: a wave's best of a node
    int fb_i[SDP_LN_WAVES][SDP_LN_NPW];
};

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
    // the bound of the whole lattice (max over its positions: sdp_lead_reduce; a NaN in the value array makes it infinite)
    const double dcol = c.ok ? __longlong_as_double((long long)__builtin_nontemporal_load(a.aux_vmax)) : (double)INFINITY;
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
        SdpLineBounds b;
        b.f1 = b.f2 = b.f3 = INFINITY;
        b.s_sum = b.p_max = b.b_max = 0.0;
        b.i1 = b.i2 = INT_MAX;
        if (usable) {
            for (int ci = gs; ci < box.total; ci += P) {
                double u[SDP_NU];
                sdp_controls_at(box, ci, u);
                const double F = sdp_line_value(ad, f, l, rows_m2, x, u, t, b.p_max, b.b_max);
                b.s_sum = b.s_sum + fabs(F);
                sdp_line_insert(b, F, ci);
            }
            // the slices of a wave meet (lanes j, j + NPW, j + 2 NPW, ..)
#pragma unroll
            for (int s = NPW; s < 64; s <<= 1) {
                const double o1 = __shfl_xor(b.f1, s, 64), o2 = __shfl_xor(b.f2, s, 64), o3 = __shfl_xor(b.f3, s, 64);
                const int j1 = __shfl_xor(b.i1, s, 64), j2 = __shfl_xor(b.i2, s, 64);
                b.s_sum = b.s_sum + __shfl_xor(b.s_sum, s, 64);
                b.p_max = sdp_ln_max(b.p_max, __shfl_xor(b.p_max, s, 64));
                b.b_max = sdp_ln_max(b.b_max, __shfl_xor(b.b_max, s, 64));
                if (j1 != INT_MAX) sdp_line_insert(b, o1, j1);
                if (SDP_LINE_TOP2) {
                    if (j2 != INT_MAX) sdp_line_insert(b, o2, j2);
                    b.f3 = sdp_ln_min(b.f3, o3);
                } else b.f2 = sdp_ln_min(b.f2, o2);
            }
        }
        // .. and the four waves (every lane reads the same four records in the same order)
        if (sl == 0) {
            SdpLineNode &o = lds.bnd[wave][j];
            o.f1 = b.f1; o.f2 = b.f2; o.f3 = b.f3; o.s_sum = b.s_sum; o.p_max = b.p_max; o.b_max = b.b_max; o.i1 = b.i1; o.i2 = b.i2;
        }
        __syncthreads();
        b.f1 = b.f2 = b.f3 = INFINITY;
        b.s_sum = b.p_max = b.b_max = 0.0;
        b.i1 = b.i2 = INT_MAX;
#pragma unroll
        for (int v = 0; v < SDP_LN_WAVES; ++v) {
            const SdpLineNode &o = lds.bnd[v][j];
            b.s_sum = b.s_sum + o.s_sum;
            b.p_max = sdp_ln_max(b.p_max, o.p_max);
            b.b_max = sdp_ln_max(b.b_max, o.b_max);
            if (o.i1 != INT_MAX) sdp_line_insert(b, o.f1, o.i1);
            if (SDP_LINE_TOP2) {
                if (o.i2 != INT_MAX) sdp_line_insert(b, o.f2, o.i2);
                b.f3 = sdp_ln_min(b.f3, o.f3);
            } else b.f2 = sdp_ln_min(b.f2, o.f2);
        }
        // the radius of the node (sdp_colres_kernel.h, shifted lattice)
        const double h_cap = ((1.0 + 2.0 * (b.p_max + c.lc)) * (3.0 + c.es)) * dcol;
        const double s_node = fma(f.ratio, b.s_sum + h_cap, h_cap);
        const bool bad = !usable || !(s_node < SDP_LN_LIMIT) || !(b.p_max + c.lc < 1073741824.0);
        const double radius = fma(f.cu, s_node, (double)(SDP_LINE_FILTER_SCALE) * b.b_max);
        const double m_hi = b.f1 + radius;                 // >= the minimum of E over the node
        const bool single = !bad && b.i1 != INT_MAX && b.f2 - radius > m_hi;
        const bool pair = SDP_LINE_TOP2 && !bad && !single && b.i1 != INT_MAX && b.i2 != INT_MAX && b.f3 - radius > m_hi;
        if (wave == 0 && sl == 0) {
            lds.cand[0][j] = (single || pair) ? (pair ? min(b.i1, b.i2) : b.i1) : INT_MAX;
            lds.cand[1][j] = pair ? max(b.i1, b.i2) : INT_MAX;
        }
        const int any_pair = __syncthreads_or((int)(pair && live));
        // the survivors' terms: an item per (slot, perturbation point, node of the tile)
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
        // near-ties or special values: the candidates (all controls of a bad node) the long way, a slice's share in
        // lattice order, compared like the reference compares
        double best = INFINITY;
        int ibest = INT_MAX;
        const bool fallback = !(single || pair);
        if (fallback) {
            for (int ci = gs; ci < box.total; ci += P) {
                double u[SDP_NU];
                sdp_controls_at(box, ci, u);
                bool cnd = bad;
                if (!cnd) {
                    double pm = 0.0, bm = 0.0;
                    const double F = sdp_line_value(ad, f, l, rows_m2, x, u, t, pm, bm);
                    cnd = !(F - radius > m_hi);
                }
                if (cnd) {
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
        __syncthreads();                                   // the tile's LDS records are free again
    }
}
