// sdp_lead_kernel.h -- Bellman backup for models with SEVERAL controlled state variables next to an
// exogenous process (gfx950, wave64; node order, no transposed layout):
//     x_k' = f_k(x, u[, t])      k <  m = SDP_LEAD_AXES     ("stocks": no perturbation)
//     x_k' = f_k(x_m.., w[, t])  k >= m                      (exogenous: no control, no stock)
// (TracedModel.controlled_axes; the cost may depend on everything; the reference's API admits any
// `dims`, stodynprog.py:57-81, multi-control lattice :655-660).  The column kernel covers m = 1; with two
// stocks its table would be W x N0 x N1 values.  Here the certified expectation-first filter of
// sdp_column_kernel.h runs on a REDUCED ARRAY IN GLOBAL MEMORY instead of an LDS table:
//
//   sdp_lead_reduce   A[n] = sum_w p_w inner_w(n),  inner_w(n) = the reference's interpolation over the
//                     trailing axes m.. of V[lead indices of n, .] at the trailing next state of (n, w)
//                     -- one pass over the grid per sweep, W x 2^(d-m) vertex loads per node (what ONE
//                     control costs the direct kernel); E[plane] = max_w prod_k (|1-lam_k| + |lam_k|) of the
//                     trailing cells of a plane (nodes with the same x_m..), vmax = max |V|.
//                     A and a copy of V are written PLANE-MAJOR ([trailing index][lead index], the last stock
//                     fastest): what the controls of a node reach is then a patch of ONE plane (N0 x N1
//                     values, 128 KiB at 128 x 128) instead of a slab of the whole array
//   sdp_sweep         one lane per node, in plane-major order: consecutive lanes = consecutive values of the
//                     last stock, so the reads of a wave are contiguous, neighbouring controls re-read the
//                     same lines (L1), and the planes an XCD works on stay in its L2.  (In node order the
//                     same kernel read the reduced array at the Infinity Cache's rate: 1.5 ms, now below.)
//     pass 1, every control:  F = fma(g, psum, multilerp_m(A; q_k, lam_k)), 2^m loads and ~25 + 12 m
//                     instructions instead of W x (2^d loads + ~6 d + 10 instructions);
//                     per node the two smallest F, sum |F|, Lp = max_u prod_k (1 + 2 |lam_k|)
//     pass 2, survivors:  the reference's W x 2^d cell evaluation (sdp_expected_cost), so J, policy and
//                     index have the reference's bits
// In real arithmetic the nested lerp of pyx:88-300 is sum over the 2^d vertices of (lead weights, free of w)
// x (trailing weights, free of u) x V, so  R(u) = g P + sum_lead-corners wl A*[corner]  exactly.
// Roundings, u the unit roundoff, first order:
//     |E - R| <= (W + 3d + 2) u S      E: per term d lerp levels of 3 roundings, g + val, . p_w, W additions
//     |F - R| <= (W + 3d + 4) u S      A: W fma + 3 (d-m) per inner; m lerp levels; the fma with g
//     S(u) = |g| Pcap + prod_k<m (1 + 2 |lam_k|) Dabs,   Dabs = Pcap max_w E_w vmax >= sum_w |p_w| sum |wt| |V|
// and with |g| Pcap <= ratio (|F| (1 + u) + |h|), |h| <= Lp Dabs:
//     S_node = ratio (sum |F| + Lp Dabs) + Lp Dabs,   radius = cu S_node,   cu = 4 (W + 3d + 4) u
// (a factor 2 over the first-order bound, as in sdp_col_lean_core).  A NaN anywhere makes F a NaN, which
// sticks in sum |F|; an infinity makes Dabs or sum |F| infinite; |p_k| >= 2^31 (x86 truncation of the
// reference, sdp_trunc_i32) makes Lp >= 2^30: all of them mark the node, which then evaluates every
// control the long way.  Dabs carries 2 tiny / cu so that the radius never drops below the smallest
// normal number.
// 8-byte reals only: the form for 4-byte reals of round 5 (a radius of ~1e-5 of the values kept so many controls that the
// sweep was 3 x slower than every control the long way) was removed in round 6; such problems run the staged tiles.
#pragma once

#if SDP_LEAD_AXES < 1 || SDP_LEAD_AXES > SDP_D || !SDP_HAS_W || SDP_LANES != 1
#error "sdp_lead_kernel.h: 1 <= SDP_LEAD_AXES <= SDP_D, a perturbation, one lane per node"
#endif
typedef double sdp_lacc;             // the reduced array, F and the bound
static_assert(sizeof(sdp_real) == 8, "sdp_lead_kernel.h: 8-byte reals");
#ifndef SDP_LEAD_COST_HAS_W
#define SDP_LEAD_COST_HAS_W 0        // 1: the cost depends on the perturbation (see sdp_lead_first)
#endif
#ifndef SDP_LEAD_UNROLL
#define SDP_LEAD_UNROLL 1            // controls of the first pass per round (2 and 4 measured the same: 0.98 ms)
#endif
#ifndef SDP_LEAD_FILTER_SCALE
#define SDP_LEAD_FILTER_SCALE 1      // test knob: multiplies the radius (any value >= 1: same bits)
#endif
constexpr int SDP_LM = SDP_LEAD_AXES, SDP_LT = SDP_D - SDP_LEAD_AXES;
// The controlled state variables need not be listed first (the reference takes the order of the state variables
// from dyn's signature, stodynprog.py:119-131): logical axis j of this kernel -- stocks first, then the exogenous
// process -- is state variable SDP_LP[j].  Only the FILTER works in the logical order (reduced array, first
// pass: free to reorder, the bound holds for any nest); the second pass evaluates the reference's nest in the
// reference's own axis order (sdp_expected_cost: the strides alone know about the plane-major copy).
#ifdef SDP_LEAD_PERM
constexpr int SDP_LP[SDP_MAXD] = SDP_LEAD_PERM;
constexpr bool SDP_LEAD_PERMUTED = true;
#else
constexpr int SDP_LP[SDP_MAXD] = {0, 1, 2, 3};
constexpr bool SDP_LEAD_PERMUTED = false;
#endif

SDP_DEV double sdp_lead_vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
SDP_DEV double sdp_lead_vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
SDP_DEV double sdp_lead_vmax_abs(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
extern "C" __device__ double __ockl_wfred_max_f64(double);

// trailing axes m.. as a grid of their own (C-order strides inside one block of trailing nodes)
struct SdpLeadGeom {
    int64_t ts;                 // nodes per block of trailing coordinates = prod orders[m..]
    int64_t ls;                 // nodes per plane = prod orders[..m-1]
    int64_t lm[SDP_LM];         // strides of the lead axes in node order
    int pm[SDP_LM];             // strides of the lead axes inside a plane (plane-major arrays)
    int64_t nstr[SDP_D];        // node-order stride of LOGICAL axis j (state variable SDP_LP[j])
    int tm[SDP_LT > 0 ? SDP_LT : 1];   // strides of the trailing axes inside the trailing index
    int ord[SDP_D];             // points of logical axis j
    sdp_real smin[SDP_LM], span[SDP_LM], rspan[SDP_LM], nm1[SDP_LM];
    int ordm2[SDP_LM];
    int pow2;
};
SDP_DEV void sdp_lead_geom(const SdpSweepArgs &a, SdpLeadGeom &g)
{
    const sdp_real *axes = (const sdp_real *)a.axes;
    int64_t phys[SDP_D];                                   // node-order strides of the state variables
    {
        int64_t acc = 1;
#pragma unroll
        for (int p = SDP_D - 1; p >= 0; --p) { phys[p] = acc; acc *= a.orders[p]; }
    }
#pragma unroll
    for (int j = 0; j < SDP_D; ++j) { g.ord[j] = a.orders[SDP_LP[j]]; g.nstr[j] = phys[SDP_LP[j]]; }
    int64_t m = 1;
    int tmul = 1;
#pragma unroll
    for (int k = SDP_D - 1; k >= SDP_LM; --k) { g.tm[k - SDP_LM] = tmul; tmul *= g.ord[k]; m *= g.ord[k]; }
    g.ts = m;
    g.pow2 = 0;
    int pm = 1;
#pragma unroll
    for (int k = SDP_LM - 1; k >= 0; --k) {
        const int ax = SDP_LP[k];
        g.lm[k] = g.nstr[k];
        g.pm[k] = pm;
        pm *= g.ord[k];
        g.smin[k] = axes[a.axis_off[ax]];
        g.span[k] = axes[a.axis_off[ax] + a.orders[ax] - 1] - g.smin[k];
        g.rspan[k] = (sdp_real)1 / g.span[k];
        g.nm1[k] = (sdp_real)(a.orders[ax] - 1);
        g.ordm2[k] = a.orders[ax] - 2;
        if (sdp_is_pow2(g.span[k])) g.pow2 |= 1 << k;
    }
    g.ls = pm;
    g.pow2 = __builtin_amdgcn_readfirstlane(g.pow2);
}
// node <-> (lead index, trailing index).  Stocks listed first: node = lead * ts + trail.
SDP_DEV void sdp_lead_split(const SdpLeadGeom &g, int64_t node, int64_t &lead, int64_t &trail)
{
    if (!SDP_LEAD_PERMUTED) { lead = node / g.ts; trail = node - lead * g.ts; return; }
    lead = 0; trail = 0;
#pragma unroll
    for (int j = 0; j < SDP_D; ++j) {
        const int i = (int)((node / g.nstr[j]) % g.ord[j]);
        if (j < SDP_LM) lead += (int64_t)i * g.pm[j];
        else trail += (int64_t)i * g.tm[j - SDP_LM];
    }
}
SDP_DEV int64_t sdp_lead_join(const SdpLeadGeom &g, int64_t lead, int64_t trail)
{
    if (!SDP_LEAD_PERMUTED) return lead * g.ts + trail;
    int64_t node = 0;
#pragma unroll
    for (int j = SDP_LM - 1; j >= 0; --j) { node += (lead % g.ord[j]) * g.nstr[j]; lead /= g.ord[j]; }
#pragma unroll
    for (int j = SDP_D - 1; j >= SDP_LM; --j) { node += (trail % g.ord[j]) * g.nstr[j]; trail /= g.ord[j]; }
    return node;
}
// node-order offset of the trailing block of a lead index (all trailing indices zero)
SDP_DEV int64_t sdp_lead_base(const SdpLeadGeom &g, int64_t lead)
{
    if (!SDP_LEAD_PERMUTED) return lead * g.ts;
    int64_t base = 0;
#pragma unroll
    for (int j = SDP_LM - 1; j >= 0; --j) { base += (lead % g.ord[j]) * g.nstr[j]; lead /= g.ord[j]; }
    return base;
}
#if SDP_LEAD_AXES < SDP_D
// the trailing axes as a grid of their own over the value array in NODE order (their own strides there)
SDP_DEV void sdp_lead_trail_grid(const SdpSweepArgs &a, const SdpLeadGeom &geo, SdpGrid<sdp_real, SDP_LT> &tg)
{
    const sdp_real *axes = (const sdp_real *)a.axes;
    sdp_real smin[SDP_LT], smax[SDP_LT];
    int32_t ord[SDP_LT];
#pragma unroll
    for (int k = 0; k < SDP_LT; ++k) {
        const int ax = SDP_LP[SDP_LM + k];
        ord[k] = a.orders[ax];
        smin[k] = axes[a.axis_off[ax]];
        smax[k] = axes[a.axis_off[ax] + ord[k] - 1];
    }
    sdp_make_grid<sdp_real, SDP_LT>(tg, ord, smin, smax);
#pragma unroll
    for (int k = 0; k < SDP_LT; ++k) tg.M[k] = (int)geo.nstr[SDP_LM + k];
}
#endif

// ---- the reduced array ------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256) sdp_lead_reduce(SdpSweepArgs a)
{
    const sdp_real *__restrict__ V = (const sdp_real *)a.V;
    sdp_lacc *__restrict__ A = (sdp_lacc *)a.aux_a;
    sdp_real *__restrict__ Vt = (sdp_real *)a.aux_v;
    sdp_real *__restrict__ E = (sdp_real *)a.aux_e;
    const sdp_real *__restrict__ wgrid = (const sdp_real *)a.wgrid;
    const sdp_real *__restrict__ proba = (const sdp_real *)a.proba;
    const sdp_real t = (sdp_real)a.t_k;
    SdpLeadGeom geo;
    sdp_lead_geom(a, geo);
#if SDP_LEAD_AXES < SDP_D
    SdpGrid<sdp_real, SDP_LT> tg;
    sdp_lead_trail_grid(a, geo, tg);
#endif
    sdp_real vmax = (sdp_real)0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (the launch covers the lead indices [aux_begin, aux_end): the whole grid on one GPU, a rank's slab of the
    // first stock plus what its controls reach when the backup is sharded)
    for (int64_t node = a.node_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; node < a.node_end; node += stride) {
        sdp_real x[SDP_D];
        sdp_node_coords(a, node, x);
        int64_t lead, trail;
        sdp_lead_split(geo, node, lead, trail);
        sdp_lacc acc = (sdp_lacc)0;
        sdp_real emax = (sdp_real)0;
        for (int wi = 0; wi < a.W; ++wi) {
#if SDP_LEAD_AXES < SDP_D
            sdp_real xt[SDP_LT];
            sdp_model_trails(x, wgrid[wi], t, xt);
            SdpCell<sdp_real, SDP_LT, sdp_real> c;
            sdp_real ew = (sdp_real)1;
#pragma unroll
            for (int k = 0; k < SDP_LT; ++k) {
                sdp_locate_axis<sdp_real, SDP_LT, sdp_real>(tg, k, xt[k], c);      // (the reference's cell: pyx:75-81)
                ew = ew * (fabs(c.oml[k]) + fabs(c.lam[k]));
            }
            const sdp_real inner = SdpLerp<sdp_real, SDP_LT, sdp_real, 0>::eval(V + sdp_lead_base(geo, lead), tg, c, 0);
            emax = ew > emax || ew != ew ? ew : emax;                              // (a NaN sticks)
#else
            const sdp_real inner = V[node];
            emax = (sdp_real)1;
#endif
            acc = fma((sdp_lacc)proba[wi], (sdp_lacc)inner, acc);
        }
        const sdp_real vn = V[node];
        A[trail * geo.ls + lead] = acc;                   // plane-major: [trailing index][lead index]
        Vt[trail * geo.ls + lead] = vn;
        if (lead == a.aux_begin) E[trail] = emax;
        vmax = (sdp_real)sdp_lead_vmax_abs((double)vmax, (double)vn);
    }
    const double vmax_w = __ockl_wfred_max_f64((double)vmax);
    if ((threadIdx.x & 63) == 0)
        atomicMax(a.aux_vmax, (unsigned long long)__double_as_longlong(vmax_w)); // (>= 0: ordered as integers)
}

// ---- the sweep ---------------------------------------------------------------------------------
// multilinear interpolation of A over the lead axes, last lead axis innermost, fused (a filter value)
template <int K>
struct SdpLeadLerp {
    static SDP_DEV sdp_lacc eval(const sdp_lacc *__restrict__ A, const int *off, const int *lm,
                                 const sdp_real *lam, int base)
    {
        const sdp_lacc lo = SdpLeadLerp<K + 1>::eval(A, off, lm, lam, base + off[K]);
        const sdp_lacc hi = SdpLeadLerp<K + 1>::eval(A, off, lm, lam, base + off[K] + lm[K]);
        return fma((sdp_lacc)lam[K], hi - lo, lo);
    }
};
template <>
struct SdpLeadLerp<SDP_LM> {
    static SDP_DEV sdp_lacc eval(const sdp_lacc *__restrict__ A, const int *, const int *,
                                 const sdp_real *, int base) { return A[base]; }
};

struct SdpLeadConst {
    sdp_lacc psum, pcap, ratio, cu, floor;
    const sdp_cst_real *p, *wg;    // weights and points (scalar loads)
    int W;
    bool ok;
};
SDP_DEV void sdp_lead_const(const SdpSweepArgs &a, SdpLeadConst &f)
{
    const sdp_cst_real *p = (const sdp_cst_real *)a.proba;
    sdp_lacc ps = (sdp_lacc)0, pa = (sdp_lacc)0;
    for (int w = 0; w < a.W; ++w) {
        ps = ps + (sdp_lacc)p[w];
        pa = pa + fabs((sdp_lacc)p[w]);
    }
    f.psum = ps;
    f.p = p;
    f.wg = (const sdp_cst_real *)a.wgrid;
    f.W = a.W;
    f.pcap = pa > (sdp_lacc)1 ? pa : (sdp_lacc)1;
    f.ratio = f.pcap / fabs(ps);                   // (psum = 0: infinite -> every node takes the long way)
    f.cu = (sdp_lacc)SDP_LEAD_FILTER_SCALE * (sdp_lacc)(4 * (a.W + 3 * SDP_D + 4)) * (sdp_lacc)0x1p-53;
    f.floor = (sdp_lacc)2 * (sdp_lacc)2.2250738585072014e-308 / f.cu;
    f.ok = pa <= (sdp_lacc)1024;                   // (false for a NaN)
}

// the control lattice in C order (control 0 slowest) without a division per point: the flat index i of the
// walk stands for the same point as sdp_controls_at(box, i, .), and the values are sdp_control_value's
struct SdpLeadWalk {
    int k[SDP_NU];
    sdp_real u[SDP_NU];
};
SDP_DEV void sdp_lead_walk_begin(const SdpBox &b, SdpLeadWalk &w)
{
#pragma unroll
    for (int c = 0; c < SDP_NU; ++c) {
        w.k[c] = 0;
        w.u[c] = sdp_control_value(b, c, 0);
    }
}
SDP_DEV void sdp_lead_walk_next(const SdpBox &b, SdpLeadWalk &w)
{
    bool carry = true;
#pragma unroll
    for (int c = SDP_NU - 1; c >= 0; --c) {
        if (carry) {
            const int k = w.k[c] + 1;
            carry = k >= b.n[c];
            w.k[c] = carry ? 0 : k;
            w.u[c] = sdp_control_value(b, c, w.k[c]);
        }
    }
}

// F of one control and the product of the (1 + 2 |lam_k|); q_k, lam_k, g as the reference computes them
// (`A`: the node's plane of the reduced array)
// With a cost that depends on the perturbation (SDP_LEAD_COST_HAS_W; the stocks still must not): the
// expectation G = sum_w p_w g_w of the cost with the reference's own g_w takes the place of g psum
// (W cost evaluations per control: the 2^d loads and the lerps of the long way are what is saved), and
// gmax collects Gabs = max(sum_w |p_w g_w|, max_w |g_w|), which stands where |g| Pcap stood in the bound
// (the raw magnitude too: a tiny weight must not hide a g_w that overflows g_w + val on the reference's path).
// [lmin, lmax]: lead indices (positions inside a plane) the node's controls have read so far
SDP_DEV sdp_lacc sdp_lead_first(const sdp_lacc *__restrict__ A, const SdpLeadGeom &geo, const SdpLeadConst &f,
                                const sdp_real *x, const sdp_real *u, sdp_real t, sdp_lacc &lp, sdp_lacc &gmax,
                                int &lmin, int &lmax)
{
    sdp_real xl[SDP_LM], lam[SDP_LM];
    int off[SDP_LM];
    sdp_model_leads(x, u, t, xl);
    sdp_lacc prod = (sdp_lacc)1;
#pragma unroll
    for (int k = 0; k < SDP_LM; ++k) {
        const sdp_real sn = sdp_div_span<sdp_real>(xl[k] - geo.smin[k], geo.span[k], geo.rspan[k], (geo.pow2 >> k) & 1);
        const sdp_real p = sn * geo.nm1[k];
        const int q = max(min((int)p, geo.ordm2[k]), 0);       // (saturating conversion; NaN -> 0)
        lam[k] = p - (sdp_real)q;
        off[k] = q * geo.pm[k];
        prod = prod * fma((sdp_lacc)2, (sdp_lacc)fabs(lam[k]), (sdp_lacc)1);      // (the fused lerp of F rounds too: 1 + 2 |lam|)
    }
    lp = sdp_lead_vmax(lp, prod);
    {
        int lo = 0, span = 0;
#pragma unroll
        for (int k = 0; k < SDP_LM; ++k) { lo += off[k]; span += geo.pm[k]; }
        lmin = min(lmin, lo);
        lmax = max(lmax, lo + span);
    }
    const sdp_lacc h = SdpLeadLerp<0>::eval(A, off, geo.pm, lam, 0);
#if SDP_LEAD_COST_HAS_W
    sdp_lacc G = (sdp_lacc)0, gabs = (sdp_lacc)0, graw = (sdp_lacc)0;
    for (int w = 0; w < f.W; ++w) {
        const sdp_lacc pw = (sdp_lacc)f.p[w];
        const sdp_lacc gw = (sdp_lacc)sdp_model_cost(x, u, f.wg[w], t);
        G = fma(pw, gw, G);
        gabs = fma(fabs(pw), fabs(gw), gabs);
        graw = sdp_lead_vmax_abs(graw, gw);
    }
    gmax = sdp_lead_vmax(gmax, sdp_lead_vmax(gabs, graw));
    return G + h;
#else
    (void)gmax;
    return fma((sdp_lacc)sdp_model_cost(x, u, (sdp_real)0, t), f.psum, h);
#endif
}

extern "C" __global__ void __launch_bounds__(256) sdp_sweep(SdpSweepArgs a)
{
    const sdp_lacc *__restrict__ Aall = (const sdp_lacc *)a.aux_a;
    const sdp_real *__restrict__ Vt = (const sdp_real *)a.aux_v;
    const sdp_real *__restrict__ E = (const sdp_real *)a.aux_e;
    SdpLeadGeom geo;
    sdp_lead_geom(a, geo);
    // the grid of the second pass: the reference's cells and weights, the strides of the plane-major copy of V
    SdpGrid<sdp_real, SDP_D> grid;
    sdp_grid_from_args(a, grid);
    {
        // (grid.M is indexed by STATE VARIABLE: sdp_expected_cost keeps the reference's axis order)
#pragma unroll
        for (int k = SDP_D - 1; k >= SDP_LM; --k) grid.M[SDP_LP[k]] = (int)geo.ls * geo.tm[k - SDP_LM];
#pragma unroll
        for (int k = 0; k < SDP_LM; ++k) grid.M[SDP_LP[k]] = geo.pm[k];
    }
    SdpLeadConst fc;
    sdp_lead_const(a, fc);
    const sdp_real t = (sdp_real)a.t_k;
    sdp_trap_unless(a.aux_a != nullptr && a.aux_e != nullptr && a.aux_v != nullptr);
    const sdp_lacc vmax = __longlong_as_double((long long)*a.aux_vmax);

    // the second pass of a node whose controls reach beyond the reduced part of the grid (sharded backups: the
    // host's guess of the reach was too small) reads V itself, in node order
    SdpGrid<sdp_real, SDP_D> grid_v;
    sdp_grid_from_args(a, grid_v);

    // XCD-aware walk over tiles of 256 consecutive PLANE-MAJOR positions: an XCD takes a contiguous eighth,
    // i.e. whole planes, whose reduced values then stay in its L2.  A launch over part of the nodes (a rank's
    // slab of the first stock, a phase of it) walks the lead indices of that part only.
    // (with the stocks not listed first a node range is no range of lead indices: every position, filtered below)
    const int64_t lead_lo = SDP_LEAD_PERMUTED ? 0 : a.node_begin / geo.ts;
    const int64_t lead_hi = SDP_LEAD_PERMUTED ? geo.ls : (a.node_end + geo.ts - 1) / geo.ts, n_l = lead_hi - lead_lo;
    const int64_t n_pos = n_l * geo.ts;
    const int64_t n_tiles = (n_pos + blockDim.x - 1) / blockDim.x;
    const int xcd = blockIdx.x & 7;
    const int64_t per_xcd = (n_tiles + 7) / 8;
    const int64_t t_end = min((int64_t)(xcd + 1) * per_xcd, n_tiles);
    const int64_t stride = gridDim.x >> 3;
    for (int64_t tile = (int64_t)xcd * per_xcd + (blockIdx.x >> 3); tile < t_end; tile += stride) {
        const int64_t pos = tile * blockDim.x + threadIdx.x;
        if (pos >= n_pos) continue;
        const int64_t trail = pos / n_l, lead = lead_lo + (pos - trail * n_l);
        const int64_t node = sdp_lead_join(geo, lead, trail);
        if (node < a.node_begin || node >= a.node_end) continue;
        sdp_real x[SDP_D];
        SdpBox box;
        sdp_node_coords(a, node, x);
        sdp_load_box(a, node, box);
        const sdp_lacc *__restrict__ A = Aall + trail * geo.ls;
        const sdp_lacc dabs = fc.pcap * ((sdp_lacc)E[trail] * vmax) + fc.floor;
        // pass 1
        sdp_lacc f1 = INFINITY, f2 = INFINITY, fsum = (sdp_lacc)0, lp = (sdp_lacc)0, gmax = (sdp_lacc)0;
        int i1 = INT_MAX, lmin = INT_MAX, lmax = INT_MIN;
        SdpLeadWalk walk;
        sdp_lead_walk_begin(box, walk);
#pragma unroll SDP_LEAD_UNROLL
        for (int ci = 0; ci < box.total; ++ci) {
            const sdp_lacc F = sdp_lead_first(A, geo, fc, x, walk.u, t, lp, gmax, lmin, lmax);
            fsum = fsum + fabs(F);                         // (a NaN sticks)
            f2 = sdp_lead_vmin(f2, sdp_lead_vmax(f1, F));
            i1 = F < f1 ? ci : i1;
            f1 = sdp_lead_vmin(f1, F);
            sdp_lead_walk_next(box, walk);
        }
        const sdp_lacc h_cap = lp * dabs;
#if SDP_LEAD_COST_HAS_W
        const sdp_lacc s_node = fsum == fsum ? gmax + h_cap : (sdp_lacc)NAN;     // (a NaN of any F sticks in the sum)
#else
        const sdp_lacc s_node = fma(fc.ratio, fsum + h_cap, h_cap);
#endif
        // the reduced array and the plane-major copy of V hold this rank's part of the grid: a control that reads
        // outside it saw stale values -- the node then takes every control the long way on V itself
        const bool outside = (int64_t)lmin < a.aux_begin || (int64_t)lmax >= a.aux_end;
        const bool bad = outside || !fc.ok || !(s_node < (sdp_lacc)0x1p1000) || !(lp < (sdp_lacc)1073741824.0);
        const sdp_lacc radius = fc.cu * s_node;
        const sdp_lacc m_hi = f1 + radius;                 // >= the minimum of E over the node
        const bool single = !bad && i1 != INT_MAX && f2 - radius > m_hi;
        // pass 2: the reference's operations on the survivors, in lattice order
        sdp_real best = INFINITY;
        int ibest = INT_MAX;
        const int first = single ? i1 : 0, last = single ? i1 + 1 : box.total;
        if (single) sdp_controls_at(box, i1, walk.u);
        else sdp_lead_walk_begin(box, walk);
        for (int ci = first; ci < last; ++ci) {
            bool cand = single || bad;
            if (!cand) {
                sdp_lacc lq = (sdp_lacc)0, gq = (sdp_lacc)0;
                int l0 = 0, l1 = 0;
                cand = !(sdp_lead_first(A, geo, fc, x, walk.u, t, lq, gq, l0, l1) - radius > m_hi);
            }
            if (cand) {
                const sdp_real jc = outside ? sdp_expected_cost(a, grid_v, (const sdp_real *)a.V, x, walk.u, t)
                                            : sdp_expected_cost(a, grid, Vt, x, walk.u, t);
                if (ibest == INT_MAX || sdp_better_seq(jc, best)) { best = jc; ibest = ci; }
            }
            if (!single) sdp_lead_walk_next(box, walk);
        }
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
