// sdp_device.h -- device-side building blocks (gfx950, wave64) shared by the
// built-in kernels and by generated-model code objects:
//   * uniform-grid multilinear interpolation with linear extrapolation, op for
//     op as reference multilinear_cython.pyx:51-300 (true division, truncating
//     cast with x86 semantics, clamp of the cell index only, nested lerp with
//     the last axis innermost);
//   * numpy-semantics scalar helpers used by generated model code;
//   * (value,index) first-occurrence argmin across the lanes of a wave segment.
// Compiled with -ffp-contract=off: one IEEE operation per source operator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "sdp_kernel_args.h"

#define SDP_DEV __device__ __forceinline__

// a launch the code object was not built for (the host validates `sdp_meta` first, so this
// never fires in a correct program): abort the kernel instead of returning stale results
SDP_DEV void sdp_trap_unless(bool ok) { if (!ok) __builtin_trap(); }

// J[node] = v, on this GPU and -- direct exchange of a sharded backup -- in the J buffer of every rank that
// reads the node (`unit`: its column in the column layout, where a mask says who does; ignored otherwise).
// Plain stores over xGMI into IPC-mapped buffers: they are complete when the kernel is, and the ranks meet
// (one tiny all-reduce per backup, sdp_hip.hip:finish_pushes) before anybody reads J or overwrites V.
template <typename real>
SDP_DEV void sdp_store_J(const SdpSweepArgs &a, int64_t node, int64_t unit, real v)
{
    ((real *)a.J)[node] = v;
    if (a.n_peer) {
        const unsigned m = a.peer_mask ? (unsigned)a.peer_mask[unit] : 0xffu;
#pragma unroll
        for (int q = 0; q < SDP_MAX_PEERS; ++q)
            if (((m >> q) & 1u) && a.peer_J[q]) ((real *)a.peer_J[q])[node] = v;
    }
}

// ---------------------------------------------------------------------------
// <int>(p) of the Cython source is cvttsd2si / cvttss2si on the reference's
// x86-64 build: NaN and out-of-range values give INT_MIN (then clamped to 0).
// ---------------------------------------------------------------------------
SDP_DEV int sdp_trunc_i32(double p) { return (fabs(p) < 2147483648.0) ? (int)p : INT_MIN; }
SDP_DEV int sdp_trunc_i32(float p) { return (fabsf(p) < 2147483648.0f) ? (int)p : INT_MIN; }

// `(s - smin) / span` of pyx:75 when the divisor is a power of two -- grids on [0, 1],
// [-4, 4], [0, 2^k] -- equals the product with its reciprocal BIT FOR BIT: x / 2^k and
// x * 2^-k are the same real number and both operations round it correctly (also into the
// subnormals).  One instruction instead of the ~16 of an IEEE division; any other
// divisor takes the true division.  `pow2` is wave-uniform: a scalar branch.
#ifndef SDP_NO_POW2
#define SDP_NO_POW2 0            // 1: always divide (A/B runs)
#endif
SDP_DEV bool sdp_is_pow2(double v)
{
    if (SDP_NO_POW2) return false;
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned e = (unsigned)(b >> 52);          // sign + exponent
    return (b & 0x000FFFFFFFFFFFFFull) == 0 && e >= 1 && e <= 2046;
}
SDP_DEV bool sdp_is_pow2(float v)
{
    if (SDP_NO_POW2) return false;
    const unsigned b = __float_as_uint(v);
    const unsigned e = b >> 23;
    return (b & 0x007FFFFFu) == 0 && e >= 1 && e <= 254;
}
template <typename real>
SDP_DEV real sdp_div_span(real num, real span, real rspan, bool pow2)
{
    if (pow2) return num * rspan;
    return num / span;
}

// Grid constants of one interpolation problem, kept in registers/SGPRs.
template <typename real, int D>
struct SdpGrid {
    real smin[D];
    real span[D];   // smax - smin              (pyx:75, denominator)
    real rspan[D];  // 1 / span, used only where span is a power of two (exact then)
    int pow2;       // bit k: span[k] is a power of two
    real nm1[D];    // (real)(order - 1)
    int ordm2[D];   // order - 2
    int M[D];       // C-order strides, M[D-1] = 1 (pyx:164-165)
    real shift;     // subtracted from every vertex value when SdpLerp<.., SHIFT = true>
};

// Per-point cell: integer offset of the lower corner along each axis and the
// barycentric weights.  `wide` is the type the lerp tree is evaluated in:
//   * the compiled reference evaluates `(1-lam)` with a C double literal
//     (Cython emits `1.0`), so its float specialisation runs the whole tree in
//     double except the innermost lam*v product, rounding to float once on the
//     store -> wide = double reproduces it bit for bit for float AND double;
//   * wide = real (pure float tree) is the fast variant of the fp32 sweep.
template <typename real, int D, typename wide = double>
struct SdpCell {
    int off[D];     // M[k] * q[k]
    real lam[D];
    wide oml[D];    // 1 - lam[k]
};

template <typename real, int D, typename wide>
SDP_DEV void sdp_locate_axis(const SdpGrid<real, D> &g, int k, real s, SdpCell<real, D, wide> &c)
{
    const real sn = sdp_div_span<real>(s - g.smin[k], g.span[k], g.rspan[k], (g.pow2 >> k) & 1);   // pyx:75
    const real p = sn * g.nm1[k];
    const int q = max(min(sdp_trunc_i32(p), g.ordm2[k]), 0);       // pyx:78
    c.lam[k] = p - (real)q;                                        // pyx:81 (unclamped)
    c.oml[k] = (wide)1 - (wide)c.lam[k];
    c.off[k] = g.M[k] * q;
}

// SHIFT = true reads every vertex as V[...] - g.shift: the relative-DP shift
// `J -= J[ref]` of the previous policy-evaluation step (stodynprog.py:760-762)
// applied at the point of use, one rounding per vertex exactly like the
// in-place subtraction of the reference, so the shifted array never has to be
// written out between two steps.
template <typename real, int D, typename wide, int K, bool SHIFT = false>
struct SdpLerp {
    static SDP_DEV wide eval(const real *__restrict__ V, const SdpGrid<real, D> &g,
                             const SdpCell<real, D, wide> &c, int base)
    {
        const wide lo = SdpLerp<real, D, wide, K + 1, SHIFT>::eval(V, g, c, base + c.off[K]);
        const wide hi = SdpLerp<real, D, wide, K + 1, SHIFT>::eval(V, g, c, base + c.off[K] + g.M[K]);
        return c.oml[K] * lo + (wide)c.lam[K] * hi;                // pyx:88,140,208,300
    }
};
// innermost axis: the two vertex loads; lam*v is a real x real product
template <typename real, int D, typename wide, bool SHIFT>
struct SdpLerp<real, D, wide, D - 1, SHIFT> {
    static SDP_DEV wide eval(const real *__restrict__ V, const SdpGrid<real, D> &g,
                             const SdpCell<real, D, wide> &c, int base)
    {
        real lo = V[base + c.off[D - 1]];
        real hi = V[base + c.off[D - 1] + g.M[D - 1]];
        if (SHIFT) { lo = lo - g.shift; hi = hi - g.shift; }
        return c.oml[D - 1] * (wide)lo + (wide)(c.lam[D - 1] * hi);
    }
};

template <typename real, int D, typename wide = double, bool SHIFT = false>
SDP_DEV real sdp_interp_point(const real *__restrict__ V, const SdpGrid<real, D> &g,
                              const real *pt)
{
    SdpCell<real, D, wide> c;
#pragma unroll
    for (int k = 0; k < D; ++k) sdp_locate_axis<real, D, wide>(g, k, pt[k], c);
    return (real)SdpLerp<real, D, wide, 0, SHIFT>::eval(V, g, c, 0);
}

template <typename real, int D>
SDP_DEV void sdp_make_grid(SdpGrid<real, D> &g, const int32_t *orders, const real *smin,
                           const real *smax)
{
    int m = 1;
    g.shift = (real)0;
    g.pow2 = 0;
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        g.smin[k] = smin[k];
        g.span[k] = smax[k] - smin[k];
        g.rspan[k] = (real)1 / g.span[k];
        if (sdp_is_pow2(g.span[k])) g.pow2 |= 1 << k;
        g.nm1[k] = (real)(orders[k] - 1);
        g.ordm2[k] = orders[k] - 2;
        g.M[k] = m;
        m *= orders[k];
    }
    g.pow2 = __builtin_amdgcn_readfirstlane(g.pow2);       // the same in every lane: keep it scalar
}

// ---------------------------------------------------------------------------
// numpy-semantics helpers for generated model code
// ---------------------------------------------------------------------------
template <typename real> SDP_DEV bool sdp_isnan(real a) { return a != a; }
// np.minimum / np.maximum propagate NaN (first NaN operand is returned)
template <typename real> SDP_DEV real sdp_npmin(real a, real b) { return (sdp_isnan(a) || a <= b) ? a : b; }
template <typename real> SDP_DEV real sdp_npmax(real a, real b) { return (sdp_isnan(a) || a >= b) ? a : b; }
// np.fmin / np.fmax ignore a NaN operand
template <typename real> SDP_DEV real sdp_npfmin(real a, real b) { return (sdp_isnan(b) || a <= b) ? a : b; }
template <typename real> SDP_DEV real sdp_npfmax(real a, real b) { return (sdp_isnan(b) || a >= b) ? a : b; }
template <typename real> SDP_DEV real sdp_npsign(real a)
{
    return a > (real)0 ? (real)1 : (a < (real)0 ? (real)-1 : (a == (real)0 ? (real)0 : a));
}
SDP_DEV double sdp_nppymod(double a, double b)
{
    double m = fmod(a, b);
    if (b != 0.0 && m != 0.0 && ((m < 0.0) != (b < 0.0))) m += b;
    else if (m == 0.0) m = copysign(0.0, b);
    return m;
}
SDP_DEV float sdp_nppymod(float a, float b)
{
    float m = fmodf(a, b);
    if (b != 0.0f && m != 0.0f && ((m < 0.0f) != (b < 0.0f))) m += b;
    else if (m == 0.0f) m = copysignf(0.0f, b);
    return m;
}
template <typename real> SDP_DEV real sdp_npfloordiv(real a, real b)
{
    // numpy npy_divmod: floor of the quotient, consistent with the Python-style remainder
    const real m0 = fmod(a, b);
    real div = (a - m0) / b;
    if (b != (real)0 && m0 != (real)0 && ((m0 < (real)0) != (b < (real)0))) div -= (real)1;
    if (div != (real)0) {
        real fl = floor(div);
        if (div - fl > (real)0.5) fl += (real)1;
        return fl;
    }
    return copysign((real)0, a / b);
}

// ---------------------------------------------------------------------------
// first-occurrence argmin (numpy argmin, stodynprog.py:686): strictly smaller
// wins, the first NaN wins over everything, equal values keep the lower index.
// ---------------------------------------------------------------------------
template <typename real>
SDP_DEV bool sdp_better_seq(real cand, real best)          // candidate has the HIGHER index
{
    return (cand < best) || (sdp_isnan(cand) && !sdp_isnan(best));
}
template <typename real>
SDP_DEV bool sdp_better_idx(real cv, int ci, real bv, int bi)
{
    const bool cn = sdp_isnan(cv), bn = sdp_isnan(bv);
    if (cn != bn) return cn;
    if (cn) return ci < bi;
    return (cv < bv) || (cv == bv && ci < bi);
}

// wavefront-wide integer min / max (DPP row operations, no LDS traffic); every lane gets the result
extern "C" __device__ int __ockl_wfred_min_i32(int);
extern "C" __device__ int __ockl_wfred_max_i32(int);
SDP_DEV int sdp_wave_min(int v) { return __ockl_wfred_min_i32(v); }
SDP_DEV int sdp_wave_max(int v) { return __ockl_wfred_max_i32(v); }

SDP_DEV double sdp_shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
SDP_DEV float sdp_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }

// Butterfly over the L (power of two, <= 64) consecutive lanes of a segment;
// every lane of the segment ends with the segment's (min value, first index).
template <typename real, int L>
SDP_DEV void sdp_seg_argmin(real &v, int &i)
{
#pragma unroll
    for (int m = 1; m < L; m <<= 1) {
        const real ov = sdp_shfl_xor(v, m);
        const int oi = __shfl_xor(i, m, 64);
        if (sdp_better_idx(ov, oi, v, i)) { v = ov; i = oi; }
    }
}
