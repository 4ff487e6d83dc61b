"""The error radii of the certified filters, checked in EXACT arithmetic (no GPU): the lean first pass of 8-byte reals
(section 3.1c), the short first passes of round 4 (8-byte, and the wide one of 4-byte reals), the wide first pass of
4-byte reals, the shifted lattice (3.1d) and the reduced array (3.3c) -- every form whose radius docs/NOTEBOOK.md derives to
first order with a factor to spare.  The first block below explains the method on the lean pass.


The filter of csrc/sdp_column_kernel.h skips a control when its short value F -- two fused operations on a table
reduced over the perturbation -- lies further than a radius above the smallest F of the node; that is sound if
|E - F| <= radius for every control, E being the reference's value (W x 6 separately rounded operations,
stodynprog.py:677-681 on multilinear_cython.pyx:88).  docs/NOTEBOOK.md section 3.1c derives the radius to first order in
the unit roundoff with a factor 2 to spare; the GPU tests back it empirically (same bits at the proven radius and at
half of it).  Here both values are computed with Python floats, operation by operation as the kernels round
(`fma` = one rounding of the exact a*b + c through fractions.Fraction), their difference is taken EXACTLY, and
compared with the radius the kernel would use -- on random tables of every magnitude regime the special-value
tests use short of overflow, on weights that are negative or do not sum to one, on positions far outside the grid
(extrapolation: |lam| >> 1), for the plain reduction order and for the resident-chunk one (tail first, then head).
VERDICT r03, "what's weak" 1(b).
"""
from fractions import Fraction

import numpy as np
import pytest

U = 2.0 ** -53
TINY = 2.2250738585072014e-308


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))          # correctly rounded (round-half-even)


def reference_value(T, p, g, q0, lam0):
    """E: sdp_col_expected_cost<1> / stodynprog.py:677-681, sequential in w"""
    oml0 = 1.0 - lam0
    acc = 0.0
    for w in range(len(p)):
        val = oml0 * T[w][q0] + lam0 * T[w][q0 + 1]
        jc = g + val
        acc = acc + jc * p[w]
    return acc


def reduced_table(T, p, chunked):
    """A[r] as sdp_col_filter_reduce accumulates it (acc + p_w * v, separately rounded), or as the resident-chunk
    kernel does: the tail's partial sum first, then the head, then their sum"""
    W, N0 = T.shape
    A = np.zeros(N0)
    for r in range(N0):
        if chunked:
            C = (W + 1) // 2
            tail = 0.0
            for w in range(C, W):
                tail = tail + p[w] * T[w][r]
            head = 0.0
            for w in range(C):
                head = head + p[w] * T[w][r]
            A[r] = head + tail
        else:
            acc = 0.0
            for w in range(W):
                acc = acc + p[w] * T[w][r]
            A[r] = acc
    return A


def filter_constants(p):
    """sdp_col_filter_setup"""
    ps = 0.0
    pa = 0.0
    for v in p:
        ps = ps + v
        pa = pa + abs(v)
    pcap = pa if pa > 1.0 else 1.0
    cu = 1.0 * float(2 * (len(p) + 8)) * 2.0 ** -52
    return dict(psum=ps, pcap=pcap, cu=cu, floor=2.0 * TINY / cu, ratio=pcap / abs(ps))


def node_check(T, p, controls, chunked):
    """controls: (position p along axis 0, cost g) of every control of one node.  Returns (worst |E - F| / radius,
    radius, scale) with everything but the final division exact."""
    W, N0 = T.shape
    fc = filter_constants(p)
    A = reduced_table(T, p, chunked)
    dcol = 0.0
    for r in range(N0):
        big = max(abs(T[w][r]) for w in range(W))
        dcol = max(dcol, fc['pcap'] * big + fc['floor'])
    F, E, lmax, fsum = [], [], 0.0, 0.0
    for pos, g in controls:
        q0 = max(min(int(pos), N0 - 2), 0)                 # pyx:78 (trunc, clamp)
        lam0 = pos - float(q0)                             # pyx:81
        lmax = max(lmax, abs(lam0))
        f = fma(g, fc['psum'], fma(lam0, A[q0 + 1] - A[q0], A[q0]))     # sdp_col_lean_core
        fsum = fsum + abs(f)
        F.append(f)
        E.append(reference_value(T, p, g, q0, lam0))
    h_cap = (1.0 + 2.0 * lmax) * dcol
    s_node = fma(fc['ratio'], fsum + h_cap, h_cap)
    radius = fc['cu'] * s_node
    worst = max(abs(Fraction(e) - Fraction(f)) for e, f in zip(E, F))
    return float(worst / Fraction(radius)), radius, s_node


def random_problem(rng, regime):
    W = int(rng.integers(1, 9))
    N0 = int(rng.integers(3, 14))
    T = rng.standard_normal((W, N0))
    if regime == 'large':
        T *= 10.0 ** rng.uniform(100, 290)
    elif regime == 'small':
        T *= 10.0 ** rng.uniform(-300, -100)
    elif regime == 'mixed':
        T *= 10.0 ** rng.uniform(-12, 12, size=T.shape)
    elif regime == 'cancel':
        T = 1e6 + 1e-6 * T                                  # A[q0+1] - A[q0] cancels almost completely
    p = np.abs(rng.standard_normal(W)) + 1e-3
    p /= p.sum()
    if regime == 'weights':
        p = rng.standard_normal(W) * 3.7                   # negative weights, a sum far from one
        if abs(p.sum()) < 0.2:
            p[0] += 1.0
    scale = float(np.abs(T).max())
    n_u = int(rng.integers(2, 12))
    controls = []
    for _ in range(n_u):
        kind = rng.integers(0, 4)
        if kind == 0:
            pos = float(rng.uniform(0, N0 - 1))
        elif kind == 1:
            pos = float(rng.integers(0, N0))               # exactly on a node
        elif kind == 2:
            pos = float(rng.uniform(-3 * N0, 4 * N0))      # extrapolation: |lam| up to ~3 N0
        else:
            pos = float(rng.uniform(0, N0 - 1)) * (1.0 + U)
        g = float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-3, 3)
        controls.append((pos, g))
    return T, p, controls


@pytest.mark.parametrize('regime', ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'])
@pytest.mark.parametrize('chunked', [False, True])
def test_the_radius_covers_the_difference_exactly(regime, chunked):
    rng = np.random.default_rng(['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'].index(regime) * 2 + int(chunked))
    worst = 0.0
    for _ in range(300):
        T, p, controls = random_problem(rng, regime)
        ratio, radius, s_node = node_check(T, p, controls, chunked)
        assert np.isfinite(radius) and radius > 0.0
        assert ratio <= 1.0, (regime, chunked, ratio, T.shape, len(controls))
        worst = max(worst, ratio)
    # the bound is meant to be generous (a factor ~2 over first order, a node-wide sum in place of a per-control
    # term): the worst case seen stays well inside it -- and is not absurdly far from it either
    assert worst < 0.6, worst


def test_a_radius_a_thousand_times_smaller_would_not_cover_it():
    """(the check above has teeth: the same difference against 1e-3 of the radius fails somewhere)"""
    rng = np.random.default_rng(7)
    seen = 0.0
    for _ in range(300):
        T, p, controls = random_problem(rng, 'cancel')
        ratio, _, _ = node_check(T, p, controls, False)
        seen = max(seen, ratio)
    assert seen > 1e-3, seen


# ---------------------------------------------------------------------------
# The short first pass of the resident-chunk kernel (csrc/sdp_colres_kernel.h, SDP_COL_LEAN2): x0' = X + a(u),
# cost = K + h(u).  It orders F'' = pack(fma(h, psum, lerp), index) -- an approximation of E - K P*, P* the EXACT sum
# of the weights -- and bounds the node by S = Pcap (|K| + max |h|) + (1 + 2 L) D with L from the column's smallest and
# largest a.  Checked here: |E - K P* - F''| <= (cu + 2^(bits+2) u) S for every control, exactly.
# ---------------------------------------------------------------------------
import struct


def pack_index(F, ci, mask):
    (b,) = struct.unpack('<q', struct.pack('<d', F))
    lo = (b & 0xffffffff & ~mask) | ci
    (out,) = struct.unpack('<d', struct.pack('<q', (b & ~0xffffffff) | lo))
    return out


def cell_of(xn0, nm1, n0):
    p = xn0 * nm1                                            # (grid [0, 1]: axis mode 2; pyx:75)
    q0 = max(min(int(p), n0 - 2), 0)                         # pyx:78
    return q0, p - float(q0)                                 # pyx:81


def short_pass_check(T, p, X, K, a, h, chunked, sign=1.0):
    """a, h: the control table of the column (one entry per control).  Returns (worst |E - K P* - F''| / radius, radius)."""
    W, N0 = T.shape
    fc = filter_constants(p)
    A = reduced_table(T, p, chunked)
    nm1 = float(N0 - 1)
    dcol = 0.0
    for r in range(N0):
        big = max(abs(T[w][r]) for w in range(W))
        dcol = max(dcol, fc['pcap'] * big + fc['floor'])
    n = len(a)
    bits = max((n - 1).bit_length(), 1)
    mask = (1 << bits) - 1
    lam_lo = cell_of(X + min(a), nm1, N0)[1]
    lam_hi = cell_of(X + max(a), nm1, N0)[1]
    L = max(1.0, abs(lam_lo), abs(lam_hi))
    s_node = fma(fc['pcap'], abs(K) + max(abs(v) for v in h), (1.0 + 2.0 * L) * dcol)
    radius = (fc['cu'] + 2.0 ** (bits + 1 - 52)) * s_node
    p_exact = sum(Fraction(v) for v in p)
    worst = Fraction(0)
    for ci in range(n):
        q0, lam0 = cell_of(X + a[ci], nm1, N0)
        assert abs(lam0) <= L
        g = K + sign * h[ci]                                 # the reference's cost, one rounding
        E = reference_value(T, p, g, q0, lam0)
        F = fma(sign * h[ci], fc['psum'], fma(lam0, A[q0 + 1] - A[q0], A[q0]))
        Fp = pack_index(F, ci, mask)
        assert abs(Fraction(Fp) - Fraction(F)) <= Fraction(2.0 ** (bits + 1 - 53)) * Fraction(s_node) * (1 + Fraction(3, 2 ** 53))
        worst = max(worst, abs(Fraction(E) - Fraction(K) * p_exact - Fraction(Fp)))
    return float(worst / Fraction(radius)), radius


@pytest.mark.parametrize('regime', ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'])
@pytest.mark.parametrize('chunked', [False, True])
def test_the_short_pass_radius_covers_the_difference_exactly(regime, chunked):
    rng = np.random.default_rng(100 + ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'].index(regime) * 2 + int(chunked))
    worst = 0.0
    for trial in range(250):
        T, p, _ = random_problem(rng, regime)
        W, N0 = T.shape
        scale = float(np.abs(T).max())
        n = int(rng.integers(1, 70))
        X = float(rng.uniform(0, 1))
        spread = float(10.0 ** rng.uniform(-3, 0.7))          # controls within a cell ... far outside the grid
        a = [float(v) for v in rng.uniform(-spread, spread, size=n)]
        if trial % 5 == 0:
            a[0] = float(rng.integers(0, N0)) / (N0 - 1) - X  # (about) exactly on a node
        K = float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-3, 3)
        h = [float(v) * scale * 10.0 ** rng.uniform(-3, 3) for v in rng.standard_normal(n)]
        sign = -1.0 if trial % 3 == 0 else 1.0                # cost = K - h
        ratio, radius = short_pass_check(T, p, X, K, a, h, chunked, sign)
        assert np.isfinite(radius) and radius > 0.0
        assert ratio <= 1.0, (regime, chunked, trial, ratio)
        worst = max(worst, ratio)
    assert worst < 0.8, worst


BNB_DELTA = 2.0 ** -20


def block_bound_check(T, p, X, K, a, h, chunked, sign, block, lead='add'):
    """The branch and bound of the short first pass (sdp_short_bnb of csrc/sdp_colfilter_kernel.h, round 5), as the kernels
    evaluate it.  The control table's wave keeps per block of `block` controls where it STARTS -- the smallest
    pa = +-a k of its controls, k = (N0 - 1) / span, moved down by DELTA -- and the smallest fl(+-h psum); the last
    block's end is its largest pa + DELTA (sdp_col_phase_u).  A node adds pX = +-(X -+ smin) k, evaluates the
    interpolant L of the reduced table at the n_blocks + 1 ends (clamped cell, unclamped lam0) and bounds block b by
        LB = hp_b + min( L(P_b), L(P_b+1), A[q_b + 1], A[q_b+1], the rows q_b + 2 .. q_b+1 - 1 ).
    Only where the blocks are in order (each ends before the next starts).  Checked here, exactly: LB lies below the
    PACKED F' of every control of the block, up to the 16 u S_node the skip test allows for.  Returns the largest
    (LB - F') / (16 u S_node) seen (<= 1 keeps every skipped control above f1 + 2 radius), or None when the blocks
    are not in order (the kernels then take the full pass)."""
    W, N0 = T.shape
    fc = filter_constants(p)
    A = reduced_table(T, p, chunked)
    nm1 = float(N0 - 1)                                       # (grid [0, 1]: smin = 0, span = 1, k = N0 - 1)
    dcol = max(fc['pcap'] * max(abs(T[w][r]) for w in range(W)) + fc['floor'] for r in range(N0))
    n = len(a)
    bits = max((n - 1).bit_length(), 1)
    mask = (1 << bits) - 1
    where = {'add': lambda v: X + v, 'sub': lambda v: X - v, 'rsub': lambda v: v - X}[lead]
    lam_lo = cell_of(where(min(a)), nm1, N0)[1]
    lam_hi = cell_of(where(max(a)), nm1, N0)[1]
    L = max(1.0, abs(lam_lo), abs(lam_hi))
    s_node = fma(fc['pcap'], abs(K) + max(abs(v) for v in h), (1.0 + 2.0 * L) * dcol)
    # the records of the table's wave
    blocks = [range(b0, min(b0 + block, n)) for b0 in range(0, n, block)]
    pa = [(-v if lead == 'sub' else v) * nm1 for v in a]
    lo = [min(pa[c] for c in m) for m in blocks]
    hi = [max(pa[c] for c in m) for m in blocks]
    if any(hi[b] + 2 * BNB_DELTA > lo[b + 1] for b in range(len(blocks) - 1)):     # (each block ends 2 DELTA before the next starts)
        return None
    starts = [v - BNB_DELTA for v in lo] + [hi[-1] + BNB_DELTA]
    hp = [min((sign * h[c]) * fc['psum'] for c in m) for m in blocks]
    # the node
    pX = -(X * nm1) if lead == 'rsub' else X * nm1
    assert (abs(X) + max(abs(v) for v in a)) < 2.0 ** 30 / nm1  # (what the kernels check before they rely on these positions)

    def at(P):
        q = max(min(int(P), N0 - 2), 0)
        return q, fma(P - float(q), A[q + 1] - A[q], A[q])
    ends = [at(pX + v) for v in starts]
    worst = Fraction(-10 ** 9)
    for b, members in enumerate(blocks):
        (qa, La), (qb, Lb) = ends[b], ends[b + 1]
        m = min(La, Lb, A[qa + 1], A[qb])
        for r in range(qa + 2, qb):
            m = min(m, A[r])
        lbv = hp[b] + m
        for ci in members:
            q0, lam0 = cell_of(where(a[ci]), nm1, N0)
            # the kernel's own position of the control lies inside the block's interval of the bounds
            assert pX + starts[b] <= float(q0) + lam0 <= pX + starts[b + 1] or q0 in (0, N0 - 2)
            F = fma(sign * h[ci], fc['psum'], fma(lam0, A[q0 + 1] - A[q0], A[q0]))
            Fp = pack_index(F, ci, mask)
            # (the packing itself is inside the RADIUS: here the unpacked value decides, plus the packing's own bound)
            worst = max(worst, (Fraction(lbv) - Fraction(F)) / (16 * Fraction(U) * Fraction(s_node)))
            assert abs(Fraction(Fp) - Fraction(F)) <= Fraction(2.0 ** (bits + 1 - 53)) * Fraction(s_node) * (1 + Fraction(3, 2 ** 53))
    return float(worst)


@pytest.mark.parametrize('regime', ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'])
@pytest.mark.parametrize('lead', ['add', 'sub', 'rsub'])
def test_the_block_bound_of_the_branch_and_bound_lies_below_every_control_of_its_block(regime, lead):
    rng = np.random.default_rng(300 + ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'].index(regime) * 3
                                + ['add', 'sub', 'rsub'].index(lead))
    worst, checked = -1e9, 0
    for trial in range(120):
        T, p, _ = random_problem(rng, regime)
        W, N0 = T.shape
        scale = float(np.abs(T).max())
        n = int(rng.integers(1, 90))
        X = float(rng.uniform(0, 1))
        spread = float(10.0 ** rng.uniform(-3, 0.7))          # controls within a cell ... far outside the grid
        # an ordinary lattice: a monotone in the control, steps of at least 4 DELTA rows
        a = [float(v) for v in -spread + np.cumsum(rng.uniform(0.1, 1.0, size=n)) * (2 * spread / n) + np.arange(n) * 4 * BNB_DELTA / (N0 - 1)]
        if lead == 'sub':
            a = a[::-1]                                       # (x0' = X - a: the positions still run upwards)
        if trial % 7 == 3:
            a = [float(v) for v in rng.permutation(a)]        # any order: blocks that overlap -- no branch and bound there
        if lead == 'rsub':
            a = [v + 2 * X for v in a]
        K = float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-3, 3)
        h = [float(v) * scale * 10.0 ** rng.uniform(-3, 3) for v in rng.standard_normal(n)]
        sign = -1.0 if trial % 3 == 0 else 1.0
        ratio = block_bound_check(T, p, X, K, a, h, bool(trial % 4 == 1), sign, 8 if trial % 5 else 16, lead)
        if ratio is None:
            assert trial % 7 == 3 and n > 8
            continue
        checked += 1
        assert ratio <= 1.0, (regime, lead, trial, ratio)
        worst = max(worst, ratio)
    assert checked >= 90 and worst < 0.5, (checked, worst)     # (a handful of roundings against sixteen)


def test_the_packed_index_comes_back_and_the_order_survives():
    rng = np.random.default_rng(5)
    for _ in range(2000):
        F = float(rng.standard_normal()) * 10.0 ** rng.uniform(-300, 300)
        bits = int(rng.integers(1, 14))
        ci = int(rng.integers(0, 1 << bits))
        Fp = pack_index(F, ci, (1 << bits) - 1)
        (b,) = struct.unpack('<q', struct.pack('<d', Fp))
        assert (b & ((1 << bits) - 1)) == ci
        assert abs(Fp - F) <= 2.0 ** bits * abs(F) * 2.0 ** -52
        assert (Fp < 0) == (F < 0)


# ---------------------------------------------------------------------------
# The short WIDE first pass (4-byte reals, SDP_COL_WIDE2 of csrc/sdp_colfilter_kernel.h): the reference runs in 4-byte
# arithmetic, F' in 8-byte arithmetic on A[r] accumulated in 8-byte reals, one bound per node:
#     S = (Gc + Pcap) [ |K| + max |h| + (1 + 2 L) max |T| ],   radius = 1.001 2^-24 (S + floor) + 2^(bits-51) S
# ---------------------------------------------------------------------------
f32 = np.float32


def reference_value32(T, p, g, q0, lam0):
    """the reference's operations on 4-byte reals (multilinear_cython.pyx with floats; stodynprog.py:677-681)"""
    oml0 = f32(f32(1) - lam0)
    acc = f32(0)
    for w in range(len(p)):
        val = f32(f32(oml0 * T[w][q0]) + f32(lam0 * T[w][q0 + 1]))
        jc = f32(g + val)
        acc = f32(acc + f32(jc * p[w]))
    return acc


def wide_short_check(T, p, X, K, a, h, sign=1.0):
    W, N0 = T.shape
    nm1 = f32(N0 - 1)
    # sdp_col_filter_setup / sdp_col_filter_reduce, wide form
    pa, gc, ps64 = f32(0), f32(0), 0.0
    for w in range(W):
        pa = f32(pa + abs(p[w]))
        nw = f32(W + 3 if w == 0 else W - w + 4)
        gc = f32(gc + f32(nw * abs(p[w])))
        ps64 = ps64 + float(p[w])
    gc = f32(gc * f32(1.0001))
    pcap = pa if pa > f32(1) else f32(1)
    cu = f32(f32(2 * (W + 8)) * f32(2.0 ** -23))
    floor = f32(f32(2) * f32(1.17549435e-38) / cu)
    A = np.zeros(N0)
    for r in range(N0):
        acc = 0.0
        for w in range(W):
            acc = acc + float(p[w]) * float(T[w][r])          # (the product of two 4-byte reals is exact in 8 bytes)
        A[r] = acc
    tmax = f32(np.abs(T).max())

    def cell(xn0):
        pos = f32(xn0 * nm1)
        q0 = max(min(int(pos), N0 - 2), 0)
        return q0, f32(pos - f32(q0))
    n = len(a)
    bits = max((n - 1).bit_length(), 1)
    mask = (1 << bits) - 1
    lam_lo, lam_hi = cell(f32(X + min(a)))[1], cell(f32(X + max(a)))[1]
    L = max(f32(1), abs(lam_lo), abs(lam_hi))
    habs = max(abs(v) for v in h)
    s_node = f32(f32(gc + pcap) * f32(f32(abs(K) + habs) + f32(f32(f32(1) + f32(f32(2) * L)) * tmax)))
    radius = 1.001 * 2.0 ** -24 * (float(s_node) + float(floor)) + float(s_node) * 2.0 ** (bits - 51)
    p_exact = sum(Fraction(float(v)) for v in p)
    worst = Fraction(0)
    for ci in range(n):
        q0, lam0 = cell(f32(X + a[ci]))
        assert abs(lam0) <= L
        g = f32(K + f32(sign) * h[ci])
        E = reference_value32(T, p, g, q0, lam0)
        F = fma(float(f32(sign) * h[ci]), ps64, fma(float(lam0), A[q0 + 1] - A[q0], A[q0]))
        Fp = pack_index(F, ci, mask)
        worst = max(worst, abs(Fraction(float(E)) - Fraction(float(K)) * p_exact - Fraction(Fp)))
    return float(worst / Fraction(radius)), radius


WIDE_DELTA = f32(2.0 ** -8)


def wide_block_bound_check(T, p, X, K, a, h, sign, block):
    """The branch and bound of the short WIDE first pass (4-byte reals: sdp_short_bnb<.., true>).  Positions and the control
    table are 4-byte reals -- the blocks' starts pa - DELTA, the node's pX, their sum --, the reduced table A[r], the
    smallest +-h psum of a block, L at the ends and F' are 8-byte reals; DELTA = 2^-8 rows covers the 4-byte roundings of
    the kernel's own positions.  Exactly: every control's own (4-byte) position lies inside its block's interval of
    the bounds, and LB lies below its F' up to the 2^-40 S_node the skip test allows for.  Returns the largest
    (LB - F') / (2^-40 S_node), or None when the blocks are not 2 DELTA apart (no branch and bound then)."""
    W, N0 = T.shape
    nm1 = f32(N0 - 1)
    pa_w, gc, ps64 = f32(0), f32(0), 0.0
    for w in range(W):
        pa_w = f32(pa_w + abs(p[w]))
        nw = f32(W + 3 if w == 0 else W - w + 4)
        gc = f32(gc + f32(nw * abs(p[w])))
        ps64 = ps64 + float(p[w])
    gc = f32(gc * f32(1.0001))
    pcap = pa_w if pa_w > f32(1) else f32(1)
    A = np.zeros(N0)
    for r in range(N0):
        acc = 0.0
        for w in range(W):
            acc = acc + float(p[w]) * float(T[w][r])
        A[r] = acc
    tmax = f32(np.abs(T).max())

    def cell(xn0):
        pos = f32(xn0 * nm1)
        q0 = max(min(int(pos), N0 - 2), 0)
        return q0, f32(pos - f32(q0))
    n = len(a)
    bits = max((n - 1).bit_length(), 1)
    mask = (1 << bits) - 1
    lam_lo, lam_hi = cell(f32(X + min(a)))[1], cell(f32(X + max(a)))[1]
    L = max(f32(1), abs(lam_lo), abs(lam_hi))
    habs = max(abs(v) for v in h)
    s_node = f32(f32(gc + pcap) * f32(f32(abs(K) + habs) + f32(f32(f32(1) + f32(f32(2) * L)) * tmax)))
    # the records of the table's wave (grid [0, 1]: k = N0 - 1)
    blocks = [range(b0, min(b0 + block, n)) for b0 in range(0, n, block)]
    pa = [f32(v * nm1) for v in a]
    lo = [min(pa[c] for c in m) for m in blocks]
    hi = [max(pa[c] for c in m) for m in blocks]
    if any(f32(hi[b] + f32(2) * WIDE_DELTA) > lo[b + 1] for b in range(len(blocks) - 1)):
        return None
    starts = [f32(v - WIDE_DELTA) for v in lo] + [f32(hi[-1] + WIDE_DELTA)]
    hp = [min(float(f32(sign) * h[c]) * ps64 for c in m) for m in blocks]
    assert float(abs(X)) + float(max(abs(v) for v in a)) < 2.0 ** 13 / float(nm1)     # (the kernels' own condition)
    pX = f32(X * nm1)

    def at(P32):
        P = float(P32)
        q = max(min(int(P), N0 - 2), 0)
        return q, fma(P - float(q), A[q + 1] - A[q], A[q])
    ends = [at(f32(pX + v)) for v in starts]
    worst = Fraction(-10 ** 9)
    for b, members in enumerate(blocks):
        (qa, La), (qb, Lb) = ends[b], ends[b + 1]
        m = min(La, Lb, A[qa + 1], A[qb])
        for r in range(qa + 2, qb):
            m = min(m, A[r])
        lbv = hp[b] + m
        for ci in members:
            q0, lam0 = cell(f32(X + a[ci]))
            assert float(f32(pX + starts[b])) <= float(q0) + float(lam0) <= float(f32(pX + starts[b + 1])) or q0 in (0, N0 - 2), \
                (float(f32(pX + starts[b])), float(q0) + float(lam0), float(f32(pX + starts[b + 1])))
            F = fma(float(f32(sign) * h[ci]), ps64, fma(float(lam0), A[q0 + 1] - A[q0], A[q0]))
            worst = max(worst, (Fraction(lbv) - Fraction(F)) / (Fraction(2.0 ** -40) * Fraction(float(s_node))))
    return float(worst)


@pytest.mark.parametrize('regime', ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'])
def test_the_block_bound_of_the_wide_branch_and_bound_lies_below_every_control_of_its_block(regime):
    rng = np.random.default_rng(500 + ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'].index(regime))
    worst, checked = -1e9, 0
    for trial in range(100):
        W = int(rng.integers(1, 9))
        N0 = int(rng.integers(3, 14))
        T = rng.standard_normal((W, N0))
        if regime == 'large':
            T *= 10.0 ** rng.uniform(10, 28)
        elif regime == 'small':
            T *= 10.0 ** rng.uniform(-30, -10)
        elif regime == 'mixed':
            T *= 10.0 ** rng.uniform(-4, 4, size=T.shape)
        elif regime == 'cancel':
            T = 1e3 + 1e-2 * T
        T = T.astype(f32)
        p = np.abs(rng.standard_normal(W)) + 1e-3
        p /= p.sum()
        if regime == 'weights':
            p = rng.standard_normal(W) * 3.7
            if abs(p.sum()) < 0.2:
                p[0] += 1.0
        p = p.astype(f32)
        scale = float(np.abs(T).max())
        n = int(rng.integers(1, 70))
        X = f32(rng.uniform(0, 1))
        spread = float(10.0 ** rng.uniform(-1.5, 0.5))
        # an ordinary lattice: a monotone in the control, steps of at least 6 DELTA rows
        a = [f32(v) for v in -spread + np.cumsum(rng.uniform(0.1, 1.0, size=n)) * (2 * spread / n) + np.arange(n) * 6 * float(WIDE_DELTA) / (N0 - 1)]
        K = f32(float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-2, 2))
        h = [f32(float(v) * scale * 10.0 ** rng.uniform(-2, 2)) for v in rng.standard_normal(n)]
        sign = -1.0 if trial % 3 == 0 else 1.0
        ratio = wide_block_bound_check(T, p, X, K, a, h, sign, 8 if trial % 5 else 16)
        if ratio is None:
            continue
        checked += 1
        assert ratio <= 1.0, (regime, trial, ratio)
        worst = max(worst, ratio)
    assert checked >= 80 and worst < 0.5, (checked, worst)


@pytest.mark.parametrize('regime', ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'])
def test_the_short_wide_pass_radius_covers_the_difference_exactly(regime):
    rng = np.random.default_rng(200 + ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'].index(regime))
    worst = 0.0
    with np.errstate(all='ignore'):
        for trial in range(250):
            W = int(rng.integers(1, 9))
            N0 = int(rng.integers(3, 14))
            T = rng.standard_normal((W, N0))
            if regime == 'large':
                T *= 10.0 ** rng.uniform(10, 28)
            elif regime == 'small':
                T *= 10.0 ** rng.uniform(-30, -10)
            elif regime == 'mixed':
                T *= 10.0 ** rng.uniform(-4, 4, size=T.shape)
            elif regime == 'cancel':
                T = 1e3 + 1e-2 * T
            T = T.astype(f32)
            p = np.abs(rng.standard_normal(W)) + 1e-3
            p /= p.sum()
            if regime == 'weights':
                p = rng.standard_normal(W) * 3.7
                if abs(p.sum()) < 0.2:
                    p[0] += 1.0
            p = p.astype(f32)
            scale = float(np.abs(T).max())
            n = int(rng.integers(1, 70))
            X = f32(rng.uniform(0, 1))
            spread = float(10.0 ** rng.uniform(-3, 0.7))
            a = [f32(v) for v in rng.uniform(-spread, spread, size=n)]
            K = f32(float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-3, 3))
            h = [f32(float(v) * scale * 10.0 ** rng.uniform(-3, 3)) for v in rng.standard_normal(n)]
            sign = -1.0 if trial % 3 == 0 else 1.0
            ratio, radius = wide_short_check(T, p, X, K, a, h, sign)
            assert np.isfinite(radius) and radius > 0.0
            assert ratio <= 1.0, (regime, trial, ratio)
            worst = max(worst, ratio)
    assert worst < 0.9, worst


# ---------------------------------------------------------------------------
# The wide first pass of section 3.1c (4-byte reals, models WITHOUT the additive shape, per-node control boxes):
#     F = fma(g, P, fma(oml0, A[q0], lam0 A[q0+1]))   in 8-byte arithmetic, q0 / lam0 / oml0 / g the reference's 4-byte values
#     bound(u) = Gc |g| + |oml0| B[q0] + |lam0| B[q0+1],   B[r] = sum_w n_w |p_w| |T[w][r]|,   n_w = W - w + 4 (W + 3 for w = 0)
#     radius = 1.001 2^-24 (max_u bound(u) + floor)
# (sdp_col_wide_core, sdp_col_filter_reduce, sdp_col_filter_nodes of csrc/sdp_colfilter_kernel.h)
# ---------------------------------------------------------------------------
def wide_check(T, p, controls):
    """controls: (position p as the reference computes it in 4-byte reals, cost g) per control"""
    W, N0 = T.shape
    gc, ps64 = f32(0), 0.0
    nw = [f32(W + 3 if w == 0 else W - w + 4) for w in range(W)]
    for w in range(W):
        gc = f32(gc + f32(nw[w] * abs(p[w])))
        ps64 = ps64 + float(p[w])
    gc = f32(gc * f32(1.0001))
    cu = f32(f32(2 * (W + 8)) * f32(2.0 ** -23))
    floor = f32(f32(2) * f32(1.17549435e-38) / cu)
    A, B = np.zeros(N0), np.zeros(N0, dtype=f32)
    for r in range(N0):
        acc, bs = 0.0, f32(0)
        for w in range(W):
            acc = acc + float(p[w]) * float(T[w][r])
            bs = f32(np.float64(f32(nw[w] * abs(p[w]))) * np.float64(abs(T[w][r])) + np.float64(bs))     # fma in 4-byte reals
        A[r], B[r] = acc, bs
    p_exact = sum(Fraction(float(v)) for v in p)
    smax, worst = f32(0), []
    for pos, g in controls:
        q0 = max(min(int(pos), N0 - 2), 0)
        lam0 = f32(pos - f32(q0))
        oml0 = f32(f32(1) - lam0)
        E = reference_value32(T, p, g, q0, lam0)
        F = fma(float(g), ps64, fma(float(oml0), A[q0], float(lam0) * A[q0 + 1]))
        inner = f32(np.float64(abs(oml0)) * np.float64(B[q0]) + np.float64(f32(abs(lam0) * B[q0 + 1])))
        bound = f32(np.float64(abs(g)) * np.float64(gc) + np.float64(inner))
        smax = max(smax, bound)
        worst.append(abs(Fraction(float(E)) - Fraction(F)))
    radius = 1.001 * 2.0 ** -24 * (float(smax) + float(floor))
    return float(max(worst) / Fraction(radius)), radius


@pytest.mark.parametrize('regime', ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'])
def test_the_wide_pass_radius_covers_the_difference_exactly(regime):
    rng = np.random.default_rng(300 + ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'].index(regime))
    worst = 0.0
    with np.errstate(all='ignore'):
        for trial in range(250):
            W = int(rng.integers(1, 9))
            N0 = int(rng.integers(3, 14))
            T = rng.standard_normal((W, N0))
            if regime == 'large':
                T *= 10.0 ** rng.uniform(10, 28)
            elif regime == 'small':
                T *= 10.0 ** rng.uniform(-30, -10)
            elif regime == 'mixed':
                T *= 10.0 ** rng.uniform(-4, 4, size=T.shape)
            elif regime == 'cancel':
                T = 1e3 + 1e-2 * T
            T = T.astype(f32)
            p = np.abs(rng.standard_normal(W)) + 1e-3
            p /= p.sum()
            if regime == 'weights':
                p = rng.standard_normal(W) * 3.7
                if abs(p.sum()) < 0.2:
                    p[0] += 1.0
            p = p.astype(f32)
            scale = float(np.abs(T).max())
            controls = []
            for _ in range(int(rng.integers(1, 12))):
                kind = rng.integers(0, 3)
                pos = f32(rng.uniform(0, N0 - 1) if kind == 0 else (rng.integers(0, N0) if kind == 1 else rng.uniform(-3 * N0, 4 * N0)))
                controls.append((pos, f32(float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-3, 3))))
            ratio, radius = wide_check(T, p, controls)
            assert np.isfinite(radius) and radius > 0.0
            assert ratio <= 1.0, (regime, trial, ratio)
            worst = max(worst, ratio)
    assert 0.01 < worst < 0.9, worst          # (tight as intended: the 4-byte radius follows the reference's roundings one by one)


# ---------------------------------------------------------------------------
# The filter on the SHIFTED LATTICE (8-byte reals; a perturbation that reaches the stock through a final sum:
# x0' = (X + a_u) - b_w; docs/NOTEBOOK.md section 3.1d; sdp_col_phase_shift / sdp_col_shift_col / sdp_col_shift_reduce /
# sdp_col_lean_core of csrc/sdp_colfilter_kernel.h).  Here the filter value differs from the reference's even in exact
# arithmetic -- G is tabulated at whole positions and interpolated -- and the radius is cu S_node + max B'[q0]:
# both parts are checked together, exactly.  Grid [0, 1] (axis mode 2), one term b.
# ---------------------------------------------------------------------------
def shifted_check(T, p, X, a, g, b, order, chain=None, leaf_sum=True):
    """a, g: per control; b: per perturbation point (x0' = (X + a_u) - b_w); order: the order in which the partial
    sums of the perturbation points reach the lattice (LDS atomics: any order is possible).
    chain: the REFERENCE adds the same three leaves in another nesting -- 'x+(w-u)': X + ((-b_w) - (-a_u)),
    '(x-w)+u': (X - b_w) + a_u -- and the pass regroups them (TracedModel.lead_split, SDP_COL_SHIFT_CHAIN):
    its a is still fl(X + a_u), and the sum of the magnitudes of the w-free leaves enters the bound through L."""
    W, N0 = T.shape
    nm1 = float(N0 - 1)
    fc = filter_constants(p)
    # sdp_col_phase_shift
    q_w, f_w, c_w, pbabs = [], [], [], 0
    for w in range(W):
        pb = (-b[w]) * nm1                                   # the signed sum of the b_i, in rows
        fl = float(np.floor(pb))
        q_w.append(int(fl))
        f_w.append(pb - fl)                                  # exact, in [0, 1)
        c_w.append(abs(p[w]) * (f_w[-1] * (1.0 - f_w[-1])))
        pbabs = max(pbabs, int(abs(abs(b[w]) * nm1)) + 1)
    flmax, nflmin = max(q_w), max(-q for q in q_w)
    kmin, rows = -(flmax + 1), N0 + flmax + nflmin + 1
    pbmax = float(max(abs(flmax), abs(nflmin), pbabs) + 1)
    lc = float(rows) + float(abs(kmin)) + pbmax + 0.0 + float(N0 + 1)
    es = float(1 + 2 * (flmax + nflmin + 2))
    # sdp_col_shift_reduce (general branch: the clamps are no-ops inside the axis)
    Ap, Bp, big = np.zeros(rows), np.zeros(rows), 0.0
    for ki in range(rows):
        k = kmin + ki
        for w in order:
            j = k + q_w[w]
            q = max(min(j, N0 - 2), 0)
            lam = float(j - q) + f_w[w]
            t0, t1 = T[w][q], T[w][q + 1]
            t2 = T[w][q + 2] if q + 2 < N0 else T[w][q + 1]
            Ap[ki] = fma(p[w], fma(lam, t1 - t0, t0), Ap[ki])
            d2 = (t2 - t1) - (t1 - t0)
            Bp[ki] = fma(c_w[w], abs(d2) if 0 <= j <= N0 - 3 else 0.0, Bp[ki])
            big = max(big, abs(t0), abs(t1))
    dcol = fc['pcap'] * big + fc['floor']
    # first pass: sdp_col_lean_core on the lattice
    F, E, lmax, fsum, bmax = [], [], 0.0, 0.0, 0.0
    for u in range(len(a)):
        xa = X + a[u]                                        # sdp_model_lead_a
        pk = xa * nm1 - float(kmin)
        q0 = max(min(int(pk), rows - 2), 0)
        lam0 = pk - float(q0)
        lmax = max(lmax, abs(lam0))
        if chain and leaf_sum:
            lmax = max(lmax, (abs(X) + abs(a[u])) * (abs(nm1 * 1.0) * 1.002))      # sdp_model_lead_aabs x the rows per unit (span 1)
        bmax = max(bmax, Bp[q0])
        f = fma(g[u], fc['psum'], fma(lam0, Ap[q0 + 1] - Ap[q0], Ap[q0]))
        fsum = fsum + abs(f)
        F.append(f)
        # the reference: its own position per perturbation point (pyx:75-81), stodynprog.py:677-681
        acc = 0.0
        for w in range(W):
            if chain == 'x+(w-u)':
                s = (X + ((-b[w]) - (-a[u]))) * nm1
            elif chain == '(x-w)+u':
                s = ((X - b[w]) + a[u]) * nm1
            else:
                s = (xa - b[w]) * nm1
            qr = max(min(int(s), N0 - 2), 0)
            lr = s - float(qr)
            val = (1.0 - lr) * T[w][qr] + lr * T[w][qr + 1]
            acc = acc + (g[u] + val) * p[w]
        E.append(acc)
    h_cap = ((1.0 + 2.0 * (lmax + lc)) * (3.0 + es)) * dcol
    s_node = fma(fc['ratio'], fsum + h_cap, h_cap)
    radius = fma(fc['cu'], s_node, bmax)
    worst = max(abs(Fraction(e) - Fraction(f)) for e, f in zip(E, F))
    return float(worst / Fraction(radius)), float(Fraction(bmax) / Fraction(radius))


@pytest.mark.parametrize('chain', [None, 'x+(w-u)', '(x-w)+u'])
@pytest.mark.parametrize('regime', ['smooth', 'rough', 'cancel', 'weights', 'large'])
def test_the_shifted_lattice_radius_covers_the_difference_exactly(regime, chain):
    rng = np.random.default_rng(400 + ['smooth', 'rough', 'cancel', 'weights', 'large'].index(regime))
    worst, model_part = 0.0, 0.0
    for trial in range(200):
        W = int(rng.integers(1, 8))
        N0 = int(rng.integers(4, 16))
        r = np.arange(N0) / (N0 - 1.0)
        if regime == 'smooth':                               # a cost-to-go with curvature: B' is what decides
            T = np.stack([(r - rng.uniform(0, 1)) ** 2 * rng.uniform(0.5, 3) + rng.uniform(-1, 1) * r for _ in range(W)])
        elif regime == 'cancel':
            T = 1e6 + 1e-6 * rng.standard_normal((W, N0))
        else:
            T = rng.standard_normal((W, N0))
            if regime == 'large':
                T *= 10.0 ** rng.uniform(100, 250)
        p = np.abs(rng.standard_normal(W)) + 1e-3
        p /= p.sum()
        if regime == 'weights':
            p = rng.standard_normal(W) * 2.1
            if abs(p.sum()) < 0.2:
                p[0] += 1.0
        scale = float(np.abs(T).max())
        n = int(rng.integers(1, 12))
        X = float(rng.uniform(0, 1))
        spread = float(10.0 ** rng.uniform(-2, 0.5))          # controls inside the grid ... far outside it
        a = [float(v) for v in rng.uniform(-spread, spread, size=n)]
        g = [float(v) * scale * 10.0 ** rng.uniform(-3, 2) for v in rng.standard_normal(n)]
        b = [float(v) for v in rng.uniform(-1, 1, size=W) * 10.0 ** rng.uniform(-2.5, 0.3)]
        if trial % 4 == 0:
            b[0] = float(rng.integers(-3, 4)) / (N0 - 1)      # a shift of a whole number of rows
        order = list(rng.permutation(W))
        if chain and trial % 3 == 0:                          # leaves that nearly cancel: the sum of their magnitudes is what counts
            X = float(rng.uniform(0.5, 1))
            a = [-X + float(v) for v in rng.uniform(-spread, spread, size=n) * 1e-3]
        ratio, share = shifted_check(T, p, X, a, g, b, order, chain)
        assert ratio <= 1.0, (regime, trial, ratio, chain)
        worst, model_part = max(worst, ratio), max(model_part, share)
    assert worst > 0.05, worst                               # the chord bound B' is close to what the lerp leaves out ...
    if regime in ('smooth', 'rough'):
        assert model_part > 0.9, model_part                  # ... and it is what the radius is made of there


# The SHORT first pass on the shifted lattice (round 6; SDP_COL_LEAN2 with SDP_COL_SHIFT of csrc/sdp_colres_kernel.h):
# x0' = (X + a_u) - b_w with a_u and the control's part h_u of the cost = K + h_u from the column's control table.  The pass
# orders F' = fma(+-h, psum, lerp(A', pa)) with the control's index in its low mantissa bits, bounds the cost by |K| + max |h|
# and every |lam0| by the positions of the column's smallest and largest a:
#     S = Pcap (|K| + max |h|) + (1 + 2 (L + Lc)) (3 + Es) D,    radius = (cu + 2^(bits+2) u) S + max B'[q0]
def shifted_short_check(T, p, X, K, a, h, b, order, sign=1.0):
    W, N0 = T.shape
    nm1 = float(N0 - 1)
    fc = filter_constants(p)
    q_w, f_w, c_w, pbabs = [], [], [], 0
    for w in range(W):
        pb = (-b[w]) * nm1
        fl = float(np.floor(pb))
        q_w.append(int(fl))
        f_w.append(pb - fl)
        c_w.append(abs(p[w]) * (f_w[-1] * (1.0 - f_w[-1])))
        pbabs = max(pbabs, int(abs(abs(b[w]) * nm1)) + 1)
    flmax, nflmin = max(q_w), max(-q for q in q_w)
    kmin, rows = -(flmax + 1), N0 + flmax + nflmin + 1
    pbmax = float(max(abs(flmax), abs(nflmin), pbabs) + 1)
    lc = float(rows) + float(abs(kmin)) + pbmax + 0.0 + float(N0 + 1)
    es = float(1 + 2 * (flmax + nflmin + 2))
    Ap, Bp, big = np.zeros(rows), np.zeros(rows), 0.0
    for ki in range(rows):
        k = kmin + ki
        for w in order:
            j = k + q_w[w]
            q = max(min(j, N0 - 2), 0)
            lam = float(j - q) + f_w[w]
            t0, t1 = T[w][q], T[w][q + 1]
            t2 = T[w][q + 2] if q + 2 < N0 else T[w][q + 1]
            Ap[ki] = fma(p[w], fma(lam, t1 - t0, t0), Ap[ki])
            d2 = (t2 - t1) - (t1 - t0)
            Bp[ki] = fma(c_w[w], abs(d2) if 0 <= j <= N0 - 3 else 0.0, Bp[ki])
            big = max(big, abs(t0), abs(t1))
    dcol = fc['pcap'] * big + fc['floor']

    def cell(xa):                                            # sdp_lean2_cell on the lattice
        pk = xa * nm1 - float(kmin)
        q0 = max(min(int(pk), rows - 2), 0)
        return q0, pk - float(q0)
    n = len(a)
    bits = max((n - 1).bit_length(), 1)
    mask = (1 << bits) - 1
    L = max(1.0, abs(cell(X + min(a))[1]), abs(cell(X + max(a))[1]))
    h_cap = ((1.0 + 2.0 * (L + lc)) * (3.0 + es)) * dcol
    s_node = fma(fc['pcap'], abs(K) + max(abs(v) for v in h), h_cap)
    p_exact = sum(Fraction(v) for v in p)
    worst, bmax = Fraction(0), 0.0
    for ci in range(n):
        xa = X + a[ci]
        q0, lam0 = cell(xa)
        assert abs(lam0) <= L
        bmax = max(bmax, Bp[q0])
        F = fma(sign * h[ci], fc['psum'], fma(lam0, Ap[q0 + 1] - Ap[q0], Ap[q0]))
        Fp = pack_index(F, ci, mask)
        assert abs(Fraction(Fp) - Fraction(F)) <= Fraction(2.0 ** (bits + 1 - 53)) * Fraction(s_node) * (1 + Fraction(3, 2 ** 53))
        g = K + sign * h[ci]                                 # the reference's cost, one rounding
        acc = 0.0
        for w in range(W):
            sr = (xa - b[w]) * nm1
            qr = max(min(int(sr), N0 - 2), 0)
            lr = sr - float(qr)
            val = (1.0 - lr) * T[w][qr] + lr * T[w][qr + 1]
            acc = acc + (g + val) * p[w]
        worst = max(worst, abs(Fraction(acc) - Fraction(K) * p_exact - Fraction(Fp)))
    radius = fma(fc['cu'] + 2.0 ** (bits + 1 - 52), s_node, bmax)
    return float(worst / Fraction(radius)), float(Fraction(bmax) / Fraction(radius))


@pytest.mark.parametrize('regime', ['smooth', 'rough', 'cancel', 'weights', 'large'])
def test_the_short_pass_on_the_shifted_lattice_covers_the_difference_exactly(regime):
    rng = np.random.default_rng(450 + ['smooth', 'rough', 'cancel', 'weights', 'large'].index(regime))
    worst, model_part = 0.0, 0.0
    for trial in range(200):
        W = int(rng.integers(1, 8))
        N0 = int(rng.integers(4, 16))
        r = np.arange(N0) / (N0 - 1.0)
        if regime == 'smooth':
            T = np.stack([(r - rng.uniform(0, 1)) ** 2 * rng.uniform(0.5, 3) + rng.uniform(-1, 1) * r for _ in range(W)])
        elif regime == 'cancel':
            T = 1e6 + 1e-6 * rng.standard_normal((W, N0))
        else:
            T = rng.standard_normal((W, N0))
            if regime == 'large':
                T *= 10.0 ** rng.uniform(100, 250)
        p = np.abs(rng.standard_normal(W)) + 1e-3
        p /= p.sum()
        if regime == 'weights':
            p = rng.standard_normal(W) * 2.1
            if abs(p.sum()) < 0.2:
                p[0] += 1.0
        scale = float(np.abs(T).max())
        n = int(rng.integers(1, 70))
        X = float(rng.uniform(0, 1))
        spread = float(10.0 ** rng.uniform(-2, 0.5))
        a = [float(v) for v in rng.uniform(-spread, spread, size=n)]
        K = float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-3, 3)
        h = [float(v) * scale * 10.0 ** rng.uniform(-3, 2) for v in rng.standard_normal(n)]
        b = [float(v) for v in rng.uniform(-1, 1, size=W) * 10.0 ** rng.uniform(-2.5, 0.3)]
        if trial % 4 == 0:
            b[0] = float(rng.integers(-3, 4)) / (N0 - 1)
        order = list(rng.permutation(W))
        ratio, share = shifted_short_check(T, [float(v) for v in p], X, K, a, h, b, order, -1.0 if trial % 3 == 0 else 1.0)
        assert ratio <= 1.0, (regime, trial, ratio)
        worst, model_part = max(worst, ratio), max(model_part, share)
    assert worst > 0.05, worst
    if regime in ('smooth', 'rough'):
        assert model_part > 0.9, model_part


def _shift_lattice(T, p, b, order):
    """sdp_col_phase_shift / sdp_col_shift_col / sdp_col_shift_reduce, as in shifted_check"""
    W, N0 = T.shape
    nm1 = float(N0 - 1)
    fc = filter_constants(p)
    q_w, f_w, c_w, pbabs = [], [], [], 0
    for w in range(W):
        pb = (-b[w]) * nm1
        fl = float(np.floor(pb))
        q_w.append(int(fl))
        f_w.append(pb - fl)
        c_w.append(abs(p[w]) * (f_w[-1] * (1.0 - f_w[-1])))
        pbabs = max(pbabs, int(abs(abs(b[w]) * nm1)) + 1)
    flmax, nflmin = max(q_w), max(-q for q in q_w)
    kmin, rows = -(flmax + 1), N0 + flmax + nflmin + 1
    pbmax = float(max(abs(flmax), abs(nflmin), pbabs) + 1)
    lc = float(rows) + float(abs(kmin)) + pbmax + 0.0 + float(N0 + 1)
    es = float(1 + 2 * (flmax + nflmin + 2))
    Ap, Bp, big = np.zeros(rows), np.zeros(rows), 0.0
    for ki in range(rows):
        k = kmin + ki
        for w in order:
            j = k + q_w[w]
            q = max(min(j, N0 - 2), 0)
            lam = float(j - q) + f_w[w]
            t0, t1 = T[w][q], T[w][q + 1]
            t2 = T[w][q + 2] if q + 2 < N0 else T[w][q + 1]
            Ap[ki] = fma(p[w], fma(lam, t1 - t0, t0), Ap[ki])
            d2 = (t2 - t1) - (t1 - t0)
            Bp[ki] = fma(c_w[w], abs(d2) if 0 <= j <= N0 - 3 else 0.0, Bp[ki])
            big = max(big, abs(t0), abs(t1))
    return dict(fc=fc, nm1=nm1, kmin=kmin, rows=rows, lc=lc, es=es, Ap=Ap, Bp=Bp, dcol=fc['pcap'] * big + fc['floor'])


def shifted_block_bound_check(T, p, X, K, a, h, b, order, sign, block):
    """The branch and bound of the short first pass ON THE SHIFTED LATTICE (sdp_short_bnb with SDP_COL_SHIFT, round 6).  As
    block_bound_check, on the lattice's table A' -- and a control's F' is off from the real R - K P* by its OWN cell's
    chord bound B'[q0] on top of the rounding radius, so a block is ruled out against the guess g only when
        LB - max B'[cells of the block]  >  F'_g + B'[q0(g)] + 2 (cu + 2^(bits+2) u) S + 24 u S.
    Checked here, exactly: LB - bm lies below F'_c - B'[q0(c)] of every control c of the block up to the 24 u S the test
    allows for, and bm covers the control's cell.  Returns the largest ((LB - bm) - (F'_c - B'_c)) / (24 u S), or None
    when the blocks are not in order."""
    N0 = T.shape[1]
    lat = _shift_lattice(T, p, b, order)
    fc, nm1, kmin, rows, Ap, Bp = lat['fc'], lat['nm1'], lat['kmin'], lat['rows'], lat['Ap'], lat['Bp']

    def cell(xa):
        pk = xa * nm1 - float(kmin)
        q0 = max(min(int(pk), rows - 2), 0)
        return q0, pk - float(q0)
    n = len(a)
    bits = max((n - 1).bit_length(), 1)
    mask = (1 << bits) - 1
    L = max(1.0, abs(cell(X + min(a))[1]), abs(cell(X + max(a))[1]))
    h_cap = ((1.0 + 2.0 * (L + lat['lc'])) * (3.0 + lat['es'])) * lat['dcol']
    s_node = fma(fc['pcap'], abs(K) + max(abs(v) for v in h), h_cap)
    blocks = [range(b0, min(b0 + block, n)) for b0 in range(0, n, block)]
    pa = [v * nm1 for v in a]
    lo = [min(pa[c] for c in m) for m in blocks]
    hi = [max(pa[c] for c in m) for m in blocks]
    if any(hi[k] + 2 * BNB_DELTA > lo[k + 1] for k in range(len(blocks) - 1)):
        return None
    starts = [v - BNB_DELTA for v in lo] + [hi[-1] + BNB_DELTA]
    hp = [min((sign * h[c]) * fc['psum'] for c in m) for m in blocks]
    pX = X * nm1 - float(kmin)                               # (the lattice's first position taken off the node's part)
    assert (abs(X) + max(abs(v) for v in a)) < 2.0 ** 29 / nm1

    def at(P):
        q = max(min(int(P), rows - 2), 0)
        return q, fma(P - float(q), Ap[q + 1] - Ap[q], Ap[q])
    ends = [at(pX + v) for v in starts]
    worst = Fraction(-10 ** 9)
    for k, members in enumerate(blocks):
        (qa, La), (qb, Lb) = ends[k], ends[k + 1]
        m = min(La, Lb, Ap[qa + 1], Ap[qb])
        bm = max(Bp[qa], Bp[min(qa + 1, rows - 1)], Bp[qb])
        for r in range(qa + 2, qb):
            m = min(m, Ap[r])
            bm = max(bm, Bp[r])
        lbv = (hp[k] + m) - bm
        for ci in members:
            q0, lam0 = cell(X + a[ci])
            assert qa <= q0 <= qb                           # the block's rows cover the control's cell: bm >= B'[q0]
            assert Bp[q0] <= bm
            F = fma(sign * h[ci], fc['psum'], fma(lam0, Ap[q0 + 1] - Ap[q0], Ap[q0]))
            Fp = pack_index(F, ci, mask)
            worst = max(worst, (Fraction(lbv) - (Fraction(F) - Fraction(Bp[q0]))) / (24 * Fraction(U) * Fraction(s_node)))
            assert abs(Fraction(Fp) - Fraction(F)) <= Fraction(2.0 ** (bits + 1 - 53)) * Fraction(s_node) * (1 + Fraction(3, 2 ** 53))
    return float(worst)


@pytest.mark.parametrize('regime', ['smooth', 'rough', 'cancel', 'weights', 'large'])
def test_the_block_bound_on_the_shifted_lattice_lies_below_every_control_of_its_block(regime):
    rng = np.random.default_rng(470 + ['smooth', 'rough', 'cancel', 'weights', 'large'].index(regime))
    worst, checked = -1e9, 0
    for trial in range(150):
        W = int(rng.integers(1, 8))
        N0 = int(rng.integers(4, 24))
        r = np.arange(N0) / (N0 - 1.0)
        if regime == 'smooth':
            T = np.stack([(r - rng.uniform(0, 1)) ** 2 * rng.uniform(0.5, 3) + rng.uniform(-1, 1) * r for _ in range(W)])
        elif regime == 'cancel':
            T = 1e6 + 1e-6 * rng.standard_normal((W, N0))
        else:
            T = rng.standard_normal((W, N0))
            if regime == 'large':
                T *= 10.0 ** rng.uniform(100, 250)
        p = np.abs(rng.standard_normal(W)) + 1e-3
        p /= p.sum()
        if regime == 'weights':
            p = rng.standard_normal(W) * 2.1
            if abs(p.sum()) < 0.2:
                p[0] += 1.0
        scale = float(np.abs(T).max())
        n = int(rng.integers(1, 90))
        X = float(rng.uniform(0, 1))
        spread = float(10.0 ** rng.uniform(-2, 0.5))
        a = [float(v) for v in -spread + np.cumsum(rng.uniform(0.1, 1.0, size=n)) * (2 * spread / n) + np.arange(n) * 4 * BNB_DELTA / (N0 - 1)]
        if trial % 7 == 3:
            a = [float(v) for v in rng.permutation(a)]
        K = float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-3, 3)
        h = [float(v) * scale * 10.0 ** rng.uniform(-3, 2) for v in rng.standard_normal(n)]
        b = [float(v) for v in rng.uniform(-1, 1, size=W) * 10.0 ** rng.uniform(-2.5, 0.3)]
        if trial % 4 == 0:
            b[0] = float(rng.integers(-3, 4)) / (N0 - 1)
        ratio = shifted_block_bound_check(T, [float(v) for v in p], X, K, a, h, b, list(rng.permutation(W)),
                                          -1.0 if trial % 3 == 0 else 1.0, 8 if trial % 5 else 16)
        if ratio is None:
            assert trial % 7 == 3 and n > 8
            continue
        checked += 1
        assert ratio <= 1.0, (regime, trial, ratio)
        worst = max(worst, ratio)
    assert checked >= 110 and worst < 0.5, (checked, worst)


@pytest.mark.parametrize('chain', ['x+(w-u)', '(x-w)+u'])
def test_a_regrouped_chain_needs_the_sum_of_its_leaves_in_the_bound(chain):
    """leaves that cancel far outside the grid (x = 1e3 .. 1e6, u = x - 1 .. x + 1 on a grid [0, 1]) and tables that are
    straight lines: no chord error, the roundings decide.  The reference's nesting and the regrouped sum then differ by
    roundings of the LEAVES' magnitude, not of the position's: the bound holds with the sum of the leaves' magnitudes in
    L (SDP_COL_SHIFT_CHAIN) and fails by an order of magnitude without it."""
    rng = np.random.default_rng(77)
    worst, worst_without = 0.0, 0.0
    for trial in range(300):
        W, N0 = int(rng.integers(1, 8)), int(rng.integers(4, 16))
        T = np.stack([np.linspace(rng.uniform(-1, 1), rng.uniform(-1, 1), N0) for _ in range(W)])
        if trial % 3 == 0:
            T = T * 10.0 ** rng.uniform(100, 250)
        p = np.abs(rng.standard_normal(W)) + 1e-3
        p /= p.sum()
        n = int(rng.integers(1, 12))
        X = float(rng.uniform(0.5, 1) * 10.0 ** rng.uniform(3, 6))
        a = [-X + float(v) for v in rng.uniform(-1, 1, size=n)]
        g = [float(v) * float(np.abs(T).max()) for v in rng.standard_normal(n)]
        b = [float(v) for v in rng.uniform(-1, 1, size=W) * 0.1]
        order = list(rng.permutation(W))
        ratio, _ = shifted_check(T, p, X, a, g, b, order, chain)
        assert ratio <= 1.0, (trial, ratio)
        worst = max(worst, ratio)
        worst_without = max(worst_without, shifted_check(T, p, X, a, g, b, order, chain, leaf_sum=False)[0])
    assert 0.0 < worst <= 1.0, worst                         # (loose by the size of the leaves: the price of not knowing how they cancel)
    assert worst_without > 5.0, worst_without


# ---------------------------------------------------------------------------
# The filter on the REDUCED ARRAY (csrc/sdp_lead_kernel.h; docs/NOTEBOOK.md section 3.3c): two controlled stocks next to one
# exogenous axis (d = 3, m = 2), grids [0, 1].  sdp_lead_reduce: A[i0, i1] = sum_w p_w inner_w(i0, i1) with the reference's
# lerp along the trailing axis; first pass F = fma(g, psum, bilerp(A)) with fused lerps; bound per node
#     S = ratio (sum |F| + Lp Dabs) + Lp Dabs,   Lp = max_u prod_k (1 + 2 |lam_k|),   Dabs = Pcap (max_w E_w) max |V| + floor,
#     radius = 4 (W + 3 d + 4) u S.
# The reference: the full nest of multilinear_cython.pyx per perturbation point (axis 0 outermost), stodynprog.py:677-681.
# ---------------------------------------------------------------------------
def lead_check(V, p, lam2, q2, controls):
    """V[n0][n1][n2]; per perturbation point the trailing cell (q2[w], lam2[w]); controls: (pos0, pos1, g) per control"""
    n0, n1, n2 = V.shape
    W, d = len(p), 3
    ps, pa = 0.0, 0.0
    for v in p:
        ps, pa = ps + v, pa + abs(v)
    pcap = pa if pa > 1.0 else 1.0
    ratio = pcap / abs(ps)
    cu = 1.0 * float(4 * (W + 3 * d + 4)) * 2.0 ** -53
    floor = 2.0 * TINY / cu
    oml2 = [1.0 - l for l in lam2]
    A = np.zeros((n0, n1))
    for i0 in range(n0):
        for i1 in range(n1):
            acc = 0.0
            for w in range(W):
                inner = oml2[w] * V[i0][i1][q2[w]] + lam2[w] * V[i0][i1][q2[w] + 1]       # pyx:88 (one axis)
                acc = fma(p[w], inner, acc)
            A[i0][i1] = acc
    emax = max(abs(oml2[w]) + abs(lam2[w]) for w in range(W))
    dabs = pcap * (emax * float(np.abs(V).max())) + floor
    F, E, lp, fsum = [], [], 0.0, 0.0
    for pos0, pos1, g in controls:
        q0 = max(min(int(pos0), n0 - 2), 0)
        q1 = max(min(int(pos1), n1 - 2), 0)
        l0, l1 = pos0 - float(q0), pos1 - float(q1)
        lp = max(lp, (1.0 * fma(2.0, abs(l0), 1.0)) * fma(2.0, abs(l1), 1.0))
        lo = fma(l1, A[q0][q1 + 1] - A[q0][q1], A[q0][q1])                  # SdpLeadLerp: last lead axis innermost, fused
        hi = fma(l1, A[q0 + 1][q1 + 1] - A[q0 + 1][q1], A[q0 + 1][q1])
        f = fma(g, ps, fma(l0, hi - lo, lo))
        fsum = fsum + abs(f)
        F.append(f)
        o0, o1 = 1.0 - l0, 1.0 - l1
        acc = 0.0
        for w in range(W):
            def z(i0, i1):
                return oml2[w] * V[i0][i1][q2[w]] + lam2[w] * V[i0][i1][q2[w] + 1]
            val = o0 * (o1 * z(q0, q1) + l1 * z(q0, q1 + 1)) + l0 * (o1 * z(q0 + 1, q1) + l1 * z(q0 + 1, q1 + 1))
            acc = acc + (g + val) * p[w]
        E.append(acc)
    h_cap = lp * dabs
    s_node = fma(ratio, fsum + h_cap, h_cap)
    radius = cu * s_node
    worst = max(abs(Fraction(e) - Fraction(f)) for e, f in zip(E, F))
    return float(worst / Fraction(radius)), radius


@pytest.mark.parametrize('regime', ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'])
def test_the_reduced_array_radius_covers_the_difference_exactly(regime):
    rng = np.random.default_rng(500 + ['ordinary', 'large', 'small', 'mixed', 'cancel', 'weights'].index(regime))
    worst = 0.0
    for trial in range(120):
        n0, n1, n2 = (int(v) for v in rng.integers(3, 8, size=3))
        W = int(rng.integers(1, 7))
        V = rng.standard_normal((n0, n1, n2))
        if regime == 'large':
            V *= 10.0 ** rng.uniform(100, 280)
        elif regime == 'small':
            V *= 10.0 ** rng.uniform(-300, -100)
        elif regime == 'mixed':
            V *= 10.0 ** rng.uniform(-10, 10, size=V.shape)
        elif regime == 'cancel':
            V = 1e6 + 1e-6 * V
        p = np.abs(rng.standard_normal(W)) + 1e-3
        p /= p.sum()
        if regime == 'weights':
            p = rng.standard_normal(W) * 2.9
            if abs(p.sum()) < 0.2:
                p[0] += 1.0
        pos2 = [float(v) for v in rng.uniform(-1.5, n2 + 0.5, size=W)]          # (the exogenous axis extrapolates too)
        q2 = [max(min(int(v), n2 - 2), 0) for v in pos2]
        lam2 = [v - float(q) for v, q in zip(pos2, q2)]
        scale = float(np.abs(V).max())
        controls = []
        for _ in range(int(rng.integers(1, 10))):
            kind = rng.integers(0, 3)
            pos = [float(rng.uniform(0, n - 1)) if kind == 0 else (float(rng.integers(0, n)) if kind == 1 else float(rng.uniform(-2 * n, 3 * n)))
                   for n in (n0, n1)]
            controls.append((pos[0], pos[1], float(rng.standard_normal()) * scale * 10.0 ** rng.uniform(-3, 3)))
        ratio, radius = lead_check(V, [float(v) for v in p], lam2, q2, controls)
        assert np.isfinite(radius) and radius > 0.0
        assert ratio <= 1.0, (regime, trial, ratio)
        worst = max(worst, ratio)
    assert worst < 0.6, worst


# ---------------------------------------------------------------------------
# The LINE kernel (csrc/sdp_line_kernel.h, round 6): ONE state variable, x' = (X + a_u) - b_w, the shifted lattice with the
# value array itself as the table.  Its bounds are PER CONTROL and local: e = B'[q0] + cu (ratio (|F| + lf Cq) + lf Cq +
# (Lc + |lam0| + PA) Dv) for the first level (chord of the reduced table), e2 = 2 cu (the same without B') for the second
# (G itself in fused arithmetic).  Both are checked against the reference's value exactly, on axes whose span is and is not
# a power of two and that do not start at zero.
# ---------------------------------------------------------------------------
def line_check(V, p, X, a, g, b, smin, span, chain=None):
    N0 = len(V)
    nm1 = float(N0 - 1)
    fc = filter_constants(p)
    W = len(p)
    m, e_ = np.frexp(span)
    pow2 = m == 0.5
    rspan = 1.0 / span
    div = (lambda v: v * rspan) if pow2 else (lambda v: v / span)
    # sdp_line_setup: the shifts
    q_w, f_w, c_w, pbabs = [], [], [], 0
    for w in range(W):
        pb = div(-b[w]) * nm1
        pa_ = abs(div(abs(b[w])) * nm1)
        fl = float(np.floor(pb))
        q_w.append(int(fl))
        f_w.append(pb - fl)
        c_w.append(abs(p[w]) * (f_w[-1] * (1.0 - f_w[-1])))
        pbabs = max(pbabs, int(pa_) + 1)
    flmax, nflmin = max(q_w), max(-q for q in q_w)
    kmin, rows = -(flmax + 1), N0 + flmax + nflmin + 1
    pbmax = float(max(abs(flmax), abs(nflmin), pbabs) + 1)
    p0 = abs(smin) * (nm1 / abs(span))
    lc = float(rows) + float(abs(kmin)) + pbmax + p0 * 1.001 + float(N0 + 1)
    # sdp_lead_reduce
    Ap, Bp, Cp, dvcol = np.zeros(rows), np.zeros(rows), np.zeros(rows), 0.0
    for ki in range(rows):
        k = kmin + ki
        acc = bnd = big = dv = 0.0
        lmax = 1.0
        for w in range(W):
            j = k + q_w[w]
            q = max(min(j, N0 - 2), 0)
            lam = float(j - q) + f_w[w]
            t0, t1 = V[q], V[q + 1]
            t2 = V[q + 2] if q + 2 < N0 else V[q + 1]
            acc = fma(p[w], fma(lam, t1 - t0, t0), acc)
            d2 = (t2 - t1) - (t1 - t0)
            bnd = fma(c_w[w], abs(d2) if 0 <= j <= N0 - 3 else 0.0, bnd)
            big = max(big, abs(t0), abs(t1), abs(t2), abs(V[q - 1] if q > 0 else V[q]))
            dv = max(dv, abs(t1 - t0), abs(t2 - t1))
            lmax = max(lmax, abs(lam))
        Ap[ki], Bp[ki] = acc, bnd
        Cp[ki] = (3.0 + 2.0 * (lmax + 1.0)) * (fc['pcap'] * big + fc['floor'])
        dvcol = max(dvcol, fc['pcap'] * (dv * 1.01) + fc['floor'])
    worst1 = worst2 = share = Fraction(0)
    for u in range(len(a)):
        xa = X + a[u]                                        # sdp_model_lead_a
        pa_abs = (abs(X) + abs(a[u])) * (abs(nm1 * rspan) * 1.002) if chain else 0.0
        pp = div(xa - smin) * nm1
        pk = pp - float(kmin)
        q0 = max(min(int(pk), rows - 2), 0)
        lam0 = pk - float(q0)
        F = fma(g[u], fc['psum'], fma(lam0, Ap[q0 + 1] - Ap[q0], Ap[q0]))
        lc_q = fma(3.0, abs(lam0), 2.0) * max(Cp[q0], Cp[q0 + 1])
        rnd = fma(fc['ratio'], abs(F) + lc_q, lc_q) + ((lc + abs(lam0)) + pa_abs) * dvcol
        e1 = fma(fc['cu'], rnd, Bp[q0])
        # second level: sdp_line_value2
        pf = float(np.floor(pp))
        fr = pp - pf
        k = int(pf)
        acc = 0.0
        for w in range(W):
            fs = fr + f_w[w]
            carry = 1 if fs >= 1.0 else 0
            jw = k + q_w[w] + carry
            q = max(min(jw, N0 - 2), 0)
            lam = float(jw - q) + (fs - float(carry))
            acc = fma(p[w], fma(lam, V[q + 1] - V[q], V[q]), acc)
        F2 = fma(g[u], fc['psum'], acc)
        e2 = (2.0 * fc['cu']) * (rnd + fc['ratio'] * abs(F2 - F))
        # the reference: its own position per perturbation point (pyx:75-81), stodynprog.py:677-681
        E = 0.0
        for w in range(W):
            if chain == 'x+(w-u)':
                xn = X + ((-b[w]) - (-a[u]))
            elif chain == '(x-w)+u':
                xn = (X - b[w]) + a[u]
            else:
                xn = xa - b[w]
            sref = div(xn - smin) * nm1
            qr = max(min(int(sref), N0 - 2), 0)
            lr = sref - float(qr)
            val = (1.0 - lr) * V[qr] + lr * V[qr + 1]
            E = E + (g[u] + val) * p[w]
        d1, d2_ = abs(Fraction(E) - Fraction(F)), abs(Fraction(E) - Fraction(F2))
        worst1 = max(worst1, d1 / Fraction(e1))
        worst2 = max(worst2, d2_ / Fraction(e2))
        share = max(share, Fraction(Bp[q0]) / Fraction(e1))
    return float(worst1), float(worst2), float(share)


LINE_AXES = [(0.0, 1.0), (-8.0, 32.0), (1000.0, 2.0), (0.0, 10.0), (-3.0, 9.0), (0.25, 0.7)]


@pytest.mark.parametrize('chain', [None, 'x+(w-u)', '(x-w)+u'])
@pytest.mark.parametrize('regime', ['smooth', 'rough', 'cancel', 'weights', 'large', 'fine'])
def test_the_line_kernels_bounds_cover_the_difference_exactly(regime, chain):
    rng = np.random.default_rng(900 + ['smooth', 'rough', 'cancel', 'weights', 'large', 'fine'].index(regime))
    worst1 = worst2 = share = 0.0
    for trial in range(150):
        W = int(rng.integers(1, 8))
        N0 = int(rng.integers(4, 16)) if regime != 'fine' else int(rng.integers(40, 90))
        smin, span = LINE_AXES[trial % len(LINE_AXES)]
        r = np.arange(N0) / (N0 - 1.0)
        if regime in ('smooth', 'fine'):                     # a cost-to-go with curvature: the chord bound decides
            V = (r - rng.uniform(0, 1)) ** 2 * rng.uniform(0.5, 3) + rng.uniform(-1, 1) * r + rng.uniform(-2, 2)
        elif regime == 'cancel':
            V = 1e6 + 1e-6 * rng.standard_normal(N0)
        else:
            V = rng.standard_normal(N0)
            if regime == 'large':
                V = V * 10.0 ** rng.uniform(100, 250)
        p = np.abs(rng.standard_normal(W)) + 1e-3
        p /= p.sum()
        if regime == 'weights':
            p = rng.standard_normal(W) * 2.1
            if abs(p.sum()) < 0.2:
                p[0] += 1.0
        scale = float(np.abs(V).max())
        n = int(rng.integers(1, 12))
        X = smin + span * float(rng.uniform(0, 1))
        reach = span * float(10.0 ** rng.uniform(-2, 0.5))   # controls inside the grid ... far outside it
        a = [float(v) for v in rng.uniform(-reach, reach, size=n)]
        g = [float(v) * scale * 10.0 ** rng.uniform(-3, 2) for v in rng.standard_normal(n)]
        b = [span * float(v) for v in rng.uniform(-1, 1, size=W) * 10.0 ** rng.uniform(-2.5, 0.3)]
        if trial % 4 == 0:
            b[0] = span * float(rng.integers(-3, 4)) / (N0 - 1)      # a shift of a whole number of rows
        if trial % 5 == 0:
            a[0] = span * float(rng.integers(0, N0)) / (N0 - 1) - (X - smin)       # a control that lands on a node (up to roundings)
        if chain and trial % 3 == 0:                          # leaves that nearly cancel
            a = [-X + smin + span * float(v) for v in rng.uniform(0, 1, size=n)]
        w1, w2, sh = line_check(V, p, X, a, g, b, smin, span, chain)
        assert w1 <= 1.0, (regime, chain, trial, 'first level', w1)
        assert w2 <= 1.0, (regime, chain, trial, 'second level', w2)
        worst1, worst2, share = max(worst1, w1), max(worst2, w2), max(share, sh)
    assert worst1 > 0.05, worst1                              # the chord bound is close to what the lerp leaves out ...
    if regime in ('smooth', 'rough', 'fine'):
        assert share > 0.9, share                             # ... and it is what the first level's interval is made of there
    assert 0.0 < worst2 < 0.6, worst2                         # the second level: roundings only, the documented factor 2 to spare


def test_the_line_kernels_position_term_is_needed_on_a_long_axis():
    """a value array that is a steep straight line (no chord error) on an axis whose positions are large: what separates
    the reference's value from the filter's is the rounding of the POSITIONS times the slope -- the (Lc + |lam0|) Dv term;
    without it the local bound (magnitudes around the control's cell) is exceeded"""
    rng = np.random.default_rng(5)
    worst = 0.0
    for trial in range(60):
        N0, W = 200, 3
        V = np.linspace(0.0, 1.0, N0) * 1e6 + 1e-9 * rng.standard_normal(N0)
        V -= V[N0 // 2]                                       # small values around the controls' cells, a steep slope
        p = np.array([0.3, 0.4, 0.3])
        X = 1000.0 + 1.0
        a = [float(v) for v in rng.uniform(-0.01, 0.01, size=6)]
        g = [0.0] * 6
        b = [float(v) for v in rng.uniform(-0.004, 0.004, size=W)]
        w1, w2, _ = line_check(V, p, X, a, g, b, 1000.0, 2.0)
        assert w1 <= 1.0 and w2 <= 1.0
        worst = max(worst, w2)
    assert worst > 1e-3, worst
