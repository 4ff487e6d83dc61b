"""HIP multilinear interpolation (through the C ABI) against the golden
vectors of the compiled reference and against the CPU oracle: bit-exact for
float64 and float32 (rows a4-a7 of SURVEY section 8)."""
import numpy as np
import pytest

from conftest import golden
from oracle import c_oracle
from stodynprog_amd.dolointerpolation import (multilinear_interpolation,
                                              MultilinearInterpolator)
from stodynprog_amd import MlinInterpolator

pytestmark = pytest.mark.gpu


def test_golden_cases_bit_exact(gpu):
    g = golden('g1_interp')
    for c in range(int(g['n_cases'])):
        p = 'c{:02d}_'.format(c)
        out = multilinear_interpolation(g[p + 'smin'], g[p + 'smax'], g[p + 'orders'],
                                        np.ascontiguousarray(g[p + 'values']),
                                        np.ascontiguousarray(g[p + 's']))
        ref = g[p + 'out']
        assert out.dtype == ref.dtype and out.shape == ref.shape
        assert np.array_equal(out, ref, equal_nan=True), 'case {}'.format(c)


def test_reference_unit_test_1d(gpu):
    # reference stodynprog/tests/test_dolointerp.py:17-40
    smin, smax, orders = np.array([0.]), np.array([2.]), np.array([3])
    grid = np.linspace(smin[0], smax[0], orders[0])
    values = np.ascontiguousarray(np.atleast_2d(grid ** 2))
    pts = np.ascontiguousarray(np.atleast_2d(np.linspace(smin[0], smax[0], 5)))
    out = multilinear_interpolation(smin, smax, orders, values, pts)
    assert np.all(np.abs(out - np.array([0, 0.5, 1, 2.5, 4])) < 1e-10)


def test_reference_unit_test_r2r2(gpu):
    # reference stodynprog/tests/test_dolointerp.py:45-93
    def f(x):
        return np.vstack([np.sqrt(x[0, :] ** 2 + x[1, :] ** 2),
                          np.power(x[0, :] ** 3 + x[1, :] ** 3, 1.0 / 3.0)])
    interp = MultilinearInterpolator([1, 1], [2, 2], [5, 5])
    interp.set_values(f(interp.grid))
    grid_points = np.array([[1, 1], [1, 2], [2, 1], [2, 2]]).T
    random_points = np.random.default_rng(5).random((2, 6)) + 1
    for pts, tol in ((grid_points, 1e-9), (random_points, 0.01)):
        assert np.all(np.abs(interp(pts) - f(pts)) < tol)
    g = golden('g1_interp')
    assert np.array_equal(interp(g['mli_pts']), g['mli_out'])


def test_extrapolation_and_cast_edge_cases(gpu):
    g = golden('g1_interp')
    out = multilinear_interpolation(np.array([0.]), np.array([2.]), np.array([3]),
                                    np.array([[0., 1., 4.]]),
                                    np.ascontiguousarray(g['ex_s']))
    assert np.array_equal(out, g['ex_out'], equal_nan=True)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('d', [1, 2, 3, 4])
def test_large_random_against_oracle(gpu, d, dtype):
    rng = np.random.default_rng(100 + d)
    orders = np.array([50, 51, 52, 11][:d], dtype=np.int64)
    smin = rng.uniform(-1, 0, d).astype(dtype)
    smax = (smin + rng.uniform(1, 2, d)).astype(dtype)
    values = rng.standard_normal((3, int(np.prod(orders)))).astype(dtype)
    n = 300_001                                     # ragged: not a multiple of the block size
    s = np.vstack([rng.uniform(smin[k] - 0.2, smax[k] + 0.2, n) for k in range(d)]).astype(dtype)
    out = multilinear_interpolation(smin, smax, orders, values, np.ascontiguousarray(s))
    ref = c_oracle.mlinterp(smin, smax, orders, values, s)
    assert np.array_equal(out, ref)


def test_empty_inputs(gpu):
    out = multilinear_interpolation(np.array([0.]), np.array([1.]), np.array([4]),
                                    np.zeros((2, 4)), np.zeros((1, 0)))
    assert out.shape == (2, 0)
    out = multilinear_interpolation(np.array([0.]), np.array([1.]), np.array([4]),
                                    np.zeros((0, 4)), np.zeros((1, 3)))
    assert out.shape == (0, 3)


def test_argument_rules_of_the_compiled_reference(gpu):
    f8, f4 = np.float64, np.float32
    smin, smax, orders = np.array([0.]), np.array([1.]), np.array([4])
    vals, s = np.zeros((1, 4)), np.zeros((1, 3))
    with pytest.raises(ValueError) as e:          # SURVEY 3.2 [measured]: dtype mismatch
        multilinear_interpolation(smin, smax, orders, vals.astype(f4), s.astype(f4))
    assert "Buffer dtype mismatch, expected 'float' but got 'double'" in str(e.value)
    with pytest.raises(ValueError) as e:
        multilinear_interpolation(smin, smax, orders.astype(np.int32), vals, s)
    assert "expected 'long'" in str(e.value)
    with pytest.raises(Exception) as e:           # d = 5 (pyx:47)
        multilinear_interpolation(np.zeros(5), np.ones(5), np.full(5, 2), np.zeros((1, 32)),
                                  np.zeros((5, 1)))
    assert str(e.value) == "Can't interpolate in dimension strictly greater than 5"
    with pytest.raises(ValueError):
        multilinear_interpolation(smin, smax, orders, np.zeros((1, 5)), s)
    out = multilinear_interpolation(smin.astype(f4), smax.astype(f4), orders, vals.astype(f4),
                                    s.astype(f4))
    assert out.dtype == f4


def test_mlin_interpolator_broadcast_wrapper(gpu):
    # reference stodynprog.py:255-290: variadic coordinates, broadcast output shape
    xg, yg, zg = np.linspace(0, 1, 10), np.linspace(-1, 1, 11), np.linspace(2, 3, 12)
    X, Y, Z = np.meshgrid(xg, yg, zg, indexing='ij')
    val = X + 2 * Y - Z * Y
    it = MlinInterpolator(xg, yg, zg)
    it.set_values(val)
    assert it(0, 0, 2.5).shape == ()
    out = it([0, 1], np.array([[0, 1]]).T, 2.5)
    assert out.shape == (2, 2)
    # trilinear function is reproduced exactly at nodes, and linearly extrapolated outside
    assert np.allclose(it(X, Y, Z), val, atol=1e-13)
    assert np.isclose(it(1.5, 0.0, 2.0), 1.5)
    ref = c_oracle.mlinterp(it._xmin, it._xmax, it._xshape, it.values,
                            np.array([[0.33], [0.2], [2.9]]))
    assert it(0.33, 0.2, 2.9) == ref[0, 0]


def test_pickled_interpolator_evaluates_after_reload(gpu, tmp_path):
    from stodynprog_amd import compat
    xg, yg = np.linspace(0, 1, 6), np.linspace(-1, 1, 5)
    X, Y = np.meshgrid(xg, yg, indexing='ij')
    it = MlinInterpolator(xg, yg)
    it.set_values(X * 2 - Y)
    before = it(0.37, 0.12)
    compat.dump_interpolator(it, str(tmp_path / 'it.dat'))
    again = compat.load_interpolator(str(tmp_path / 'it.dat'))
    assert again(0.37, 0.12) == before
    assert pickle_roundtrip(it)(0.37, 0.12) == before


def pickle_roundtrip(obj):
    import pickle
    return pickle.loads(pickle.dumps(obj))


def test_power_of_two_spans_multiply_by_the_reciprocal_bit_exactly(gpu):
    """`(s - smin) / span` (pyx:75) is computed as a product with 1/span when span is a
    power of two (sdp_div_span, csrc/sdp_device.h): the same real number, rounded the same
    way -- checked against the C oracle (true division) on spans 1, 8, 0.5, 2^-30, 2^40 and,
    for contrast, 10 and 3; queries include huge, tiny, subnormal, infinite and NaN values"""
    rng = np.random.default_rng(12)
    for dt in (np.float64, np.float32):
        tiny = np.finfo(dt).tiny
        for lo, span in ((0., 1.), (-4., 8.), (0.25, 0.5), (1., 2. ** -30), (-7., 2. ** 40), (0., 10.), (-1., 3.)):
            smin = np.array([lo, lo], dtype=dt)
            smax = np.array([lo + span, lo + span], dtype=dt)
            orders = np.array([7, 5])
            values = rng.standard_normal((2, 35)).astype(dt)
            base = lo + span * rng.uniform(-0.5, 1.5, (2, 4000))
            special = np.array([[lo, lo + span, lo + tiny, lo - tiny * 3, 1e300 if dt == np.float64 else 1e38,
                                 -1e300 if dt == np.float64 else -1e38, np.inf, -np.inf, np.nan, 5e-324],
                                [lo + span / 3] * 10])
            s = np.ascontiguousarray(np.concatenate([base, special], axis=1).astype(dt))
            with np.errstate(all='ignore'):
                out = multilinear_interpolation(smin, smax, orders, values, s)
                ref = c_oracle.mlinterp(smin, smax, orders, values, s)
            assert np.array_equal(out, ref, equal_nan=True), (dt, lo, span)
