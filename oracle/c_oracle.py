"""ctypes loader for the C oracle (oracle/sdp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsdp_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "sdp_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        for suf, ct in (("f64", C.c_double), ("f32", C.c_float)):
            f = getattr(_lib, "oracle_mlinterp_" + suf)
            f.restype = C.c_int
            f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        _lib.oracle_vi_tab_f64.restype = C.c_int
        _lib.oracle_vi_tab_f64.argtypes = [
            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_vi_synth3d_f64.restype = C.c_int
        _lib.oracle_vi_synth3d_f64.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int64,
            C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def mlinterp(smin, smax, orders, values, s):
    """Same contract as multilinear_interpolation (multilinear_cython.pyx:17)."""
    values = np.ascontiguousarray(values)
    dt = values.dtype
    if dt not in (np.float64, np.float32):
        raise TypeError("values must be float32 or float64")
    smin = np.ascontiguousarray(smin, dtype=dt)
    smax = np.ascontiguousarray(smax, dtype=dt)
    orders = np.ascontiguousarray(orders, dtype=np.int64)
    s = np.ascontiguousarray(s, dtype=dt)
    d, n_s = s.shape
    n_v = values.shape[0]
    out = np.zeros((n_v, n_s), dtype=dt)
    f = lib().oracle_mlinterp_f64 if dt == np.float64 else lib().oracle_mlinterp_f32
    rc = f(d, _p(smin), _p(smax), _p(orders), _p(values), n_v, _p(s), n_s, _p(out))
    if rc != 0:
        raise Exception("Can't interpolate in dimension strictly greater than 5")
    return out


def vi_tab(smin, smax, orders, V, cell_off, W, proba, x_next, g):
    """Tabulated backup of a batch of nodes -> (J, idx, margin)."""
    smin = np.ascontiguousarray(smin, dtype=np.float64)
    smax = np.ascontiguousarray(smax, dtype=np.float64)
    orders = np.ascontiguousarray(orders, dtype=np.int64)
    V = np.ascontiguousarray(V, dtype=np.float64).ravel()
    cell_off = np.ascontiguousarray(cell_off, dtype=np.int64)
    x_next = np.ascontiguousarray(x_next, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    proba = np.ascontiguousarray(proba, dtype=np.float64) if W > 0 else None
    n = len(cell_off) - 1
    J = np.zeros(n)
    idx = np.zeros(n, dtype=np.int64)
    margin = np.zeros(n)
    rc = lib().oracle_vi_tab_f64(len(orders), _p(smin), _p(smax), _p(orders), _p(V),
                                 n, _p(cell_off), W, _p(proba), _p(x_next), _p(g),
                                 _p(J), _p(idx), _p(margin))
    assert rc == 0
    return J, idx, margin


def vi_synth3d(grids, V, par, u_min, u_max, U, wgrid, proba, node_ids=None,
               n_nodes=None, n_threads=1):
    """Synthetic benchmark sweep on selected nodes -> (J, idx, margin)."""
    g = [np.ascontiguousarray(x, dtype=np.float64) for x in grids]
    smin = np.array([x[0] for x in g])
    smax = np.array([x[-1] for x in g])
    orders = np.array([len(x) for x in g], dtype=np.int64)
    V = np.ascontiguousarray(V, dtype=np.float64).ravel()
    par = np.ascontiguousarray(par, dtype=np.float64)
    wgrid = np.ascontiguousarray(wgrid, dtype=np.float64)
    proba = np.ascontiguousarray(proba, dtype=np.float64)
    if node_ids is not None:
        node_ids = np.ascontiguousarray(node_ids, dtype=np.int64)
        n = len(node_ids)
    else:
        n = int(n_nodes)
    J = np.zeros(n)
    idx = np.zeros(n, dtype=np.int64)
    margin = np.zeros(n)
    rc = lib().oracle_vi_synth3d_f64(_p(smin), _p(smax), _p(orders), _p(g[0]), _p(g[1]),
                                     _p(g[2]), _p(V), _p(par), u_min, u_max, U,
                                     len(wgrid), _p(wgrid), _p(proba), n, _p(node_ids),
                                     _p(J), _p(idx), _p(margin), n_threads)
    assert rc == 0
    return J, idx, margin


def max_threads():
    return lib().oracle_max_threads()
