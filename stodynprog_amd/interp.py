"""Multilinear interpolation front-ends backed by the HIP kernel.

Host-side mirrors of the reference's three interpolation entry points:
  * multilinear_interpolation(smin, smax, orders, values, s)
        reference stodynprog/dolointerpolation/multilinear_cython.pyx:17-49
  * MultilinearInterpolator, mlinspace
        reference stodynprog/dolointerpolation/multilinear.py:15-91
  * MlinInterpolator
        reference stodynprog/stodynprog.py:255-290
All of them call sdp_mlinterp_{f64,f32} of libsdp_hip.so (include/sdp_hip.h);
there is no host implementation.
"""
import numpy as np

from . import _native as nat

__all__ = ['multilinear_interpolation', 'MultilinearInterpolator', 'mlinspace',
           'MlinInterpolator']

_CNAME = {np.dtype(np.float64): 'double', np.dtype(np.float32): 'float'}


def _floating_dtype(values):
    dt = np.asarray(values).dtype
    if dt not in _CNAME:
        raise TypeError('No matching signature found')      # Cython fused-type dispatch error
    return dt


def multilinear_interpolation(smin, smax, orders, values, s):
    """Interpolate `values` (n_v x S, given on the uniform grid smin..smax with
    `orders` points per axis, last axis fastest) at the points `s` (d x n_s).
    Linear extrapolation outside the grid.  Returns an (n_v, n_s) array of the
    dtype of `values`.

    Argument rules of the compiled reference are kept: every real argument
    must have the dtype of `values` (float32 or float64), `orders` must be
    int64, `values` and `s` must be 2-D C-contiguous; d must be 1..4.
    """
    dt = _floating_dtype(values)
    for name, arr, nd in (('smin', smin, 1), ('smax', smax, 1), ('values', values, 2),
                          ('s', s, 2)):
        arr = np.asarray(arr)
        if arr.dtype != dt:
            raise ValueError("Buffer dtype mismatch, expected '{}' but got '{}'".format(
                _CNAME[dt], _CNAME.get(arr.dtype, str(arr.dtype))))
        if arr.ndim != nd:
            raise ValueError('Buffer has wrong number of dimensions (expected {}, got {})'
                             .format(nd, arr.ndim))
    orders = np.asarray(orders)
    if orders.dtype != np.int64:
        raise ValueError("Buffer dtype mismatch, expected 'long' but got '{}'".format(
            orders.dtype))
    values = np.asarray(values)
    s = np.asarray(s)
    if not values.flags.c_contiguous or not s.flags.c_contiguous:
        raise ValueError('ndarray is not C-contiguous')
    d, n_s = s.shape
    n_v = values.shape[0]
    smin = np.ascontiguousarray(smin)
    smax = np.ascontiguousarray(smax)
    orders = np.ascontiguousarray(orders)
    if d >= 1 and d <= 4:
        if len(smin) < d or len(smax) < d or len(orders) < d:
            raise ValueError('smin, smax and orders need one entry per dimension')
        if values.shape[1] != int(np.prod(orders[:d])):
            raise ValueError('values has {} columns but the grid has {} nodes'.format(
                values.shape[1], int(np.prod(orders[:d]))))
    out = np.zeros((n_v, n_s), dtype=dt)
    f = nat.lib().sdp_mlinterp_f64 if dt == np.float64 else nat.lib().sdp_mlinterp_f32
    nat.check(f(d, nat.ptr(smin), nat.ptr(smax), nat.ptr(orders), nat.ptr(values), n_v,
                nat.ptr(s), n_s, nat.ptr(out)))
    return out


def mlinspace(smin, smax, orders):
    """(d, S) array enumerating the nodes of the Cartesian grid, last axis
    fastest (reference multilinear.py:15-21)."""
    if len(orders) == 1:
        return np.atleast_2d(np.linspace(np.ravel(smin)[0], np.ravel(smax)[0],
                                         int(np.ravel(orders)[0]))).copy()
    axes = [np.linspace(smin[i], smax[i], orders[i]) for i in range(len(orders))]
    meshes = np.meshgrid(*axes, indexing='ij')
    return np.vstack([m.flatten() for m in meshes])


class MultilinearInterpolator(object):
    """dolo-style interpolator object (reference multilinear.py:23-91).

    smin, smax, orders : grid bounds and number of points along each dimension
    values : (n_v, S) array, each row a function sampled on the grid with the
             last index varying fastest
    """
    def __init__(self, smin, smax, orders, values=None, dtype=np.float64):
        self.smin = np.array(smin, dtype=dtype)
        self.smax = np.array(smax, dtype=dtype)
        self.orders = np.array(orders, dtype=np.int64)
        self.d = len(orders)
        self.dtype = dtype
        self._nodes = None                 # the grid nodes, enumerated on first use of `.grid`
        if values is not None:
            self.set_values(values)

    @property
    def grid(self):
        """(d, S) coordinates of the grid nodes (the attribute name is API: multilinear.py:79-84)"""
        nodes = self._nodes
        if nodes is None:
            nodes = self._nodes = mlinspace(self.smin, self.smax, self.orders)
        return nodes

    def set_values(self, values):
        self.values = np.ascontiguousarray(values, dtype=self.dtype)

    def interpolate(self, s):
        s = np.ascontiguousarray(s, dtype=self.dtype)
        return multilinear_interpolation(self.smin, self.smax, self.orders, self.values, s)

    def __call__(self, s):
        return self.interpolate(s)


class _DeviceValues(object):
    """sdp_interp handle: value rows uploaded once, evaluated many times."""

    def __init__(self, smin, smax, orders, values):
        import ctypes as C
        values = np.ascontiguousarray(values)
        self.dtype = values.dtype
        self.n_v = values.shape[0]
        self.d = len(orders)
        smin = np.ascontiguousarray(smin, dtype=np.float64)
        smax = np.ascontiguousarray(smax, dtype=np.float64)
        orders = np.ascontiguousarray(orders, dtype=np.int64)
        h = C.c_void_p()
        nat.check(nat.lib().sdp_interp_create(nat.np_real(self.dtype), self.d, nat.ptr(smin),
                                              nat.ptr(smax), nat.ptr(orders), nat.ptr(values),
                                              self.n_v, C.byref(h)))
        self.h = h

    def eval(self, s):
        s = np.ascontiguousarray(s, dtype=self.dtype)
        out = np.empty((self.n_v, s.shape[1]), dtype=self.dtype)
        nat.check(nat.lib().sdp_interp_eval(self.h, nat.ptr(s), s.shape[1], nat.ptr(out)))
        return out

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                nat.lib().sdp_interp_destroy(self.h)
                self.h = None
        except Exception:
            pass


class _DeviceTab(object):
    """sdp_tab handle over an interpolator's values: uploaded once, then any number of
    tabulated backups (DPSolver._value_at_state_vect, one call per node in user code)."""

    def __init__(self, interp):
        import ctypes as C
        V = np.ascontiguousarray(interp.values, dtype=float).ravel()
        smin = np.ascontiguousarray(interp._xmin, dtype=float)
        smax = np.ascontiguousarray(interp._xmax, dtype=float)
        orders = np.ascontiguousarray(interp._xshape, dtype=np.int64)
        h = C.c_void_p()
        nat.check(nat.lib().sdp_tab_create(interp.ndim, nat.ptr(smin), nat.ptr(smax),
                                           nat.ptr(orders), nat.ptr(V), C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                nat.lib().sdp_tab_destroy(self.h)
                self.h = None
        except Exception:
            pass


def device_tab(interp):
    """the (cached) sdp_tab handle of a MlinInterpolator; dropped by set_values"""
    tab = interp.__dict__.get('_tab')
    if tab is None:
        tab = interp.__dict__['_tab'] = _DeviceTab(interp)
    return tab


class MlinInterpolator:
    """Variadic-coordinate interpolator over grid vectors (reference
    stodynprog.py:255-290).  Only the first entry, last entry and length of
    each grid vector are used: the grid is taken as uniform.

    Instances pickle with the reference's attribute names (ndim, _xmin, _xmax,
    _xshape, values); the device copy of the values is made on first call.
    """

    def __init__(self, *x_grid):
        self.ndim = len(x_grid)
        self._xmin = np.array([x[0] for x in x_grid])
        self._xmax = np.array([x[-1] for x in x_grid])
        self._xshape = np.array([len(x) for x in x_grid], dtype=np.int64)
        self.values = None

    def set_values(self, values):
        assert values.ndim == self.ndim
        assert values.shape == tuple(self._xshape)
        self.values = np.ascontiguousarray(np.atleast_2d(values.ravel()))
        self.__dict__.pop('_dev', None)
        self.__dict__.pop('_tab', None)

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_dev', None)                 # device handles do not pickle
        state.pop('_tab', None)
        return state

    def __call__(self, *x_interp):
        """evaluate at coordinates `x_interp`; the output has the shape of the
        broadcast coordinate inputs."""
        assert len(x_interp) == self.ndim
        x_mesh = np.broadcast_arrays(*x_interp)
        shape = x_mesh[0].shape
        dt = self.values.dtype
        x_stack = np.ascontiguousarray(np.vstack([np.asarray(x, dtype=dt).ravel() for x in x_mesh]))
        if self.ndim > 4:
            raise Exception("Can't interpolate in dimension strictly greater than 5")  # pyx:47
        dev = self.__dict__.get('_dev')
        if dev is None:
            dev = self.__dict__['_dev'] = _DeviceValues(self._xmin, self._xmax, self._xshape,
                                                       self.values)
        return dev.eval(x_stack).reshape(shape)
