"""Reading (and writing) interpolator pickles in the reference's format.

The reference's Searev example saves its optimal policy as a pickled
`stodynprog.stodynprog.MlinInterpolator` (reference
examples/20 Searev storage control/storage_control.py:217-223, read back at
storage_simulation.py:180-182; the file P_sto_law.dat is a Python 2,
protocol 0 pickle).  `load_interpolator` opens such files -- old-style class
instance, latin-1 numpy payloads -- as a `stodynprog_amd.MlinInterpolator`;
`dump_interpolator` writes a file the reference can load back.
"""
import pickle

import numpy as np

from .interp import MlinInterpolator

__all__ = ['load_interpolator', 'dump_interpolator']

_REFERENCE_PATHS = {('stodynprog.stodynprog', 'MlinInterpolator'),
                    ('stodynprog_amd.interp', 'MlinInterpolator'),
                    ('stodynprog', 'MlinInterpolator')}


# what an interpolator pickle of the reference actually needs (Python 2 old-style
# instance or Python 3 object + numpy arrays); anything else is refused: a .dat
# file is data, not a program
_ALLOWED_GLOBALS = {
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
    ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
    ('numpy', 'ndarray'), ('numpy', 'dtype'),
    ('copy_reg', '_reconstructor'), ('copyreg', '_reconstructor'),
    ('__builtin__', 'object'), ('builtins', 'object'),
    ('_codecs', 'encode'),                       # numpy's py3 protocol-2 payload: encode(str, 'latin1')
}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _REFERENCE_PATHS:
            return MlinInterpolator
        if (module, name) in _ALLOWED_GLOBALS:
            return super(_Unpickler, self).find_class(module, name)
        raise pickle.UnpicklingError(
            'interpolator pickle refers to {}.{}: only MlinInterpolator instances made of '
            'numpy arrays are accepted'.format(module, name))


def load_interpolator(file):
    """Load a pickled MlinInterpolator written by the reference (Python 2 or 3)
    or by this package.  `file`: path or binary file object."""
    if isinstance(file, (str, bytes)):
        with open(file, 'rb') as f:
            return load_interpolator(f)
    obj = _Unpickler(file, encoding='latin1').load()
    if not isinstance(obj, MlinInterpolator):
        raise TypeError('the pickle holds a {}, not an MlinInterpolator'.format(type(obj).__name__))
    # normalise what the sweep and the C ABI expect
    obj.ndim = int(obj.ndim)
    obj._xmin = np.asarray(obj._xmin, dtype=float)
    obj._xmax = np.asarray(obj._xmax, dtype=float)
    obj._xshape = np.asarray(obj._xshape, dtype=np.int64)
    if obj.values is not None:
        obj.values = np.ascontiguousarray(np.atleast_2d(obj.values), dtype=float)
    return obj


def dump_interpolator(interp, file, protocol=2):
    """Pickle `interp` so that `pickle.load` under the reference package gives a
    `stodynprog.stodynprog.MlinInterpolator` with the same attributes."""
    if isinstance(file, (str, bytes)):
        with open(file, 'wb') as f:
            return dump_interpolator(interp, f, protocol)
    # An instance pickle only records the class path and the attribute dict:
    # write it as a plain object of the reference's class path.
    state = interp.__getstate__()
    shim = type('MlinInterpolator', (object,), {})
    shim.__module__ = 'stodynprog.stodynprog'
    obj = shim.__new__(shim)
    obj.__dict__.update(state)
    import copyreg
    import io
    buf = io.BytesIO()
    p = pickle.Pickler(buf, protocol)
    p.dispatch_table = copyreg.dispatch_table.copy()
    p.dispatch_table[shim] = lambda o: (copyreg._reconstructor, (shim, object, None), o.__dict__)
    # make the shim importable by name during pickling
    import sys
    import types
    mod = sys.modules.get('stodynprog.stodynprog')
    created = []
    if mod is None:
        pkg = sys.modules.get('stodynprog')
        if pkg is None:
            pkg = types.ModuleType('stodynprog')
            sys.modules['stodynprog'] = pkg
            created.append('stodynprog')
        mod = types.ModuleType('stodynprog.stodynprog')
        sys.modules['stodynprog.stodynprog'] = mod
        created.append('stodynprog.stodynprog')
    had = getattr(mod, 'MlinInterpolator', None)
    mod.MlinInterpolator = shim
    try:
        p.dump(obj)
    finally:
        if had is None:
            try:
                delattr(mod, 'MlinInterpolator')
            except AttributeError:
                pass
        else:
            mod.MlinInterpolator = had
        for name in created:
            sys.modules.pop(name, None)
    file.write(buf.getvalue())
