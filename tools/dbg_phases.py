import sys, io, contextlib
sys.path.insert(0, '/root/repo')
import numpy as np
from stodynprog_amd import models
def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)
for trial in range(3):
    _, a = models.storage_ar1(n_E=1200, n_P=1000, steps=(1.0, 0.1))
    _, b = models.storage_ar1(n_E=1200, n_P=1000, steps=(1.0, 0.1))
    V = np.random.default_rng(21).standard_normal(a._state_grid_shape)
    Ja, pa = quiet(a.value_iteration, V, False)
    Jb, pb = quiet(b.value_iterations, V, 1, False)
    bad = Ja != Jb
    print('trial', trial, 'kernel', a.backend_info['kernel'], a.backend_info.get('filter_form'), 'bad', int(bad.sum()), 'of', bad.size)
    if bad.any():
        r, c = np.nonzero(bad)
        print(' rows', r.min(), r.max(), 'cols', c.min(), c.max(), 'distinct cols', len(np.unique(c)), 'distinct rows', len(np.unique(r)))
        print(' cols sample', np.unique(c)[:40])
        print(' rows sample', np.unique(r)[:40])
        print(' Jb finite there', np.isfinite(Jb[bad]).all(), 'Ja sample', Ja[bad][:5])
    Ja2, _ = quiet(a.value_iteration, V, False)
    print('  second call bad', int((Ja2 != Jb).sum()))
