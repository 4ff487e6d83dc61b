"""System description layer: the drop-in `SysDescription` class.

Host-side mirror of the reference's description API (reference
stodynprog/stodynprog.py:19-247): same constructor, properties, attribute
names, validation rules and error texts, so user scripts only change their
import line.  Pure Python; the callables stored here are traced into device
code by `DPSolver` (see trace.py / codegen.py), never called on the GPU.
"""
from __future__ import division, print_function
import inspect

__all__ = ['SysDescription', '_enforce_sig_len', '_zero_cost']


def _zero_cost(*x):
    """Default terminal cost g(x) = 0 (reference sdp.py:19-21)."""
    return 0.


def _positional_names(fun):
    """Names of the positional parameters and the **kwargs catch-all name."""
    spec = inspect.getfullargspec(fun)      # getargspec is gone in Python >= 3.11
    return list(spec.args), spec.varkw


def _enforce_sig_len(fun, args, with_params, shortname=None):
    """Check that `fun` takes exactly len(args) positional arguments, and that
    it has a **kwargs catch-all iff the system carries parameters.

    Returns True, or raises ValueError with the reference's message format
    (sdp.py:24-53; text pinned by the reference's tests/test_stodynprog.py:58):
        "<shortname>'<name>' should accept N args (a, b), not M"
    """
    names, varkw = _positional_names(fun)
    prefix = '' if shortname is None else shortname
    prefix += "'{:s}' ".format(fun.__name__)
    if len(names) != len(args):
        raise ValueError(prefix + 'should accept {:d} args ({:s}), not {:d}'.format(
            len(args), ', '.join(args), len(names)))
    if with_params and varkw is None:
        raise ValueError(prefix + 'should accept extra keyword arguments')
    if not with_params and varkw is not None:
        raise ValueError(prefix + 'should not accept extra keyword arguments')
    return True


class SysDescription(object):
    """Dynamical system x_{k+1} = f(x_k, u_k, w_k) with instant cost
    g(x_k, u_k, w_k) and admissible control box U(x_k), as seen by the
    dynamic-programming solver.

    dims : (n_state, n_control[, n_perturb]).
    """

    def __init__(self, dims, stationnary=True, name='', params=None):
        self.name = name
        self.stationnary = bool(stationnary)
        self.params = params if params is not None else {}
        if len(dims) == 3:
            n_state, n_control, n_perturb = dims
        elif len(dims) == 2:
            (n_state, n_control), n_perturb = dims, 0
        else:
            raise ValueError('dims tuple should be of len 2 or 3')
        self.state = ['x{:d}'.format(i + 1) for i in range(n_state)]
        self.control = ['u{:d}'.format(i + 1) for i in range(n_control)]
        self.perturb = ['w{:d}'.format(i + 1) for i in range(n_perturb)]
        # expected signature of dyn and cost; time index first when time dependent
        self._dyn_args = self.state + self.control + self.perturb
        if not self.stationnary:
            self._dyn_args.insert(0, 'time_k')
        self._dyn = None
        self._cost = None
        self._control_box = None
        self._terminal_cost = _zero_cost
        self._perturb_laws = None

    # -- properties ---------------------------------------------------------
    @property
    def stochastic(self):
        """True when the system has at least one perturbation variable."""
        return len(self.perturb) > 0

    @property
    def dyn(self):
        """dynamics function x_{k+1} = f_k(x_k, u_k, w_k)"""
        return self._dyn

    @dyn.setter
    def dyn(self, dyn):
        if _enforce_sig_len(dyn, self._dyn_args, bool(self.params), 'dynamics function'):
            self._dyn = dyn
        # variable names are taken over from the signature of `dyn` (sdp.py:119-131)
        names, _ = _positional_names(dyn)
        self._dyn_args = names
        if not self.stationnary:
            names = names[1:]
        ns, nc, nw = len(self.state), len(self.control), len(self.perturb)
        self.state = names[:ns]
        self.control = names[ns:ns + nc]
        self.perturb = names[ns + nc:ns + nc + nw]

    @property
    def control_box(self):
        """admissible controls U_k(x_k) as a box: ((u1_min, u1_max), ...)"""
        return self._control_box

    @control_box.setter
    def control_box(self, control_box):
        args = list(self.state)
        if not self.stationnary:
            args.insert(0, 'time_k')
        if _enforce_sig_len(control_box, args, bool(self.params),
                            'control description function'):
            self._control_box = control_box

    @property
    def cost(self):
        """instant cost function g_k(x_k, u_k, w_k)"""
        return self._cost

    @cost.setter
    def cost(self, cost):
        if _enforce_sig_len(cost, self._dyn_args, bool(self.params), 'cost function'):
            self._cost = cost

    @property
    def terminal_cost(self):
        """terminal cost function g(x_K)"""
        return self._terminal_cost

    @terminal_cost.setter
    def terminal_cost(self, cost):
        names, _ = _positional_names(cost)
        if len(names) != len(self.state):
            raise ValueError('cost function should accept '
                             '{:d} args instead of {:d}'.format(len(self.state), len(names)))
        self._terminal_cost = cost

    @property
    def perturb_laws(self):
        """distribution laws of the perturbations w_k"""
        return self._perturb_laws

    @perturb_laws.setter
    def perturb_laws(self, laws):
        if len(laws) != len(self.perturb):
            raise ValueError('{:d} perturbation laws should be provided'
                             .format(len(self.perturb)))
        self._perturb_laws = laws
        # a law with a density is continuous, one with a mass function discrete
        kinds = []
        for law in laws:
            try:
                law.pdf(0)
                kinds.append('continuous')
                continue
            except AttributeError:
                pass
            try:
                law.pmf(0)
                kinds.append('discrete')
            except AttributeError:
                raise ValueError('perturbation law {:s} should either have a pdf '
                                 'or a pmf method'.format(repr(law)))
        self.perturb_types = kinds

    # -- reporting ------------------------------------------------------------
    def print_summary(self):
        """summary information about the dynamical system"""
        print('Dynamical system "{}" description'.format(self.name))
        print('* behavioral properties: {}, {}'.format(
            'stationnary' if self.stationnary else 'time dependent',
            'stochastic' if self.stochastic else 'deterministic'))
        print('* functions:')
        funcs = [('dynamics', self.dyn), ('cost', self.cost),
                 ('control box', self.control_box)]
        width = max(len(label) for label, _ in funcs) + 1
        for label, fun in funcs:
            where = ('None (to be defined)' if fun is None
                     else '{0.__module__}.{0.__name__}'.format(fun))
            print('  - {0:{width}}: {1}'.format(label, where, width=width))
        print('* variables')
        vects = [('state', self.state), ('control', self.control)]
        if self.stochastic:
            vects.append(('perturbation', self.perturb))
        width = max(len(label) for label, _ in vects) + 1
        for label, vect in vects:
            print('  - {0:{width}}: {1} (dim {2:d})'.format(
                label, ', '.join(vect), len(vect), width=width))

    def __repr__(self):
        return '<SysDescription "{:s}" at 0x{:x}>'.format(self.name, id(self))
