"""Symbolic tracing of the user's model callables.

The reference evaluates `dyn`, `cost` (numpy expressions written by the user)
inside its hot loop (reference stodynprog/stodynprog.py:674,676).  Device code
cannot call Python, so `DPSolver` calls each callable ONCE with `Sym` proxies
for x, u, w (and t_k); the proxies record every arithmetic operator and numpy
ufunc into an expression DAG, which codegen.py turns into a HIP device
function with one IEEE operation per recorded operator, in the recorded order
(contraction off), so results match numpy bit for bit for + - * / sqrt abs,
comparisons and selects.

Lookup tables are traceable through `np.interp(x, xp, fp)` (concrete tables).
Anything that needs a concrete value (`if x > 0:`, `float(x)`, `max(a, b)`,
indexing an array with x) raises TraceError; the solver then reports the model
as not traceable (tabulated mode evaluates such callables on the host).
"""
import math
import numbers

import numpy as np

__all__ = ['TraceError', 'Graph', 'Sym', 'trace_model', 'trace_box', 'TracedBox', 'callable_fingerprint']


class TraceError(Exception):
    """The callable cannot be expressed as a straight-line expression DAG."""


# dependency bits
# x0 (the leading state axis) is tracked apart from the other state variables
DEP_X, DEP_U, DEP_W, DEP_T, DEP_XR = 1, 2, 4, 8, 16

# op -> (arity, result kind)   kind: 'r' real, 'b' bool
_OPS = {
    'add': (2, 'r'), 'sub': (2, 'r'), 'mul': (2, 'r'), 'div': (2, 'r'),
    'neg': (1, 'r'), 'abs': (1, 'r'), 'sqrt': (1, 'r'), 'square': (1, 'r'),
    'recip': (1, 'r'),
    'pow': (2, 'r'), 'min': (2, 'r'), 'max': (2, 'r'), 'fmin': (2, 'r'), 'fmax': (2, 'r'),
    'floor': (1, 'r'), 'ceil': (1, 'r'), 'trunc': (1, 'r'), 'rint': (1, 'r'),
    'sign': (1, 'r'), 'exp': (1, 'r'), 'exp2': (1, 'r'), 'expm1': (1, 'r'),
    'log': (1, 'r'), 'log2': (1, 'r'), 'log10': (1, 'r'), 'log1p': (1, 'r'),
    'sin': (1, 'r'), 'cos': (1, 'r'), 'tan': (1, 'r'), 'asin': (1, 'r'),
    'acos': (1, 'r'), 'atan': (1, 'r'), 'sinh': (1, 'r'), 'cosh': (1, 'r'),
    'tanh': (1, 'r'), 'atan2': (2, 'r'), 'hypot': (2, 'r'), 'fmod': (2, 'r'),
    'pymod': (2, 'r'), 'floordiv': (2, 'r'), 'cbrt': (1, 'r'),
    'lt': (2, 'b'), 'le': (2, 'b'), 'gt': (2, 'b'), 'ge': (2, 'b'),
    'eq': (2, 'b'), 'ne': (2, 'b'),
    'and': (2, 'b'), 'or': (2, 'b'), 'xor': (2, 'b'), 'not': (1, 'b'),
    'isnan': (1, 'b'), 'isfinite': (1, 'b'), 'isinf': (1, 'b'),
    'select': (3, 'r'), 'bselect': (3, 'b'), 'b2r': (1, 'r'),
    'interp1': (1, 'r'),         # np.interp(x, xp, fp): node.value = index into Graph.tables
}

# ops whose device result is the correctly rounded IEEE result (bit-exact vs numpy)
EXACT_OPS = {'add', 'sub', 'mul', 'div', 'neg', 'abs', 'sqrt', 'square', 'recip',
             'min', 'max', 'fmin', 'fmax', 'floor', 'ceil', 'trunc', 'rint', 'sign',
             'lt', 'le', 'gt', 'ge', 'eq', 'ne', 'and', 'or', 'xor', 'not', 'isnan',
             'isfinite', 'isinf', 'select', 'bselect', 'b2r', 'var', 'const', 'bconst',
             'interp1'}


class Node(object):
    __slots__ = ('id', 'op', 'args', 'value', 'kind', 'deps')

    def __init__(self, id, op, args, value, kind, deps):
        self.id, self.op, self.args, self.value = id, op, args, value
        self.kind, self.deps = kind, deps


class Graph(object):
    """Hash-consed expression DAG (common sub-expressions are shared)."""

    def __init__(self, unique_consts=False):
        self.nodes = []
        self._memo = {}
        # step-by-step traces (trace_model(t_value=...)): every occurrence of a real
        # constant is its own node, so that the DAG's shape cannot depend on which
        # VALUES happen to coincide (data of a night hour equal to a literal 0)
        self.unique_consts = unique_consts
        self.tables = []          # lookup tables of np.interp nodes: (xp, fp, left, right)
        self.scalar_pow = set()   # ids of nodes standing for a numpy SCALAR power (see _power)
        self.fork = None          # trace_box: the decisions of the path being traced (_Fork)

    def _intern(self, key, op, args, value, kind, deps):
        n = self._memo.get(key)
        if n is None:
            n = Node(len(self.nodes), op, args, value, kind, deps)
            self.nodes.append(n)
            self._memo[key] = n
        return n

    def var(self, name, dep):
        return self._intern(('var', name), 'var', (), name, 'r', dep)

    def const(self, value):
        value = float(value)
        # key on the bit pattern: -0.0 and 0.0, and NaN payloads, stay distinct
        key = ('const', np.float64(value).tobytes())
        if self.unique_consts:
            key = ('const', len(self.nodes))
        return self._intern(key, 'const', (), value, 'r', 0)

    def bconst(self, value):
        return self._intern(('bconst', bool(value)), 'bconst', (), bool(value), 'b', 0)

    def interp(self, x, xp, fp, left, right):
        """np.interp(x, xp, fp, left, right) with concrete 1-D tables"""
        key = (xp.tobytes(), fp.tobytes(), np.float64(left).tobytes(), np.float64(right).tobytes())
        for tid, t in enumerate(self.tables):
            if t[4] == key:
                break
        else:
            tid = len(self.tables)
            self.tables.append((xp, fp, float(left), float(right), key))
        return self._intern(('interp1', x.id, tid), 'interp1', (x,), tid, 'r', x.deps)

    def op(self, op, *args):
        arity, kind = _OPS[op]
        assert len(args) == arity, op
        deps = 0
        for a in args:
            deps |= a.deps
        return self._intern((op,) + tuple(a.id for a in args), op, tuple(args), None,
                            kind, deps)


def _is_plain_number(v):
    if isinstance(v, (bool, np.bool_)):
        return False
    if isinstance(v, numbers.Real):
        return True
    if isinstance(v, np.ndarray) and v.ndim == 0 and v.dtype.kind in 'fiu':
        return True
    return False


class Sym(object):
    """Proxy standing for a real (or boolean) array of lattice shape."""
    __array_priority__ = 1000.
    __slots__ = ('g', 'n')

    def __init__(self, graph, node):
        self.g, self.n = graph, node

    # ---- lifting ------------------------------------------------------------
    def _lift(self, other):
        if isinstance(other, Sym):
            if other.g is not self.g:
                raise TraceError('mixing symbols of two different traces')
            return other.n
        if isinstance(other, (bool, np.bool_)):
            return self.g.bconst(other)
        if _is_plain_number(other):
            return self.g.const(float(other))
        if isinstance(other, np.ndarray) and other.size == 1 and other.dtype.kind in 'fiu':
            return self.g.const(float(other.reshape(())))
        if isinstance(other, np.ndarray) and other.size == 1 and other.dtype.kind == 'b':
            return self.g.bconst(bool(other.reshape(())))
        raise TraceError('cannot trace operand of type {}'.format(type(other).__name__))

    def _real(self, node):
        """coerce a bool node to real (numpy: True -> 1.0 in arithmetic)"""
        return self.g.op('b2r', node) if node.kind == 'b' else node

    def _bin(self, op, a, b):
        a, b = self._lift(a), self._lift(b)
        if op in ('and', 'or', 'xor'):
            if a.kind != 'b' or b.kind != 'b':
                raise TraceError('bitwise operator on non-boolean operands')
        elif op in ('eq', 'ne') and a.kind == 'b' and b.kind == 'b':
            x = self.g.op('xor', a, b)
            return Sym(self.g, x if op == 'ne' else self.g.op('not', x))
        else:
            a, b = self._real(a), self._real(b)
        return Sym(self.g, self.g.op(op, a, b))

    def _un(self, op, a):
        a = self._lift(a)
        if op == 'not':
            if a.kind != 'b':
                raise TraceError('logical not of a non-boolean')
        elif op not in ('isnan', 'isfinite', 'isinf') or a.kind == 'b':
            a = self._real(a)
        return Sym(self.g, self.g.op(op, a))

    # ---- arithmetic ------------------------------------------------------------
    def __add__(self, o): return self._bin('add', self, o)
    def __radd__(self, o): return self._bin('add', o, self)
    def __sub__(self, o): return self._bin('sub', self, o)
    def __rsub__(self, o): return self._bin('sub', o, self)
    def __mul__(self, o): return self._bin('mul', self, o)
    def __rmul__(self, o): return self._bin('mul', o, self)
    def __truediv__(self, o): return self._bin('div', self, o)
    def __rtruediv__(self, o): return self._bin('div', o, self)
    def __floordiv__(self, o): return self._bin('floordiv', self, o)
    def __rfloordiv__(self, o): return self._bin('floordiv', o, self)
    def __mod__(self, o): return self._bin('pymod', self, o)
    def __rmod__(self, o): return self._bin('pymod', o, self)
    def __neg__(self): return self._un('neg', self)
    def __pos__(self): return self
    def __abs__(self): return self._un('abs', self)

    def __pow__(self, o):
        return _power(self, o)

    def __rpow__(self, o):
        return _power(o, self)

    # ---- comparisons / logic ------------------------------------------------------
    def __lt__(self, o): return self._bin('lt', self, o)
    def __le__(self, o): return self._bin('le', self, o)
    def __gt__(self, o): return self._bin('gt', self, o)
    def __ge__(self, o): return self._bin('ge', self, o)
    def __eq__(self, o): return self._bin('eq', self, o)
    def __ne__(self, o): return self._bin('ne', self, o)
    __hash__ = None
    def __and__(self, o): return self._bin('and', self, o)
    def __rand__(self, o): return self._bin('and', o, self)
    def __or__(self, o): return self._bin('or', self, o)
    def __ror__(self, o): return self._bin('or', o, self)
    def __xor__(self, o): return self._bin('xor', self, o)
    def __rxor__(self, o): return self._bin('xor', o, self)
    def __invert__(self): return self._un('not', self)

    # ---- things that need a concrete value -------------------------------------------
    def __bool__(self):
        fork = getattr(self.g, 'fork', None)
        if fork is not None:
            # trace_box: the callable is run once per PATH through its data-dependent branches (see _Fork)
            return fork.decide(self)
        raise TraceError('the truth value of a symbolic expression is needed '
                         '(Python `if`/`max`/`min` on state, control or perturbation)')
    __nonzero__ = __bool__

    def __float__(self):
        raise TraceError('float() of a symbolic expression')

    def __int__(self):
        raise TraceError('int() of a symbolic expression')

    def __index__(self):
        raise TraceError('symbolic expression used as an index')

    def __len__(self):
        raise TraceError('len() of a symbolic expression')

    def __iter__(self):
        raise TraceError('iteration over a symbolic expression')

    def __getitem__(self, key):
        if key is Ellipsis or key == ():
            return self
        raise TraceError('indexing a symbolic expression')

    # ---- ndarray-like conveniences used in model code --------------------------------------
    @property
    def shape(self):
        raise TraceError('.shape of a symbolic expression')

    def astype(self, dtype, **kw):
        dt = np.dtype(dtype)
        if dt.kind == 'f':
            return Sym(self.g, self._real(self.n))
        if dt.kind == 'b' and self.n.kind == 'b':
            return self
        raise TraceError('astype({}) is not traceable'.format(dt))

    def clip(self, a_min=None, a_max=None, **kw):
        return _clip(self, a_min, a_max)

    def copy(self):
        return self

    # ---- numpy protocols --------------------------------------------------------------------
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != '__call__' or kwargs.get('out') is not None:
            raise TraceError('ufunc method {}.{} is not traceable'.format(ufunc.__name__, method))
        extra = set(kwargs) - {'dtype', 'casting', 'order', 'subok', 'where'}
        if extra or kwargs.get('where', True) is not True:
            raise TraceError('ufunc keyword(s) {} not traceable'.format(sorted(extra)))
        name = ufunc.__name__
        sym = next(i for i in inputs if isinstance(i, Sym))
        if name in _UFUNC_BIN:
            return sym._bin(_UFUNC_BIN[name], inputs[0], inputs[1])
        if name in _UFUNC_UN:
            return sym._un(_UFUNC_UN[name], inputs[0])
        if name in ('power', 'float_power'):
            return _power(inputs[0], inputs[1], ufunc=True)
        if name == 'heaviside':
            x, h0 = inputs
            # 0 for x < 0, h0 for x == 0, 1 for x > 0, NaN for NaN (numpy's definition)
            return _where(sym._bin('lt', x, 0.0), 0.0,
                          _where(sym._bin('gt', x, 0.0), 1.0,
                                 _where(sym._bin('eq', x, 0.0), h0, x)))
        if name == 'positive':
            return sym
        if name == 'reciprocal':
            return sym._bin('div', 1.0, inputs[0])
        if name in ('logical_and', 'logical_or', 'logical_xor', 'bitwise_and',
                    'bitwise_or', 'bitwise_xor'):
            op = name.split('_')[1]
            a, b = (_as_bool(sym, i) for i in inputs)
            return Sym(sym.g, sym.g.op(op, a, b))
        if name in ('logical_not', 'invert'):
            return Sym(sym.g, sym.g.op('not', _as_bool(sym, inputs[0])))
        raise TraceError('numpy ufunc `{}` is not traceable'.format(name))

    def __array_function__(self, func, types, args, kwargs):
        name = func.__name__
        if name == 'where' and len(args) == 3 and not kwargs:
            return _where(*args)
        if name == 'clip':
            a = args[0]
            lo = args[1] if len(args) > 1 else kwargs.get('a_min', kwargs.get('min'))
            hi = args[2] if len(args) > 2 else kwargs.get('a_max', kwargs.get('max'))
            return _clip(a, lo, hi)
        if name in ('asarray', 'asanyarray', 'ascontiguousarray', 'array', 'copy',
                    'squeeze', 'ravel', 'atleast_1d', 'real'):
            a = args[0]
            if isinstance(a, Sym):
                return a
        if name in ('abs', 'absolute'):
            return abs(args[0])
        if name in ('zeros_like', 'ones_like', 'full_like') and isinstance(args[0], Sym):
            fill = {'zeros_like': 0., 'ones_like': 1.}.get(name)
            if fill is None:
                fill = args[1] if len(args) > 1 else kwargs['fill_value']
            return Sym(args[0].g, args[0]._lift(fill))
        if name in ('isnan', 'isfinite', 'isinf'):
            return args[0]._un(name, args[0])
        if name == 'interp':
            return _interp(*args, **kwargs)
        if name == 'select':
            return _select(*args, **kwargs)
        raise TraceError('numpy function `{}` is not traceable'.format(name))


_UFUNC_BIN = {
    'add': 'add', 'subtract': 'sub', 'multiply': 'mul', 'divide': 'div',
    'true_divide': 'div', 'minimum': 'min', 'maximum': 'max', 'fmin': 'fmin',
    'fmax': 'fmax', 'less': 'lt', 'less_equal': 'le', 'greater': 'gt',
    'greater_equal': 'ge', 'equal': 'eq', 'not_equal': 'ne', 'arctan2': 'atan2',
    'hypot': 'hypot', 'fmod': 'fmod', 'remainder': 'pymod', 'mod': 'pymod',
    'floor_divide': 'floordiv',
}
_UFUNC_UN = {
    'negative': 'neg', 'absolute': 'abs', 'fabs': 'abs', 'sqrt': 'sqrt', 'square': 'square',
    'floor': 'floor', 'ceil': 'ceil', 'trunc': 'trunc', 'rint': 'rint', 'sign': 'sign',
    'exp': 'exp', 'exp2': 'exp2', 'expm1': 'expm1', 'log': 'log', 'log2': 'log2',
    'log10': 'log10', 'log1p': 'log1p', 'sin': 'sin', 'cos': 'cos', 'tan': 'tan',
    'arcsin': 'asin', 'arccos': 'acos', 'arctan': 'atan', 'sinh': 'sinh', 'cosh': 'cosh',
    'tanh': 'tanh', 'cbrt': 'cbrt', 'isnan': 'isnan', 'isfinite': 'isfinite',
    'isinf': 'isinf',
}


def _any_sym(*vals):
    for v in vals:
        if isinstance(v, Sym):
            return v
    raise TraceError('no symbolic operand')


def _as_bool(sym, v):
    n = sym._lift(v)
    if n.kind == 'b':
        return n
    return sym.g.op('ne', n, sym.g.const(0.0))      # numpy truthiness of a real


def _power(base, expo, ufunc=False):
    """`base ** expo` (operator) or `np.power(base, expo)` (ufunc=True), as numpy
    evaluates them:
      * ndarray ** python scalar takes numpy's fast paths (number.c,
        fast_scalar_power): 2 -> square, 1 -> copy, 0.5 -> sqrt, -1 -> reciprocal,
        0 -> ones; every other exponent runs the pow loop;
      * np.power only special-cases the exponent 2 (x * x in the loop);
      * a sub-expression that depends on the state (and time) only is a numpy
        SCALAR when the reference evaluates the callable node by node
        (stodynprog.py:674-676): scalar ** goes through libm pow, which differs
        from the correctly rounded x*x / sqrt / 1/x in ~0.09 % of the values by
        one ulp [measured].  The device keeps the correctly rounded operation (the
        closest it can get) and the node is flagged: the model then reports
        `scalar_pow` among its inexact operations instead of claiming bit parity."""
    sym = _any_sym(base, expo)
    if _is_plain_number(expo):
        e = float(expo)
        b = sym._real(sym._lift(base))
        op = None
        if e == 2.0:
            op = 'square'
        elif not ufunc:
            if e == 1.0:
                return Sym(sym.g, b)
            if e == 0.0:
                return Sym(sym.g, sym.g.const(1.0))
            op = {0.5: 'sqrt', -1.0: 'recip'}.get(e)
        if op is not None:
            node = sym.g.op(op, b)
            if not ufunc and (b.deps & (DEP_U | DEP_W)) == 0:
                sym.g.scalar_pow.add(node.id)
            return Sym(sym.g, node)
    return sym._bin('pow', base, expo)


def _where(cond, a, b):
    sym = _any_sym(cond, a, b)
    g = sym.g
    if isinstance(cond, (bool, np.bool_)):
        pick = a if cond else b
        return pick if isinstance(pick, Sym) else Sym(g, sym._lift(pick))
    c = _as_bool(sym, cond)
    an, bn = sym._lift(a), sym._lift(b)
    if an.kind == 'b' and bn.kind == 'b':
        return Sym(g, g.op('bselect', c, an, bn))
    return Sym(g, g.op('select', c, sym._real(an), sym._real(bn)))


def _select(condlist, choicelist, default=0):
    """np.select: the first true condition wins -- nested selects, last to first"""
    if len(condlist) != len(choicelist) or not condlist:
        raise TraceError('np.select needs condition and choice lists of equal non-zero length')
    out = default
    for cond, choice in zip(reversed(list(condlist)), reversed(list(choicelist))):
        out = _where(cond, choice, out)
    return out


def _interp(x, xp, fp, left=None, right=None, period=None):
    """np.interp with a symbolic abscissa and concrete tables (an efficiency
    curve, a tariff): recorded as one `interp1` node; the device function
    generated for it (codegen.INTERP_SOURCE) repeats numpy's arithmetic
    (numpy/_core/src/multiarray/compiled_base.c, arr_interp) operation for
    operation, so the node is bit-exact like + - * /."""
    if not isinstance(x, Sym) or isinstance(xp, Sym) or isinstance(fp, Sym):
        raise TraceError('np.interp is traceable for a symbolic x and concrete tables only')
    if period is not None:
        raise TraceError('np.interp(period=...) is not traceable')
    try:
        xp = np.ascontiguousarray(xp, dtype=np.float64)
        fp = np.ascontiguousarray(fp, dtype=np.float64)
    except (TypeError, ValueError):
        raise TraceError('np.interp tables must be real arrays')
    if xp.ndim != 1 or fp.ndim != 1 or xp.size != fp.size or xp.size == 0:
        raise TraceError('np.interp tables must be 1-D, of equal non-zero length')
    if isinstance(left, Sym) or isinstance(right, Sym):
        raise TraceError('np.interp left/right must be numbers')
    left = fp[0] if left is None else float(left)
    right = fp[-1] if right is None else float(right)
    return Sym(x.g, x.g.interp(x._real(x.n), xp, fp, left, right))


def _clip(a, lo, hi):
    # np.clip(a, lo, hi) == minimum(maximum(a, lo), hi)
    sym = _any_sym(a, lo, hi)
    out = a
    if lo is not None:
        out = sym._bin('max', out, lo)
    if hi is not None:
        out = sym._bin('min', out, hi)
    return out


class TracedModel(object):
    """Result of tracing one (dyn, cost) pair."""

    def __init__(self, graph, x_next, cost, n_state, n_control, n_perturb, time_dep):
        self.graph = graph
        self.x_next = x_next          # list of Node (real), one per state axis
        self.cost = cost              # Node (real)
        self.n_state, self.n_control, self.n_perturb = n_state, n_control, n_perturb
        self.time_dep = time_dep
        self.t_value = None           # concrete time index of a step-by-step trace
        self.param_index = None       # node id -> slot, after lift_constants()

    def lift_constants(self):
        """Turn every real constant of the live DAG into a kernel parameter
        (slot order = recording order): codegen then emits `sdp_model_prm[slot]`
        instead of a literal, so models that differ only by constants -- the time
        steps of a finite-horizon problem with time-indexed data -- share one
        code object.  Returns the parameter values of THIS model."""
        consts = [n for n in self.live_nodes() if n.op == 'const']
        self.param_index = {n.id: i for i, n in enumerate(consts)}
        return [n.value for n in consts]

    def param_values(self):
        return [n.value for n in self.live_nodes() if n.op == 'const']

    def structure_key(self):
        """Text identifying the live DAG up to the values of its real
        constants (two models with equal keys generate the same source once
        their constants are lifted)."""
        kept = getattr(self, '_structure_key', None)
        if kept is not None and kept[0] == (len(self.graph.nodes), self.param_index is not None):
            return kept[1]
        live = self.live_nodes()
        pos = {n.id: i for i, n in enumerate(live)}
        parts = []
        for n in live:
            if n.op == 'const':
                parts.append('c')
            elif n.op in ('var', 'bconst'):
                parts.append('{}:{}'.format(n.op, n.value))
            elif n.op == 'interp1':
                import hashlib
                digest = hashlib.sha256(b''.join(self.graph.tables[n.value][4])).hexdigest()[:16]
                parts.append('interp1[{}]({})'.format(digest, pos[n.args[0].id]))
            else:
                parts.append('{}({})'.format(n.op, ','.join(str(pos[a.id]) for a in n.args)))
        outs = [pos[n.id] for n in self.x_next] + [pos[self.cost.id]]
        key = ';'.join(parts) + '|' + ','.join(map(str, outs))
        self._structure_key = ((len(self.graph.nodes), self.param_index is not None), key)     # (the live DAG of a trace does not change)
        return key

    def live_nodes(self):
        """Nodes reachable from the outputs, in topological (creation) order."""
        seen = set()
        stack = list(self.x_next) + [self.cost]
        while stack:
            n = stack.pop()
            if n.id in seen:
                continue
            seen.add(n.id)
            stack.extend(n.args)
        return [n for n in self.graph.nodes if n.id in seen]

    @property
    def bit_exact(self):
        """True when every op is correctly rounded on the device (so the model
        evaluates bit-identically to numpy on the host)."""
        return not self.inexact_ops()

    def inexact_ops(self):
        live = self.live_nodes()
        ops = {n.op for n in live if n.op not in EXACT_OPS}
        if any(n.id in self.graph.scalar_pow for n in live):
            ops.add('scalar_pow')
        return sorted(ops)

    def slice_nodes(self, outputs):
        """Nodes reachable from `outputs` only, in topological order."""
        seen = set()
        stack = list(outputs)
        while stack:
            n = stack.pop()
            if n.id in seen:
                continue
            seen.add(n.id)
            stack.extend(n.args)
        return [n for n in self.graph.nodes if n.id in seen]

    def slice_nodes_until(self, outputs, stop_ids):
        """Nodes reachable from `outputs` without passing through a node of `stop_ids`
        (those nodes themselves excluded), in topological order."""
        seen = set()
        stack = [n for n in outputs if n.id not in stop_ids]
        while stack:
            n = stack.pop()
            if n.id in seen:
                continue
            seen.add(n.id)
            stack.extend(a for a in n.args if a.id not in stop_ids)
        return [n for n in self.graph.nodes if n.id in seen]

    def control_uniform_frontier(self, lead=None):
        """(`lead`: the node standing for x0' -- its w-free part a of `lead_split` when the
        perturbation reaches x0'.)  Sub-expressions of x0' and of the cost that depend on the control (and possibly on
        x_1.., t) but neither on the leading state variable x0 nor on w -- so they take the same
        value at every node of a column along axis 0 -- and feed an expression that does depend on
        x0 (or are x0' / the cost themselves).  With a control lattice shared by the nodes of a
        column they can be evaluated once per (column, control) instead of once per (node,
        control): the same operations on the same operands, hence the same bits
        (csrc/sdp_colfilter_kernel.h, SDP_COL_UTAB).  Returns the nodes in recording order, or None
        when there is none or a boolean is among them."""
        outs = [lead if lead is not None else self.x_next[0], self.cost]
        nodes = self.slice_nodes(outs)

        def uniform(n):
            return (n.deps & (DEP_X | DEP_W)) == 0 and (n.deps & DEP_U) != 0
        front = set()
        for n in nodes:
            if not uniform(n):
                front.update(a.id for a in n.args if uniform(a))
        front.update(o.id for o in outs if uniform(o))
        fr = [n for n in nodes if n.id in front]
        if not fr or any(n.kind != 'r' for n in fr):
            return None
        return fr

    def additive_control_split(self, frontier):
        """The shape the short first pass of the certified filter can use (csrc/sdp_colres_kernel.h,
        SDP_COL_LEAN2): x0' = X(x, t) +- a  and  cost = K(x, t) +- h  (or h alone, or no control in it), a and h
        nodes of the control table `frontier` (control_uniform_frontier), X and K free of the control and of the
        perturbation.  The argmin over the controls of a node does not see K, |K| + max |h| bounds the cost, and
        x0' is monotone in a -- so the smallest and the largest a of the column bound every position of the node.
        Returns None or a dict: lead = (X node, slot of a, form), cost = (K node or None, slot of h or None, form);
        form: 'add' (X + a), 'sub' (X - a) or 'rsub' (a - X)."""
        slot = {n.id: k for k, n in enumerate(frontier)}
        if self.cost_depends_on_w:
            return None
        lead_node = self.x_next[0]
        if self.lead_depends_on_w:
            # round 6: a perturbation that reaches x0' through FINAL sums, x0' = (X +- a) +- b_1(w) ..: the w-free part has the
            # shape (the shifted lattice's short pass); a regrouped chain's a* is not a value the reference computes: not taken
            sp = self.lead_split()
            if sp is None or self.lead_split_chain() is not None:
                return None
            lead_node = sp[0]

        def split(node, need_control):
            if node.id in slot:
                return (None, slot[node.id], 'add')
            if not (node.deps & DEP_U):
                return None if need_control else (node, None, 'add')
            if node.op not in ('add', 'sub') or node.kind != 'r':
                return None
            l, r = node.args
            for part, other, form in ((r, l, node.op), (l, r, 'add' if node.op == 'add' else 'rsub')):
                if part.id in slot and not (other.deps & (DEP_U | DEP_W)) and other.kind == 'r':
                    return (other, slot[part.id], form)
            return None
        lead = split(lead_node, True)
        cost = split(self.cost, False)
        if lead is None or cost is None or lead[0] is None:
            return None
        return dict(lead=lead, cost=cost)

    @property
    def storage_separable(self):
        """True when the next values of all state axes but the LEADING one
        depend neither on the control nor on the leading state variable:
            x0' = f0(x, u, w[, t])       x_k' = f_k(x_1.., w[, t])  for k >= 1
        i.e. a controlled stock next to an exogenous process driven by the
        noise -- the shape of every storage-control example of the reference
        (there f0 does not even depend on w).  The partial interpolation over
        axes 1.. is then a function of (row of axis 0, w) only and is shared by
        all controls and by all nodes of a column along axis 0 -- see
        csrc/sdp_column_kernel.h."""
        if self.n_state < 2:
            return False
        trail = 0
        for n in self.x_next[1:]:
            trail |= n.deps
        return (trail & (DEP_U | DEP_X)) == 0

    @property
    def column_shareable(self):
        """True when the next values of all state axes but the leading one do not
        depend on the LEADING state variable (they may depend on the control):
            x0' = f0(x, u, w[, t])       x_k' = f_k(x_1.., u, w[, t])  for k >= 1
        The nodes of a column along axis 0 then share, control by control, the partial
        interpolation over axes 1.. -- storage_separable is the special case without u.
        (The column kernel can use it when the nodes of a column also share their
        control values, i.e. the admissible box does not depend on x0.)"""
        if self.n_state < 2:
            return False
        trail = 0
        for n in self.x_next[1:]:
            trail |= n.deps
        return (trail & DEP_X) == 0

    @property
    def trail_depends_on_u(self):
        trail = 0
        for n in self.x_next[1:]:
            trail |= n.deps
        return bool(trail & DEP_U)

    def separable_axis_hint(self):
        """index k > 0 of a state variable that would make the model
        storage-separable if it came first (all other next-state values
        independent of the control and of x_k), or None.  Needs per-variable
        dependency sets, so the graph is walked once per candidate."""
        if self.n_state < 2 or self.storage_separable:
            return None
        reach = {}

        def leaves(node):
            if node.id not in reach:
                if node.op == 'var':
                    reach[node.id] = frozenset([node.value])
                else:
                    acc = frozenset()
                    for a in node.args:
                        acc = acc | leaves(a)
                    reach[node.id] = acc
            return reach[node.id]

        for k in range(1, self.n_state):
            ok = True
            for j, n in enumerate(self.x_next):
                if j == k:
                    continue
                deps = leaves(n)
                if 'x%d' % k in deps or any(v.startswith('u') for v in deps):
                    ok = False
                    break
            if ok:
                return k
        return None

    @property
    def lead_depends_on_w(self):
        """does x0' depend on the perturbation?  (if not, its interpolation
        cell is located once per control instead of once per lattice cell)"""
        return bool(self.x_next[0].deps & DEP_W)

    @property
    def cost_depends_on_w(self):
        return bool(self.cost.deps & DEP_W)

    def var_reach(self, node, _memo=None):
        """names of the variables (`x0`, `u1`, `w0`, `t`) `node` depends on"""
        memo = self.__dict__.setdefault('_reach', {})
        stack = [node]
        while stack:
            n = stack[-1]
            if n.id in memo:
                stack.pop()
                continue
            if n.op == 'var':
                memo[n.id] = frozenset([n.value])
                stack.pop()
                continue
            todo = [a for a in n.args if a.id not in memo]
            if todo:
                stack.extend(todo)
                continue
            acc = frozenset()
            for a in n.args:
                acc = acc | memo[a.id]
            memo[n.id] = acc
            stack.pop()
        return memo[node.id]

    def controlled_axes(self):
        """m such that the first m state variables are 'stocks' -- their next values may depend on the
        whole state and on the control but NOT on the perturbation -- and the others an exogenous
        process: next values that depend on x_m.., the perturbation and the time only:
            x_k' = f_k(x, u[, t])  for k < m        x_k' = f_k(x_m.., w[, t])  for k >= m
        (storage_separable is m = 1 with a stock that may also see w).  The expectation over w then
        commutes with the interpolation along the first m axes -- csrc/sdp_lead_kernel.h.  Returns the
        smallest such m >= 1, or None."""
        d = self.n_state
        reach = [self.var_reach(n) for n in self.x_next]
        for m in range(1, d + 1):
            lead_vars = set('x%d' % j for j in range(m))
            ok = all(not any(v.startswith('w') for v in reach[k]) for k in range(m))
            ok = ok and all(not (reach[k] & lead_vars) and not any(v.startswith('u') for v in reach[k])
                            for k in range(m, d))
            if ok:
                return m
        return None

    def controlled_order(self):
        """(m, order): the state variables split into m 'stocks' -- next values that may depend on the whole
        state and on the control but not on the perturbation -- and an exogenous process -- next values that
        depend on the exogenous variables, the perturbation and the time only -- WHEREVER they are listed
        (the reference takes the order of the state variables from dyn's signature, stodynprog.py:119-131);
        `order` lists the stocks first, then the exogenous variables, each group in the listed order.
        controlled_axes() is the case order == (0, 1, ..).  None when no such split exists."""
        d = self.n_state
        reach = [self.var_reach(n) for n in self.x_next]
        # the largest closed set of variables that evolve on their own: start from those the control does not
        # reach, drop what depends on a variable outside the set, repeat
        exo = set(k for k in range(d) if not any(v.startswith('u') for v in reach[k]))
        changed = True
        while changed:
            changed = False
            for k in sorted(exo):
                if any(v.startswith('x') and int(v[1:]) not in exo for v in reach[k]):
                    exo.discard(k)
                    changed = True
        stocks = [k for k in range(d) if k not in exo]
        if not stocks or any(any(v.startswith('w') for v in reach[k]) for k in stocks):
            return None
        return len(stocks), tuple(stocks + sorted(exo))

    LEAD_SPLIT_MAX_TERMS = 4

    def lead_split(self):
        """x0' = a(x, u[, t]) +- b_1(x_1.., w[, t]) +- b_2 ..  with the sums as the LAST operations of
        the recorded expression, a innermost: `x + u - w` of the inventory example (reference
        doc/example_inventory.py:31-33), `x + u - 0.5 * w - 0.1 * y`.  Returns (a, [(b_1, sign_1), ..])
        in the order the reference adds them, sign = +1 / -1, or None.  The perturbation then moves
        the axis-0 position of every control of a column by the same amounts, so the expectation over
        w can still be taken on the table before the controls are looked at
        (csrc/sdp_colfilter_kernel.h, SDP_COL_SHIFT).  `a` is one sub-expression, evaluated as the
        reference evaluates it: the first pass then starts from the reference's own value of it.

        Round 5: a chain of sums in ANY nesting, `x + (w - u)`, `(x - 0.1 * y) - (u - w)`, is taken too
        (`lead_split_chain`): its leaves are regrouped into a sum of the w-free ones, a*, and a sum of the others --
        a* is then NOT a value the reference computes; what the regrouping costs is part of the error radius."""
        cached = getattr(self, '_lead_split_cache', None)
        if cached is None:
            res, chain = self._lead_split_final(), None
            if res is None:
                got = self._lead_split_any_nesting()
                if got is not None:
                    res, chain = got
            cached = self._lead_split_cache = (res, chain)
        return cached[0]

    def lead_split_chain(self):
        """None when `lead_split` found the final-sum form (or nothing); else the regrouped chain's
        (w-free leaves [(node, sign), ..], number of additions / subtractions of the whole chain)."""
        self.lead_split()
        return self._lead_split_cache[1]

    def _lead_split_final(self):
        top = self.x_next[0]
        if not (top.deps & DEP_W):
            return None

        def is_b(n):
            return (n.deps & (DEP_X | DEP_U)) == 0

        terms = []
        node = top
        while node.deps & DEP_W:
            if node.op not in ('add', 'sub') or len(terms) == self.LEAD_SPLIT_MAX_TERMS:
                return None
            l, r = node.args
            if is_b(r):
                terms.append((r, 1 if node.op == 'add' else -1))
                node = l
            elif node.op == 'add' and is_b(l):
                terms.append((l, 1))
                node = r
            else:
                return None
        terms.reverse()
        if not any(b.deps & DEP_W for b, _ in terms):
            return None
        return node, terms

    LEAD_CHAIN_MAX_LEAVES = 4       # (three additions: the radius' constant 14 of the positions covers 2 x 3 + 6.2 roundings)

    def _lead_split_any_nesting(self):
        """x0' = a tree of real + / - whose leaves are either free of w (they may see x, u) or free of x0 and u (they
        may see x_1.., w), at most LEAD_CHAIN_MAX_LEAVES of them.  The leaves keep the reference's operations; the
        tree is regrouped as a* = the signed sum of the w-free leaves (new nodes of the graph, left to right) and
        the signed sum of the others.  Returns ((a*, [(b, sign), ..]), (w-free leaves, additions)) or None."""
        top = self.x_next[0]
        if not (top.deps & DEP_W) or top.kind != 'r':
            return None
        leaves = []

        def walk(n, sign):
            if not (n.deps & DEP_W):
                leaves.append((n, sign, 'a'))
                return True
            if (n.deps & (DEP_X | DEP_U)) == 0:
                leaves.append((n, sign, 'b'))
                return True
            if n.op not in ('add', 'sub') or n.kind != 'r':
                return False
            l, r = n.args
            return walk(l, sign) and walk(r, sign if n.op == 'add' else -sign)
        if not walk(top, 1) or len(leaves) > self.LEAD_CHAIN_MAX_LEAVES:
            return None
        a_leaves = [(n, sg) for n, sg, what in leaves if what == 'a']
        b_leaves = [(n, sg) for n, sg, what in leaves if what == 'b']
        if not a_leaves or not b_leaves or not any(n.deps & (DEP_X | DEP_U) for n, _ in a_leaves):
            return None
        if len(b_leaves) > self.LEAD_SPLIT_MAX_TERMS:
            return None
        a_leaves.sort(key=lambda item: item[1] < 0)          # (a positive leaf first, if there is one: no negation needed)
        g = self.graph
        first, sg = a_leaves[0]
        a_star = first if sg > 0 else g.op('neg', first)
        for n, sg in a_leaves[1:]:
            a_star = g.op('add' if sg > 0 else 'sub', a_star, n)
        return (a_star, b_leaves), (a_leaves, len(leaves) - 1)


def trace_model(dyn, cost, n_state, n_control, n_perturb, params=None, stationnary=True,
                t_value=None):
    """Trace `dyn` and `cost` with symbolic x, u, w (and t_k first when the
    system is time dependent), in the argument order of sdp.py:668-672.

    t_value: for a time-dependent system, trace ONE time step with the concrete
    time index `t_value` instead of a symbol.  Callables that look data up by
    time (`p['P_req_data'][k]`, reference examples/01 .../det_storage_control.py:89)
    cannot take a symbolic k; traced step by step their data become constants
    of the DAG, which `TracedModel.lift_constants` turns into kernel parameters
    so that all time steps of one structure share one code object."""
    params = params or {}
    g = Graph(unique_consts=t_value is not None)
    xs = [Sym(g, g.var('x%d' % i, DEP_X if i == 0 else DEP_XR)) for i in range(n_state)]
    us = [Sym(g, g.var('u%d' % i, DEP_U)) for i in range(n_control)]
    ws = [Sym(g, g.var('w%d' % i, DEP_W)) for i in range(n_perturb)]
    args = xs + us + ws
    if not stationnary:
        args = [Sym(g, g.var('t', DEP_T)) if t_value is None else t_value] + args
    some = xs[0] if xs else Sym(g, g.const(0.0))

    def as_real_node(v, what):
        if isinstance(v, Sym):
            return some._real(v.n)
        try:
            return some._real(some._lift(v))
        except TraceError:
            raise TraceError('{} returned a value of type {} that does not depend '
                             'symbolically on its arguments'.format(what, type(v).__name__))

    try:
        out = dyn(*args, **params)
    except TraceError:
        raise
    except Exception as e:      # e.g. numpy failing on an object array
        raise TraceError('dynamics function not traceable: {}: {}'.format(type(e).__name__, e))
    if isinstance(out, Sym) or _is_plain_number(out):
        out = (out,)
    try:
        out = tuple(out)
    except TypeError:
        raise TraceError('dynamics function should return a tuple of next-state values')
    if len(out) != n_state:
        raise TraceError('dynamics function returned {} values for {} state variables'
                         .format(len(out), n_state))
    x_next = [as_real_node(v, 'dynamics function') for v in out]
    try:
        c = cost(*args, **params)
    except TraceError:
        raise
    except Exception as e:
        raise TraceError('cost function not traceable: {}: {}'.format(type(e).__name__, e))
    cnode = as_real_node(c, 'cost function')
    model = TracedModel(g, x_next, cnode, n_state, n_control, n_perturb,
                        (not stationnary) and t_value is None)
    model.t_value = t_value
    return model


# ---------------------------------------------------------------------------
# The admissible box of the controls (reference stodynprog.py:432-463: `control_box(*state_k, **params)` is called at
# every node of every sweep).  DPSolver keeps a table of it, so the table has to be what those calls return AT EVERY
# NODE: the callback is traced like dyn and cost, and a trace IS that proof -- the DAG, evaluated with numpy on the
# whole grid, performs per node exactly the IEEE operations the scalar call performs (EXACT_OPS only; anything else
# sends the solver back to calling the callback node by node, like the reference).
#   * `np.max((a, b))` / `np.min(..)` / `np.amax` / `np.amin` of a tuple or list of operands -- the idiom of every box of the
#     reference's examples (AR1 notebook cell 15, searev/storage_control.py:76-78) -- is scalar-only as written.  The
#     callback is run on a copy of itself whose GLOBALS name a stand-in for the numpy module (every global bound to
#     numpy itself: `np`, `numpy`, ..): the stand-in forwards everything to numpy except these four, which fold the
#     operands with the `max` / `min` of the DAG (np.maximum / np.minimum: what the reduction over a short float array
#     computes, NaN included).  Nothing outside that one function object sees the stand-in -- no rebinding of numpy's
#     own attributes (round 5 patched np.max / np.min process-wide while the callback ran).
#   * Python control flow on the state (`if E > 5:`, the builtins `max(a, b)` / `min(a, b)`) asks a symbolic
#     condition for its truth value.  The callback is then traced once per PATH: every such question is answered
#     from a script of decisions (first run: always True), and after the run the last undecided True is flipped
#     and the callback runs again, until every path has been taken (at most _FORK_MAX_PATHS); the paths' results
#     are merged with `select` nodes on the recorded conditions.  Per node the merged DAG picks what the scalar
#     call would have returned -- the callback is assumed to be a pure function of its arguments, as the whole
#     tracer assumes.
# ---------------------------------------------------------------------------
_FORK_MAX_PATHS = 64


class _Fork(object):
    """decisions of one path: `script` is the prefix to replay, `taken` what this run met: (condition node, decision)"""

    def __init__(self, script):
        self.script = list(script)
        self.taken = []

    def decide(self, sym):
        n = sym.n
        if n.kind != 'b':
            n = sym.g.op('ne', n, sym.g.const(0.0))          # truthiness of a real
        if n.op == 'bconst':
            return bool(n.value)
        for cond, d in self.taken:                            # the same question again on this path: the same answer
            if cond is n:
                return d
        k = len(self.taken)
        d = self.script[k] if k < len(self.script) else True
        self.taken.append((n, d))
        return d


class _NumpyStandIn(object):
    """what the traced copy of a box callback sees under the names that are bound to the numpy module in its globals"""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def _fold(op, a, args, kw):
        real = getattr(np, op)
        if args or kw or not isinstance(a, (tuple, list)) or not any(isinstance(v, Sym) for v in a):
            return real(a, *args, **kw)
        sym = _any_sym(*a)
        out = a[0]
        for v in a[1:]:
            if not (isinstance(v, Sym) or _is_plain_number(v)):
                raise TraceError('np.{} of nested operands is not traceable'.format(op))
            out = sym._bin('max' if op in ('max', 'amax') else 'min', out, v)      # maximum.reduce / minimum.reduce, left to right
        return out if isinstance(out, Sym) else Sym(sym.g, sym._lift(out))

    def max(self, a, *args, **kw): return self._fold('max', a, args, kw)
    def amax(self, a, *args, **kw): return self._fold('amax', a, args, kw)
    def min(self, a, *args, **kw): return self._fold('min', a, args, kw)
    def amin(self, a, *args, **kw): return self._fold('amin', a, args, kw)


_NP_STAND_IN = _NumpyStandIn()


def _with_numpy_stand_in(fn):
    """a copy of the Python function `fn` (plain function, bound method or functools.partial of one) whose globals
    name _NP_STAND_IN wherever fn's name the numpy module; anything else is returned as it is"""
    import functools
    import types
    if isinstance(fn, functools.partial):
        inner = _with_numpy_stand_in(fn.func)
        return fn if inner is fn.func else functools.partial(inner, *fn.args, **(fn.keywords or {}))
    if isinstance(fn, types.MethodType):
        inner = _with_numpy_stand_in(fn.__func__)
        return fn if inner is fn.__func__ else types.MethodType(inner, fn.__self__)
    if not isinstance(fn, types.FunctionType):
        return fn
    glob = fn.__globals__
    names = [k for k in fn.__code__.co_names if glob.get(k) is np]
    if not names:
        return fn
    g2 = dict(glob)
    for k in names:
        g2[k] = _NP_STAND_IN
    copy = types.FunctionType(fn.__code__, g2, fn.__name__, fn.__defaults__, fn.__closure__)
    copy.__kwdefaults__ = fn.__kwdefaults__
    return copy


class TracedBox(object):
    """Result of tracing a `control_box` callback: per control the nodes of its lower and upper end."""

    def __init__(self, graph, ends, n_state, time_dep, t_value, paths):
        self.graph, self.ends, self.n_state = graph, ends, n_state
        self.time_dep, self.t_value, self.paths = time_dep, t_value, paths

    def live_nodes(self):
        seen, order = set(), []
        stack = [n for pair in self.ends for n in pair]
        # (ids increase from operands to results: the live nodes in id order are a valid evaluation order)
        while stack:
            n = stack.pop()
            if n.id in seen:
                continue
            seen.add(n.id)
            stack.extend(n.args)
        for n in self.graph.nodes:
            if n.id in seen:
                order.append(n)
        return order

    def inexact_ops(self):
        """operations of the live DAG whose whole-grid numpy evaluation is not guaranteed to repeat the scalar call bit
        for bit (transcendentals; a numpy SCALAR power, see _power): a box with any of them is not taken from its trace"""
        bad = sorted({n.op for n in self.live_nodes() if n.op not in EXACT_OPS})
        if any(n.id in self.graph.scalar_pow for n in self.live_nodes()):
            bad.append('scalar_pow')
        return bad

    def signature(self):
        """structure AND constants of the live DAG (tables included): two traces with the same signature give the same
        box at every node -- what DPSolver compares from call to call instead of probing nodes"""
        sig = []
        for n in self.live_nodes():
            if n.op == 'const':
                sig.append(('c', np.float64(n.value).tobytes()))
            elif n.op in ('var', 'bconst'):
                sig.append((n.op, n.value))
            elif n.op == 'interp1':
                sig.append((n.op, n.args[0].id, self.graph.tables[n.value][4]))
            else:
                sig.append((n.op,) + tuple(a.id for a in n.args))
        sig.append(tuple((lo.id, hi.id) for lo, hi in self.ends))
        return tuple(sig)

    def evaluate(self, x, t=None):
        """the box on numpy arrays (any mutually broadcastable shapes, e.g. an open grid): [(lo, hi)] per control"""
        vals = {'x%d' % i: v for i, v in enumerate(x)}
        vals['t'] = t
        env = {}
        with np.errstate(all='ignore'):
            for n in self.live_nodes():
                if n.op == 'var':
                    env[n.id] = np.asarray(vals[n.value], dtype=float)
                elif n.op == 'const':
                    env[n.id] = np.float64(n.value)
                elif n.op == 'bconst':
                    env[n.id] = np.bool_(n.value)
                elif n.op == 'interp1':
                    xp, fp, left, right = self.graph.tables[n.value][:4]
                    env[n.id] = np.interp(env[n.args[0].id], xp, fp, left, right)
                else:
                    env[n.id] = _NP_EVAL[n.op](*[env[a.id] for a in n.args])
        return [(env[lo.id], env[hi.id]) for lo, hi in self.ends]


def trace_box(control_box, n_state, n_control, params=None, stationnary=True, t_value=None):
    """Trace `control_box(*state, **params)` (reference stodynprog.py:438-440; the time index first for a time-dependent
    system: symbolic, or the concrete `t_value`).  Returns a TracedBox or raises TraceError."""
    params = params or {}
    g = Graph()
    xs = [Sym(g, g.var('x%d' % i, DEP_X if i == 0 else DEP_XR)) for i in range(n_state)]
    args = list(xs)
    if not stationnary:
        args = [Sym(g, g.var('t', DEP_T)) if t_value is None else t_value] + args
    some = xs[0] if xs else Sym(g, g.const(0.0))
    fn = _with_numpy_stand_in(control_box)

    def run(script):
        g.fork = _Fork(script)
        try:
            out = fn(*args, **params)
            out = tuple(out)
            if len(out) != n_control:
                raise TraceError('control_box returned {} intervals for {} controls'.format(len(out), n_control))
            ends = []
            for iv in out:
                lo, hi = iv
                pair = []
                for v in (lo, hi):
                    n = v.n if isinstance(v, Sym) else some._lift(v)
                    pair.append(some._real(n))
                ends.append(tuple(pair))
            return ends, g.fork.taken
        except TraceError:
            raise
        except Exception as e:
            raise TraceError('control_box not traceable: {}: {}'.format(type(e).__name__, e))
        finally:
            g.fork = None

    # depth-first over the decision tree: a leaf is a path with its result; merged bottom-up with selects
    leaves = []                                   # (decisions, conds, ends)
    script = []
    while True:
        ends, taken = run(script)
        leaves.append(([d for _, d in taken], [c for c, _ in taken], ends))
        if len(leaves) > _FORK_MAX_PATHS:
            raise TraceError('control_box branches on the state along more than {} paths'.format(_FORK_MAX_PATHS))
        dec = [d for _, d in taken]
        while dec and dec[-1] is False:
            dec.pop()
        if not dec:
            break
        dec[-1] = False
        script = dec

    def merge(group, depth):
        # every leaf of `group` shares its first `depth` decisions
        if len(group) == 1 and len(group[0][0]) == depth:
            return group[0][2]
        cond = group[0][1][depth]
        yes = [l for l in group if l[0][depth]]
        no = [l for l in group if not l[0][depth]]
        if not yes or not no or any(l[1][depth] is not cond for l in group):
            raise TraceError('control_box is not a pure function of its arguments (its branches changed between two runs)')
        a, b = merge(yes, depth + 1), merge(no, depth + 1)
        return [tuple(g.op('select', cond, u, v) if u is not v else u for u, v in zip(pa, pb)) for pa, pb in zip(a, b)]

    ends = merge(leaves, 0)
    return TracedBox(g, ends, n_state, (not stationnary) and t_value is None, t_value, len(leaves))


# ---------------------------------------------------------------------------
# Is a callable still what it was when it was traced?  The reference evaluates dyn, cost and control_box at call time
# (stodynprog.py:440, 674-676), so data they read may change between two calls; DPSolver therefore traced them afresh
# on every call (~0.1 ms each: as much as the kernels of the reference's own problem sizes).  A FINGERPRINT of a
# callable is everything a pure Python function's result can depend on besides its arguments, by VALUE: its code object,
# its defaults, the contents of its closure cells and of the globals its code names -- numbers, strings, small arrays
# (bytes), tuples / lists / dicts of those, other Python functions (recursively), library modules and builtins by
# identity.  Anything else -- an object with attributes, a module that is not a library, an array beyond 64 KiB,
# deep nesting -- has NO fingerprint (None) and the callable is traced on every call as before: the cache can only
# ever return the trace of a callable that would trace to the same graph.
# ---------------------------------------------------------------------------
_FP_LIBRARIES = ('numpy', 'math', 'cmath', 'scipy', 'operator', 'functools', 'itertools', 'builtins')
_FP_MAX_ARRAY_BYTES = 1 << 16
_FP_MAX_ITEMS = 512


class _NoFingerprint(Exception):
    pass


def _fp_value(v, depth, budget):
    import types
    budget[0] -= 1
    if budget[0] < 0 or depth > 4:
        raise _NoFingerprint
    if v is None or isinstance(v, (bool, int, float, complex, str, bytes)):
        return (type(v).__name__, v if v == v else 'nan')
    if isinstance(v, np.generic):
        return ('g', v.dtype.str, v.tobytes())
    if isinstance(v, np.ndarray):
        if v.nbytes > _FP_MAX_ARRAY_BYTES or v.dtype.kind == 'O':
            raise _NoFingerprint
        return ('a', v.dtype.str, v.shape, v.tobytes())
    if isinstance(v, (tuple, list)):
        return (type(v).__name__,) + tuple(_fp_value(x, depth + 1, budget) for x in v)
    if isinstance(v, dict):
        try:
            keys = sorted(v)
        except TypeError:
            raise _NoFingerprint
        return ('d',) + tuple((_fp_value(k, depth + 1, budget), _fp_value(v[k], depth + 1, budget)) for k in keys)
    if isinstance(v, types.ModuleType):
        if v.__name__.split('.')[0] in _FP_LIBRARIES:
            return ('m', v.__name__)
        raise _NoFingerprint
    if isinstance(v, (types.BuiltinFunctionType, np.ufunc)) or (isinstance(v, type) and v.__module__ in ('builtins', 'numpy')):
        return ('b', getattr(v, '__module__', None), getattr(v, '__qualname__', getattr(v, '__name__', None)), id(v))
    if isinstance(v, (types.FunctionType, types.MethodType)) or type(v).__name__ == 'partial':
        return _fp_callable(v, depth + 1, budget)
    raise _NoFingerprint


def _fp_callable(fn, depth, budget):
    import functools
    import types
    if isinstance(fn, functools.partial):
        return ('partial', _fp_callable(fn.func, depth, budget), _fp_value(tuple(fn.args), depth, budget),
                _fp_value(dict(fn.keywords or {}), depth, budget))
    if isinstance(fn, types.MethodType):
        raise _NoFingerprint                               # (the instance's attributes are out of sight)
    if not isinstance(fn, types.FunctionType):
        raise _NoFingerprint
    code = fn.__code__
    if getattr(fn.__module__, 'split', None) and (fn.__module__ or '').split('.')[0] in _FP_LIBRARIES:
        return ('lib', fn.__module__, fn.__qualname__, id(code))
    parts = ['f', id(code), code.co_code, _fp_value(code.co_consts if not any(isinstance(c, types.CodeType) for c in code.co_consts)
                                                   else tuple(c for c in code.co_consts if not isinstance(c, types.CodeType)), depth, budget)]
    for c in code.co_consts:
        if isinstance(c, types.CodeType):                    # a nested def / lambda / comprehension: its names count too
            parts.append(('inner', c.co_code, tuple(c.co_names)))
    parts.append(_fp_value(fn.__defaults__, depth, budget))
    parts.append(_fp_value(fn.__kwdefaults__, depth, budget))
    for cell in fn.__closure__ or ():
        try:
            parts.append(_fp_value(cell.cell_contents, depth, budget))
        except ValueError:                                  # an empty cell
            parts.append(('empty',))
    glob = fn.__globals__
    names = set(code.co_names)
    for c in code.co_consts:
        if isinstance(c, types.CodeType):
            names.update(c.co_names)
    for name in sorted(names):
        if name in glob:
            v = glob[name]
            if v is fn:
                continue
            parts.append((name, _fp_value(v, depth, budget)))
        # (a name that is neither a global nor a builtin is an attribute name: it belongs to a value seen elsewhere)
    return tuple(parts)


def callable_fingerprint(*callables_and_data):
    """hashable image of everything the results of these callables (pure Python functions) and data (params dicts,
    numbers) can depend on besides their arguments, or None when that cannot be established (see above)"""
    import types
    import functools
    budget = [_FP_MAX_ITEMS]
    try:
        out = []
        for c in callables_and_data:
            if isinstance(c, (types.FunctionType, types.MethodType, functools.partial)):
                out.append(_fp_callable(c, 0, budget))
            else:
                out.append(_fp_value(c, 0, budget))
        fp = tuple(out)
        hash(fp)
        return fp
    except (_NoFingerprint, TypeError, RecursionError):
        return None


def evaluate(model, x, u, w, t=None):
    """Reference interpreter of a traced model with numpy (host).  Used by the
    tests to check that tracing preserved the callable's semantics."""
    env = {}
    vals = {}
    for i, v in enumerate(x):
        vals['x%d' % i] = v
    for i, v in enumerate(u):
        vals['u%d' % i] = v
    for i, v in enumerate(w):
        vals['w%d' % i] = v
    vals['t'] = t
    f = _NP_EVAL
    with np.errstate(all='ignore'):
        for n in model.live_nodes():
            if n.op == 'var':
                env[n.id] = np.asarray(vals[n.value], dtype=float)
            elif n.op == 'const':
                env[n.id] = np.float64(n.value)
            elif n.op == 'bconst':
                env[n.id] = np.bool_(n.value)
            elif n.op == 'interp1':
                xp, fp, left, right = model.graph.tables[n.value][:4]
                env[n.id] = np.interp(env[n.args[0].id], xp, fp, left, right)
            else:
                env[n.id] = f[n.op](*[env[a.id] for a in n.args])
    return [env[n.id] for n in model.x_next], env[model.cost.id]


_NP_EVAL = {
    'add': np.add, 'sub': np.subtract, 'mul': np.multiply, 'div': np.divide,
    'neg': np.negative, 'abs': np.abs, 'sqrt': np.sqrt, 'square': np.square,
    'recip': lambda a: 1.0 / a, 'pow': np.power, 'min': np.minimum, 'max': np.maximum,
    'fmin': np.fmin, 'fmax': np.fmax, 'floor': np.floor, 'ceil': np.ceil,
    'trunc': np.trunc, 'rint': np.rint, 'sign': np.sign, 'exp': np.exp, 'exp2': np.exp2,
    'expm1': np.expm1, 'log': np.log, 'log2': np.log2, 'log10': np.log10,
    'log1p': np.log1p, 'sin': np.sin, 'cos': np.cos, 'tan': np.tan, 'asin': np.arcsin,
    'acos': np.arccos, 'atan': np.arctan, 'sinh': np.sinh, 'cosh': np.cosh,
    'tanh': np.tanh, 'atan2': np.arctan2, 'hypot': np.hypot, 'fmod': np.fmod,
    'pymod': np.mod, 'floordiv': np.floor_divide, 'cbrt': np.cbrt,
    'lt': np.less, 'le': np.less_equal, 'gt': np.greater, 'ge': np.greater_equal,
    'eq': np.equal, 'ne': np.not_equal, 'and': np.logical_and, 'or': np.logical_or,
    'xor': np.logical_xor, 'not': np.logical_not, 'isnan': np.isnan,
    'isfinite': np.isfinite, 'isinf': np.isinf,
    'select': np.where, 'bselect': np.where,
    'b2r': lambda a: np.asarray(a, dtype=float),
}
assert not math.isnan(0.0)
