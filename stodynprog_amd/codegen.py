"""HIP source generation for a traced model.

Turns the expression DAG recorded by trace.py into the device function
`sdp_model_cell` (one statement per recorded operator, in recording order) and
wraps it in a translation unit that includes the hand-written kernels of
csrc/sdp_sweep_kernel.h.  The unit is compiled to a gfx950 code object with
    hipcc --genco --offload-arch=gfx950 -O3 -ffp-contract=off
so that every operator is one IEEE operation, as in numpy on the host
(reference stodynprog.py:674-677 evaluates the same expressions with numpy).
"""
import hashlib
import re
import math
import os

import numpy as np

from .trace import TracedModel, DEP_X

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')


# ---------------------------------------------------------------------------
# Diagnostic / A-B switches.  The product has none: nothing in this module reads
# the environment.  Tests and the tools under tools/ hand an explicit dict
# (`DPSolver.debug_defines`, `bench.py --debug-define K=V`) to the planning
# functions below as `debug`; the names are those of the macros of csrc/*.h they
# set (plus a few planning switches: SDP_COL_FILTER / SDP_COL_SHIFT / SDP_COL_UTAB /
# SDP_LEAD_FILTER = 0 switch a form off, SDP_COL_THREADS / SDP_COL_WCHUNK /
# SDP_STG_CU / SDP_COL_WPAIR force a shape).  A solver that carries such a dict says
# so in backend_info['debug_defines'], and bench.py in config.debug_defines.
# ---------------------------------------------------------------------------
# macros copied into the generated unit as they are (integers)
DEBUG_INT_MACROS = ('SDP_COL_SHARE_X2', 'SDP_COL_MIN_WAVES', 'SDP_COL_BATCH', 'SDP_COL_WMODE', 'SDP_COL_UNROLL_U', 'SDP_COL_UNROLL_W',
                    'SDP_COL_A_GROUP', 'SDP_COL_A_ORDER', 'SDP_COL_FILTER_UNROLL', 'SDP_COL_B_PRIO',
                    'SDP_COL_FILTER_TOP2', 'SDP_COL_TILE', 'SDP_COL_FILTER_RUNROLL', 'SDP_COL_LEAN', 'SDP_COL_WIDE',
                    'SDP_COL_A_WIDE_LOADS', 'SDP_COLU_WIDE_LOADS', 'SDP_COLU_A_GROUP', 'SDP_COL_LDS_PAD', 'SDP_COL_HOIST',
                    'SDP_COL_TAIL_KEEP', 'SDP_COL_KEEP_BATCH', 'SDP_COL_KEEP_LOADS', 'SDP_COL_TAIL_HOLD', 'SDP_SHORT_GROUP', 'SDP_BNB_CHUNK')
# (SDP_COL_LEAN2 = 0 keeps the resident-chunk kernel on the first pass of section 3.1c: an A/B switch of short_pass_source)
# (SDP_COL_WRES is a planning switch: it sizes the LDS image -- column_config)
# every name a `debug` dict may carry (a typo must not pass silently)
DEBUG_NAMES = frozenset(DEBUG_INT_MACROS + (
    'SDP_STAMP', 'SDP_NO_POW2', 'SDP_EXTRA_DEFINES', 'SDP_COL_FILTER_SCALE', 'SDP_LEAD_FILTER_SCALE',
    'SDP_LEAD_UNROLL', 'SDP_COL_A_LW', 'SDP_COL_FILTER', 'SDP_COL_SHIFT', 'SDP_COL_UTAB', 'SDP_LEAD_FILTER',
    'SDP_COL_THREADS', 'SDP_COL_WCHUNK', 'SDP_STG_CU', 'SDP_COL_WPAIR', 'SDP_COL_WRES', 'SDP_COL_LEAN2', 'SDP_COL_BNB',
    'SDP_LINE_FILTER', 'SDP_LINE_FILTER_SCALE', 'SDP_LINE_TOP2'))


def check_debug(debug):
    """normalised copy of a debug dict (values as strings), or None; unknown names raise"""
    if not debug:
        return None
    unknown = sorted(set(debug) - DEBUG_NAMES)
    if unknown:
        raise ValueError('unknown debug define(s): {}'.format(', '.join(unknown)))
    return {k: str(v) for k, v in debug.items() if v is not None and str(v) != ''} or None


def _dbg(debug, name, default=None):
    if not debug:
        return default
    v = debug.get(name)
    return default if v is None or str(v) == '' else str(v)

_BIN_INFIX = {'add': '+', 'sub': '-', 'mul': '*', 'div': '/',
              'lt': '<', 'le': '<=', 'gt': '>', 'ge': '>=', 'eq': '==', 'ne': '!=',
              'and': '&&', 'or': '||', 'xor': '!='}
_FUNC1 = {'abs': 'fabs', 'sqrt': 'sqrt', 'floor': 'floor', 'ceil': 'ceil', 'trunc': 'trunc',
          'rint': 'rint', 'exp': 'exp', 'exp2': 'exp2', 'expm1': 'expm1', 'log': 'log',
          'log2': 'log2', 'log10': 'log10', 'log1p': 'log1p', 'sin': 'sin', 'cos': 'cos',
          'tan': 'tan', 'asin': 'asin', 'acos': 'acos', 'atan': 'atan', 'sinh': 'sinh',
          'cosh': 'cosh', 'tanh': 'tanh', 'cbrt': 'cbrt'}
_FUNC2 = {'pow': 'pow', 'atan2': 'atan2', 'hypot': 'hypot', 'fmod': 'fmod',
          'min': 'sdp_npmin', 'max': 'sdp_npmax', 'fmin': 'sdp_npfmin', 'fmax': 'sdp_npfmax',
          'pymod': 'sdp_nppymod', 'floordiv': 'sdp_npfloordiv'}


def real_literal(value):
    """Exact C++ literal for a double constant (cast to the kernel's real)."""
    if math.isnan(value):
        return '(sdp_real)NAN'
    if math.isinf(value):
        return '(sdp_real)INFINITY' if value > 0 else '(-(sdp_real)INFINITY)'
    return '(sdp_real)({})'.format(float(value).hex())


def _emit_node(n, name, param_index=None):
    a = [name(x) for x in n.args]
    op = n.op
    if op == 'const':
        if param_index is not None and n.id in param_index:
            return 'sdp_model_prm[{}]'.format(param_index[n.id])     # lifted constant
        return real_literal(n.value)
    if op == 'bconst':
        return 'true' if n.value else 'false'
    if op == 'var':
        v = n.value
        if v == 't':
            return 't'
        return '{}[{}]'.format(v[0], v[1:]) if v[0] in 'xu' else 'w'
    if op in _BIN_INFIX:
        return '({} {} {})'.format(a[0], _BIN_INFIX[op], a[1])
    if op in _FUNC1:
        return '{}({})'.format(_FUNC1[op], a[0])
    if op in _FUNC2:
        return '{}({}, {})'.format(_FUNC2[op], a[0], a[1])
    if op == 'neg':
        return '(-{})'.format(a[0])
    if op == 'square':
        return '({0} * {0})'.format(a[0])
    if op == 'recip':
        return '((sdp_real)1 / {})'.format(a[0])
    if op == 'sign':
        return 'sdp_npsign({})'.format(a[0])
    if op == 'not':
        return '(!{})'.format(a[0])
    if op == 'isnan':
        return '({0} != {0})'.format(a[0])
    if op == 'isinf':
        return '(fabs({}) == (sdp_real)INFINITY)'.format(a[0])
    if op == 'isfinite':
        return '(fabs({}) < (sdp_real)INFINITY)'.format(a[0])
    if op in ('select', 'bselect'):
        return '({} ? {} : {})'.format(a[0], a[1], a[2])
    if op == 'b2r':
        return '({} ? (sdp_real)1 : (sdp_real)0)'.format(a[0])
    if op == 'interp1':
        return 'sdp_interp1_{}({})'.format(n.value, a[0])
    raise NotImplementedError(op)


# np.interp on the device: numpy's arr_interp (compiled_base.c) restated --
# range checks first (binary_search_with_guess), the largest j with
# xp[j] <= x, then the same three floating-point operations and the same
# fall-backs for non-finite results.
INTERP_SOURCE = '''SDP_DEV sdp_real sdp_np_interp(sdp_real x, const sdp_real *xp, const sdp_real *fp, int n,
                                sdp_real left, sdp_real right)
{
    if (n == 1) return x < xp[0] ? left : (x > xp[0] ? right : fp[0]);
    if (x != x) return x;
    if (x > xp[n - 1]) return right;
    if (x < xp[0]) return left;
    int imin = 0, imax = n;
    while (imin < imax) {
        const int imid = imin + ((imax - imin) >> 1);
        if (x >= xp[imid]) imin = imid + 1; else imax = imid;
    }
    const int j = imin - 1;
    if (j == n - 1) return fp[j];
    if (xp[j] == x) return fp[j];
    const sdp_real slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    sdp_real r = slope * (x - xp[j]) + fp[j];
    if (r != r) {
        r = slope * (x - xp[j + 1]) + fp[j + 1];
        if (r != r && fp[j] == fp[j + 1]) r = fp[j];
    }
    return r;
}
'''


def interp_tables_source(model):
    """Tables and per-table wrappers of the np.interp nodes of `model`."""
    used = sorted({n.value for n in model.live_nodes() if n.op == 'interp1'})
    if not used:
        return ''
    out = [INTERP_SOURCE]
    for tid in used:
        xp, fp, left, right = model.graph.tables[tid][:4]
        out.append('__constant__ sdp_real sdp_tabx_{}[{}] = {{{}}};'.format(
            tid, xp.size, ', '.join(real_literal(v) for v in xp)))
        out.append('__constant__ sdp_real sdp_tabf_{}[{}] = {{{}}};'.format(
            tid, fp.size, ', '.join(real_literal(v) for v in fp)))
        out.append('SDP_DEV sdp_real sdp_interp1_{0}(sdp_real x)\n{{\n    return sdp_np_interp('
                   'x, sdp_tabx_{0}, sdp_tabf_{0}, {1}, {2}, {3});\n}}'.format(
                       tid, xp.size, real_literal(left), real_literal(right)))
    return '\n'.join(out) + '\n'


def _emit_body(model, nodes, lines, names=None):
    """statements for `nodes` (topological order); returns the name map
    (`names`: nodes that already have a name -- leaves of a partial evaluation)"""
    names = dict(names or {})

    def name(n):
        return names[n.id]

    for n in nodes:
        expr = _emit_node(n, name, getattr(model, 'param_index', None))
        if n.op in ('const', 'bconst', 'var'):
            names[n.id] = expr              # inline leaves
            continue
        var = 'v{}'.format(n.id)
        ctype = 'bool' if n.kind == 'b' else 'sdp_real'
        lines.append('    const {} {} = {};'.format(ctype, var, expr))
        names[n.id] = var
    return names


def model_function_source(model):
    """C++ text of sdp_model_cell for a TracedModel: the whole dyn + cost."""
    assert isinstance(model, TracedModel)
    if model.n_perturb > 1:
        raise NotImplementedError('only 0 or 1 perturbation variable is supported '
                                  '(as reference stodynprog.py:679-683)')
    lines = ['SDP_DEV void sdp_model_cell(const sdp_real *x, const sdp_real *u, sdp_real w,',
             '                            sdp_real t, sdp_real *xn, sdp_real &g)',
             '{',
             '    (void)x; (void)u; (void)w; (void)t;']
    names = _emit_body(model, model.live_nodes(), lines)
    for k, n in enumerate(model.x_next):
        lines.append('    xn[{}] = {};'.format(k, names[n.id]))
    lines.append('    g = {};'.format(names[model.cost.id]))
    lines.append('}')
    return '\n'.join(lines)


def separable_functions_source(model):
    """C++ text of the three slices used by the column kernel
    (csrc/sdp_column_kernel.h) for a storage-separable model."""
    assert model.column_shareable
    out = []
    lines = ['SDP_DEV sdp_real sdp_model_lead(const sdp_real *x, const sdp_real *u, sdp_real w,',
             '                                sdp_real t)',
             '{', '    (void)x; (void)u; (void)w; (void)t;']
    names = _emit_body(model, model.slice_nodes([model.x_next[0]]), lines)
    lines += ['    return {};'.format(names[model.x_next[0].id]), '}']
    out.append('\n'.join(lines))
    lines = ['SDP_DEV void sdp_model_trail(const sdp_real *x, const sdp_real *u, sdp_real w, sdp_real t,',
             '                                sdp_real *xn)',
             '{', '    (void)x; (void)u; (void)w; (void)t;']
    names = _emit_body(model, model.slice_nodes(model.x_next[1:]), lines)
    for k, n in enumerate(model.x_next[1:]):
        lines.append('    xn[{}] = {};'.format(k + 1, names[n.id]))
    lines.append('}')
    out.append('\n'.join(lines))
    lines = ['SDP_DEV sdp_real sdp_model_cost(const sdp_real *x, const sdp_real *u, sdp_real w,',
             '                                sdp_real t)',
             '{', '    (void)x; (void)u; (void)w; (void)t;']
    names = _emit_body(model, model.slice_nodes([model.cost]), lines)
    lines += ['    return {};'.format(names[model.cost.id]), '}']
    out.append('\n'.join(lines))
    split = model.lead_split()
    if split is not None:
        # x0' = a(x, u) +- b_1(x_1.., w) +- b_2 ..: a, the sum B of the signed b_i in the reference's order, and
        # the sum of the |b_i| (for the error bound), each b_i with the reference's own operations
        # (SDP_COL_SHIFT of csrc/sdp_colfilter_kernel.h)
        a_node, terms = split
        lines = ['SDP_DEV sdp_real sdp_model_lead_a(const sdp_real *x, const sdp_real *u, sdp_real t)',
                 '{', '    (void)x; (void)u; (void)t;']
        names = _emit_body(model, model.slice_nodes([a_node]), lines)
        lines += ['    return {};'.format(names[a_node.id]), '}']
        out.append('\n'.join(lines))
        chain = model.lead_split_chain()
        if chain is not None:
            # a chain of sums in another nesting, regrouped (TracedModel.lead_split): a is the sum of the chain's w-free
            # leaves, no value of the reference's; the sum of their magnitudes bounds what the regrouping costs
            # (SDP_COL_SHIFT_CHAIN of csrc/sdp_colfilter_kernel.h)
            leaves = [n for n, _ in chain[0]]
            lines = ['SDP_DEV sdp_real sdp_model_lead_aabs(const sdp_real *x, const sdp_real *u, sdp_real t)',
                     '{', '    (void)x; (void)u; (void)t;']
            names = _emit_body(model, model.slice_nodes(leaves), lines)
            lines += ['    return {};'.format(' + '.join('fabs({})'.format(names[n.id]) for n in leaves)), '}']
            out.append('\n'.join(lines))
        lines = ['SDP_DEV void sdp_model_lead_b(const sdp_real *x, sdp_real w, sdp_real t, sdp_real &b, sdp_real &babs)',
                 '{', '    (void)x; (void)w; (void)t;']
        names = _emit_body(model, model.slice_nodes([b for b, _ in terms]), lines)
        b0, s0 = terms[0]
        lines.append('    b = {}{};'.format('-' if s0 < 0 else '', names[b0.id]))
        lines.append('    babs = fabs({});'.format(names[b0.id]))
        for bn, sg in terms[1:]:
            lines.append('    b = b {} {};'.format('+' if sg > 0 else '-', names[bn.id]))
            lines.append('    babs = babs + fabs({});'.format(names[bn.id]))
        lines.append('}')
        out.append('\n'.join(lines))
    return '\n\n'.join(out)


def lead_functions_source(model, m, order=None):
    """C++ text of the slices csrc/sdp_lead_kernel.h uses for a model with `m` controlled stocks next to an
    exogenous process (TracedModel.controlled_order): `order` lists the state variables, stocks first
    (default: as they are listed)."""
    d = model.n_state
    order = tuple(order) if order is not None else tuple(range(d))
    leads = [model.x_next[k] for k in order[:m]]
    trails = [model.x_next[k] for k in order[m:]]
    out = []
    lines = ['SDP_DEV void sdp_model_leads(const sdp_real *x, const sdp_real *u, sdp_real t, sdp_real *xl)',
             '{', '    (void)x; (void)u; (void)t;']
    names = _emit_body(model, model.slice_nodes(leads), lines)
    for k, n in enumerate(leads):
        lines.append('    xl[{}] = {};'.format(k, names[n.id]))
    lines.append('}')
    out.append('\n'.join(lines))
    if m < d:
        lines = ['SDP_DEV void sdp_model_trails(const sdp_real *x, sdp_real w, sdp_real t, sdp_real *xt)',
                 '{', '    (void)x; (void)w; (void)t;']
        names = _emit_body(model, model.slice_nodes(trails), lines)
        for k, n in enumerate(trails):
            lines.append('    xt[{}] = {};'.format(k, names[n.id]))
        lines.append('}')
        out.append('\n'.join(lines))
    lines = ['SDP_DEV sdp_real sdp_model_cost(const sdp_real *x, const sdp_real *u, sdp_real w,',
             '                                sdp_real t)',
             '{', '    (void)x; (void)u; (void)w; (void)t;']
    names = _emit_body(model, model.slice_nodes([model.cost]), lines)
    lines += ['    return {};'.format(names[model.cost.id]), '}']
    out.append('\n'.join(lines))
    return '\n\n'.join(out)


def lead_filter_applies(model, dtype, min_axes=2, debug=None):
    """Several controlled state variables next to an exogenous process (TracedModel.controlled_axes
    >= 2; one stock is the column kernel's case), a perturbation that does not reach them (the cost may see
    it), 8-byte reals (a form for 4-byte reals existed in round 5: its radius of ~1e-5 of the values kept so many
    controls that the sweep was 3 x SLOWER than every control the long way, 8.9 against 2.6 ms; removed in round 6):
    the node-order sweep with the certified filter on an array reduced over w (csrc/sdp_lead_kernel.h).  Returns the
    number of controlled axes, or 0.  (`debug`: SDP_LEAD_FILTER = 0 switches it off, A/B runs.)"""
    if _dbg(debug, 'SDP_LEAD_FILTER', '1') == '0':
        return 0
    if model.n_perturb != 1 or np.dtype(dtype).itemsize != 8:
        return 0
    m = model.controlled_axes()
    return int(m) if m is not None and m >= min_axes else 0


def lead_order(model, dtype, debug=None):
    """(m, order) for the reduced-array sweep of a model whose stocks are NOT listed first (the order of the
    state variables is the user's: reference stodynprog.py:119-131), or None: TracedModel.controlled_order
    with at least one exogenous variable, 8-byte reals, one perturbation."""
    if _dbg(debug, 'SDP_LEAD_FILTER', '1') == '0' or model.n_perturb != 1 or np.dtype(dtype).itemsize != 8:
        return None
    co = model.controlled_order()
    if co is None or co[1] == tuple(range(model.n_state)) or co[0] >= model.n_state:
        return None
    return co


UTAB_MAX_VALUES = 4          # tabulated sub-expressions per control
UTAB_MAX_BYTES = 4096        # per parity buffer of the table in LDS


def control_table_plan(model, dtype, per_node, max_controls, debug=None):
    """Can the column-uniform sub-expressions of x0' and of the cost be tabulated once per
    (column, control) (SDP_COL_UTAB of csrc/sdp_colfilter_kernel.h)?  Needs a control lattice that
    is the same at every node (constant box) and small enough for LDS.  Returns the frontier
    nodes (TracedModel.control_uniform_frontier) or None.  (`debug`: SDP_COL_UTAB = 0 switches it
    off, A/B runs.)"""
    if _dbg(debug, 'SDP_COL_UTAB', '1') == '0' or per_node or model.cost_depends_on_w:
        return None
    lead = None
    if model.lead_depends_on_w:
        split = model.lead_split()
        if split is None:
            return None
        lead = split[0]
    fr = model.control_uniform_frontier(lead)
    if fr is None or len(fr) > UTAB_MAX_VALUES:
        return None
    if len(fr) * int(max_controls) * np.dtype(dtype).itemsize > UTAB_MAX_BYTES:
        return None
    return fr


def control_table_source(model, frontier):
    """C++ text of sdp_model_utab (the tabulated values of one control) and of
    sdp_model_lead_tab / sdp_model_cost_tab (x0' and the cost from them)."""
    slot = {n.id: k for k, n in enumerate(frontier)}
    out = []
    lines = ['SDP_DEV void sdp_model_utab(const sdp_real *x, const sdp_real *u, sdp_real t, sdp_real *tab)',
             '{', '    (void)x; (void)u; (void)t;']
    names = _emit_body(model, model.slice_nodes(frontier), lines)
    for n in frontier:
        lines.append('    tab[{}] = {};'.format(slot[n.id], names[n.id]))
    lines.append('}')
    out.append('\n'.join(lines))
    pre = {n.id: 'tab[{}]'.format(slot[n.id]) for n in frontier}
    split = model.lead_split() if model.lead_depends_on_w else None
    for fname, node in (('sdp_model_lead_tab', split[0] if split else model.x_next[0]),     # (shifted table: the w-free half)
                        ('sdp_model_cost_tab', model.cost)):
        lines = ['SDP_DEV sdp_real {}(const sdp_real *x, const sdp_real *tab, sdp_real t)'.format(fname),
                 '{', '    (void)x; (void)tab; (void)t;']
        names = _emit_body(model, model.slice_nodes_until([node], set(slot)), lines, pre)
        lines += ['    return {};'.format(names[node.id]), '}']
        out.append('\n'.join(lines))
    chain = model.lead_split_chain() if split else None
    if chain is not None:              # (a regrouped chain of sums: the sum of the w-free leaves' magnitudes, from the table too)
        leaves = [n for n, _ in chain[0]]
        lines = ['SDP_DEV sdp_real sdp_model_lead_aabs_tab(const sdp_real *x, const sdp_real *tab, sdp_real t)',
                 '{', '    (void)x; (void)tab; (void)t;']
        names = _emit_body(model, model.slice_nodes_until(leaves, set(slot)), lines, pre)
        lines += ['    return {};'.format(' + '.join('fabs({})'.format(names[n.id]) for n in leaves)), '}']
        out.append('\n'.join(lines))
    return '\n\n'.join(out)


def short_pass_source(model, frontier, macro='SDP_COL_LEAN2'):
    """C++ text for the short first pass of the resident-chunk kernel (SDP_COL_LEAN2 of csrc/sdp_colres_kernel.h), or
    None when the model does not have the shape: x0' = X(x, t) +- a, cost = K(x, t) +- h with a, h in the control
    table (TracedModel.additive_control_split).  X and K as functions of the node alone, the slots of a and h, and
    how they enter."""
    sp = model.additive_control_split(frontier)
    if sp is None:
        return None
    (x_node, a_slot, a_form), (k_node, h_slot, h_form) = sp['lead'], sp['cost']
    out = ['#define {} 1          // x0\' = X(x) +- a(u), cost = K(x) +- h(u): the short first pass'.format(macro),
           '#define SDP_LEAN2_A_SLOT {}'.format(int(a_slot)),
           '#define SDP_LEAN2_H_SLOT {}'.format(-1 if h_slot is None else int(h_slot)),
           '#define SDP_LEAN2_LEAD(X, A) {}'.format({'add': '((X) + (A))', 'sub': '((X) - (A))', 'rsub': '((A) - (X))'}[a_form]),
           '#define SDP_LEAN2_FORM {}            // 0: X + a, 1: X - a, 2: a - X'.format({'add': 0, 'sub': 1, 'rsub': 2}[a_form]),
           '#define SDP_LEAN2_A_FIXED {}         // a depends on the control alone: the same lattice of positions in every column'.format(
               0 if (frontier[int(a_slot)].deps & DEP_X) else 1),
           '#define SDP_LEAN2_HNEG {}           // the part of the cost that depends on the control enters negated (K - h)'.format(
               1 if (h_slot is not None and h_form == 'sub') else 0)]
    for fname, node in (('sdp_model_lead_x', x_node), ('sdp_model_cost_x', k_node)):
        lines = ['SDP_DEV sdp_real {}(const sdp_real *x, sdp_real t)'.format(fname), '{', '    (void)x; (void)t;']
        if node is None:
            lines += ['    return (sdp_real)0;', '}']
        else:
            names = _emit_body(model, model.slice_nodes([node]), lines)
            lines += ['    return {};'.format(names[node.id]), '}']
        out.append('\n'.join(lines))
    return '\n'.join(out)


def lanes_for(max_controls):
    """Lanes per state node: the power of two covering the largest control
    lattice, capped at the 64 lanes of a wavefront."""
    lanes = 1
    while lanes < max_controls and lanes < 64:
        lanes *= 2
    return lanes


def _prologue_lines(model, real, lanes, debug):
    """what every generated unit starts with: the type and shape macros, the diagnostic
    switches of an explicit `debug` dict (none in the product), the device helpers, the
    lifted constants, the interpolation tables and sdp_model_cell"""
    lines = ['// generated by stodynprog_amd.codegen -- do not edit',
             '#define SDP_REAL {}'.format(real),
             '#define SDP_D {}'.format(model.n_state),
             '#define SDP_NU {}'.format(model.n_control),
             '#define SDP_HAS_W {}'.format(1 if model.n_perturb else 0),
             '#define SDP_LANES {}'.format(int(lanes))]
    if _dbg(debug, 'SDP_STAMP') in ('1', '2', '3'):
        lines.append('#define SDP_STAMP {}     // diagnostic build: in-kernel clock stamps (tools/clock_probe.py, '
                     'tools/phase_probe.py)'.format(int(_dbg(debug, 'SDP_STAMP'))))
    if _dbg(debug, 'SDP_NO_POW2') == '1':
        lines.append('#define SDP_NO_POW2 1  // A/B: true division also for power-of-two spans')
    for kv in (_dbg(debug, 'SDP_EXTRA_DEFINES') or '').split(','):
        if '=' in kv:
            lines.append('#define {} {}   // SDP_EXTRA_DEFINES (diagnostic builds)'.format(*kv.split('=', 1)))
    lines += ['#include "sdp_device.h"', 'typedef SDP_REAL sdp_real;', '']
    if getattr(model, 'param_index', None):
        # lifted constants (TracedModel.lift_constants): set per launch through
        # sdp_problem_set_params; uniform loads from constant memory
        lines += ['#define SDP_NPARAMS {}'.format(len(model.param_index)),
                  'extern "C" { __constant__ sdp_real sdp_model_prm[SDP_NPARAMS]; }     // a definition', '']
    tables = interp_tables_source(model)
    if tables:
        lines.append(tables)
    lines += [model_function_source(model), '']
    return lines


def _column_lines(model, dtype, column, col_cfg, window, per_control, filtered, utab, debug, wres=0):
    """the macros and model slices of a unit that includes csrc/sdp_column_kernel.h"""
    rs = np.dtype(dtype).itemsize
    wpair = use_wpair(model, dtype, debug) and window is None
    shifted = bool(filtered and column_shift_applies(model, dtype, debug=debug))
    lines = ['#define SDP_COST_HAS_W {}'.format(1 if model.cost_depends_on_w else 0),
             '#define SDP_LEAD_HAS_W {}'.format(1 if model.lead_depends_on_w else 0),
             '#define SDP_TRAIL_HAS_U {}'.format(1 if model.trail_depends_on_u else 0),
             '#define SDP_COL_N0 {}'.format(int(column[0])),
             '#define SDP_COL_W {}'.format(max(int(column[1]), 1)),
             '#define SDP_COL_THREADS {}'.format(col_cfg[0]),
             '#define SDP_COL_WPAIR {}'.format(1 if wpair else 0)]
    if wres:
        lines.append('#define SDP_COL_WRES {}         // resident-chunk form: perturbation points the table holds at a time'.format(int(wres)))
        if rs == 8 and shifted and not _dbg(debug, 'SDP_COL_UNROLL_W'):
            # the second pass of the shifted lattice's resident-chunk kernel one batch of perturbation points at a time: unrolled
            # four times it kept 8 points' cells, table entries and weights per survivor in flight (two survivors in half of the
            # waves), in registers the kernel does not have at three waves per SIMD -- round 6, same box: 2.45 -> 2.23 ms, and
            # with the tail held on top 2.12 ms (boxes 9-11 of profiles/r06_column_ab.txt: 48 -> 22 spilled registers).  The
            # benchmark's kernel (no cell per point, one survivor) does not care: 948.8 against 954.3 sweeps/s over 50 steps
            # (box 20; the 5 % of box 12 were the A/B tool's first-position penalty); 4-byte reals: 7.51 against 7.72 ms.
            lines.append('#define SDP_COL_UNROLL_W 1')
    if filtered and int(col_cfg[0]) <= 256 and not _dbg(debug, 'SDP_COL_MIN_WAVES'):
        if wres:
            # (at most 4: 128 registers per lane.  Round 4 asked for up to 8 for small tables -- 64 registers, which these
            # kernels do not fit in: 90 - 200 spilled registers in the suite's small models, and spill code the compiler
            # got wrong, see spill_hazards)
            lines.append('#define SDP_COL_MIN_WAVES {}    // as many waves per SIMD as the LDS image admits workgroups per CU, at most 4'.format(
                max(1, min(4, (COLUMN_LDS_MAX // int(col_cfg[1])) * int(col_cfg[0]) // 256))))
        else:
            lines.append('#define SDP_COL_MIN_WAVES 1    // small workgroups of the filtered kernel: no register cap')
    if filtered:
        lines.append('#define SDP_COL_FILTER 1')
        if shifted:
            lines += ['#define SDP_COL_SHIFT 1        // x0\' = a(x, u) +- b_1(x_1.., w) +- ..: first pass on the shifted lattice',
                      '#define SDP_COL_SHIFT_TERMS {}'.format(len(model.lead_split()[1])),
                      '#define SDP_COL_SHIFT_CHAIN {}        // additions of a chain of sums that was regrouped (0: the final-sum form)'.format(
                          model.lead_split_chain()[1] if model.lead_split_chain() is not None else 0),
                      '#define SDP_COL_SHIFT_ROWS {}'.format(int(col_cfg[2]))]
        if _dbg(debug, 'SDP_COL_FILTER_SCALE'):
            lines.append('#define SDP_COL_FILTER_SCALE {}'.format(float(_dbg(debug, 'SDP_COL_FILTER_SCALE'))))
    # how the table build deals its entries to the threads
    if per_control is None and not _dbg(debug, 'SDP_COL_A_ORDER'):
        wide = column_wide_loads(column[0], dtype, window, debug)
        order = column_build_order(int(col_cfg[0]), column[1], int(window[2]) if window is not None else column[0],
                                   (16 // rs) if wide else 1)
        if order[0] == 2:
            lw = int(_dbg(debug, 'SDP_COL_A_LW') or order[1])
            lines += ['#define SDP_COL_A_ORDER 2',
                      '#define SDP_COL_A_LW {}'.format(lw)]
            if rs == 4 and int(column[0]) >= 512 and not wres and window is None and not _dbg(debug, 'SDP_COL_A_GROUP'):
                # (config 5, 512^3 in 4-byte reals, same box: 7.50 -> 7.36 ms, with four blocks' bounds per stage of
                # the branch and bound 7.24 ms -- boxes 25-26 of profiles/r06_column_ab.txt; 8-byte reals: worse)
                lines.append('#define SDP_COL_A_GROUP 8      // table entries per thread whose vertex loads are issued together')
            if wide and not _dbg(debug, 'SDP_COL_A_WIDE_LOADS'):
                lines.append('#define SDP_COL_A_WIDE_LOADS 1')

                # resident chunks: the tail is built once, its entries wait in the registers of the threads that made
                # them (SDP_COL_TAIL_HOLD of csrc/sdp_colres_kernel.h; round 6: 1.235 -> 1.048 ms on the benchmark, same
                # bits) -- where they are whole rounds of points and rows and at most 32 registers per thread
                # (on the shifted lattice too since the second pass is no longer unrolled there: 2.23 -> 2.12 ms, box 11 of
                # profiles/r06_column_ab.txt; with the unrolled pass it had lost, 2.96 against 2.71 ms)
                if (wres and rs == 8 and not wpair and window is None and _dbg(debug, 'SDP_COL_TAIL_HOLD') is None
                        and not _dbg(debug, 'SDP_COL_TAIL_KEEP')):
                    tail, groups = int(column[1]) - int(wres), int(col_cfg[0]) // lw
                    if (tail > 0 and tail % groups == 0 and int(column[0]) % (2 * lw) == 0
                            and (tail // groups) * (int(column[0]) // (2 * lw)) * 4 <= 32):
                        lines.append('#define SDP_COL_TAIL_HOLD 1    // the tail of the table is built once and held in registers')
    if window is not None:
        lines.append('#define SDP_COL_ROWS {}'.format(int(window[2])))
    if per_control is not None:
        lines.append('#define SDP_COL_WCHUNK {}'.format(int(per_control[2])))
        if int(column[0]) % (16 // rs) == 0 and not _dbg(debug, 'SDP_COLU_WIDE_LOADS'):
            lines.append('#define SDP_COLU_WIDE_LOADS 1')
    for k in DEBUG_INT_MACROS:                                   # tuning knobs of A/B runs (explicit dict only)
        if _dbg(debug, k):
            lines.append('#define {} {}'.format(k, int(_dbg(debug, k))))
    lines += [separable_functions_source(model), '']
    if utab is not None and filtered:
        lines += ['#define SDP_COL_UTAB {}'.format(len(utab[0])),
                  '#define SDP_COL_UTAB_N {}'.format(int(utab[1])),
                  control_table_source(model, utab[0]), '']
        # the short first passes: 8-byte reals in the resident-chunk kernel, 4-byte reals (wide form) in the full-table one
        short = None
        if not model.cost_depends_on_w and window is None and per_control is None \
                and _dbg(debug, 'SDP_COL_LEAN2') != '0' and (not shifted or rs == 8):
            # (round 6: on the shifted lattice too -- final sums only, additive_control_split)
            if rs == 8 and wres and _dbg(debug, 'SDP_COL_LEAN') != '0':
                short = short_pass_source(model, utab[0], 'SDP_COL_LEAN2')
                if short and _dbg(debug, 'SDP_COL_BNB') != '0':
                    short += ('\n#define SDP_COL_BNB 1          // the short first pass as a certified branch and bound over '
                              'blocks of controls (sdp_lean2_bnb)')

            elif rs == 4 and not wres and _dbg(debug, 'SDP_COL_WIDE') != '0' and _dbg(debug, 'SDP_COL_LEAN') in (None, '0') \
                    and _dbg(debug, 'SDP_COL_FILTER_TOP2') in (None, '1'):
                short = short_pass_source(model, utab[0], 'SDP_COL_WIDE2')
                if short and _dbg(debug, 'SDP_COL_BNB') != '0':
                    short += ('\n#define SDP_COL_BNB 1          // the short wide first pass as a certified branch and bound over '
                              'blocks of controls (sdp_short_bnb)')
                    if int(column[0]) >= 512 and not _dbg(debug, 'SDP_BNB_CHUNK'):
                        short += '\n#define SDP_BNB_CHUNK 4        // blocks whose bounds are evaluated together (see SDP_COL_A_GROUP above)'
        if short:
            lines += [short, '']
    lines += ['#include "sdp_column_kernel.h"    // also brings in sdp_sweep_kernel.h', '']
    return lines


def line_filter_applies(model, dtype, W, debug=None):
    """ONE state variable whose perturbation enters x' through final sums (`x + u - w`, the shop inventory of the
    reference's tutorial, doc/example_inventory.py:31-33), a cost that does not see the perturbation, 8-byte reals:
    the certified filter on the shifted lattice with the value array itself as the table (csrc/sdp_line_kernel.h).
    (`debug`: SDP_LINE_FILTER = 0 switches it off, A/B runs.)"""
    if _dbg(debug, 'SDP_LINE_FILTER', '1') == '0':
        return False
    return bool(model.n_state == 1 and model.n_perturb == 1 and 1 <= int(W) <= 1024 and np.dtype(dtype).itemsize == 8
                and not model.time_dep and model.lead_depends_on_w and not model.cost_depends_on_w
                and model.lead_split() is not None)


def line_functions_source(model):
    """C++ text of the slices csrc/sdp_line_kernel.h uses: a, the signed sum B of the b_i with the sum of their
    magnitudes, the cost -- each with the reference's own operations (as separable_functions_source emits them
    for the column kernel's shifted lattice)"""
    a_node, terms = model.lead_split()
    out = []
    lines = ['SDP_DEV sdp_real sdp_model_lead_a(const sdp_real *x, const sdp_real *u, sdp_real t)',
             '{', '    (void)x; (void)u; (void)t;']
    names = _emit_body(model, model.slice_nodes([a_node]), lines)
    lines += ['    return {};'.format(names[a_node.id]), '}']
    out.append('\n'.join(lines))
    chain = model.lead_split_chain()
    if chain is not None:
        leaves = [n for n, _ in chain[0]]
        lines = ['SDP_DEV sdp_real sdp_model_lead_aabs(const sdp_real *x, const sdp_real *u, sdp_real t)',
                 '{', '    (void)x; (void)u; (void)t;']
        names = _emit_body(model, model.slice_nodes(leaves), lines)
        lines += ['    return {};'.format(' + '.join('fabs({})'.format(names[n.id]) for n in leaves)), '}']
        out.append('\n'.join(lines))
    lines = ['SDP_DEV void sdp_model_lead_b(const sdp_real *x, sdp_real w, sdp_real t, sdp_real &b, sdp_real &babs)',
             '{', '    (void)x; (void)w; (void)t;']
    names = _emit_body(model, model.slice_nodes([b for b, _ in terms]), lines)
    b0, s0 = terms[0]
    lines.append('    b = {}{};'.format('-' if s0 < 0 else '', names[b0.id]))
    lines.append('    babs = fabs({});'.format(names[b0.id]))
    for bn, sg in terms[1:]:
        lines.append('    b = b {} {};'.format('+' if sg > 0 else '-', names[bn.id]))
        lines.append('    babs = babs + fabs({});'.format(names[bn.id]))
    lines.append('}')
    out.append('\n'.join(lines))
    lines = ['SDP_DEV sdp_real sdp_model_cost(const sdp_real *x, const sdp_real *u, sdp_real w,',
             '                                sdp_real t)',
             '{', '    (void)x; (void)u; (void)w; (void)t;']
    names = _emit_body(model, model.slice_nodes([model.cost]), lines)
    lines += ['    return {};'.format(names[model.cost.id]), '}']
    out.append('\n'.join(lines))
    return '\n\n'.join(out)


def translation_unit(model, dtype, lanes, column=None, staged=None, window=None,
                     per_control=None, filtered=False, utab=None, lead_axes=0, col_cfg=None, debug=None, wres=0,
                     lead_perm=None, line=0):
    """column: None for the generic node-order kernels, or (N0, W[, controls, columns]) to also
    build the column kernels of csrc/sdp_column_kernel.h for a storage-separable
    model on a grid with N0 points along axis 0 and W perturbation points.
    col_cfg: the tuple `column_config` returned for exactly this unit (the caller plans once and
    hands the plan over; None: planned here, same arguments).
    staged: None, or the dict of `staged_config` to also build the LDS-staged
    generic kernel of csrc/sdp_staged_kernel.h (node order, any traceable model).
    window: None, or the tuple of `column_window_config` (column kernel whose
    table holds a window of rows of axis 0).
    per_control: None, or the tuple of `column_percontrol_config` (column kernel
    that rebuilds its table for every control).
    filtered: column kernel with the certified expectation-first filter (SDP_COL_FILTER of
    csrc/sdp_colfilter_kernel.h; see `column_filter_applies`).
    utab: None, or (frontier nodes, capacity of the control table in controls) of `control_table_plan`
    (filtered kernel only): the first pass reads the column-uniform sub-expressions from a table.
    debug: None (the product), or a dict of diagnostic switches (see DEBUG_NAMES)."""
    debug = check_debug(debug)
    real = {'float64': 'double', 'float32': 'float'}[np.dtype(dtype).name]
    head = _prologue_lines(model, real, lanes, debug)
    if column is not None:
        assert model.column_shareable
        if per_control is not None:
            col_cfg = per_control
        elif window is not None:
            col_cfg = window
        elif col_cfg is None:
            col_cfg = column_config(column[0], column[1], model.n_state, dtype,
                                    use_wpair(model, dtype, debug), filtered,
                                    max_controls=column[2] if len(column) > 2 else None,
                                    n_columns=column[3] if len(column) > 3 else None,
                                    shift=filtered and column_shift_applies(model, dtype, debug=debug),
                                    utab_values=(utab_reals(len(utab[0]), utab[1]) if utab is not None and filtered else 0),
                                    debug=debug)
        if col_cfg is None:
            raise ValueError('the column kernel does not fit this grid (its table exceeds the LDS of a CU)')
        head += _column_lines(model, dtype, column, col_cfg, window, per_control, filtered, utab, debug,
                              wres if (window is None and per_control is None) else 0)
    elif staged is not None:
        tile = tuple(staged['tile']) + (1,) * (4 - len(staged['tile']))
        head += ['#define SDP_STG_THREADS {}'.format(int(staged['threads']))] + [
            '#define SDP_STG_T{} {}'.format(k, int(tile[k])) for k in range(4)] + [
            '#define SDP_STG_CU {}'.format(int(staged['cu'])),
            '#define SDP_STG_CW {}'.format(int(staged['cw'])),
            '#define SDP_STG_CAP {}'.format(int(staged['cap'])),
            '#include "sdp_staged_kernel.h"    // also brings in sdp_sweep_kernel.h',
            '']
    elif line:
        # one state variable with noise in its sums: the filter on the shifted lattice, the value array as its table
        chain = model.lead_split_chain()
        head += ['#define SDP_LINE 1',
                 '#define SDP_LINE_W {}'.format(int(line)),
                 '#define SDP_LINE_CHAIN {}        // additions of a chain of sums that was regrouped (0: the final-sum form)'.format(
                     chain[1] if chain is not None else 0)]
        if _dbg(debug, 'SDP_LINE_FILTER_SCALE'):
            head.append('#define SDP_LINE_FILTER_SCALE {}'.format(float(_dbg(debug, 'SDP_LINE_FILTER_SCALE'))))
        if _dbg(debug, 'SDP_LINE_TOP2') is not None:
            head.append('#define SDP_LINE_TOP2 {}'.format(int(_dbg(debug, 'SDP_LINE_TOP2'))))
        head += [line_functions_source(model), '',
                 '#include "sdp_sweep_kernel.h"    // brings in sdp_line_kernel.h', '']
    elif lead_axes:
        # several controlled state variables: node-order sweep with the filter on an array reduced over w
        head += ['#define SDP_LEAD_AXES {}'.format(int(lead_axes)),
                 '#define SDP_LEAD_COST_HAS_W {}'.format(1 if model.cost_depends_on_w else 0)]
        if lead_perm is not None and tuple(lead_perm) != tuple(range(model.n_state)):
            full = tuple(lead_perm) + tuple(range(model.n_state, 4))
            head.append('#define SDP_LEAD_PERM {{{}}}     // state variable of logical axis j: the stocks are not listed first'.format(
                ', '.join(str(int(k)) for k in full)))
        if _dbg(debug, 'SDP_LEAD_FILTER_SCALE'):
            head.append('#define SDP_LEAD_FILTER_SCALE {}'.format(float(_dbg(debug, 'SDP_LEAD_FILTER_SCALE'))))
        if _dbg(debug, 'SDP_LEAD_UNROLL'):
            head.append('#define SDP_LEAD_UNROLL {}'.format(int(_dbg(debug, 'SDP_LEAD_UNROLL'))))
        head += [lead_functions_source(model, int(lead_axes), lead_perm), '',
                 '#include "sdp_sweep_kernel.h"    // brings in sdp_lead_kernel.h', '']
    else:
        head += ['#include "sdp_sweep_kernel.h"', '']
    return '\n'.join(head)


# ---------------------------------------------------------------------------
# LDS-staged generic kernel (csrc/sdp_staged_kernel.h): tile / chunk planning
# ---------------------------------------------------------------------------
STAGED_TILES = {1: (512,), 2: (16, 32), 3: (8, 8, 8), 4: (4, 4, 4, 8)}
STAGED_LDS_BYTES = 76 * 1024         # per workgroup: two 512-thread workgroups per CU


def staged_row_stride(length, tile_last, d, rs):
    """sdp_stg_row_stride of csrc/sdp_staged_kernel.h"""
    if d == 1 or tile_last >= 32 or rs != 8:
        return length | 1
    return (length + tile_last - 1) // (2 * tile_last) * (2 * tile_last) + tile_last


def staged_config(model, state_grid, perturb_grid, box, dtype, t_value=0.0, n_samples=24, debug=None):
    """Tile shape, chunk sizes and LDS budget of the staged kernel for `model`
    on this discretisation.  The chunk (controls x perturbation points whose
    next states share one staged box) is the largest whose box -- measured by
    evaluating the traced dynamics with numpy at the corners of sample tiles
    and chunks, exactly what the kernel does per chunk -- fits the budget for
    most samples.  A bad choice costs time only: cells outside the staged box
    read global memory (csrc/sdp_staged_kernel.h).
    box: dict with lo, hi, n ([nu][S] or [nu][1]) and per_node."""
    from .trace import evaluate
    d = model.n_state
    rs = np.dtype(dtype).itemsize
    shape = tuple(len(g) for g in state_grid)
    tile = STAGED_TILES[d]
    cap = STAGED_LDS_BYTES // rs
    W = len(perturb_grid[0]) if perturb_grid and model.n_perturb else 0
    wg = np.asarray(perturb_grid[0], dtype=float) if W else np.zeros(1)
    Wn = max(W, 1)
    nu = model.n_control
    lo, hi, n = box['lo'], box['hi'], box['n']
    S = int(np.prod(shape))
    rng = np.random.default_rng(2024)
    smin = np.array([g[0] for g in state_grid], dtype=float)
    span = np.array([g[-1] - g[0] for g in state_grid], dtype=float)
    nm1 = np.array([len(g) - 1 for g in state_grid], dtype=float)
    tiles_per = [-(-shape[k] // tile[k]) for k in range(d)]

    def control_values(flat_nodes, ci):
        """numpy.linspace point `ci` (flat C-order lattice index, clipped) of each node"""
        col = flat_nodes if box['per_node'] else np.zeros_like(flat_nodes)
        nn = n[:, col].astype(np.int64)
        tot = np.prod(nn, axis=0)
        flat = np.minimum(ci, tot - 1)
        out = []
        for c in range(nu - 1, -1, -1):
            k = flat % nn[c]
            flat = flat // nn[c]
            l, h = lo[c, col], hi[c, col]
            step = np.where(nn[c] > 1, (h - l) / np.maximum(nn[c] - 1, 1), 0.0)
            out.append(np.where(k == nn[c] - 1, h, k * step + l))
        return out[::-1], tot

    def volume(cu, cw):
        vols = []
        for _ in range(n_samples):
            torg = [int(rng.integers(0, tiles_per[k])) * tile[k] for k in range(d)]
            # corner nodes of the tile (clipped to the grid)
            idx = np.array(np.meshgrid(*[[torg[k], min(torg[k] + tile[k], shape[k]) - 1]
                                         for k in range(d)], indexing='ij')).reshape(d, -1)
            flat = np.ravel_multi_index(tuple(idx), shape)
            x = [np.asarray(state_grid[k], dtype=float)[idx[k]] for k in range(d)]
            _, tot = control_values(flat, 0)
            c0 = int(rng.integers(0, max(int(tot.max()) - cu, 0) + 1))
            w0 = int(rng.integers(0, max(Wn - cw, 0) + 1))
            qlo = np.full(d, np.inf)
            qhi = np.full(d, -np.inf)
            for ci in (c0, c0 + cu - 1):
                u, _ = control_values(flat, ci)
                for wi in (w0, min(w0 + cw, Wn) - 1):
                    with np.errstate(all='ignore'):
                        xn, _ = evaluate(model, x, u, [wg[wi]] if model.n_perturb else [], t_value)
                    for k in range(d):
                        p = (np.asarray(xn[k], dtype=float) - smin[k]) / span[k] * nm1[k]
                        q = np.clip(np.floor(np.nan_to_num(p, nan=0.0, posinf=1e9, neginf=-1e9)),
                                    0, shape[k] - 2)
                        qlo[k] = min(qlo[k], q.min())
                        qhi[k] = max(qhi[k], q.max())
            ext = [min(qhi[k] + 1, shape[k] - 2) + 2 - max(qlo[k] - 1, 0) for k in range(d)]
            ext[-1] = staged_row_stride(int(ext[-1]), tile[-1], d, rs)
            vols.append(float(np.prod(ext)))
        return float(np.percentile(vols, 80))

    cws = sorted({max(1, Wn >> s) for s in range(0, 8)}, reverse=True)
    best = None
    # controls interleaved per chunk (what depends on w alone is shared by them).  Measured on
    # the control-coupled 256^3 x 64 x 32 problem: 4 -> 93.7 ms, 2 -> 120.3 ms (4 spills five
    # registers at the 128-VGPR bound two workgroups per CU impose; still faster)
    cus = (4, 2, 1)
    if _dbg(debug, 'SDP_STG_CU'):                       # A/B runs
        cus = (int(_dbg(debug, 'SDP_STG_CU')),)
    for cu in cus:
        for cw in cws:
            if best is not None and cu * cw <= best[0] * best[1]:
                continue
            if volume(cu, cw) <= cap:
                best = (cu, cw)
                break                          # smaller cw only shrinks the chunk
    if best is None:
        best = (1, 1)
    return dict(threads=int(np.prod(tile)), tile=tile, cu=best[0], cw=best[1], cap=int(cap))


COLUMN_LDS_MAX = 160 * 1024          # LDS of one gfx950 CU


def use_wpair(model, dtype, debug=None):
    """float32 tables interleave perturbation points 2k and 2k+1 (SDP_COL_WPAIR of
    csrc/sdp_column_kernel.h): one 8-byte LDS read serves two lattice cells.
    Needs a perturbation and an x0' that does not depend on it (one axis-0 cell
    per control).  Same bits as the plain layout; measured on 256^3 x 64 x 32:
    exact 6.05 -> 5.77 ms, fused 4.43 -> 3.66 ms.  (A table of (T[r], T[r+1])
    pairs was also measured for 4-byte reals and did not pay: not kept.)"""
    if _dbg(debug, 'SDP_COL_WPAIR'):                    # A/B runs
        return bool(int(_dbg(debug, 'SDP_COL_WPAIR')))
    return (np.dtype(dtype).itemsize == 4 and model.n_perturb > 0
            and not model.lead_depends_on_w and not model.trail_depends_on_u)


def column_shift_applies(model, dtype, table=None, debug=None):
    """The certified filter with a perturbation that reaches x0' additively (`x + u - w`): the first
    pass then reads a table reduced over w on a lattice that the perturbation points have SHIFTED
    (SDP_COL_SHIFT of csrc/sdp_colfilter_kernel.h).  8-byte reals only: in 4-byte reals the rounding of
    the positions alone would put most controls inside the radius.  (`debug`: SDP_COL_SHIFT = 0 switches
    it off, A/B runs.)"""
    if _dbg(debug, 'SDP_COL_SHIFT', '1') == '0':
        return False
    ok = bool(model.n_perturb > 0 and model.lead_depends_on_w and model.lead_split() is not None
              and dtype is not None and np.dtype(dtype).itemsize == 8)
    if ok and table is not None:       # (n0, w, n_state): the shifted lattice must fit LDS beside the table
        ok = column_config(table[0], table[1], table[2], dtype, False, True, shift=True,
                           utab_values=UTAB_MAX_BYTES // np.dtype(dtype).itemsize + 256,    # (whatever the control table takes, block statistics included)
                           debug=debug) is not None
    return ok


def column_filter_applies(model, window=None, per_control=None, dtype=None, table=None, debug=None):
    """Can phase B of the column kernel run the certified expectation-first filter
    (SDP_COL_FILTER of csrc/sdp_colfilter_kernel.h)?  It needs a perturbation that reaches
    neither x0' nor the cost -- then the expectation commutes with the lerp along axis 0 and
    all but the surviving controls of a node are decided on a table reduced over w -- and the
    plain full-column table with the reference's arithmetic.  Same bits as without it
    (the survivors are re-evaluated with the reference's operations).  (`debug`: SDP_COL_FILTER = 0
    switches it off, A/B runs.)"""
    if _dbg(debug, 'SDP_COL_FILTER', '1') == '0':
        return False
    # (a cost that depends on the perturbation is fine since round 3: the first pass accumulates its
    # expectation with the reference's own values, sdp_col_cost_expect; x0' must still not depend on it)
    # (and so is a perturbation that reaches x0' through a final sum, in 8-byte reals: column_shift_applies)
    return bool(model.n_perturb > 0 and (not model.lead_depends_on_w or column_shift_applies(model, dtype, table, debug))
                and not model.trail_depends_on_u and window is None
                and per_control is None)


def column_config(n0, w, n_state, dtype, wpair=False, filtered=False, max_controls=None, n_columns=None,
                  shift=False, utab_values=0, debug=None, wres=0):
    """Compile-time shape of the column kernel for a grid with n0 points along
    axis 0 and w perturbation points: (threads, lds_bytes), or None if the
    table does not fit the LDS of a CU.  512-thread workgroups while two of
    them fit a CU, else 1024 threads (one workgroup then has to fill the CU's
    wave slots alone).  `wpair`: the table holds whole pairs of perturbation
    points (an odd count is rounded up).  `utab_values`: reals per parity buffer of the
    control table (SDP_COL_UTAB x SDP_COL_UTAB_N; 0: none).  `lds_bytes` is sizeof(SdpColLds)
    of the unit these arguments generate (`_column_lds` lays the struct out member by member;
    tests/test_trace_codegen.py compiles the largest accepted sizes against the struct's
    static_assert).
    `filtered` (certified filter, SDP_COL_FILTER): one lane per node of the column where that
    fits a workgroup (n0 <= 512) -- the lanes of a node then share nothing, the second pass
    is not repeated on them and fewer waves meet at the barriers; measured on 256^3 x 64 x 32
    fp64: 2.56 ms with 512 threads, 2.32 ms with 256 (and no register cap: 165 VGPRs).
    Long control lattices (`max_controls`, with `n_columns` columns in the grid) get more lanes per
    node: a lane walks its share of a node's controls one after the other, and at the reference's own
    problem sizes that share is what takes the time -- storage-AR1 41 x 61 nodes x <= 8001 controls:
    64 threads 4.13 ms, 512: 0.59, 1024: 0.37 (61 columns: one workgroup per CU at most anyway); Searev
    31 x 61 x 61 x <= 2201: 64 threads 1.91 ms, 512: 0.65, 1024: 0.74.  About 128 controls per lane,
    512 threads at most while the grid has columns for every CU, else 1024."""
    rs = np.dtype(dtype).itemsize
    w = max(int(w), 1)
    tw = w + (w & 1) if wpair else w
    sizes = (512, 1024)
    if filtered and n0 <= 512:
        first = max(64, (int(n0) + 63) // 64 * 64)
        if max_controls:
            lanes = 1
            while lanes * 128 < int(max_controls) and lanes < 64:
                lanes *= 2
            cap = 1024 if (n_columns is not None and int(n_columns) < 256) else 512
            first = max(first, min(first * lanes, cap))
        sizes = (first,) + tuple(t for t in sizes if t > first)
    if _dbg(debug, 'SDP_COL_THREADS'):                  # A/B runs (a CU then holds as many workgroups as fit)
        sizes = (int(_dbg(debug, 'SDP_COL_THREADS')),)
    if shift:
        # (threads, lds_bytes, rows of the shifted lattice): as many rows as LDS has room for beside the table
        # and the control table while two workgroups share a CU -- at least n0 + n0 / 8, so that
        # shifts spread over an eighth of the axis still fit --, else one workgroup per CU and 2 n0 rows
        # (`wres`: the resident-chunk form -- the table holds that many points -- may get a third workgroup per CU)
        for per_cu in ((3, 2, 1) if wres else (2, 1)):
            for threads in sizes:
                if per_cu >= 2 and threads > 512:
                    continue
                base = _column_lds(int(wres) or tw, w, n0, n_state, rs, threads, shift=True, shift_rows=0, utab_values=utab_values)
                rows = min(2 * int(n0) + 16, (COLUMN_LDS_MAX // per_cu - base - 1024) // (2 * rs))
                if rows >= int(n0) + max(8, int(n0) // 8):
                    return (threads, _column_lds(int(wres) or tw, w, n0, n_state, rs, threads, shift=True, shift_rows=rows,
                                                 utab_values=utab_values), int(rows))
        return None
    for threads in sizes:
        # (`wres`: the resident-chunk form, whose table holds that many perturbation points; its workgroup has
        # one lane per node, as many of them per CU as the image allows)
        lds = _column_lds(int(wres) or tw, w, n0, n_state, rs, threads, reduced=filtered, utab_values=utab_values,
                          partial_minima=not filtered)
        if lds * (2 if threads <= 512 else 1) <= COLUMN_LDS_MAX:
            return threads, lds
    return None


def column_resident_points(model, n0, w, n_state, dtype, filtered, shift, wpair, threads, utab_values=0, debug=None):
    """Perturbation points the LDS table of the filtered column kernel holds at a time (SDP_COL_WRES of
    csrc/sdp_colres_kernel.h), or 0 for the plain kernel with the whole W x n0 table.
    The whole table lets a CU hold two workgroups when it takes 64 KiB, and the kernel is bound by what there
    is to overlap (one workgroup per CU: 3.39 ms, two: 1.67 on the benchmark problem).  Half of the points
    resident -- the rest built twice -- halves the table: chosen when it takes the CU from < 3 to >= 3
    workgroups.  Needs the lean first pass of 8-byte reals without the shifted lattice, a cost that does not
    see the perturbation, one lane per node.  (`debug`: SDP_COL_WRES = 0 switches it off, k > 0 forces k.)"""
    rs = np.dtype(dtype).itemsize
    w = int(w)
    ok = (filtered and rs == 8 and not wpair and not model.cost_depends_on_w and w >= 4
          and int(threads) >= int(n0) and _dbg(debug, 'SDP_COL_LEAN', '-1') != '0')
    forced = _dbg(debug, 'SDP_COL_WRES')
    if forced is not None:
        k = int(forced)
        return k if (ok and 0 < k < w and 2 * k >= w) else 0
    if not ok:
        return 0
    half = (w + 1) // 2
    if shift:
        # (the shifted lattice sits beside the table: ask the planner of that image)
        both = [column_config(n0, w, n_state, dtype, wpair, True, shift=True, utab_values=utab_values, debug=debug, wres=k)
                for k in (0, half)]
        if None in both:
            return 0
        whole, part = both[0][1], both[1][1]
    else:
        whole = _column_lds(w, w, n0, n_state, rs, threads, reduced=True, utab_values=utab_values)
        part = _column_lds(half, w, n0, n_state, rs, threads, reduced=True, utab_values=utab_values)
    if COLUMN_LDS_MAX // whole < 3 <= COLUMN_LDS_MAX // part:
        return half if RESIDENT_CHUNKS_DEFAULT else 0
    return 0


# measured on the benchmark problem (256^3 x 64 x 32, 8-byte reals), same box: whole table, two workgroups per CU
# 1.65 - 1.67 ms; 16 of 32 points resident, four per CU 1.555 ms (three: 1.62, two: 1.92 -- the chunked form costs
# 15 % at equal occupancy and wins by what it lets overlap); profiles/r04_column_ab.txt
RESIDENT_CHUNKS_DEFAULT = True


def column_wide_loads(n0, dtype, window, debug=None):
    """16-byte vertex loads in the table build (SDP_COL_A_WIDE_LOADS of csrc/sdp_column_kernel.h): a lane
    takes 16 / sizeof(real) adjacent rows.  Needs whole groups of rows, no row window, exact arithmetic.
    (`debug`: SDP_COL_A_WIDE_LOADS = 0 switches it off, 1 forces it for 4-byte reals: A/B runs.)"""
    if _dbg(debug, 'SDP_COL_A_WIDE_LOADS', '1') == '0':
        return False
    rpl = 16 // np.dtype(dtype).itemsize
    if rpl != 2 and _dbg(debug, 'SDP_COL_A_WIDE_LOADS') != '1':
        # 4-byte reals: four rows per lane, but the pair layout of their table scatters the stores --
        # measured on 512^3 fp32: 11.5 ms against 11.2 without (not used)
        return False
    return int(n0) % rpl == 0 and window is None


def column_build_order(threads, w, rows, rows_per_lane=1):
    """How phase A of the column kernel deals the W x rows table entries to the threads
    (SDP_COL_A_ORDER / SDP_COL_A_LW of csrc/sdp_column_kernel.h): (2, lanes_per_w) when the
    perturbation points fill the workgroup's thread groups -- a thread then keeps its w, reads
    the trailing cell once and walks the rows with constant address steps (measured on MI355X:
    256^3 x 64 x 32 fp64 -2 %, fp32 -3.5 %) -- else (0, 0): entries dealt round-robin (with 9
    points on 16 groups of 32 lanes the first form leaves 44 % of the threads idle: +3 %)."""
    w = max(int(w), 1)
    best = (0, 0, 0.0)
    # (with 16-byte loads 32 lanes per point measured best: 256^3 x 64 x 32 fp64 1.65 ms, 64: 1.74, 16: 1.74)
    for lw in ((32, 64, 16, 8) if rows_per_lane > 1 else (64, 32, 16, 8)):
        if lw * rows_per_lane > rows or threads % lw:
            continue
        groups = threads // lw
        util = w / float(-(-w // groups) * groups)
        if util > best[2] + 1e-9:
            best = (2, lw, util)
    return (best[0], best[1]) if best[2] >= 0.85 else (0, 0)


def bnb_words(n_controls):
    """reals of block statistics kept beside a control table of `n_controls` controls (SDP_BNB_WORDS of
    csrc/sdp_column_kernel.h: four per block of 8 controls, larger blocks beyond 64 of them)"""
    n = max(int(n_controls), 1)
    block = 8
    while (n + block - 1) // block > 64:
        block *= 2
    return 4 * ((n + block - 1) // block + 1)


def utab_reals(n_values, n_controls):
    """reals per parity buffer of a control table of `n_values` sub-expressions x `n_controls` controls, with the
    block statistics of the branch and bound (what `utab_values` of the planning functions counts; 0: no table)"""
    return int(n_values) * int(n_controls) + bnb_words(n_controls) if n_values else 0


def _column_lds(tw, w, rows, n_state, rs, threads, reduced=False, shift=False, shift_rows=0, utab_values=0,
                partial_minima=False):
    """sizeof(SdpColLds) of csrc/sdp_column_kernel.h, member by member with the alignment rules of
    the C++ struct.  tw: perturbation points the table holds (SDP_COL_TW; SDP_COL_WRES of the resident-chunk
    form; SDP_COL_WCHUNK of the table per control); rows: rows of axis 0 the table holds;
    reduced: with the reduced table `ad` of the certified filter (full-column table) -- one real per row in the
    lean form of 8-byte reals, 16 bytes per row otherwise;
    shift: the shifted lattice -- `ad` then has `shift_rows` rows of two reals -- and the shifts of the
    perturbation points; utab_values: reals per parity buffer of the control table (0: the struct's two);
    partial_minima: the unfiltered sweep's per-thread partial minima (SDP_COL_LDS_PART)."""
    dt = n_state - 1
    part = threads if partial_minima else 1
    members = [(rs, tw * rows, 16),                          # T
               (rs, w * dt, rs), (rs, w * dt, rs),           # w_lam, w_oml
               (rs, 1, rs), (rs, 1, rs),                     # pw, gw (SDP_COL_WMODE 2 only: not generated)
               (rs, part, rs), (4, part, 4),                 # part_J, part_i
               (4, w * dt, 4),                               # w_off
               (4, 4, 4), (4, 1, 4), (8, 2, 8),              # win, next_unit, dcol
               (rs, 2 * (int(utab_values) + 4 if utab_values else 2), 16)]        # utab[2][.. + 4 column statistics]
    if reduced or shift:
        per_row = 4 if rs == 4 else (2 if shift else 1)      # SDP_COL_LDS_AD (8-byte reals: the lean form is the default)
        members.append((rs, per_row * int(shift_rows if shift else rows), 16))    # ad
    if shift:
        members += [(4, 2 * w, 4), (rs, 2 * w, rs), (rs, 2 * w, rs), (4, 8, 4)]                   # sh_q, sh_f, sh_c, sh_k
    off = 0
    for size, count, align in members:
        off = (off + align - 1) // align * align + size * int(count)
    return (off + 15) // 16 * 16


def column_percontrol_config(n0, w, n_state, dtype, debug=None):
    """Shape of the column kernel with a table per control (SDP_TRAIL_HAS_U of
    csrc/sdp_column_kernel.h): (threads, lds_bytes, w_chunk).  One thread per node of
    a column (at most 512; longer columns are split over workgroups), and a table of
    `w_chunk` perturbation points at a time, about 32 KiB: four to five workgroups
    then share a CU and hide each other's barriers and load latencies."""
    rs = np.dtype(dtype).itemsize
    w = max(int(w), 1)
    threads = min(512, max(64, (int(n0) + 63) // 64 * 64))
    wchunk = max(1, min(w, (32 * 1024) // (int(n0) * rs)))
    if _dbg(debug, 'SDP_COL_WCHUNK'):                   # A/B runs
        wchunk = max(1, min(w, int(_dbg(debug, 'SDP_COL_WCHUNK'))))
    lds = _column_lds(wchunk, w, n0, n_state, rs, threads)
    if lds > COLUMN_LDS_MAX:
        return None
    return threads, lds, wchunk


def column_window_config(n0, w, n_state, dtype, reach_rows):
    """Row-window shape of the column kernel for a grid whose full W x n0 table
    does not fit the LDS of a CU (SDP_COL_ROWS of csrc/sdp_column_kernel.h):
    (threads, lds_bytes, rows, seg_nodes) or None.  A unit is a segment of
    `seg_nodes` nodes of a column; its table holds `rows` rows of axis 0, enough
    for the segment plus `reach_rows` (the rows the controls of ONE node span,
    measured by the caller) plus the interpolation partner and a margin.  Two
    512-thread workgroups per CU while that leaves segments of >= 64 nodes,
    else one 1024-thread workgroup."""
    rs = np.dtype(dtype).itemsize
    w = max(int(w), 1)
    for threads, wgs in ((512, 2), (1024, 1)):
        budget = COLUMN_LDS_MAX // wgs
        fixed = _column_lds(w, w, 0, n_state, rs, threads)
        rows = (budget - fixed) // (w * rs)
        rows = min(rows // 32 * 32, n0)
        seg = (rows - int(reach_rows) - 4) // 64 * 64
        if rows >= 128 and seg >= 64:
            return threads, _column_lds(w, w, rows, n_state, rs, threads), int(rows), int(min(seg, n0))
    return None


def column_lds_bytes(n0, w, n_state, dtype):
    """LDS bytes of the column kernel's image (struct SdpColLds), plain table."""
    cfg = column_config(n0, w, n_state, dtype)
    return cfg[1] if cfg else COLUMN_LDS_MAX + 1


HIPCC_FLAGS = ['--genco', '--offload-arch=gfx950', '-O3', '-ffp-contract=off',
               '-fno-fast-math', '-std=c++17', '-I', CSRC]


_HEADERS = ('sdp_kernel_args.h', 'sdp_device.h', 'sdp_sweep_kernel.h', 'sdp_column_kernel.h', 'sdp_colfilter_kernel.h', 'sdp_colres_kernel.h',
            'sdp_colfull_kernel.h', 'sdp_colu_kernel.h',
            'sdp_lead_kernel.h', 'sdp_staged_kernel.h')
_digest_cache = {}


def _headers_digest():
    """sha256 state over the kernel headers (re-read only when a file changes)"""
    stamp = tuple(os.stat(os.path.join(CSRC, fn)).st_mtime_ns for fn in _HEADERS)
    h = _digest_cache.get(stamp)
    if h is None:
        h = hashlib.sha256()
        for fn in _HEADERS:
            with open(os.path.join(CSRC, fn), 'rb') as f:
                h.update(f.read())
        _digest_cache.clear()
        _digest_cache[stamp] = h
    return h.copy()


_compiler_id = None
HIPCC_PATH = '/opt/rocm/bin/hipcc'       # (_native.py, which reads HIPCC from the environment, sets it)


def compiler_identity():
    """What `hipcc --version` prints (HIP version, clang version and commit), once per process: part of every
    cache key, so that code objects built by another compiler -- a cache that travelled from a container with a
    different ROCm -- are never picked up (the key named the source, headers and flags only).
    Without a runnable hipcc the identity is a fixed string: nothing can be compiled then anyway."""
    global _compiler_id
    if _compiler_id is None:
        import subprocess
        try:
            out = subprocess.run([HIPCC_PATH, '--version'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 timeout=60).stdout.decode(errors='replace')
            # (the lines that name the compiler; not the installation directory, which may be a symlink)
            keep = [ln.strip() for ln in out.splitlines() if ln.startswith(('HIP version', 'AMD clang version', 'clang version'))]
            _compiler_id = '\n'.join(keep) or 'hipcc without a version'
        except (OSError, subprocess.SubprocessError):
            _compiler_id = 'no hipcc'
    return _compiler_id


BUILD_RULES = 'spill fence 1'       # what _native.compile_model does beyond one hipcc call (below): part of the key


def source_key(source):
    """Cache key of a code object: generated text + kernel headers + flags + the compiler's identity."""
    h = _headers_digest()
    # (headers only some units include enter the key of those units alone: an edit there leaves the other code objects valid)
    if '#define SDP_LINE ' in source:
        with open(os.path.join(CSRC, 'sdp_line_kernel.h'), 'rb') as f:
            h.update(f.read())
    h.update(source.encode())
    h.update(' '.join(HIPCC_FLAGS[:-1]).encode())
    h.update(compiler_identity().encode())
    h.update(BUILD_RULES.encode())
    return h.hexdigest()[:24]


# ---------------------------------------------------------------------------
# Spill code and the execution mask.  Round 5 found a kernel of the test suite (resident chunks, long first pass,
# 128 threads under a budget of 64 vector registers) that never ended: the compiler (ROCm 7.2's clang 20) had put the
# store of a spilled vector register -- threadIdx.x -- at the top of a block where divergent paths JOIN, before the
# instruction that restores the execution mask (`s_or_b64 exec, exec, s[..]`).  A store to scratch memory writes the
# active lanes only: a wavefront that had skipped the branch altogether stored nothing, reloaded whatever its scratch
# memory held, took itself for wave 0 and never claimed the workgroup's next unit.  Which kernels are hit depends on
# the register allocation alone -- three unrelated one-line changes each made it come and go -- so the build LOOKS:
# a code object whose kernels use scratch memory or accumulation registers (the two places spilled vector registers
# go) is compiled to assembly once more and scanned for exactly this pattern; a kernel that shows it is rebuilt with
# fewer waves per SIMD asked of the register allocator (more registers, less spill code) until it is clean, or refused.
# ---------------------------------------------------------------------------
_ASM_FUNC = re.compile(r'^([A-Za-z_]\w*):\s*; @')
_ASM_BLOCK = re.compile(r'^(\.LBB\d+_\d+):|^; %bb\.(\d+):')
_ASM_SPILL = re.compile(r'^\s+(?:scratch_(?:store|load)\w*|buffer_(?:store|load)\w*|v_accvgpr_(?:write|read)\w*)\b.*;.*\b(?:Spill|Reload)\b')
_ASM_END_CF = re.compile(r'^\s+s_or_b64\s+exec,\s*exec,')
_ASM_EXEC_WRITE = re.compile(r'^\s+s_\w+\s+exec(?:_lo|_hi)?\b|^\s+s_\w*saveexec\w*\b|^\s+v_cmpx')


def spill_hazards(asm):
    """[(kernel, block, line number of the mask restore, [spill instructions before it]), ...] of an assembly listing
    (hipcc -S): spill stores / reloads of vector registers between the top of a basic block and the `s_or_b64 exec,
    exec, ..` that re-activates the lanes joining there.  Those instructions run for the lanes of ONE incoming path."""
    out, func, block, pending, at_top = [], None, None, [], False
    for n, line in enumerate(asm.splitlines(), 1):
        m = _ASM_FUNC.match(line)
        if m:
            func, block, pending, at_top = m.group(1), 'entry', [], False     # (the entry block: every launched lane is active)
            continue
        m = _ASM_BLOCK.match(line)
        if m:
            block, pending, at_top = m.group(1) or '%bb.' + m.group(2), [], True
            continue
        if not at_top:
            continue
        if _ASM_SPILL.match(line):
            pending.append(line.strip())
        elif _ASM_END_CF.match(line):
            if pending:
                out.append((func, block, n, pending))
            at_top = False
        elif _ASM_EXEC_WRITE.match(line):
            at_top = False
    return out


def _msgpack_ints_after(blob, key):
    out, i = [], blob.find(key)
    while i >= 0:
        j = i + len(key)
        b = blob[j] if j < len(blob) else 0
        out.append(b if b <= 0x7f else int.from_bytes(blob[j + 1:j + 1 + {0xcc: 1, 0xcd: 2, 0xce: 4}.get(b, 0)], 'big') if b in (0xcc, 0xcd, 0xce) else -1)
        i = blob.find(key, j)
    return out


def code_object_may_spill(blob):
    """True when a kernel of the code object (its bytes) uses scratch memory or accumulation registers, by its own
    metadata (the msgpack note: .private_segment_fixed_size, .agpr_count), or when the metadata cannot be read."""
    scratch = _msgpack_ints_after(blob, b'.private_segment_fixed_size')
    agprs = _msgpack_ints_after(blob, b'.agpr_count')
    if not scratch or len(agprs) != len(scratch):
        return True
    return any(v != 0 for v in scratch) or any(v != 0 for v in agprs)
