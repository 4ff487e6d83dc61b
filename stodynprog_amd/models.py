"""Problem definitions used by the tests, the golden-vector generator and bench.py.

Each problem is written the way a user of the library writes one: plain Python
callables made of numpy expressions, attached to a `SysDescription`.  They
restate the reference's documented examples (BASELINE.json configs 1-3) and
define the synthetic benchmark problem (configs 4-5); nothing here is device
code -- `DPSolver` traces these callables and generates the device model.

Sources restated (paths relative to the reference checkout):
  inventory    doc/example_inventory.py:28-91
  storage_ar1  examples/howto storage-AR1.ipynb, code cells 7-23
  searev       examples/20 Searev storage control/storage_control.py:34-117
               and searev_data.py:17-81
  nas_demo     stodynprog/stodynprog.py:879-957
"""
import numpy as np


# ----------------------------------------------------------------------------
# Minimal perturbation laws (duck-typed like scipy.stats frozen laws:
# sdp.py:196-207 only looks for .pdf / .pmf)
# ----------------------------------------------------------------------------
class NormalLaw(object):
    """Gaussian law exposing `.pdf`, evaluated with the same operator sequence
    as scipy.stats.norm(loc, scale).pdf so discretised weights are identical."""

    def __init__(self, loc=0., scale=1.):
        self.loc, self.scale = float(loc), float(scale)

    def pdf(self, x):
        z = (np.asarray(x, dtype=float) - self.loc) / self.scale
        return np.exp(-z ** 2 / 2.0) / np.sqrt(2 * np.pi) / self.scale

    def __repr__(self):
        return 'NormalLaw(loc={:g}, scale={:g})'.format(self.loc, self.scale)


class DiscreteLaw(object):
    """Finite discrete law exposing `.pmf` (like scipy.stats.rv_discrete)."""

    def __init__(self, values, probas):
        self.values = np.asarray(values, dtype=float)
        self.probas = np.asarray(probas, dtype=float)

    def pmf(self, x):
        x = np.asarray(x, dtype=float)
        out = np.zeros(x.shape)
        for v, p in zip(self.values, self.probas):
            out = np.where(x == v, p, out)
        return out

    def __repr__(self):
        return 'DiscreteLaw({} values)'.format(len(self.values))


def _classes(api):
    """(SysDescription, DPSolver) of the requested implementation: this
    package by default, or any module exposing the same two names (the golden
    generator passes the reference package)."""
    if api is None:
        from . import SysDescription, DPSolver
        return SysDescription, DPSolver
    return api.SysDescription, api.DPSolver


# ----------------------------------------------------------------------------
# 1. Inventory control (config 1)
# ----------------------------------------------------------------------------
def inventory(api=None, h=0.5, p=3, c=1):
    """Shop inventory: 1 state (stock), 1 control (order), discrete demand."""
    SysDescription, DPSolver = _classes(api)
    shop = SysDescription((1, 1, 1), name='Shop Inventory')

    def stock_dyn(x, u, w):
        return (x + u - w,)
    shop.dyn = stock_dyn
    shop.perturb_laws = [DiscreteLaw([0, 1, 2, 3], [0.2, 0.4, 0.3, 0.1])]

    def order_box(x):
        return ((0, 10),)
    shop.control_box = order_box

    def shop_cost(x, u, w):
        return np.where(x > 0, x * h, -x * p) + u * c
    shop.cost = shop_cost

    solver = DPSolver(shop)
    solver.discretize_state(-3, 6, 10)
    solver.discretize_perturb(0, 3, 4)
    solver.control_steps = (1,)
    return shop, solver


def inventory_fine(api=None, n_x=600, n_u=257, n_w=16, h=0.5, p=3., c=1.):
    """The shop inventory (ONE state variable, `x + u - w`: reference doc/example_inventory.py:31-33, cost :59-65) on a
    finer grid with a continuous demand: a size where a 1-D problem is worth a kernel."""
    SysDescription, DPSolver = _classes(api)
    shop = SysDescription((1, 1, 1), name='Shop inventory, fine grid')

    def stock_dyn(x, u, w):
        return (x + u - w,)
    shop.dyn = stock_dyn
    shop.perturb_laws = [NormalLaw(2.0, 0.8)]

    def order_box(x):
        return ((0., 8.),)
    shop.control_box = order_box

    def shop_cost(x, u, w):
        return np.where(x > 0, x * h, -x * p) + u * c
    shop.cost = shop_cost

    solver = DPSolver(shop)
    solver.discretize_state(-8., 24., n_x)
    solver.discretize_perturb(0., 4., n_w)
    solver.control_steps = (8. / (n_u - 1),)
    return shop, solver


def inventory_markov(api=None, n_x=128, n_d=32, n_w=9, h=0.5, p=3., c=1., x_max=24., order_max=10.,
                     order_step=0.25):
    """The shop inventory next to an exogenous demand level (two state variables): the stock follows
    `x + u - demand` as in the reference's example (doc/example_inventory.py:31-33, cost :59-65), the
    demand of a period is a mean-reverting level plus noise.  The perturbation reaches the stock, not
    the cost: the column kernel filters it on the shifted lattice (csrc/sdp_colfilter_kernel.h, SDP_COL_SHIFT)."""
    SysDescription, DPSolver = _classes(api)
    shop = SysDescription((2, 1, 1), name='Shop inventory, Markov demand')
    mean, corr, sigma = 2.0, 0.7, 0.6

    def shop_dyn(x, d, u, w):
        return (x + u - (d + w), mean + corr * (d - mean) + 0.5 * w)
    shop.dyn = shop_dyn
    shop.perturb_laws = [NormalLaw(0, sigma)]

    def order_box(x, d):
        return ((0., order_max),)
    shop.control_box = order_box

    def shop_cost(x, d, u, w):
        return np.where(x > 0, x * h, -x * p) + u * c
    shop.cost = shop_cost

    solver = DPSolver(shop)
    solver.discretize_state(-8., x_max, n_x, 0., 4., n_d)
    solver.discretize_perturb(-3 * sigma, 3 * sigma, n_w)
    solver.control_steps = (order_step,)
    return shop, solver


# ----------------------------------------------------------------------------
# 2. Energy storage facing an AR(1) mismatch (config 2)
# ----------------------------------------------------------------------------
def storage_ar1(api=None, n_E=41, n_P=61, n_w=9, steps=(0.001, 0.1),
                dt=1., p_scale=1., p_corr=0.8, E_rated=10., P_rated=4., P_tol=0.5):
    SysDescription, DPSolver = _classes(api)
    innov_scale = p_scale * np.sqrt(1 - p_corr)     # sic: the notebook uses 1-p_corr
    tol = 0.9 * P_tol
    sto = SysDescription((2, 2, 1), name='Storage + AR(1)')

    def sto_dyn(E, P_mis, P_sto, P_cur, innov):
        return (E + P_sto * dt, p_corr * P_mis + innov)
    sto.dyn = sto_dyn

    def sto_box(E, P_mis):
        lo = np.max((-E / dt, -P_rated))
        hi = np.min(((E_rated - E) / dt, +P_rated))
        return ((lo, hi), (0, 0))
    sto.control_box = sto_box

    def sto_cost(E, P_mis, P_sto, P_cur, innov):
        P_dev = P_mis - P_cur - P_sto
        above = (P_dev - tol) ** 2
        mid = 0. * P_dev
        under = (P_dev + tol) ** 2
        cost = np.where(P_dev > tol, above, mid)
        cost = np.where(P_dev < -tol, under, cost)
        return cost
    sto.cost = sto_cost
    sto.perturb_laws = [NormalLaw(0, innov_scale)]

    solver = DPSolver(sto)
    p_mis_max = 4 * p_scale
    solver.discretize_state(0, E_rated, n_E, -p_mis_max, p_mis_max, n_P)
    solver.discretize_perturb(-4 * innov_scale, 4 * innov_scale, n_w)
    solver.control_steps = tuple(steps)
    return sto, solver


def storage_ar1_empirical_policy(solver, dt=1., E_rated=10., P_rated=4.):
    """'P_sto = P_mis whenever feasible' (notebook cell 26) on the full grid."""
    E, P_mis = solver.state_grid_full
    lo = np.maximum(-E / dt, -P_rated)
    hi = np.minimum((E_rated - E) / dt, +P_rated)
    pol = np.zeros(E.shape + (2,))
    pol[..., 0] = np.where(P_mis < lo, lo, np.where(P_mis > hi, hi, P_mis))
    return pol


# ----------------------------------------------------------------------------
# 3. SEAREV wave-energy converter + storage (config 3)
# ----------------------------------------------------------------------------
SEAREV = dict(c1=1.9799, c2=-0.9879, innov_std=0.00347, E_rated=10, P_rated=1.1,
              a=0.0, dt=0.1, power_max=1.1, damp=4.e6, torque_max=2e6)


def searev_power(speed, damp=SEAREV['damp'], torque_max=SEAREV['torque_max'],
                 power_max=SEAREV['power_max']):
    """Power take-off (MW) as a function of speed: damping torque, clipped to
    +-torque_max, power clipped to power_max (searev_data.py:70-81)."""
    tor = speed * damp
    tor = np.where(tor > torque_max, torque_max, tor)
    tor = np.where(tor < -torque_max, -torque_max, tor)
    P_prod = tor * speed / 1e6
    return np.where(P_prod > power_max, power_max, P_prod)


def searev(api=None, n_E=31, n_S=61, n_A=61, n_w=9, step=0.001):
    SysDescription, DPSolver = _classes(api)
    k = SEAREV
    c1, c2, dt, a = k['c1'], k['c2'], k['dt'], k['a']
    E_rated, P_rated, power_max = k['E_rated'], k['P_rated'], k['power_max']
    wec = SysDescription((3, 1, 1), name='Searev + Storage')

    def wec_dyn(E_sto, Speed, Accel, P_sto, innov):
        E_n = E_sto + (P_sto - a * abs(P_sto)) * dt
        S_n = (c1 + c2) * Speed - dt * c2 * Accel + innov
        A_n = (c1 + c2 - 1) / dt * Speed - c2 * Accel + innov / dt
        return (E_n, S_n, A_n)
    wec.dyn = wec_dyn

    def wec_box(E_sto, Speed, Accel):
        lo = np.max((-E_sto / (1 + a) / dt, -P_rated))
        hi = np.min(((E_rated - E_sto) / (1 - a) / dt, P_rated))
        return ((lo, hi),)
    wec.control_box = wec_box

    def wec_cost(E_sto, Speed, Accel, P_sto, innov):
        P_grid = searev_power(Speed) - P_sto
        return (P_grid / power_max) ** 2
    wec.cost = wec_cost
    wec.perturb_laws = [NormalLaw(0, k['innov_std'])]

    solver = DPSolver(wec)
    solver.discretize_state(0, E_rated, n_E,
                            -4 * .254, 4 * 0.254, n_S,
                            -4 * .227, 4 * .227, n_A)
    solver.discretize_perturb(-3 * k['innov_std'], 3 * k['innov_std'], n_w)
    solver.control_steps = (step,)
    return wec, solver


def searev_linear_policy(solver):
    """Heuristic initial law P_sto = P_prod - P_rated*E/E_rated
    (storage_control.py:123-134) on the full grid, shape dims+(1,)."""
    E, S, A = solver.state_grid_full
    pol = searev_power(S) - SEAREV['P_rated'] * E / SEAREV['E_rated']
    return pol[..., np.newaxis]


# ----------------------------------------------------------------------------
# 4. NaS storage demo (smoke-sized; sdp.py:879-957)
# ----------------------------------------------------------------------------
def nas_demo(api=None, n_E=51, n_P=41, n_w=11):
    SysDescription, DPSolver = _classes(api)
    E_rated, P_rated, a = 7.2, 2, 0.05
    scale, phi = 1.5, 0.8
    innov_scale = scale * np.sqrt(1 - phi ** 2)
    nas = SysDescription((2, 1, 1), name='NaS Storage')

    def nas_dyn(E, P_req, P_sto, innov):
        return (E + P_sto - a * abs(P_sto), phi * P_req + innov)
    nas.dyn = nas_dyn

    def nas_box(E, P_req):
        lo = np.max((-E / (1 + a), -P_rated))
        hi = np.min(((E_rated - E) / (1 - a), P_rated))
        return ((lo, hi),)
    nas.control_box = nas_box

    def nas_cost(E, P_req, P_sto, innov):
        P_dev = P_req - P_sto
        return P_dev ** 2
    nas.cost = nas_cost
    nas.perturb_laws = [NormalLaw(0, innov_scale)]

    solver = DPSolver(nas)
    solver.discretize_state(0, E_rated, n_E, -4 * scale, 4 * scale, n_P)
    solver.discretize_perturb(-3 * innov_scale, 3 * innov_scale, n_w)
    solver.control_steps = (.1,)
    return nas, solver


# ----------------------------------------------------------------------------
# 5. Synthetic 3-D benchmark problem (configs 4 and 5) -- FROZEN definition
# ----------------------------------------------------------------------------
# State grid linspace(0,1,N)^3; one control on the constant box [-1,1] with
# 64 points; one Gaussian perturbation on 32 points.  Storage-like axis 0 is
# driven by the control (reach +-8 cells at N=256); axes 1-2 are a stable
# linear process driven by the perturbation (Searev-shaped dynamics).  The
# cost is strictly convex in u so the minimiser is unique (exact policy
# indices).  Only + - * are used: bit-reproducible on any IEEE platform.
SYNTH = dict(b=8. / 255., a11=0.9, a12=0.1, a21=-0.1, a22=0.9, c=0.5,
             m1=0.5 * (1 - 0.9 - 0.1), m2=0.5 * (1 + 0.1 - 0.9),
             k1=1.8, k0=0.9, eps=0.1, kx=0.25, sigma=0.02,
             u_step=0.032, n_u=64)
# parameter block in the order oracle/sdp_oracle.c:oracle_vi_synth3d_f64 expects
SYNTH_PAR = [SYNTH[k] for k in ('b', 'm1', 'a11', 'a12', 'm2', 'a21', 'a22', 'c',
                                'k1', 'k0', 'eps', 'kx')]


def synthetic3d(api=None, N=256, n_w=32, stock_noise=0.0, nested=False):
    """(`stock_noise` != 0: the perturbation also reaches the stock, x0' = (x0 + b u) - stock_noise w --
    the shape of the reference's inventory example, doc/example_inventory.py:31-33, at benchmark size;
    `nested`: the same sum in another nesting, x0' = x0 + (b u - stock_noise w))"""
    SysDescription, DPSolver = _classes(api)
    p = SYNTH
    b, a11, a12, a21, a22, c = p['b'], p['a11'], p['a12'], p['a21'], p['a22'], p['c']
    m1, m2, k1, k0, eps, kx = p['m1'], p['m2'], p['k1'], p['k0'], p['eps'], p['kx']
    syn = SysDescription((3, 1, 1), name='Synthetic 3-D benchmark')

    def synth_dyn(x0, x1, x2, u, w):
        x0n = x0 + b * u
        if stock_noise:
            x0n = x0 + (b * u - stock_noise * w) if nested else x0n - stock_noise * w
        x1n = m1 + a11 * x1 + a12 * x2 + w
        x2n = m2 + a21 * x1 + a22 * x2 + c * w
        return (x0n, x1n, x2n)
    syn.dyn = synth_dyn

    def synth_box(x0, x1, x2):
        return ((-1., 1.),)
    syn.control_box = synth_box

    def synth_cost(x0, x1, x2, u, w):
        e = (k1 * x1 - k0) - u
        return e * e + eps * (u * u) + kx * x0
    syn.cost = synth_cost
    syn.perturb_laws = [NormalLaw(0, p['sigma'])]

    solver = DPSolver(syn)
    solver.discretize_state(0, 1, N, 0, 1, N, 0, 1, N)
    solver.discretize_perturb(-3 * p['sigma'], 3 * p['sigma'], n_w)
    solver.control_steps = (p['u_step'],)      # width 2 / 0.032 = 62.5 -> 64 points
    return syn, solver


def synthetic3d_coupled(api=None, N=256, n_w=32, gain=0.1, cross=0.0):
    """The synthetic benchmark with the control ALSO driving the second state
    variable (x1' += gain * u) -- no longer storage-separable: the partial
    interpolation over axes 1.. differs from control to control, so the column
    kernel's table does not apply and the LDS-staged tile kernel runs
    (csrc/sdp_staged_kernel.h).  `cross` != 0 additionally couples x1' to x0
    (a fully coupled model: no axis is exogenous)."""
    SysDescription, DPSolver = _classes(api)
    p = SYNTH
    b, a11, a12, a21, a22, c = p['b'], p['a11'], p['a12'], p['a21'], p['a22'], p['c']
    m1, m2, k1, k0, eps, kx = p['m1'], p['m2'], p['k1'], p['k0'], p['eps'], p['kx']
    syn = SysDescription((3, 1, 1), name='Synthetic 3-D benchmark, control-coupled')

    def coupled_dyn(x0, x1, x2, u, w):
        x0n = x0 + b * u
        x1n = m1 + a11 * x1 + a12 * x2 + w + gain * u
        if cross:                               # (a traced `0 * x0` would still count as a dependency)
            x1n = x1n + cross * x0
        x2n = m2 + a21 * x1 + a22 * x2 + c * w
        return (x0n, x1n, x2n)
    syn.dyn = coupled_dyn

    def synth_box(x0, x1, x2):
        return ((-1., 1.),)
    syn.control_box = synth_box

    def synth_cost(x0, x1, x2, u, w):
        e = (k1 * x1 - k0) - u
        return e * e + eps * (u * u) + kx * x0
    syn.cost = synth_cost
    syn.perturb_laws = [NormalLaw(0, p['sigma'])]

    solver = DPSolver(syn)
    solver.discretize_state(0, 1, N, 0, 1, N, 0, 1, N)
    solver.discretize_perturb(-3 * p['sigma'], 3 * p['sigma'], n_w)
    solver.control_steps = (p['u_step'],)
    return syn, solver


def synthetic3d_V0(state_grid, dtype=np.float64):
    """Closed-form initial cost-to-go on the grid (only + - * /: reproducible)."""
    x0 = np.asarray(state_grid[0], dtype=np.float64).reshape(-1, 1, 1)
    x1 = np.asarray(state_grid[1], dtype=np.float64).reshape(1, -1, 1)
    x2 = np.asarray(state_grid[2], dtype=np.float64).reshape(1, 1, -1)
    q0 = x0 - 0.5
    q1 = x1 - 0.3
    q2 = x2 - 0.7
    V = q0 * q0 + 0.5 * (q1 * q1)
    V = V + 0.25 * (q2 * q2)
    V = V + (0.3 * x0) * x1
    V = V - (0.2 * x1) * x2
    V = V + 0.1 / (1.0 + x2 * x2)
    return np.ascontiguousarray(V, dtype=dtype)


# ----------------------------------------------------------------------------
# 6. Small finite-horizon, time-dependent problem (bellman_recursion)
# ----------------------------------------------------------------------------
def finite_horizon(api=None, n_x=17, n_w=5):
    """1 state, 1 control, 1 perturbation; dynamics, cost and admissible box
    all depend on the time index (first argument, reference sdp.py:89-91)."""
    SysDescription, DPSolver = _classes(api)
    fh = SysDescription((1, 1, 1), stationnary=False, name='finite horizon')

    def fh_dyn(k, x, u, w):
        return (0.9 * x + u + w + 0.05 * k,)
    fh.dyn = fh_dyn

    def fh_cost(k, x, u, w):
        return (x - 0.1 * k) ** 2 + 0.1 * u * u

    fh.cost = fh_cost

    def fh_box(k, x):
        return ((-1., 1. + 0.5 * k),)
    fh.control_box = fh_box
    fh.perturb_laws = [NormalLaw(0, 0.1)]
    solver = DPSolver(fh)
    solver.discretize_state(-2, 2, n_x)
    solver.discretize_perturb(-0.3, 0.3, n_w)
    solver.control_steps = (0.125,)
    return fh, solver


# ----------------------------------------------------------------------------
# 7. Deterministic storage with time-indexed input data (bellman_recursion)
# ----------------------------------------------------------------------------
def pv_profile(T, dt=0.5):
    """Closed-form stand-in for the measured PV production of the reference's
    example (pv_prod.csv is not shipped): clipped day arcs with a slow
    modulation, only + - * / abs (reproducible bit for bit)."""
    k = np.arange(T, dtype=float)
    h = (k * dt) % 24.0                                  # hour of day
    arc = 1.0 - ((h - 12.5) / 6.5) * ((h - 12.5) / 6.5)  # parabola, zero at 6:00 and 19:00
    day = np.floor(k * dt / 24.0)
    scale = 0.55 + 0.15 * ((day * 3.0) % 4.0) / 3.0
    return np.where(arc > 0, arc * scale, 0.0)


def pv_storage(api=None, T=48, N_E=50, u_step=0.001, dt=0.5, loss=0.05):
    """Storage smoothing the output of a PV plant with perfect knowledge of the
    production -- the shape of the reference's finite-horizon examples
    (examples/01 Deterministic storage control/pv_storage_control.py:48-98 and
    det_storage_control.py:58-98): one state (stored energy), one control, no
    perturbation, `stationnary=False`, and a cost that LOOKS UP the production
    of the time step in a data array (`P_prod_data[k]`).  The admissible box
    uses np.max / np.min on scalar tuples like the reference's."""
    SysDescription, DPSolver = _classes(api)
    E_rated, P_rated = 2.0, 1.0
    P_prod_data = pv_profile(T, dt)
    sto = SysDescription((1, 1, 0), name='Deterministic Storage for PV', stationnary=False)

    def dyn_sto(k, E_sto, P_sto):
        E_sto_n = E_sto + (P_sto - loss * abs(P_sto)) * dt
        return (E_sto_n,)

    def admissible_controls(k, E_sto):
        P_neg = np.max((-E_sto / (1 + loss) / dt, -P_rated))
        P_pos = np.min(((E_rated - E_sto) / (1 - loss) / dt, P_rated))
        return ((P_neg, P_pos),)

    def cost_model(k, E_sto, P_sto):
        P_prod = P_prod_data[k]
        P_grid = P_prod - P_sto
        over = np.where(P_grid > 0.4, P_grid - 0.4, 0)
        neg = np.where(P_grid < 0, P_grid, 0)
        return over ** 2 + neg ** 2 + 0 * P_sto ** 2
    sto.dyn = dyn_sto
    sto.control_box = admissible_controls
    sto.cost = cost_model
    solver = DPSolver(sto)
    solver.discretize_state(0, E_rated, N_E)
    solver.control_steps = (u_step,)
    solver.P_prod_data = P_prod_data
    return sto, solver


# ----------------------------------------------------------------------------
# Two controlled stocks next to an exogenous inflow (reduced-array sweep, csrc/sdp_lead_kernel.h)
# ----------------------------------------------------------------------------
def two_reservoirs(api=None, n_a=48, n_b=48, n_y=24, n_w=9, steps=(0.125, 0.125)):
    """A cascade of two reservoirs fed by an AR(1) inflow: the upper one releases into the lower one,
    the lower one into the turbine whose output should follow a demand.  Two state variables are
    driven by the controls, the third is exogenous and carries the perturbation: the reference's API
    admits it like any other `dims` (stodynprog.py:57-81; 2-D control lattice :655-660)."""
    SysDescription, DPSolver = _classes(api)
    sysd = SysDescription((3, 2, 1), name='Two reservoirs')

    def dyn(a, b, y, u, v, w):
        return (a + (0.7 + 0.5 * y) - u, b + u - v, 0.3 + 0.7 * (y - 0.3) + w)
    sysd.dyn = dyn

    def box(a, b, y):
        return ((0., 1.), (0., 1.))
    sysd.control_box = box

    def cost(a, b, y, u, v, w):
        spill = np.where(a > 1.7, a - 1.7, 0.0 * a) + np.where(b > 1.7, b - 1.7, 0.0 * b)
        dry = np.where(a < 0.3, 0.3 - a, 0.0 * a) + np.where(b < 0.3, 0.3 - b, 0.0 * b)
        return (v - 0.8) * (v - 0.8) + 0.05 * (u - v) * (u - v) + 4.0 * spill + 8.0 * dry
    sysd.cost = cost
    sysd.perturb_laws = [NormalLaw(0, 0.1)]
    solver = DPSolver(sysd)
    solver.discretize_state(0., 2., n_a, 0., 2., n_b, -0.2, 0.8, n_y)
    solver.discretize_perturb(-0.3, 0.3, n_w)
    solver.control_steps = steps
    return sysd, solver
