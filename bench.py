#!/usr/bin/env python3
"""Benchmark of the value-iteration sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config NAME]

A step is ONE Bellman sweep (DPSolver.value_iteration's device work, reference
stodynprog/stodynprog.py:466-534) with the value array resident in HBM and
ping-ponged between sweeps.  Default workload: BASELINE.json configs[3], the
configuration the metric is quoted on (synthetic 256^3 state nodes x 64
controls x 32 perturbation points, fp64).  `--config` selects the other
BASELINE configs (ar1 = configs[1], searev = configs[2], synth512f32 =
configs[4]) for their own bench lines.

With N > 1 (one process per GPU, launched by `python -m torch.distributed.run`;
the processes themselves never import torch) the outer state axis is sharded
over the ranks and the J slabs are all-gathered with RCCL after every sweep,
inside the timed region.

Prints ONE JSON line on rank 0.
  roofline      the column kernel removes the 2^d-vertex gather per lattice cell (LDS table) and, by
                default, all but the near-minimal controls of a node (certified expectation-first
                filter), so what binds it is vector-instruction issue: bound = "valu_issue",
                achieved = 4 SIMD clocks x the vector wave-instructions of one launch (rocprofv3 PMC
                counts of this very command, profiles/pmc_<workload>.json; every instruction type
                costs 4.2-4.4 clocks in a mixed stream, profiles/r03_ubench_valu_rate.txt) / kernel
                duration (HIP events on the kernel's stream, measured here); peak = 256 CU x 4 SIMD x
                2.4 GHz issue clocks per second.  `count_source_stale` is true when the committed
                counts were taken from another version of the kernel.  With --no-filter the bound is
                "fp64_valu" (fp64 wave-instructions against the 78.6 TFLOP/s spec rate, 6.144e11 /s).
                The gather-accounted contract figure of SURVEY.md 8(d) and the HBM utilisation from
                the measured traffic are reported beside it.
  cpu_baseline  the C oracle (oracle/sdp_oracle.c, a port of the reference
                algorithm) timed on this box's host cores on a bounded sample of
                the same workload: single thread (faithful to the reference
                build, reference setup.py:18-22 has no -fopenmp) and all cores.
"""
import argparse
import glob
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

# ---- peaks (MI355X: /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0
N_SIMD = 256 * 4
CLOCK_SPEC_HZ = 2.4e9
# one wave64 fp64 VALU instruction occupies its SIMD for 4 clocks (16 lanes/clk:
# 78.6 TFLOP/s = 1024 SIMD x 16 lanes x 2 flop x 2.4 GHz); plain fp32 for 2
FP64_ISSUE_PEAK = N_SIMD * CLOCK_SPEC_HZ / 4        # 6.144e11 wave-instructions/s
FP32_ISSUE_PEAK = N_SIMD * CLOCK_SPEC_HZ / 2

WORKLOADS = {
    # name: (models.<builder>, kwargs, dtype, BASELINE.json configs[] index, label)
    'synth256': ('synthetic3d', dict(N=256), 'float64', 3,
                 'synthetic3d {n}^3 state x 64 controls x 32 perturbations'),
    'synth512f32': ('synthetic3d', dict(N=512), 'float32', 4,
                    'synthetic3d {n}^3 state x 64 controls x 32 perturbations, fp32 with policy indices'),
    'ar1': ('storage_ar1', dict(n_E=200, n_P=200, steps=(8. / 49, 0.1)), 'float64', 1,
            'storage-AR1 200x200 state (SoC, mismatch) x <=50 controls x 9 perturbations'),
    'searev': ('searev', dict(n_E=128, n_S=128, n_A=128, step=2.2 / 31), 'float64', 2,
               'Searev + storage 128^3 state x <=32 controls x 9 perturbations'),
    # not a BASELINE config: the benchmark problem with the control also driving x1
    # (x1' += 0.1 u) -- not storage-separable, runs the LDS-staged tile kernel
    # not a BASELINE config: the benchmark problem with the perturbation also reaching the stock,
    # x0' = (x0 + b u) - 0.07 w (the shape of the reference's inventory example): the certified filter
    # runs on the shifted lattice (csrc/sdp_colfilter_kernel.h, SDP_COL_SHIFT)
    'noisy256': ('synthetic3d', dict(N=256, stock_noise=0.07), 'float64', None,
                 'synthetic3d {n}^3 x 64 controls x 32 perturbations, perturbation also in the stock'),
    # the same sum in another nesting, x0' = x0 + (b u - 0.07 w): regrouped by the tracer (TracedModel.lead_split), no control table
    'noisy256_nested': ('synthetic3d', dict(N=256, stock_noise=0.07, nested=True), 'float64', None,
                        'synthetic3d {n}^3 x 64 controls x 32 perturbations, perturbation also in the stock: x0 + (b u - 0.07 w)'),
    # not a BASELINE config: two controlled stocks (a cascade of reservoirs) next to an exogenous inflow,
    # 128 x 128 x 64 nodes x 16 x 16 controls x 16 perturbation points -- the node-order sweep with the
    # certified filter on an array reduced over w (csrc/sdp_lead_kernel.h)
    'reservoirs': ('two_reservoirs', dict(n_a=128, n_b=128, n_y=64, n_w=16, steps=(1. / 15, 1. / 15)), 'float64', None,
                   'two reservoirs 128x128x64 state x 16x16 controls x 16 perturbations (two controlled state variables)'),
    # not a BASELINE config: ONE state variable (the reference's tutorial shape, x + u - w, at sizes worth a kernel): since
    # round 6 the filtered line kernel (csrc/sdp_line_kernel.h: the shifted lattice with the value array as its table);
    # round 5 ran the direct kernel (0.023 ms here, 6.6 ms on the fine grid below), rounds 2-4 the staged tiles (0.95 ms)
    'inventory1d': ('inventory_fine', dict(n_x=600, n_u=257, n_w=16), 'float64', None,
                    'shop inventory, one state variable: 600 nodes x 257 controls x 16 perturbations'),
    'inventory1d_fine': ('inventory_fine', dict(n_x=65536, n_u=4097, n_w=16), 'float64', None,
                         'shop inventory, one state variable: 65 536 nodes x 4097 controls x 16 perturbations'),
    'coupled256': ('synthetic3d_coupled', dict(N=256), 'float64', None,
                   'synthetic3d {n}^3 x 64 controls x 32 perturbations, control-coupled x1 (non-separable)'),
}


def algorithmic_bytes(S, cells, d, real_bytes, nu):
    """cells*2^d*T gathered + S*(T read + T write + 4*nu index)."""
    return cells * (2 ** d) * real_bytes + S * (2 * real_bytes + 4 * nu)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline_synth(solver, V0, models, U, budget_single=8.0, budget_all=12.0):
    """C oracle on bounded samples of random nodes: one thread, then all cores."""
    from oracle import c_oracle
    S = V0.size
    W = len(solver.perturb_grid[0])
    rng = np.random.default_rng(1)

    def run(n, threads):
        nodes = np.sort(rng.integers(0, S, n))
        t0 = time.perf_counter()
        c_oracle.vi_synth3d(solver.state_grid, V0, models.SYNTH_PAR, -1., 1., U,
                            solver.perturb_grid[0], solver.perturb_proba[0],
                            node_ids=nodes, n_threads=threads)
        return time.perf_counter() - t0

    def leg(threads, budget):
        run(64 * threads, threads)                      # warm-up (page in V0, spawn threads)
        n0 = 512 * threads
        t_probe = run(n0, threads)
        n = int(max(n0, min(S, n0 * budget / max(t_probe, 1e-6))))
        t = run(n, threads)
        cells_per_s = n * U * W / t
        return {'value': cells_per_s / (S * U * W), 'unit': 'sweeps/s', 'cores': threads,
                'lattice_cells_per_s': cells_per_s,
                'sample': '{} random state nodes x {} controls x {} perturbations of the same '
                          'problem ({:.1f} s of the C oracle on {} thread{}, scaled linearly to a '
                          'full sweep)'.format(n, U, W, t, threads, '' if threads == 1 else 's')}
    threads = c_oracle.max_threads()                   # before the 1-thread leg pins OpenMP to 1
    single = leg(1, budget_single)
    out = dict(single, kind='port', cpu_model=cpu_model(),
               note='single thread = the reference build (no OpenMP, reference setup.py:18-22); '
                    'the port has no Python/numpy staging and is ~3.7x faster per core than the '
                    'real reference (BASELINE.md section 5)')
    if threads > 1:
        out['all_cores'] = leg(threads, budget_all)
    return out


def cpu_baseline_numpy(solver, V0, cells_per_sweep, budget=15.0):
    """numpy restatement of the reference's per-node loop (oracle/vi_numpy.py,
    the reference's own structure: one Python iteration per node), one thread,
    on a bounded sample of random nodes."""
    from oracle import vi_numpy
    spec = vi_numpy.Spec.from_solver(solver)
    S = V0.size
    W = max(len(solver.perturb_grid[0]) if solver.perturb_grid else 1, 1)
    rng = np.random.default_rng(1)
    lo, hi, n = solver._box_table()
    per_node = np.prod(n.astype(np.int64), axis=0)

    def run(k):
        nodes = np.sort(rng.integers(0, S, k))
        t0 = time.perf_counter()
        vi_numpy.value_iteration(spec, V0, nodes=nodes)
        cells = float(per_node[nodes].sum() if per_node.size > 1 else per_node[0] * k) * W
        return time.perf_counter() - t0, cells
    run(8)
    t_probe, _ = run(64)
    k = int(max(64, min(S, 64 * budget / max(t_probe, 1e-6))))
    t, cells = run(k)
    return {'value': cells / t / cells_per_sweep, 'unit': 'sweeps/s', 'cores': 1, 'kind': 'port',
            'lattice_cells_per_s': cells / t, 'cpu_model': cpu_model(),
            'sample': '{} random state nodes of the same problem ({:.1f} s of oracle/vi_numpy.py, the '
                      'reference-shaped per-node numpy loop, one thread; scaled linearly by lattice '
                      'cells to a full sweep)'.format(k, t)}


def load_pmc(key):
    """Latest committed PMC summary of this workload's bench command
    (profiles/pmc_<key>.json, written by tools/summarize_prof.py)."""
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'pmc_{}.json'.format(key))))
    if not paths:
        return None, None
    try:
        with open(paths[-1]) as f:
            return json.load(f), os.path.relpath(paths[-1], ROOT)
    except (OSError, ValueError):
        return None, None


def load_issue_classes(key):
    """profiles/issue_classes_<key>.json (tools/issue_model.py), or None"""
    try:
        with open(os.path.join(ROOT, 'profiles', 'issue_classes_{}.json'.format(key))) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def load_clock():
    """in-kernel clock of the sweep kernel (s_memtime / s_memrealtime stamps of the
    diagnostic build, tools/clock_probe.py -> profiles/clock.json)"""
    try:
        with open(os.path.join(ROOT, 'profiles', 'clock.json')) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def build_solver(args, models, DPSolver, comm):
    name, kw, dtype, cfg, label = WORKLOADS[args.config]
    kw = dict(kw)
    if args.grid and name.startswith('synthetic3d'):
        kw['N'] = args.grid
    if args.dtype:
        dtype = args.dtype
    dtype = np.dtype(dtype)
    sysd, ref_solver = getattr(models, name)(**kw)
    solver = DPSolver(sysd, dtype=dtype, comm=comm)
    for attr in ('state_grid', '_state_grid_shape', '_state_ref_ind', '_state_ref',
                 'perturb_grid', 'perturb_proba', 'control_steps'):
        setattr(solver, attr, getattr(ref_solver, attr))
    if name.startswith('synthetic3d'):
        V0 = models.synthetic3d_V0(solver.state_grid, dtype=dtype)
    else:
        # smooth closed-form start (no RNG): squared distance to the grid centre
        parts = []
        for g in solver.state_grid_full:
            g = np.asarray(g, dtype=np.float64)
            parts.append(((g - g.mean()) / (g.max() - g.min())) ** 2)
        V0 = np.ascontiguousarray(np.broadcast_to(sum(parts), solver._state_grid_shape), dtype=dtype)
    n = kw.get('N', solver._state_grid_shape[0])
    return sysd, ref_solver, solver, V0, dtype, cfg, label.format(n=n), name


def clone_solver(DPSolver, sysd, solver, dtype, **attrs):
    s = DPSolver(sysd, dtype=dtype)
    for attr in ('state_grid', '_state_grid_shape', '_state_ref_ind', '_state_ref',
                 'perturb_grid', 'perturb_proba', 'control_steps'):
        setattr(s, attr, getattr(solver, attr))
    for k, v in attrs.items():
        setattr(s, k, v)
    return s


_REAL_STDOUT = []          # the program's real stdout while descriptor 1 is fenced (main)


def _trace(*a):
    if os.environ.get('SDP_BENCH_TRACE'):
        print('[bench r{}]'.format(os.environ.get('RANK', '0')), *a, file=sys.stderr, flush=True)


def run(args):
    """everything but the final print; returns the result dict on rank 0, else None"""
    from stodynprog_amd import models, dist, _native as nat
    from stodynprog_amd.solver import DPSolver

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dev_comm, _ = dist.from_env()
    else:
        dev_comm = None
        nat.require_gpu()
        nat.check(nat.lib().sdp_set_device(0))
        if os.environ.get('SDP_BENCH_SINGLE_RANK_COMM'):
            # test hook: drive the multi-GPU code path (phases, streams, tuning) on one GPU
            dev_comm = dist.RcclCommunicator(0, 1, dist.RcclCommunicator.new_unique_id())
    if args.gpus != world and rank == 0:
        print('warning: --gpus {} but WORLD_SIZE {}'.format(args.gpus, world), file=sys.stderr)

    if args.debug_define:
        # every solver of this run (the clones of the secondary figures too): an explicit dict, never the environment
        DPSolver.debug_defines = dict(kv.split('=', 1) for kv in args.debug_define)
    sysd, ref_solver, solver, V0, dtype, cfg, label, model_name = build_solver(args, models, DPSolver, dev_comm)
    if args.kernel:
        solver.kernel = args.kernel
    if args.no_filter:
        solver.certified_filter = False
    S = V0.size
    d = V0.ndim
    nu = len(sysd.control)
    W = len(solver.perturb_grid[0]) if solver.perturb_grid else 0
    bp = solver._box_plan()
    per_node = np.prod(bp['n'].astype(np.int64), axis=0)
    lattice = float(per_node.sum() if bp['per_node'] else per_node[0] * S)   # (node, control) pairs
    cells = lattice * max(W, 1)                                               # lattice cells per sweep
    U_max = int(bp['max_u'])

    def sync_all():
        nat.check(nat.lib().sdp_synchronize())
        if dev_comm is not None:
            dev_comm.barrier()
            nat.check(nat.lib().sdp_synchronize())

    def timed_region(prob):
        """W untimed + EXACTLY K timed sweeps from V0, bracketed by barriers and device
        synchronisations; (seconds, kernel ms), both the maximum over the ranks"""
        prob.set_value(V0)
        if args.warmup > 0:
            prob.bench_sweeps(args.warmup)
            prob.swap()
        sync_all()
        t0 = time.perf_counter()
        loop_ms, kernel_ms = prob.bench_sweeps(args.steps)       # K sweeps (+ exchanges)
        if dev_comm is not None:
            prob.complete()  # sparse exchange: the timed region ends, like the others, with J complete everywhere
        sync_all()
        elapsed = time.perf_counter() - t0
        if dev_comm is not None:
            elapsed = dev_comm.allreduce_max(elapsed)
            kernel_ms = dev_comm.allreduce_max(kernel_ms)
        return elapsed, kernel_ms

    if dev_comm is None:
        prob = solver._problem()
        assert solver.backend_info['max_controls'] == U_max
        # The procedure -- W untimed + K timed sweeps from V0 between barriers and synchronisations -- runs TWICE, back to back.
        # The first run finds a chip that has been idle while the host set the problem up: its clock ramps for ~25 ms of load
        # (tools/ramp_probe.py, profiles/r06_ramp_probe.txt: the same chain from the same V0 costs 1.27, 1.15, 1.12, 1.09,
        # 1.07 ms per sweep in its first five chunks of five sweeps the first time and 1.04, 1.05, 1.03, 1.02, 1.02 ms the
        # second time -- it is not the content).  `value` is the second run; the first stands beside it (`first_run_from_idle`).
        cold = timed_region(prob) if not args.single_run else None
        elapsed, kernel_ms = timed_region(prob)
        if rank != 0:
            return None
        out = report(args, locals())
        if cold is not None:
            out['first_run_from_idle'] = {
                'value': args.steps / cold[0], 'ms_per_step': 1e3 * cold[0] / args.steps, 'kernel_ms': cold[1] / args.steps,
                'note': 'the same W + K sweeps from the same V0, run first: the chip had been idle and its clock ramps '
                        '(tools/ramp_probe.py); `value` is the run that follows it (bench.py --single-run: this one only)'}
        return finish_single(args, locals(), out)
    return run_sharded(args, locals())


def run_sharded(args, env):
    """N > 1 (or the one-rank test hook): the RCCL exchange is tuned, TIMED and reported first;
    the other exchanges (grouped sends / receives; on request the ones over HIP IPC) are tried
    afterwards, each under a watchdog, and can only replace the RCCL result by a faster,
    bit-identical one.  An error or a rejected result of a candidate leaves the RCCL line and
    status 0; a candidate that HANGS (a rank that never answers) leaves the valid RCCL line on
    stdout and ends the process with status 3 -- the run is not reported as clean.  A failure of
    the RCCL path itself is an error."""
    import threading
    solver, dev_comm, V0, rank, nat = (env[k] for k in ('solver', 'dev_comm', 'V0', 'rank', 'nat'))
    sync_all, timed_region = env['sync_all'], env['timed_region']
    # Default: the exchanges that rest on the collective library alone -- the all-gather and the grouped sends / receives of
    # what each rank reads.  The exchanges over buffers mapped through HIP IPC ('direct', 'sparse', 'peer') have never
    # crossed a real xGMI link: they are opt-in (--exchanges rccl,sendrecv,direct,..) until a multi-GPU run has
    # reproduced the single-GPU J with them (advisor, round 4).
    exchanges = [e for e in (args.exchanges or os.environ.get('SDP_COMM_EXCHANGES', '') or 'rccl,sendrecv').split(',') if e]
    if 'rccl' not in exchanges:
        exchanges.insert(0, 'rccl')
    # phases per backup: each phase's exchange runs under the next phase's kernel.  Few phases leave
    # a long last exchange exposed, many add launches (measured on one GPU: ~0.03 ms per extra phase,
    # profiles/r03_fixed_cost_sharded.txt); sparse / peer writes need few.  `t`: tapered phases.
    # 'direct': the kernel stores J into the ranks that read it (sparse need lists where the model has them):
    # nothing to hide behind a second phase, so one launch per sweep comes first.
    PLANS = {'rccl': ((4, False), (2, False), (1, False), (8, False), (16, False), (4, True), (8, True)),
             'sendrecv': ((1, False), (2, False), (4, False)),
             'direct': ((1, False), (2, False)),
             'peer': ((2, False), (1, False), (4, False), (8, False)),
             'sparse': ((1, False), (2, False), (4, False))}
    # (tapered plans have uneven parts: padded to the longest part and gathered with ONE ncclAllGather since round 5 --
    # the grouped broadcasts of rounds 1-4, which stalled under the 8-rank stand-in, are gone: csrc/sdp_hip.hip gather_phase_of)
    if os.environ.get('SDP_COMM_PHASES'):
        forced = (int(os.environ['SDP_COMM_PHASES']), False)
        PLANS = {k: (forced,) for k in PLANS}
    fault = os.environ.get('SDP_BENCH_FAULT', '')          # test hook: '<exchange>:<raise|hang|reject>:<rank>'
    f_exch, f_kind, f_rank = (fault.split(':') + ['', '', ''])[:3]
    phase_times, notes = {}, []
    state = {'J_check': None}
    lock = threading.Lock()              # the watchdog thread reads notes / phase_times while this one fills them

    def note(text):
        with lock:
            notes.append(text)

    def key_of(exch, ph, taper):
        return '{}{}{}'.format(ph, 't' if taper else '', '' if exch == 'rccl' else '/' + exch)

    def configure(exch, ph, taper):
        solver.comm_phases, solver.comm_taper = ph, taper
        solver.comm_exchange = 'peer' if exch == 'sparse' else exch          # ('sendrecv': sparse by construction)
        solver.comm_sparse = exch in ('sparse', 'direct')
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return solver._problem()

    def release_trials():
        """Every plan is its own problem on the device (value / J / policy buffers, and for the peer exchanges the
        mappings of every peer's buffers).  A plan is released as soon as it has been timed, by all ranks at the
        same point and in two steps: every rank unmaps, then the buffers go (the library parks buffers that were
        ever exported instead of freeing them: csrc/sdp_hip.hip, park_exported -- nine problems with mappings, one
        after the other, ended in `hipIpcGetMemHandle: invalid argument` or a GPU page fault in a peer's stores
        about once in seven full-size runs while freed addresses were reused)."""
        sync_all()
        old = [solver._cache.pop(k_) for k_ in [k_ for k_ in list(solver._cache) if k_[0] == 'problem']]
        for prob in old:
            prob.unmap_peers()          # every rank lets go of the peers' buffers ...
        sync_all()
        for prob in old:
            prob.close()                # ... before anybody frees its own
        sync_all()

    def tune(exch):
        """3 sweeps per plan, the maximum over the ranks decides; the J of the first plan must equal
        the RCCL one bit for bit on every rank.  Returns (best plan, its time) or (None, reason)."""
        best = (None, 'no plan timed')
        # pre-flight: a rank that cannot even start says so BEFORE anybody enters the collective set-up
        pre_error = None
        try:
            if f_exch == exch and f_kind == 'raise' and str(rank) == f_rank:
                raise RuntimeError('injected failure (SDP_BENCH_FAULT)')
        except Exception as e:
            pre_error = '{}: {}'.format(type(e).__name__, e)
        if dev_comm.allreduce_max(1.0 if pre_error else 0.0) > 0:
            return None, pre_error or 'another rank failed before the set-up'
        for n, (ph, taper) in enumerate(PLANS[exch]):
            local_error = None
            t = 0.0
            try:
                if f_exch == exch and f_kind == 'hang' and str(rank) == f_rank:
                    time.sleep(1e6)
                _trace('plan', key_of(exch, ph, taper), 'configure')
                trial = configure(exch, ph, taper)
                _trace('plan', key_of(exch, ph, taper), 'configured')
                got = solver.backend_info.get('exchange') or 'rccl'        # (None with one rank)
                if got not in {'sparse': ('peer-sparse',), 'direct': ('direct-sparse', 'direct'), 'sendrecv': ('sendrecv',)}.get(exch, (exch,)):
                    local_error = ('not available on this node: {}'.format(getattr(trial, 'peer_failure', 'buffers not mappable'))
                                   if got == 'rccl' else 'does not apply to this kernel family')
                else:
                    trial.set_value(V0)
                    trial.bench_sweeps(2)
                    trial.swap()
                    _trace('plan', key_of(exch, ph, taper), 'warm')
                    sync_all()
                    t0 = time.perf_counter()
                    trial.bench_sweeps(3)
                    sync_all()
                    t = time.perf_counter() - t0
                    _trace('plan', key_of(exch, ph, taper), 'timed')
                    if n == 0:                                  # one result check per exchange
                        J_now = trial.get_value()
                        if f_exch == exch and f_kind == 'reject' and str(rank) == f_rank:
                            J_now = J_now + 1.0
                        if state['J_check'] is None:
                            state['J_check'] = J_now
                        elif not np.array_equal(J_now, state['J_check']):
                            local_error = 'rejected: J differs from the RCCL result'
            except Exception as e:                              # this rank's trouble: the ranks agree below
                local_error = '{}: {}'.format(type(e).__name__, e)
            trial = None
            try:
                _trace('plan', key_of(exch, ph, taper), 'release')
                release_trials()
            except Exception as e:
                local_error = local_error or '{}: {}'.format(type(e).__name__, e)
            # every rank gets here for every plan (a rank stuck inside a collective does not: watchdog)
            failed = dev_comm.allreduce_max(1.0 if local_error else 0.0) > 0
            if failed:
                return None, local_error or 'another rank failed'
            t = dev_comm.allreduce_max(t) / 3 * 1e3
            with lock:
                phase_times[key_of(exch, ph, taper)] = t
            _trace('candidate', key_of(exch, ph, taper), round(t, 3))
            if best[0] is None or t < best[1]:
                best = ((ph, taper), t)
        return best

    def measure(exch, plan):
        prob = configure(exch, *plan)
        elapsed, kernel_ms = timed_region(prob)
        e = dict(env, prob=prob, elapsed=elapsed, kernel_ms=kernel_ms, phase_times=dict(phase_times),
                 peer_note='; '.join(notes) or None)
        out = report(args, e) if rank == 0 else None
        if rank == 0:
            out = self_check(args, e, out)
        sync_all()
        return out, elapsed

    # ---- 1. the RCCL exchange: tuned, timed, reported.  A failure here is a failure of the run.
    plan, t_rccl = tune('rccl')
    if plan is None:
        raise RuntimeError('RCCL exchange failed: {}'.format(t_rccl))
    best_out, best_elapsed = measure('rccl', plan)
    best_cfg = ('rccl', plan)
    _trace('rccl timed', best_elapsed)

    # ---- 2. optional exchanges, each under a watchdog that prints the best result so far
    budget = float(os.environ.get('SDP_BENCH_OPTIONAL_TIMEOUT', '180'))
    for exch in [e for e in exchanges if e != 'rccl' and dev_comm.nranks > 1]:
        box = {'out': best_out}

        def bail(exch=exch, box=box):
            # A rank that never answers is a hung GPU process: rank 0 prints the valid RCCL line (with the hang in it, as
            # its own field), every rank says so on stderr, and every rank leaves with status 3 -- a hang must not
            # look like a successful run (rounds 3-5 left with 0).  Nothing is restarted.
            try:
                os.write(2, '[bench rank {}] watchdog: the optional {} exchange did not answer within {:.0f} s; leaving with '
                            'the RCCL result printed, exit status 3\n'.format(rank, exch, budget).encode())
            except OSError:
                pass
            try:
                if rank == 0 and box['out'] is not None:
                    o = box['out']
                    with lock:
                        o['config']['comm_exchange_note'] = '; '.join(notes + [
                            '{} exchange abandoned: no answer within {:.0f} s (a rank hung)'.format(exch, budget)])
                        o['config']['comm_phase_tuning_ms_per_sweep'] = dict(phase_times)
                    o['config']['optional_exchange_hang'] = exch
                    # (file descriptor 1 is fenced off while run() executes: write to the real one)
                    os.write(_REAL_STDOUT[0] if _REAL_STDOUT else 1, (json.dumps(o) + '\n').encode())
            finally:
                os._exit(3)
        dog = threading.Timer(budget, bail)
        dog.daemon = True
        dog.start()
        try:
            plan, t_opt = tune(exch)
            if plan is None:
                note('{} exchange not used: {}'.format(exch, t_opt))
            elif t_opt >= t_rccl:
                note('{} exchange not faster in tuning ({:.3f} vs {:.3f} ms per sweep)'.format(exch, t_opt, t_rccl))
            else:
                out, elapsed = measure(exch, plan)
                if dev_comm.allreduce_max(1.0 if (rank == 0 and out.get('sharded_matches_single_gpu') is not True) else 0.0) > 0:
                    note('{} exchange rejected: the sharded chain differs from the single-GPU one'.format(exch))
                elif elapsed < best_elapsed:
                    best_out, best_elapsed = out, elapsed
                    best_cfg = (exch, plan)
        finally:
            dog.cancel()
    # ---- 3. the work-equivalent figure: the same chain with every control the long way (the certified filter
    # off), sharded like the headline -- a curve that scales with the kernel, next to the absolute one.
    # Optional like the exchanges: under the same watchdog, and its failure costs a note, not the run.
    if not args.no_filter and not args.no_filter_check and dev_comm.nranks >= 1 and solver.backend_info.get('certified_filter'):
        box = {'out': best_out}

        def bail2(box=box):
            try:
                os.write(2, '[bench rank {}] watchdog: the long-way chain did not answer within {:.0f} s; leaving with the '
                            'headline result printed, exit status 3\n'.format(rank, budget).encode())
            except OSError:
                pass
            try:
                if rank == 0 and box['out'] is not None:
                    o = box['out']
                    with lock:
                        o['config']['comm_exchange_note'] = '; '.join(notes + ['long-way chain abandoned: no answer within {:.0f} s'.format(budget)])
                        o['config']['comm_phase_tuning_ms_per_sweep'] = dict(phase_times)
                    o['config']['optional_exchange_hang'] = 'long-way chain'
                    os.write(_REAL_STDOUT[0] if _REAL_STDOUT else 1, (json.dumps(o) + '\n').encode())
            finally:
                os._exit(3)
        dog = threading.Timer(budget, bail2)
        dog.daemon = True
        dog.start()
        try:
            err = None
            try:
                solver.certified_filter = False
                prob = configure(*((best_cfg[0],) + tuple(best_cfg[1])))
                elapsed, kernel_ms = timed_region(prob)
            except Exception as e:
                err = '{}: {}'.format(type(e).__name__, e)
            finally:
                solver.certified_filter = True
            if dev_comm.allreduce_max(1.0 if err else 0.0) > 0:
                note('long-way chain not timed: {}'.format(err or 'another rank failed'))
            elif rank == 0:
                best_out['every_control_the_long_way'] = {
                    'ms_per_step': elapsed * 1e3 / args.steps, 'sweeps_per_s': args.steps / elapsed,
                    'kernel_ms_per_sweep': kernel_ms / args.steps, 'n_gpus': dev_comm.nranks,
                    'exchange': solver.backend_info.get('exchange'),
                    'note': 'DPSolver.certified_filter = False, same sharding and exchange as the headline: the '
                            'reference\'s W x 6 operations for every control (work-equivalent curve)'}
        finally:
            dog.cancel()
    if rank == 0:
        best_out['config']['comm_exchange_note'] = '; '.join(notes) or None
        best_out['config']['comm_phase_tuning_ms_per_sweep'] = dict(phase_times)
    return best_out


def report(args, env):
    """the JSON line of one timed region (rank 0); `env`: the locals of run()"""
    (models, solver, sysd, prob, dev_comm, V0, dtype, cfg, label, model_name, world, rank, S, d, nu, W, cells,
     U_max, elapsed, kernel_ms) = (env[k] for k in (
         'models', 'solver', 'sysd', 'prob', 'dev_comm', 'V0', 'dtype', 'cfg', 'label', 'model_name', 'world',
         'rank', 'S', 'd', 'nu', 'W', 'cells', 'U_max', 'elapsed', 'kernel_ms'))
    phase_times, peer_note = env.get('phase_times'), env.get('peer_note')
    kernel_family = solver.backend_info.get('kernel')
    ms_per_step = elapsed * 1e3 / args.steps
    sweeps_per_s = args.steps / elapsed
    rb = dtype.itemsize
    # ---- roofline of the dominant kernel: the share of the sweep ONE launch chain of
    # this rank processes / its average duration (HIP events on the kernel's stream,
    # measured above inside the timed region)
    if prob.parts is not None:       # sharded: this rank's nodes, summed over its phase launches
        share = float((prob.parts[:, rank + 1] - prob.parts[:, rank]).sum()) / S
    else:
        share = float(prob.node_range[1] - prob.node_range[0]) / S
    k_ms = kernel_ms / args.steps
    k_s = k_ms * 1e-3
    kname = {'column': 'sdp_sweep_col', 'generic': 'sdp_sweep', 'staged': 'sdp_sweep_lds'}.get(
        kernel_family, 'sdp_sweep')
    filtered = bool(solver.backend_info.get('certified_filter'))
    pmc_key = '{}_{}_{}{}'.format(args.config if not args.grid else '{}{}'.format(model_name, args.grid),
                                  'f64' if rb == 8 else 'f32', kernel_family, '_filter' if filtered else '')
    pmc, pmc_path = load_pmc(pmc_key)
    # the counts belong to ONE version of the kernel: the summary records the key of the generated
    # source + kernel headers it was taken from (codegen.source_key, written by tools/summarize_prof.py)
    from stodynprog_amd import codegen
    source_key = codegen.source_key(solver._kernel_plan()['source'])
    count_stale = bool(pmc) and pmc.get('kernel_source_key') != source_key
    issue_peak = (N_SIMD * 2.4e9) if filtered else (FP64_ISSUE_PEAK if rb == 8 else FP32_ISSUE_PEAK)
    analytic = cells * 6 / 64.0                        # 6 operations per lattice cell, never fusable
    valu_all = valu_f64 = None
    if filtered and pmc and pmc.get('counters_mean_per_dispatch', {}).get('SQ_INSTS_VALU'):
        # Priced in SIMD issue cycles at 4 clocks per wave64 vector instruction WHATEVER its type:
        # in a mixed stream every vector instruction of this kernel's kind -- fp64 arithmetic, fp64
        # min / max / compare / convert, 32-bit integer and select work -- occupies its SIMD for
        # 4.2-4.4 clocks (tools/ubench/valu_rate.hip, profiles/r03_ubench_valu_rate.txt: the 2-clock
        # rate of plain 32-bit operations does not survive the mix), so 4 is the floor and the spec
        # rate of the fp64 instructions that make up most of the stream.  (Round 2 charged the
        # non-fp64 instructions 2 clocks: kept beside it as `frac_r02_accounting`.)
        cm = pmc['counters_mean_per_dispatch']
        valu_all = float(cm['SQ_INSTS_VALU'])
        valu_f64 = float(sum(cm.get('SQ_INSTS_VALU_{}_F64'.format(k), 0.0) for k in ('ADD', 'MUL', 'FMA'))) if rb == 8 else 0.0
        counted = 4.0 * valu_all
        count_source = ('{}: per {} dispatch (rocprofv3 --pmc of this command) SQ_INSTS_VALU {:.4g} (fp64 ADD/MUL/FMA '
                        '{:.4g} of them); issue cycles = 4 x all vector instructions'.format(
                            pmc_path, kname, valu_all, valu_f64))
        # Round 4: the same total priced BY INSTRUCTION CLASS where a class table of this very kernel is committed
        # (tools/issue_model.py: disassembly of the code object, hot loops weighted by their trip counts, the rest
        # at the slowest class; clocks per class from profiles/r03_ubench_valu_rate.txt)
        # (round 5, VERDICT r04: the line's `frac` is the uniform 4-clock figure; the class-priced one stays beside it)
        classes = load_issue_classes(pmc_key)
        class_cycles = None
        if classes and classes.get('kernel_source_key') == source_key and classes.get('issue_cycles_per_launch') \
                and classes.get('pmc_kernel_source_key') == source_key:
            class_cycles = float(classes['issue_cycles_per_launch'])
            count_source += ('; beside it `frac_priced_by_class` (profiles/issue_classes_{}.json: fp64 / VOP3 4.3 clk, '
                             'plain 32-bit VOP2 2.4 clk): {:.4g} issue cycles'.format(pmc_key, class_cycles))
    elif filtered:
        counted = None
        count_source = 'no PMC summary committed for this workload'
    elif pmc and pmc.get('valu_wave_instr'):
        counted = float(pmc['valu_wave_instr'])
        count_source = ('{}: {} per {} dispatch (rocprofv3 --pmc of this command)'
                        .format(pmc_path, pmc.get('valu_wave_instr_counters', 'SQ_INSTS_VALU_*'), kname))
    else:
        counted = analytic
        count_source = ('analytic lower bound: 6 {} operations per lattice cell (2 mul + add of the outer '
                        'lerp, cost add, weight mul, accumulate); no PMC summary committed for this '
                        'workload'.format('fp64' if rb == 8 else 'fp32'))
    instr_launch = counted * share if counted is not None else None
    achieved = instr_launch / k_s if counted is not None else None
    traffic = None
    traffic_source = None
    if pmc and pmc.get('hbm_bytes') and world == 1:
        traffic = float(pmc['hbm_bytes'])
        traffic_source = '{}: {}'.format(pmc_path, pmc.get('hbm_bytes_formula', '2 x FETCH_SIZE + WRITE_SIZE'))
    clock = load_clock()
    roof = {
        'bound': 'valu_issue' if filtered else ('fp64_valu' if rb == 8 else 'fp32_valu'),
        'achieved': achieved, 'peak': issue_peak, 'unit': 'SIMD issue cycles/s' if filtered else 'wave-instr/s',
        'frac': achieved / issue_peak if achieved is not None else None,
        'traffic': traffic, 'traffic_source': traffic_source,
        'kernel': kname, 'kernel_ms': k_ms,
        'valu_wave_instr_per_launch': (valu_all * share if valu_all is not None else instr_launch),
        'issue_cycles_per_launch': instr_launch if filtered else None, 'count_source': count_source,
        'count_source_stale': count_stale, 'kernel_source_key': source_key,
        'every_control_the_long_way_wave_instr_per_launch': analytic * share,
        'peak_source': ('spec: 256 CU x 4 SIMD x 2.4 GHz issue cycles per second' if filtered else
                        'spec: 256 CU x 4 SIMD x 2.4 GHz / {} clk per wave64 {} VALU instruction '
                        '(MI355X_MICROARCH.md: fp32 vector 157.3 TFLOP/s, fp64 half of it)'.format(
                            4 if rb == 8 else 2, 'fp64' if rb == 8 else 'fp32')),
        'why_not_hbm': 'certified expectation-first filter (docs/NOTEBOOK.md section 3): the expectation over w commutes '
                       'with the lerp along axis 0, so one lerp on a w-reduced table plus a proven error radius '
                       'decides every control but the near-minimal ones; only those run the reference\'s W x 6 '
                       'operations, and J / policy / index keep the same bits.  What is left is VALU issue (cell '
                       'location, cost, bounds per control) on top of the LDS table build; HBM is a few % '
                       'utilised (hbm block).  `reference_operations` prices the sweep at the 6 operations per '
                       'lattice cell the reference spends: above the fp64 peak by design' if filtered else
                       'the column kernel tabulates the inner lerps of a column in LDS, so the 2^d-vertex '
                       'gather per lattice cell never reaches L2/HBM; HBM is a few % utilised (hbm block) '
                       'and the 6 separately rounded operations per cell that bit-exactness forbids to '
                       'fuse bind the kernel' if kernel_family == 'column' else
                       'per-cell gathers: bound by vector-memory/LDS gather issue, see docs/NOTEBOOK.md section 4',
    }
    if filtered and valu_all is not None:
        roof['frac_uniform_4clk'] = 4.0 * valu_all * share / k_s / issue_peak
        roof['priced_by_class'] = False
        if class_cycles is not None:
            roof['frac_priced_by_class'] = class_cycles * share / k_s / issue_peak
        roof['valu_wave_instr_fp64_arith'] = valu_f64 * share
        roof['frac_r02_accounting'] = (4.0 * valu_f64 + 2.0 * (valu_all - valu_f64)) * share / k_s / issue_peak
        # what the filtered algorithm cannot do without, in wave64 instructions per launch: per control
        # the rounding-exact chain to its cell (u, x0', p, q0, lam0: 8), F (3) and the running minima (3);
        # per table entry the 2^(d-1)-vertex lerp nest (3 per lerp) and its share of the reduction (1); per
        # node one control evaluated with the reference's 6 operations per perturbation point
        floor_instr = (S * (cells / max(W, 1) / S) * 14.0 + S * max(W, 1) * (3.0 * (2 ** (d - 1) - 1) + 1.0)
                       + S * max(W, 1) * 6.0) / 64.0
        roof['filter_floor_wave_instr_per_launch'] = floor_instr * share
    if filtered:
        roof['reference_operations'] = {
            'wave_instr_per_launch': analytic * share, 'per_s': analytic * share / k_s,
            'over_fp64_issue_peak': analytic * share / k_s / FP64_ISSUE_PEAK,
            'note': 'NOT a utilisation: 6 separately rounded operations per lattice cell (what the reference, the '
                    'oracle and `--no-filter` execute) divided by this kernel\'s time'}
    if clock and clock.get('sweep_kernel_ghz') and achieved is not None:
        # (the filtered kernel draws less power and holds a higher clock than the long-way kernel)
        ghz = float(clock['sweep_kernel_ghz'] if filtered else
                    clock.get('long_way_kernel', {}).get('sweep_kernel_ghz', clock['sweep_kernel_ghz']))
        ghz = min(ghz, 2.4)      # (the stamps of the filtered kernel read 2.43 GHz, 1 % above the part's peak clock)
        peak_clk = N_SIMD * ghz * 1e9 / (1 if filtered else (4 if rb == 8 else 2))
        roof['measured_clock_ghz'] = ghz
        roof['peak_at_measured_clock'] = peak_clk
        roof['frac_at_measured_clock'] = achieved / peak_clk
        roof['clock_source'] = 'profiles/clock.json (s_memtime / s_memrealtime stamps, diagnostic build)'
        cpi = clock.get('fp64_clk_per_wave_instr_measured')
        if cpi and rb == 8 and not filtered:
            # the ubench does not reach 4 clk per fp64 wave-instruction either (8 waves/SIMD,
            # independent chains): what the pipe sustains at the clock the sweep kernel holds
            roof['fp64_clk_per_wave_instr_measured'] = cpi
            roof['frac_of_measured_issue_rate'] = achieved / (N_SIMD * ghz * 1e9 / cpi)
    if traffic:
        roof['hbm'] = {'achieved_GBps': traffic / k_s / 1e9, 'peak_GBps': HBM_PEAK_GBS,
                       'frac': traffic / k_s / 1e9 / HBM_PEAK_GBS,
                       'compulsory_bytes': S * (2 * rb + rb * nu + 4)}
    bytes_launch = algorithmic_bytes(S, cells, d, rb, nu) * share
    roof['gather_contract'] = {
        'algorithmic_bytes_per_launch': bytes_launch, 'GBps': bytes_launch / k_s / 1e9,
        'frac_of_hbm_peak': bytes_launch / k_s / 1e9 / HBM_PEAK_GBS,
        'note': 'SURVEY 8(d) contract figure: 2^d*T bytes per lattice cell as if every vertex came from '
                'HBM.  NOT a utilisation: the table removes those reads, so this exceeds 1 by design '
                '(north_star bar: >= 0.5)'}
    out = {
        'metric': 'vi_sweeps_per_sec', 'value': sweeps_per_s, 'unit': 'sweeps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64' if rb == 8 else 'f32', 'data': 'synthetic',
        'config': {'workload': ('{} (BASELINE.json configs[{}])'.format(label, cfg) if cfg is not None
                                else '{} (not a BASELINE config)'.format(label)),
                   'state_nodes': S, 'controls_max': U_max, 'perturbations': W,
                   'lattice_cells_per_sweep': cells,
                   'kernel_family': kernel_family, 'certified_filter': filtered,
                   'sharding': ('single GPU' if dev_comm is None else
                                'columns dealt in {} {}phases x {} ranks; {} of each phase of J under the '
                                'kernel of the next phase'.format(
                                    int(prob.parts.shape[0]), 'tapered ' if solver.comm_taper else '', world,
                                    {'peer': 'peer writes (HIP IPC, copy engines)',
                                     'peer-sparse': 'peer writes of the rows each rank reads (one slab of columns '
                                                    'per rank; {:.0%} of the array received per rank)'.format(
                                                        getattr(prob, 'need_fraction', 0.0))}.get(
                                        solver.backend_info.get('exchange'), 'RCCL all-gather'))),
                   'comm_exchange': None if dev_comm is None else solver.backend_info.get('exchange'),
                   'comm_exchange_note': peer_note,
                   'comm_phase_tuning_ms_per_sweep': phase_times,
                   'torch_imported': 'torch' in sys.modules,
                   'debug_defines': solver.backend_info.get('debug_defines')},
        'state_cells_per_sec': S * sweeps_per_s,
        'lattice_cells_per_sec': cells * sweeps_per_s,
        'roofline': roof,
    }
    return out


def finish_single(args, env, out):
    """single GPU, outside the timed region: the secondary figures and the CPU baseline"""
    (models, DPSolver, solver, ref_solver, sysd, prob, dev_comm, V0, dtype, model_name, world, S, cells, U_max) = (
        env[k] for k in ('models', 'DPSolver', 'solver', 'ref_solver', 'sysd', 'prob', 'dev_comm', 'V0', 'dtype',
                         'model_name', 'world', 'S', 'cells', 'U_max'))
    kernel_family = solver.backend_info.get('kernel')
    filtered = bool(solver.backend_info.get('certified_filter'))
    if world == 1 and filtered and not args.no_filter_check:
        # The filter's claim, checked in this very run (outside the timed region): the same chain of
        # sweeps with every control evaluated the long way -- the reference's W x 6 operations per
        # control -- gives the same J and the same policy indices, bit for bit; its time stands beside.
        try:
            J_f = prob.get_value()
            _, idx_f = prob.get_policy()
            long_way = clone_solver(DPSolver, sysd, solver, dtype, kernel=solver.kernel, certified_filter=False)
            lprob = long_way._problem()
            lprob.set_value(V0)
            if args.warmup > 0:
                lprob.bench_sweeps(args.warmup)
                lprob.swap()
            _, lk = lprob.bench_sweeps(args.steps)
            _, idx_l = lprob.get_policy()
            out['every_control_the_long_way'] = {
                'kernel_ms': lk / args.steps, 'sweeps_per_s': 1e3 * args.steps / lk,
                'J_identical': bool(np.array_equal(J_f, lprob.get_value(), equal_nan=True)),
                'policy_index_identical': bool(np.array_equal(idx_f, idx_l)),
                'note': 'DPSolver.certified_filter = False (bench.py --no-filter): same chain of {} sweeps, '
                        'compared after the last one over all {} nodes'.format(args.warmup + args.steps, S)}
            del lprob, J_f, idx_f, idx_l
        except Exception as e:
            out['every_control_the_long_way'] = {'error': repr(e)}
    if world == 1 and not args.no_other_configs:
        # The timed region is sweeps W+1 .. W+K of a chain from a closed-form start: with the driver's W = 5, K = 20 it lies in
        # the part where the cost-to-go is still rough (more blocks survive the branch and bound) and the clock still ramps
        # (profiles/r06_final40_summary.txt: 1.14, 1.27, 1.39 ms .. 1.05 ms from dispatch 27 on).  A value iteration runs for
        # hundreds of sweeps; what the later ones cost stands beside, outside the timed region and never in `value`.
        try:
            prob.swap()
            prob.bench_sweeps(30)
            prob.swap()
            _, sk = prob.bench_sweeps(50)
            out['later_sweeps_of_the_chain'] = {
                'kernel_ms': sk / 50, 'sweeps_per_s': 1e3 * 50 / sk,
                'note': 'sweeps {} .. {} of the same chain, kernel time by HIP events, outside the timed region (`value` is sweeps '
                        '{} .. {})'.format(args.warmup + args.steps + 31, args.warmup + args.steps + 80, args.warmup + 1,
                                           args.warmup + args.steps)}
        except Exception as e:
            out['later_sweeps_of_the_chain'] = {'error': repr(e)}
    if world == 1 and args.config == 'synth256' and not args.grid and not args.no_other_configs:
        # The other GPU configurations of BASELINE.json (configs[1], [2], [4]) in the same run, outside
        # the timed region: steady-state kernel time per sweep (sweeps 6..25 of a chain, HIP events),
        # so that their numbers are reproducible from the default command.  Parity cases, not bench
        # lines: `value` above is configs[3] alone.
        import copy
        others = {}
        for k_ in [k_ for k_ in solver._cache if k_[0] == 'problem']:
            solver._cache.pop(k_).close()
        for name in ('ar1', 'searev', 'synth512f32', 'noisy256', 'noisy256_nested', 'reservoirs', 'inventory1d', 'inventory1d_fine'):
            try:
                a2 = copy.copy(args)
                a2.config, a2.grid, a2.dtype = name, 0, None
                sysd2, ref2, s2, V2, dt2, cfg2, label2, _ = build_solver(a2, models, DPSolver, None)
                p2 = s2._problem()
                p2.set_value(V2)
                p2.bench_sweeps(5)
                p2.swap()
                _, k2 = p2.bench_sweeps(20)
                bp2 = s2._box_plan()
                per2 = np.prod(bp2['n'].astype(np.int64), axis=0)
                cells2 = float(per2.sum() if bp2['per_node'] else per2[0] * V2.size) * max(
                    len(s2.perturb_grid[0]) if s2.perturb_grid else 1, 1)
                others[name] = {'workload': ('{} (BASELINE.json configs[{}])'.format(label2, cfg2) if cfg2 is not None
                                             else '{} (not a BASELINE config)'.format(label2)),
                                'dtype': 'f64' if dt2.itemsize == 8 else 'f32',
                                'kernel_ms_per_sweep': k2 / 20, 'sweeps_per_s': 20e3 / k2,
                                'lattice_cells_per_sec': cells2 * 20e3 / k2,
                                'kernel_family': s2.backend_info.get('kernel'),
                                'certified_filter': bool(s2.backend_info.get('certified_filter')),
                                'filter_form': s2.backend_info.get('filter_form')}
                for k_ in [k_ for k_ in s2._cache if k_[0] == 'problem']:
                    s2._cache.pop(k_).close()
            except Exception as e:
                others[name] = {'error': repr(e)}
        try:
            # the one case the reference publishes a time for: storage-AR1 at the
            # notebook's size, 41 x 61 nodes x 4001..8001 controls x 9 perturbation points
            _, s3 = models.storage_ar1()
            p3 = s3._problem()
            p3.set_value(np.zeros(s3._state_grid_shape))
            p3.bench_sweeps(5)
            p3.swap()
            _, k3 = p3.bench_sweeps(20)
            others['ar1_reference_size'] = {
                'workload': 'storage-AR1 41x61 state x 4001..8001 controls x 9 perturbations (the reference notebook\'s own problem)',
                'dtype': 'f64', 'kernel_ms_per_sweep': k3 / 20, 'sweeps_per_s': 20e3 / k3,
                'reference_published_s_per_sweep': [4.89, 5.50],
                'reference_source': 'examples/howto storage-AR1.ipynb:596-608 ("a good Intel Core i7 laptop", one thread)',
                'kernel_family': s3.backend_info.get('kernel'),
                'certified_filter': bool(s3.backend_info.get('certified_filter'))}
            for k_ in [k_ for k_ in s3._cache if k_[0] == 'problem']:
                s3._cache.pop(k_).close()
        except Exception as e:
            others['ar1_reference_size'] = {'error': repr(e)}
        try:
            # The reference's own calling convention (stodynprog.py:466, 530-533): numpy arrays in and out on EVERY
            # call of value_iteration -- upload, sweep, downloads of J and the policy; PCIe-bound, never `value`.
            _, s4 = models.synthetic3d(N=256)
            J4, _u4 = s4.value_iteration(np.asarray(V0, dtype=np.float64), report_time=False)
            J4, _u4 = s4.value_iteration(J4, report_time=False)
            t4 = time.perf_counter()
            for _ in range(5):
                J4, _u4 = s4.value_iteration(J4, report_time=False)
            t4 = (time.perf_counter() - t4) / 5
            others['host_array_call_ms'] = t4 * 1e3
            others['host_array_calls_per_s'] = 1.0 / t4
            others['host_array_call_note'] = ('DPSolver.value_iteration(J_next) -> (J_k, pol_k) with numpy arrays every call '
                                              '(the drop-in signature), 256^3 fp64: 134 MB up, 268 MB down per call')
            for k_ in [k_ for k_ in s4._cache if k_[0] == 'problem']:
                s4._cache.pop(k_).close()
            del J4, _u4
        except Exception as e:
            others['host_array_call_ms'] = None
            others['host_array_call_note'] = repr(e)
        out['other_configs'] = dict(others, note='steady state (sweeps 6..25 of a chain from a closed-form start), kernel '
                                    'time by HIP events; outside the timed region of `value`')
    if not args.no_cpu_baseline and world == 1:
        try:
            if model_name == 'synthetic3d' and not args.config.startswith('noisy256'):     # (the C oracle's model has no noise in the stock)
                out['cpu_baseline'] = cpu_baseline_synth(ref_solver, np.asarray(V0, dtype=np.float64),
                                                         models, U_max)
            else:
                out['cpu_baseline'] = cpu_baseline_numpy(ref_solver, np.asarray(V0, dtype=np.float64), cells)
        except Exception as e:                       # the baseline must never hide the GPU number
            out['cpu_baseline'] = {'value': None, 'error': repr(e)}
    return out




def self_check(args, env, out):
    """sharded run, outside the timed region: rank 0 repeats the chain on its GPU alone"""
    (models, DPSolver, solver, ref_solver, sysd, prob, dev_comm, V0, dtype, model_name, world, S, cells, U_max) = (
        env[k] for k in ('models', 'DPSolver', 'solver', 'ref_solver', 'sysd', 'prob', 'dev_comm', 'V0', 'dtype',
                         'model_name', 'world', 'S', 'cells', 'U_max'))
    kernel_family = solver.backend_info.get('kernel')
    filtered = bool(solver.backend_info.get('certified_filter'))
    if True:
        # self-check of the sharded path (outside the timed region): rank 0 repeats
        # the same chain of sweeps on its GPU alone and compares J bit for bit
        try:
            J_sharded = prob.get_value()
            single = clone_solver(DPSolver, sysd, solver, dtype, kernel=solver.kernel,
                                  certified_filter=solver.certified_filter)
            sprob = single._problem()
            sprob.set_value(V0)
            if args.warmup > 0:
                sprob.bench_sweeps(args.warmup)
                sprob.swap()
            sprob.bench_sweeps(args.steps)
            out['sharded_matches_single_gpu'] = bool(np.array_equal(J_sharded, sprob.get_value()))
        except Exception as e:
            out['sharded_matches_single_gpu'] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='synth256', choices=sorted(WORKLOADS))
    ap.add_argument('--grid', type=int, default=0,
                    help='synthetic workloads: points per state axis (default: the config\'s)')
    ap.add_argument('--dtype', default=None, choices=['float64', 'float32'])
    ap.add_argument('--kernel', default=None, choices=['auto', 'generic', 'column', 'staged', 'lead', 'line'],
                    help='kernel family (default: auto)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--exchanges', default=None, metavar='LIST',
                    help="N > 1: the exchanges of J to time, comma-separated (default rccl,sendrecv: the collective library "
                         "alone; direct / sparse / peer write into buffers mapped through HIP IPC: opt-in)")
    ap.add_argument('--single-run', action='store_true',
                    help='one GPU: run the W + K sweeps once, from an idle chip (default: twice back to back, `value` is the '
                         'second run, the first is reported as first_run_from_idle)')
    ap.add_argument('--no-filter-check', action='store_true',
                    help='skip the untimed re-run of the chain with every control the long way and its comparison')
    ap.add_argument('--no-filter', action='store_true',
                    help='column kernel: evaluate every control with the reference\'s W x 6 operations instead '
                         'of the certified expectation-first filter (same bits either way; A/B runs)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the untimed steady-state runs of the other BASELINE configurations')
    ap.add_argument('--debug-define', action='append', default=[], metavar='NAME=VALUE',
                    help='diagnostic switch of the generated kernels (stodynprog_amd.codegen.DEBUG_NAMES; A/B runs of '
                         'tools/tune.py).  Recorded in config.debug_defines: such a line is not a product figure')
    args = ap.parse_args()
    rank = int(os.environ.get('RANK', '0'))
    # native libraries (RCCL's version banner, stdio-buffered until exit) must not
    # write into stdout: it carries exactly ONE line, the JSON result of rank 0
    from stodynprog_amd.dist import _stdout_to_stderr
    try:
        fence = _stdout_to_stderr()
        with fence:
            _REAL_STDOUT[:] = [fence._saved]
            out = run(args)
        _REAL_STDOUT[:] = []
    except BaseException as e:                        # a failed run still prints ONE JSON line
        import traceback
        traceback.print_exc()
        err = {'metric': 'vi_sweeps_per_sec', 'value': None, 'unit': 'sweeps/s',
               'n_gpus': int(os.environ.get('WORLD_SIZE', '1')), 'steps': args.steps,
               'warmup': args.warmup, 'error': '{}: {}'.format(type(e).__name__, e),
               'rank': rank}
        if rank == 0:
            print(json.dumps(err), flush=True)
        sys.exit(1)
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
