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
  roofline      the column kernel removes the 2^d-vertex gather per lattice cell
                (LDS table), so its binding resource is fp64 VALU issue:
                achieved = fp64 wave-instructions per launch (rocprofv3 PMC counts
                of this very command, profiles/pmc_<workload>.json) / kernel
                duration (HIP events on the kernel's stream, measured here);
                peak = 256 CU x 4 SIMD x 2.4 GHz / 4 clk = 6.144e11 /s (the 78.6
                TFLOP/s fp64-vector spec expressed in wave64 instructions).  The
                gather-accounted contract figure of SURVEY.md 8(d) and the HBM
                utilisation from the measured traffic are reported beside it.
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

    phase_times, peer_note = None, None
    if dev_comm is not None:
        # Untimed tuning of the comm/compute overlap: how many phases a backup is cut
        # into (each phase's all-gather runs under the next phase's kernel).  Few
        # phases leave a long last gather exposed, many add launches and small
        # collectives; the best count depends on the rank count and the fabric.
        # Every rank times the same candidates; the max over ranks decides, so all
        # ranks pick the same count.
        exchanges = [e for e in os.environ.get('SDP_COMM_EXCHANGES', 'rccl,peer,sparse').split(',') if e]
        if os.environ.get('SDP_COMM_PHASES'):
            solver.comm_phases = int(os.environ['SDP_COMM_PHASES'])
            solver.comm_exchange = 'peer' if exchanges[0] == 'sparse' else exchanges[0]
            solver.comm_sparse = exchanges[0] == 'sparse'
        else:
            # Second dimension: how the rows travel.  'rccl' = all-gather kernels of the
            # collective library; 'peer' = every rank copies its rows into the peers' buffers
            # (HIP IPC mappings, copy engines, no compute units); 'sparse' = 'peer' with one
            # slab of columns per rank and only the rows a peer reads sent to it.  'peer' /
            # 'sparse' are candidates only where every rank can map every peer, and only if
            # their J equals the RCCL one bit for bit on every rank after the same sweeps.
            phase_times, J_check = {}, None
            for exch in exchanges:
                if dev_comm.nranks == 1 and exch != 'rccl':
                    continue
                for ph, taper in ((2, False), (4, False), (8, False), (16, False), (4, True), (8, True)):
                    if exch == 'sparse' and taper:
                        continue                            # (slabs are cut evenly)
                    solver.comm_phases, solver.comm_taper = ph, taper
                    solver.comm_exchange = 'peer' if exch == 'sparse' else exch
                    solver.comm_sparse = exch == 'sparse'
                    with warnings.catch_warnings():
                        warnings.simplefilter('ignore')
                        trial = solver._problem()
                    got = solver.backend_info.get('exchange', 'rccl')
                    if got != {'sparse': 'peer-sparse'}.get(exch, exch):
                        _trace('exchange', exch, 'not available:', got, getattr(trial, 'peer_failure', ''))
                        peer_note = ('peer exchange unavailable on this node: {}'.format(
                                         getattr(trial, 'peer_failure', 'buffers not mappable'))
                                     if got == 'rccl' else 'sparse exchange does not apply to this kernel family')
                        break
                    trial.set_value(V0)
                    trial.bench_sweeps(2)
                    trial.swap()
                    sync_all()
                    t0 = time.perf_counter()
                    trial.bench_sweeps(3)
                    sync_all()
                    t = dev_comm.allreduce_max(time.perf_counter() - t0) / 3 * 1e3
                    key = '{}{}{}'.format(ph, 't' if taper else '', '' if exch == 'rccl' else '/' + exch)
                    if (ph, taper) == (4, False):           # one result check per exchange
                        J_now = trial.get_value()
                        if J_check is None:
                            J_check = J_now
                        elif dev_comm.allreduce_max(0.0 if np.array_equal(J_now, J_check) else 1.0) > 0:
                            peer_note = '{} exchange rejected: J differs from the RCCL result'.format(exch)
                            phase_times = {k: v for k, v in phase_times.items() if '/' + exch not in k}
                            break
                        del J_now
                    phase_times[key] = t
                    _trace('candidate', key, round(t, 3))
            best = min(phase_times, key=lambda k: (phase_times[k], k))
            plan, _, exch = best.partition('/')
            solver.comm_phases, solver.comm_taper = int(plan.rstrip('t')), plan.endswith('t')
            solver.comm_exchange = {'': 'rccl', 'sparse': 'peer'}.get(exch, exch)
            solver.comm_sparse = exch == 'sparse'
            del J_check

    prob = solver._problem()
    assert solver.backend_info['max_controls'] == U_max
    prob.set_value(V0)
    kernel_family = solver.backend_info.get('kernel')

    _trace('tuned', solver.comm_phases if dev_comm is not None else None)
    # warm-up sweeps (untimed), ping-pong like the timed ones
    if args.warmup > 0:
        prob.bench_sweeps(args.warmup)
        prob.swap()
    sync_all()
    t0 = time.perf_counter()
    loop_ms, kernel_ms = prob.bench_sweeps(args.steps)       # K sweeps (+ all-gathers)
    if dev_comm is not None:
        prob.complete()      # sparse exchange: the timed region ends, like the others, with J complete everywhere
    sync_all()
    elapsed = time.perf_counter() - t0
    _trace('timed region done')
    if dev_comm is not None:
        elapsed = dev_comm.allreduce_max(elapsed)
        kernel_ms = dev_comm.allreduce_max(kernel_ms)

    _trace('reduced')
    if rank != 0:
        return None
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
    issue_peak = (N_SIMD * 2.4e9) if filtered else (FP64_ISSUE_PEAK if rb == 8 else FP32_ISSUE_PEAK)
    analytic = cells * 6 / 64.0                        # 6 operations per lattice cell, never fusable
    valu_all = valu_f64 = None
    if filtered and pmc and pmc.get('counters_mean_per_dispatch', {}).get('SQ_INSTS_VALU'):
        # the filter leaves a mix of fp64 arithmetic (4 clk per wave64 instruction) and 32-bit /
        # conversion / select work (2 clk: MI355X_MICROARCH.md, cycle constants): priced in SIMD
        # issue cycles, fp64 ADD/MUL/FMA at 4 and EVERYTHING else at 2 (the 64-bit min / max /
        # compare / convert instructions among "everything else" cost 4: the figure is a floor)
        cm = pmc['counters_mean_per_dispatch']
        valu_all = float(cm['SQ_INSTS_VALU'])
        valu_f64 = float(sum(cm.get('SQ_INSTS_VALU_{}_F64'.format(k), 0.0) for k in ('ADD', 'MUL', 'FMA'))) if rb == 8 else 0.0
        counted = 4.0 * valu_f64 + 2.0 * (valu_all - valu_f64)
        count_source = ('{}: per {} dispatch (rocprofv3 --pmc of this command) SQ_INSTS_VALU {:.4g}, of which fp64 '
                        'ADD/MUL/FMA {:.4g}; issue cycles = 4 x fp64 + 2 x the rest'.format(
                            pmc_path, kname, valu_all, valu_f64))
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
        'analytic_min_wave_instr_per_launch': analytic * share,
        'peak_source': ('spec: 256 CU x 4 SIMD x 2.4 GHz issue cycles per second' if filtered else
                        'spec: 256 CU x 4 SIMD x 2.4 GHz / {} clk per wave64 {} VALU instruction '
                        '(MI355X_MICROARCH.md: fp32 vector 157.3 TFLOP/s, fp64 half of it)'.format(
                            4 if rb == 8 else 2, 'fp64' if rb == 8 else 'fp32')),
        'why_not_hbm': 'certified expectation-first filter (DESIGN.md section 3): the expectation over w commutes '
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
                       'per-cell gathers: bound by vector-memory/LDS gather issue, see DESIGN.md section 4',
    }
    if filtered and valu_all is not None:
        roof['valu_wave_instr_fp64_arith'] = valu_f64 * share
        roof['frac_if_every_instr_took_4clk'] = 4.0 * valu_all * share / k_s / issue_peak
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
                   'torch_imported': 'torch' in sys.modules},
        'state_cells_per_sec': S * sweeps_per_s,
        'lattice_cells_per_sec': cells * sweeps_per_s,
        'roofline': roof,
    }
    if world == 1 and kernel_family == 'column' and args.fused and not solver.backend_info.get('row_window') \
            and not solver.backend_info.get('table_per_control'):
        # secondary figure (never the headline `value`): the opt-in fused-arithmetic
        # variant of the same kernel (weight-scaled LDS table + FMAs; J within
        # ~1e-15 relative of the exact kernel)
        try:
            fs = clone_solver(DPSolver, sysd, solver, dtype, arithmetic='fused')
            for k_ in [k_ for k_ in solver._cache if k_[0] == 'problem']:
                solver._cache.pop(k_).close()               # free the exact problem's buffers first
            fprob = fs._problem()
            fprob.set_value(V0)
            fprob.bench_sweeps(max(args.warmup, 1))
            fprob.swap()
            _, fk = fprob.bench_sweeps(args.steps)
            out['fused_arithmetic'] = {'kernel_ms': fk / args.steps, 'sweeps_per_s': 1e3 * args.steps / fk,
                                       'note': 'SECONDARY figure, never the headline `value`: opt-in '
                                               'DPSolver.arithmetic="fused" (weight-scaled LDS table + 2 FMAs per '
                                               'cell instead of the reference\'s 6 separately rounded operations; J '
                                               'within ~1e-15 relative of the exact kernel, inside the 1e-10 parity '
                                               'bar but not the reference rounding sequence); LDS-read bound'}
        except Exception as e:
            out['fused_arithmetic'] = {'error': repr(e)}
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
    if world > 1 or os.environ.get('SDP_BENCH_SELFCHECK'):
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
    if not args.no_cpu_baseline and world == 1:
        try:
            if model_name == 'synthetic3d':
                out['cpu_baseline'] = cpu_baseline_synth(ref_solver, np.asarray(V0, dtype=np.float64),
                                                         models, U_max)
            else:
                out['cpu_baseline'] = cpu_baseline_numpy(ref_solver, np.asarray(V0, dtype=np.float64), cells)
        except Exception as e:                       # the baseline must never hide the GPU number
            out['cpu_baseline'] = {'value': None, 'error': repr(e)}
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
    ap.add_argument('--kernel', default=None, choices=['auto', 'generic', 'column', 'staged'],
                    help='kernel family (default: auto)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-filter-check', action='store_true',
                    help='skip the untimed re-run of the chain with every control the long way and its comparison')
    ap.add_argument('--no-filter', action='store_true',
                    help='column kernel: evaluate every control with the reference\'s W x 6 operations instead '
                         'of the certified expectation-first filter (same bits either way; A/B runs)')
    ap.add_argument('--fused', action='store_true',
                    help='also time the opt-in fused-arithmetic variant (secondary figure; off by default so '
                         'that a profile of the default command holds ONE flavour of sdp_sweep_col)')
    ap.add_argument('--no-fused', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    rank = int(os.environ.get('RANK', '0'))
    # native libraries (RCCL's version banner, stdio-buffered until exit) must not
    # write into stdout: it carries exactly ONE line, the JSON result of rank 0
    from stodynprog_amd.dist import _stdout_to_stderr
    try:
        with _stdout_to_stderr():
            out = run(args)
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
