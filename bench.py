#!/usr/bin/env python3
"""Benchmark of the value-iteration sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is ONE Bellman sweep (DPSolver.value_iteration's device work, reference
stodynprog/stodynprog.py:466-534) over the synthetic benchmark problem of
BASELINE.json config 4: 256^3 state nodes x 64 controls x 32 perturbation
points, fp64, value array resident in HBM and ping-ponged between sweeps.
With N > 1 (one process per GPU, launched by torch.distributed.run) the outer
state axis is sharded over the ranks and the J slabs are all-gathered with RCCL
after every sweep, inside the timed region.

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` is the
gather-accounted HBM-read roofline of SURVEY.md section 8(d); `cpu_baseline`
is the C oracle (oracle/sdp_oracle.c, a port of the reference algorithm) timed
on this box's host cores on a bounded slab of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_ISSUE_PEAK = 4.85e11      # fp64 VALU wave-instructions/s, measured (profiles/ubench_fp64_rate.txt)


def algorithmic_bytes(S, U, W, d, real_bytes, nu):
    """S*U*W*2^d*T gathered + S*(T read + T write + 4*nu index)."""
    return S * U * W * (2 ** d) * real_bytes + S * (2 * real_bytes + 4 * nu)


def cpu_baseline(solver, V0, models, budget_s=18.0):
    """Time the C oracle on a bounded slab of nodes with all host cores."""
    from oracle import c_oracle
    S = V0.size
    U, W = 64, len(solver.perturb_grid[0])
    threads = c_oracle.max_threads()
    rng = np.random.default_rng(1)

    def run(n):
        nodes = np.sort(rng.integers(0, S, n))
        t0 = time.perf_counter()
        c_oracle.vi_synth3d(solver.state_grid, V0, models.SYNTH_PAR, -1., 1., U,
                            solver.perturb_grid[0], solver.perturb_proba[0],
                            node_ids=nodes, n_threads=threads)
        return time.perf_counter() - t0
    run(64 * threads)                                   # warm-up (page in V0, spawn threads)
    n0 = 512 * threads
    t_probe = run(n0)
    n = int(max(n0, min(S, n0 * budget_s / max(t_probe, 1e-6))))
    t = run(n)
    cells_per_s = n * U * W / t
    return {
        'value': cells_per_s / (S * U * W), 'unit': 'sweeps/s', 'cores': threads,
        'kind': 'port',
        'lattice_cells_per_s': cells_per_s,
        'sample': '{} random state nodes x {} controls x {} perturbations of the same '
                  '256^3 problem ({:.1f} s of OpenMP C oracle, scaled linearly to a full '
                  'sweep)'.format(n, U, W, t),
    }


def run():
    """everything but the final print; returns the result dict on rank 0, else None"""
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--grid', type=int, default=256, help='points per state axis')
    ap.add_argument('--dtype', default='float64', choices=['float64', 'float32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--fused', action='store_true',
                    help='also time the opt-in fused-arithmetic variant (secondary figure)')
    ap.add_argument('--no-fused', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    from stodynprog_amd import models, dist, _native as nat
    from stodynprog_amd.solver import DPSolver

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dev_comm, host_comm = dist.from_env()
    else:
        dev_comm = host_comm = None
        nat.require_gpu()
        nat.check(nat.lib().sdp_set_device(0))
        if os.environ.get('SDP_BENCH_SINGLE_RANK_COMM'):
            # test hook: drive the multi-GPU code path (phases, streams, tuning) on one GPU
            dev_comm = dist.RcclCommunicator(0, 1, dist.RcclCommunicator.new_unique_id())
    if args.gpus != world and rank == 0:
        print('warning: --gpus {} but WORLD_SIZE {}'.format(args.gpus, world), file=sys.stderr)

    dtype = np.dtype(args.dtype)
    N = args.grid
    sysd, ref_solver = models.synthetic3d(N=N)
    solver = DPSolver(sysd, dtype=dtype, comm=dev_comm)
    solver.state_grid = ref_solver.state_grid
    solver._state_grid_shape = ref_solver._state_grid_shape
    solver._state_ref_ind = ref_solver._state_ref_ind
    solver.perturb_grid, solver.perturb_proba = ref_solver.perturb_grid, ref_solver.perturb_proba
    solver.control_steps = ref_solver.control_steps
    V0 = models.synthetic3d_V0(solver.state_grid, dtype=dtype)
    S, U, W, d, nu = V0.size, models.SYNTH['n_u'], len(solver.perturb_grid[0]), 3, 1

    def sync_all():
        nat.check(nat.lib().sdp_synchronize())
        if dev_comm is not None:
            dev_comm.barrier()
            nat.check(nat.lib().sdp_synchronize())

    phase_times = None
    if dev_comm is not None:
        # Untimed tuning of the comm/compute overlap: how many phases a backup is cut
        # into (each phase's all-gather runs under the next phase's kernel).  Few
        # phases leave a long last gather exposed, many add launches and small
        # collectives; the best count depends on the rank count and the fabric.
        # Every rank times the same candidates; the max over ranks decides, so all
        # ranks pick the same count.
        if os.environ.get('SDP_COMM_PHASES'):
            solver.comm_phases = int(os.environ['SDP_COMM_PHASES'])
        else:
            phase_times = {}
            for ph, taper in ((2, False), (4, False), (8, False), (16, False), (4, True), (8, True)):
                solver.comm_phases, solver.comm_taper = ph, taper
                trial = solver._problem()
                trial.set_value(V0)
                trial.bench_sweeps(2)
                trial.swap()
                sync_all()
                t0 = time.perf_counter()
                trial.bench_sweeps(3)
                sync_all()
                phase_times['{}{}'.format(ph, 't' if taper else '')] = \
                    dev_comm.allreduce_max(time.perf_counter() - t0) / 3 * 1e3
            best = min(phase_times, key=lambda k: (phase_times[k], k))
            solver.comm_phases, solver.comm_taper = int(best.rstrip('t')), best.endswith('t')

    prob = solver._problem()
    assert solver.backend_info['max_controls'] == U
    prob.set_value(V0)

    # warm-up sweeps (untimed), ping-pong like the timed ones
    if args.warmup > 0:
        prob.bench_sweeps(args.warmup)
        prob.swap()
    sync_all()
    t0 = time.perf_counter()
    loop_ms, kernel_ms = prob.bench_sweeps(args.steps)       # K sweeps (+ all-gathers)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dev_comm is not None:
        elapsed = dev_comm.allreduce_max(elapsed)
        kernel_ms = dev_comm.allreduce_max(kernel_ms)

    if rank != 0:
        return None
    ms_per_step = elapsed * 1e3 / args.steps
    sweeps_per_s = args.steps / elapsed
    rb = dtype.itemsize
    # roofline of the dominant kernel (sdp_sweep): algorithmic bytes of the
    # nodes ONE launch processes / its average duration (HIP events on the
    # kernel's stream, measured above inside the timed region)
    if prob.parts is not None:       # sharded: this rank's nodes, summed over its phase launches
        nodes_per_launch = int((prob.parts[:, rank + 1] - prob.parts[:, rank]).sum())
    else:
        nodes_per_launch = prob.node_range[1] - prob.node_range[0]
    bytes_launch = algorithmic_bytes(nodes_per_launch, U, W, d, rb, nu)
    k_ms = kernel_ms / args.steps
    achieved = bytes_launch / (k_ms * 1e-3) / 1e9
    # HBM-side traffic of one launch from the committed rocprofv3 PMC passes of
    # this same command (profiles/README.md): (2 x FETCH_SIZE + WRITE_SIZE) KiB
    traffic = None
    if (N, rb, world, solver.backend_info.get('kernel')) == (256, 8, 1, 'column'):
        traffic = (2 * 989401.0 + 327680.0) * 1024
    # the column kernel's real ceiling: 6 fp64 operations per lattice cell that
    # bit-exactness does not allow to fuse, against the measured fp64 VALU
    # issue rate of the chip (profiles/ubench_fp64_rate.txt)
    fp64_wave_instr = nodes_per_launch * U * W * 6 / 64.0
    fp64_rate = fp64_wave_instr / (k_ms * 1e-3)
    out = {
        'metric': 'vi_sweeps_per_sec', 'value': sweeps_per_s, 'unit': 'sweeps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64' if rb == 8 else 'f32', 'data': 'synthetic',
        'config': {'workload': 'synthetic3d {0}^3 state x {1} controls x {2} perturbations '
                               '(BASELINE.json configs[3])'.format(N, U, W),
                   'state_nodes': S, 'controls': U, 'perturbations': W,
                   'kernel_family': solver.backend_info.get('kernel'),
                   'sharding': ('single GPU' if dev_comm is None else
                                'columns dealt in {} {}phases x {} ranks; RCCL all-gather of each phase '
                                'of J under the kernel of the next phase'.format(
                                    int(prob.parts.shape[0]), 'tapered ' if solver.comm_taper else '',
                                    world)),
                   'comm_phase_tuning_ms_per_sweep': phase_times},
        'state_cells_per_sec': S * sweeps_per_s,
        'lattice_cells_per_sec': S * U * W * sweeps_per_s,
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'traffic_source': 'profiles/r01_final_summary.txt (rocprofv3 --pmc FETCH_SIZE / '
                                       'WRITE_SIZE, FETCH x2 per the gfx950 correction)' if traffic else None,
                     'fp64_valu_wave_instr_per_s': fp64_rate, 'fp64_valu_peak_measured': FP64_ISSUE_PEAK,
                     'fp64_valu_frac': fp64_rate / FP64_ISSUE_PEAK,
                     'kernel': 'sdp_sweep_col' if solver.backend_info.get('kernel') == 'column' else 'sdp_sweep',
                     'kernel_ms': k_ms,
                     'algorithmic_bytes_per_launch': bytes_launch,
                     'note': 'gather accounting (SURVEY 8d): 2^d*T bytes per lattice cell; V '
                             'is reused from L2/Infinity Cache so frac may exceed 1'},
    }
    if world == 1 and solver.backend_info.get('kernel') == 'column' and args.fused:
        # secondary figure (never the headline `value`): the opt-in fused-arithmetic
        # variant of the same kernel (weight-scaled LDS table + FMAs; J within
        # ~1e-15 relative of the exact kernel)
        try:
            fs = DPSolver(sysd, dtype=dtype)
            fs.state_grid, fs._state_grid_shape = solver.state_grid, solver._state_grid_shape
            fs._state_ref_ind = solver._state_ref_ind
            fs.perturb_grid, fs.perturb_proba = solver.perturb_grid, solver.perturb_proba
            fs.control_steps = solver.control_steps
            fs.arithmetic = 'fused'
            for k_ in [k_ for k_ in solver._cache if k_[0] == 'problem']:
                solver._cache.pop(k_).close()               # free the exact problem's buffers first
            fprob = fs._problem()
            fprob.set_value(V0)
            fprob.bench_sweeps(max(args.warmup, 1))
            fprob.swap()
            _, fk = fprob.bench_sweeps(args.steps)
            out['fused_arithmetic'] = {'kernel_ms': fk / args.steps, 'sweeps_per_s': 1e3 * args.steps / fk,
                                       'note': 'opt-in DPSolver.arithmetic="fused"; not the reference rounding sequence'}
        except Exception as e:
            out['fused_arithmetic'] = {'error': repr(e)}
    if world > 1 or os.environ.get('SDP_BENCH_SELFCHECK'):
        # self-check of the sharded path (outside the timed region): rank 0 repeats
        # the same chain of sweeps on its GPU alone and compares J bit for bit
        try:
            J_sharded = prob.get_value()
            single = DPSolver(sysd, dtype=dtype)
            single.state_grid, single._state_grid_shape = solver.state_grid, solver._state_grid_shape
            single._state_ref_ind = solver._state_ref_ind
            single.perturb_grid, single.perturb_proba = solver.perturb_grid, solver.perturb_proba
            single.control_steps = solver.control_steps
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
            out['cpu_baseline'] = cpu_baseline(ref_solver, np.asarray(V0, dtype=np.float64), models)
        except Exception as e:                       # the baseline must never hide the GPU number
            out['cpu_baseline'] = {'value': None, 'error': repr(e)}
    return out


def main():
    # native libraries (RCCL's version banner, stdio-buffered until exit) must not
    # write into stdout: it carries exactly ONE line, the JSON result of rank 0
    from stodynprog_amd.dist import _stdout_to_stderr
    with _stdout_to_stderr():
        out = run()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
