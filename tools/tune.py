#!/usr/bin/env python3
"""A/B timing of column-kernel build knobs on the GPU box (tuning aid).
usage: python tools/tune.py "K1=v K2=v" "K1=v ..." ...   (one bench run per argument; the knobs are the
diagnostic switches of stodynprog_amd.codegen.DEBUG_NAMES, handed to bench.py as --debug-define);
extra bench.py arguments from $TUNE_BENCH_ARGS (e.g. "--dtype float32 --grid 512")"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from stodynprog_amd.codegen import DEBUG_NAMES
for cfg in sys.argv[1:]:
    defs, env = [], dict(os.environ)
    for kv in cfg.split():
        if kv.split('=', 1)[0] in DEBUG_NAMES:
            defs += ['--debug-define', kv]          # a switch of the generated kernels: explicit
        else:
            env[kv.split('=', 1)[0]] = kv.split('=', 1)[1]     # a hook of bench.py itself (SDP_COMM_PHASES, ..)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '10', '--warmup', '3',
                        '--no-cpu-baseline', '--no-other-configs'] + defs +
                       os.environ.get('TUNE_BENCH_ARGS', '').split(),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        print('{:70s} {:8.3f} ms/sweep  kernel {:8.3f} ms'.format(cfg or '(default)', d['ms_per_step'], d['roofline']['kernel_ms']), flush=True)
    except Exception:
        print(cfg, 'FAILED', r.stderr.decode()[-600:], flush=True)
