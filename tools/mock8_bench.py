#!/usr/bin/env python3
"""`bench.py --gpus N` exactly as the driver launches it (env of torch.distributed.run), N ranks on ONE GPU through the
collective stand-in of the tests (tests/mock_rccl.cpp, asynchronous build, the TEST build of the library): the line
rank 0 prints, with its sharded-vs-single-GPU check.  For the record of configurations the suite has no time for
(BASELINE configs[4]: 512^3 fp32 on 8 GPUs).
usage: python tools/mock8_bench.py [N] [bench args ...]      (through gpurun)"""
import os, sys, pathlib, tempfile, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))
import test_gpu_dist as td

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
argv = sys.argv[2:] or ['--config', 'synth512f32', '--steps', '5', '--warmup', '2', '--no-cpu-baseline']
tmp = pathlib.Path(tempfile.mkdtemp(prefix='mock8_'))
mock = td._build_mock(tmp, asynchronous=True, slot_mb=int(os.environ.get('MOCK_SLOT_MB', '96')))
outs = td._run_ranks(td._with_hooks(tmp, os.path.join(root, 'bench.py')), world, dict(SDP_RCCL_LIBRARY=mock),
                     timeout=int(os.environ.get('MOCK_TIMEOUT', '1500')), argv=['--gpus', str(world)] + argv)
line = [l for l in outs[0].strip().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
print(json.dumps({k: d[k] for k in ('metric', 'value', 'n_gpus', 'ms_per_step', 'scaling', 'dtype', 'sharded_matches_single_gpu') if k in d}))
print(json.dumps(d['config'], indent=1))
print(json.dumps(d.get('every_control_the_long_way')))
