#!/usr/bin/env python3
"""Run chosen cases / exchanges of the multi-rank library test (tests/test_gpu_dist.py: LIB_WORKER) by hand:
    python tools/dist_case.py <ranks> <cases, e.g. 12> <exchanges, e.g. sparse,direct> [async 0|1] [repeats]"""
import os, pathlib, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))
import test_gpu_dist as t
world, cases, exch = int(sys.argv[1]), sys.argv[2], sys.argv[3]
asynchronous = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
tmp = pathlib.Path(tempfile.mkdtemp())
mock = t._build_mock(tmp, asynchronous)
script = tmp / 'lib_worker.py'
script.write_text(t.LIB_WORKER.format(root=root))
for rep in range(reps):
    try:
        outs = t._run_ranks(t._with_hooks(tmp, script), world, dict(SDP_RCCL_LIBRARY=mock, SDP_TEST_CASES=cases, SDP_TEST_EXCHANGES=exch), timeout=int(os.environ.get('CASE_TIMEOUT', '120')))
        print('run', rep, 'ok:', outs[0].strip().splitlines()[-1], flush=True)
    except AssertionError as e:
        print('run', rep, 'FAILED:', str(e)[-1800:], flush=True)
