#!/usr/bin/env python3
"""Run one GPU test function many times in ONE process (a flake hunt).   python tools/loop_test.py <module> <function> [n]"""
import importlib, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))
mod = importlib.import_module(sys.argv[1])
fn = getattr(mod, sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
bad = 0
t0 = time.time()
for i in range(n):
    try:
        fn(None)
    except AssertionError as e:
        bad += 1
        print('run', i, 'FAILED', str(e)[:200], flush=True)
print('{} failures in {} runs, {:.0f} s'.format(bad, n, time.time() - t0))
