#!/usr/bin/env python3
"""After an edit of the kernel headers every cached code object is stale (the key of a unit is a hash of its generated
source AND the headers it includes).  This compiles the generated sources the in-tree cache already holds (the units of the
test suite and of bench.py, brought back from GPU boxes by tools/sync_kcache.sh) under their new keys, here, on the CPU, so
that the next GPU box finds them instead of compiling for eight minutes.  Stale files are removed by
__graft_entry__.build().      usage: python tools/rekey_kcache.py [workers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from stodynprog_amd import _native as nat, codegen

workers = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, (os.cpu_count() or 2) - 1)
todo, seen = [], set()
for fn in sorted(os.listdir(nat.KCACHE)):
    if not fn.endswith('.hip'):
        continue
    with open(os.path.join(nat.KCACHE, fn)) as f:
        src = f.read()
    # round 6: the opt-in fused arithmetic went, and with it a line of every column unit
    if '#define SDP_COL_FUSED 1' in src:
        continue
    src = src.replace('#define SDP_COL_FUSED 0\n', '')
    key = codegen.source_key(src)
    if key in seen or os.path.exists(os.path.join(nat.KCACHE, key + '.hsaco')):
        continue
    seen.add(key)
    todo.append(src)
print('{} units to compile with {} workers'.format(len(todo), workers), flush=True)
failed = 0
def one(src):
    try:
        nat.compile_model(src)
        return None
    except Exception as e:
        return str(e)[:300]
with ThreadPoolExecutor(workers) as ex:
    for k, err in enumerate(ex.map(one, todo)):
        if err:
            failed += 1
            print('FAILED:', err, flush=True)
        if k % 50 == 49:
            print(k + 1, 'done', flush=True)
print('done; {} failed'.format(failed))
