"""Which HIP / HSA / RCCL shared objects end up in the process, and does a
1-rank RCCL communicator initialise, depending on whether torch is imported
before or after libsdp_hip.so?  usage: diag_libs.py torch-first|lib-first"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]

def maps():
    seen = []
    for line in open('/proc/self/maps'):
        p = line.split()[-1]
        if any(k in p for k in ('amdhip64', 'hsa-runtime', 'rccl')) and p not in seen:
            seen.append(p)
    return seen

if order == 'torch-first':
    import torch, torch.distributed
from stodynprog_amd import _native as nat
nat.lib()
if order == 'lib-first':
    import torch, torch.distributed
print(order, 'before rccl:', maps(), flush=True)
from stodynprog_amd.dist import RcclCommunicator
try:
    nat.check(nat.lib().sdp_set_device(0))
    uid = RcclCommunicator.new_unique_id()
    c = RcclCommunicator(0, 1, uid)
    c.barrier()
    print(order, 'RCCL 1-rank init OK', flush=True)
except Exception as e:
    print(order, 'RCCL init FAILED:', e, flush=True)
print(order, 'after rccl:', maps(), flush=True)
