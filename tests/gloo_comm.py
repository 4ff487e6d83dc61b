"""Host communicator over torch.distributed / gloo for the CPU tests of the slab logic (world_size 2) and the
two-process GPU test: TEST scaffolding -- the product (stodynprog_amd) never imports torch.  Implements the host
side of stodynprog_amd.dist.Communicator."""
import numpy as np

from stodynprog_amd.dist import Communicator


class GlooCommunicator(Communicator):
    """Host-side collectives over an initialised torch.distributed group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)

    def all_gather_slabs(self, J, bounds):
        """Fill the flat view of `J` (complete array, own slab valid) with the
        slabs of every rank, in place.  Slabs may have different lengths."""
        import torch
        flat = J.reshape(-1)
        mine = torch.from_numpy(np.ascontiguousarray(flat[bounds[self.rank]:bounds[self.rank + 1]]))
        for r in range(self.nranks):
            n = int(bounds[r + 1] - bounds[r])
            buf = mine.clone() if r == self.rank else torch.empty(n, dtype=mine.dtype)
            if n:
                self._dist.broadcast(buf, src=self._global_rank(r), group=self.group)
                flat[bounds[r]:bounds[r + 1]] = buf.numpy()
        return J

    def _global_rank(self, r):
        if self.group is None:
            return r
        return self._dist.get_global_rank(self.group, r)

    def broadcast_bytes(self, payload, src=0):
        obj = [payload if self.rank == src else None]
        self._dist.broadcast_object_list(obj, src=self._global_rank(src), group=self.group)
        return obj[0]

    def allreduce_max(self, value):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        return float(t[0])

    def barrier(self):
        self._dist.barrier(group=self.group)
