"""Data-parallel helpers: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" for the CPU tests).

The path shards over graphs (rows of a minibatch are independent through encoder, heads and
per-row loss).  Each rank runs the HIP step on its share of the global minibatch with the loss
scaled by the GLOBAL row counts, then ONE all-reduce(sum) of the flat fp32 gradient buffer
(+ 4 loss scalars riding at its tail) per optimizer step; Adam is replicated.  The reference has
no distributed code at all -- this is the collective the north-star adds.
"""
import os

import torch
import torch.distributed as dist


class DistContext:
    """rank / world + the collectives the update needs.  world == 1 -> every call is a no-op."""

    def __init__(self, rank=0, world=1, group=None, active=None):
        self.rank, self.world, self.group = rank, world, group
        # collectives are issued when world > 1; UPAMD_DIST_FORCE_INIT=1 also issues them for a single rank (lets the
        # RCCL code path be exercised on a one-GPU box)
        self.active = (world > 1) if active is None else active

    @classmethod
    def from_env(cls, backend=None, device=None):
        """Initialise from torchrun's environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
        world = int(os.environ.get('WORLD_SIZE', '1'))
        forced = world == 1 and os.environ.get('UPAMD_DIST_FORCE_INIT') == '1' and 'RANK' in os.environ
        if world == 1 and not forced:
            return cls(0, 1)
        rank = int(os.environ['RANK'])
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            kwargs = {}
            if device is not None and backend == 'nccl':
                kwargs['device_id'] = device
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
        return cls(rank, world, active=True)

    def close(self):
        """Tear the process group down (end of the program)."""
        if self.active and dist.is_initialized():
            dist.destroy_process_group()
        self.active = False

    def all_reduce_sum(self, tensor):
        if self.active:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    def all_reduce_max(self, tensor):
        if self.active:
            dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=self.group)
        return tensor

    def barrier(self):
        if self.active:
            dist.barrier(group=self.group)

    def broadcast(self, tensor, src=0):
        if self.active:
            dist.broadcast(tensor, src=src, group=self.group)
        return tensor


def shard_rows(rows, rank, world):
    """This rank's contiguous share of a global minibatch's row list (len(rows) % world == 0)."""
    n = len(rows)
    if n % world != 0:
        raise ValueError('global minibatch of %d rows is not divisible by %d ranks' % (n, world))
    per = n // world
    return rows[rank * per:(rank + 1) * per]


def global_counts(ctx, lists, device):
    """Element-wise sum over ranks of several equal-length integer lists (per-minibatch row counts);
    one tiny all-reduce per epoch, not per step."""
    if not ctx.active:
        return [list(x) for x in lists]
    t = torch.tensor([list(x) for x in lists], dtype=torch.float64, device=device)
    ctx.all_reduce_sum(t)
    return [row.tolist() for row in t]
