"""Data-parallel helpers: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" for the CPU tests and for several ranks sharing one GPU in the tests).

The path shards over graphs (rows of a minibatch are independent through encoder, heads and
per-row loss).  Each rank runs the HIP step on its share of the global minibatch with the loss
scaled by the GLOBAL row counts, then ONE all-reduce(sum) of the flat fp32 gradient buffer
(+ 4 loss scalars riding at its tail) per optimizer step; Adam is replicated.  The reference has
no distributed code at all -- this is the collective the north-star adds.

Two ways of feeding the ranks (``PPOUpdater(dp_mode=...)``):

* ``global`` -- every rank is handed the SAME replay batch (sampled once and broadcast with
  ``broadcast_batch``, or generated identically).  One global permutation per epoch (the numpy
  global RNG must be seeded identically on every rank -- verified), global minibatch k =
  ``order[k*B:(k+1)*B]`` exactly as the reference forms it
  (urban_planning/agents/urban_planning_agent.py:306-321), and every rank takes ``B / world`` of
  its rows (``split_minibatch``: contiguous slices, or dealt so that the ranks' edge counts
  balance).  A G-rank run therefore reproduces the reference's minibatch sequence and -- up to
  floating-point summation order -- its loss curve.
* ``local`` -- every rank owns a different shard (its own env workers) and shuffles it locally;
  the global minibatch is the union of the ranks' local ones.  Same estimator, different
  trajectory than a 1-GPU run.  Ranks may hold different numbers of rows: the number of
  minibatches per epoch is the minimum over the ranks (``agree_min``).
"""
import os
import zlib

import numpy as np
import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return True


class _StreamDone:
    """completion of work enqueued on ``stream``: ``wait()`` orders the current stream behind it"""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
        return True


class DistContext:
    """rank / world + the collectives the update needs.  world == 1 -> every call is a no-op."""

    def __init__(self, rank=0, world=1, group=None, active=None):
        self.rank, self.world, self.group = rank, world, group
        # collectives are issued when world > 1; UPAMD_DIST_FORCE_INIT=1 also issues them for a single rank (lets the
        # RCCL code path be exercised on a one-GPU box)
        self.active = (world > 1) if active is None else active
        self._backend = None

    @classmethod
    def from_env(cls, backend=None, device=None):
        """Initialise from torchrun's environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).

        The rendezvous store is opened first (``torch.distributed.rendezvous('env://')``: the launcher's agent store
        under ``torch.distributed.run``, a TCP store otherwise) and, for RCCL, used to prove that no two ranks of a
        host resolved to the same physical GPU BEFORE any communicator exists -- the reference's entry point pins
        every process to ``--gpu_index`` (urban_planning/train.py:50,54), so a plain ``torchrun -m urban_planning.train``
        lands all ranks on one device; ``drl_urban_planning_amd.launch`` maps LOCAL_RANK to a device first."""
        world = int(os.environ.get('WORLD_SIZE', '1'))
        forced = world == 1 and os.environ.get('UPAMD_DIST_FORCE_INIT') == '1' and 'RANK' in os.environ
        if world == 1 and not forced:
            return cls(0, 1)
        rank = int(os.environ['RANK'])
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if backend is None:
                backend = os.environ.get('UPAMD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            store, rank, world = next(iter(dist.rendezvous('env://', rank=rank, world_size=world)))
            if backend == 'nccl':
                assert_one_rank_per_device(exchange_idents(store, rank, world, device_ident(device)), rank)
            # ranks > 0 sit inside a collective (broadcast_batch / broadcast_object) for the whole of rank 0's env sampling and
            # evaluation under UPAMD_DP_SAMPLING=rank0: the process group's default watchdog (10 min under RCCL) would abort a
            # long sampling phase.  UPAMD_DIST_TIMEOUT_S (default 4 h) bounds a genuinely hung job instead.
            import datetime
            kwargs = {'timeout': datetime.timedelta(seconds=float(os.environ.get('UPAMD_DIST_TIMEOUT_S', '14400')))}
            if device is not None and backend == 'nccl':
                kwargs['device_id'] = device
            dist.init_process_group(backend=backend, store=store, rank=rank, world_size=world, **kwargs)
        return cls(rank, world, active=True)

    @property
    def backend(self):
        if self._backend is None:
            self._backend = dist.get_backend(self.group) if (self.active and dist.is_initialized()) else 'none'
        return self._backend

    def close(self):
        """Tear the process group down (end of the program)."""
        if self.active and dist.is_initialized():
            dist.destroy_process_group()
        self.active = False

    def _all_reduce(self, tensor, op):
        if not self.active:
            return tensor
        if tensor.is_cuda and self.backend == 'gloo':
            # gloo moves host memory: stage explicitly (used when several test ranks share one GPU)
            host = tensor.detach().cpu()
            dist.all_reduce(host, op=op, group=self.group)
            tensor.copy_(host)
        else:
            dist.all_reduce(tensor, op=op, group=self.group)
        return tensor

    def all_reduce_sum(self, tensor):
        return self._all_reduce(tensor, dist.ReduceOp.SUM)

    def all_reduce_sum_async(self, tensor):
        """Sum-all-reduce of ``tensor`` ordered behind the CURRENT stream (the caller has made that stream wait for whatever
        produces the tensor); returns an object whose ``wait()`` orders the stream that is current THEN behind the result.
        RCCL: the collective runs on the process group's own stream, nothing blocks the host.  gloo on a device tensor (test
        ranks sharing a GPU): staged through the host, so the HOST blocks until the producer is done -- the device work
        already enqueued on other streams carries on underneath, which is all the overlap a host-memory backend can give."""
        if not self.active:
            return _Done()
        if tensor.is_cuda and self.backend == 'gloo':
            stream = torch.cuda.current_stream(tensor.device)
            host = tensor.detach().cpu()                  # (synchronises the current stream only)
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            tensor.copy_(host)
            return _StreamDone(stream)
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_reduce_max(self, tensor):
        return self._all_reduce(tensor, dist.ReduceOp.MAX)

    def all_reduce_min(self, tensor):
        return self._all_reduce(tensor, dist.ReduceOp.MIN)

    def barrier(self):
        if self.active:
            dist.barrier(group=self.group)

    def broadcast(self, tensor, src=0):
        if self.active:
            if tensor.is_cuda and self.backend == 'gloo':
                host = tensor.detach().cpu()
                dist.broadcast(host, src=src, group=self.group)
                tensor.copy_(host)
            else:
                dist.broadcast(tensor, src=src, group=self.group)
        return tensor

    def broadcast_array(self, array, device='cpu', src=0):
        """Rank ``src``'s integer numpy array on every rank (same shape everywhere), e.g. an epoch's permutation."""
        if not self.active or self.world == 1:
            return array
        on = device if self.backend == 'nccl' else 'cpu'
        t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.int64)).to(on)
        dist.broadcast(t, src=src, group=self.group)
        return t.cpu().numpy()

    def broadcast_object(self, obj, device='cpu', src=0):
        """Rank ``src``'s picklable object on every rank (rollout logs, numpy RNG state)."""
        if not self.active or self.world == 1:
            return obj
        box = [obj if self.rank == src else None]
        dist.broadcast_object_list(box, src=src, group=self.group,
                                   device=torch.device(device) if self.backend == 'nccl' else None)
        return box[0]

    def gather_objects(self, obj):
        """Every rank's picklable object, in rank order, on every rank."""
        if not self.active or self.world == 1:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    # ---- small host-side agreements (float64 on the host for gloo, on `device` for nccl)
    def _scalar_tensor(self, values, device):
        on = device if (self.active and self.backend == 'nccl') else 'cpu'
        return torch.tensor(values, dtype=torch.float64, device=on)

    def agree_min(self, value, device='cpu'):
        """min over the ranks of an integer every rank holds (e.g. minibatches per epoch)."""
        if not self.active:
            return int(value)
        t = self._scalar_tensor([float(value)], device)
        self.all_reduce_min(t)
        return int(t.item())

    def same_everywhere(self, values, device='cpu'):
        """True iff every rank holds the same tuple of numbers (|v| < 2^53: exact in float64)."""
        if not self.active:
            return True
        v = [float(x) for x in values]
        lo = self._scalar_tensor(v, device)
        hi = lo.clone()
        self.all_reduce_min(lo)
        self.all_reduce_max(hi)
        return bool(torch.equal(lo, hi))


def device_ident(device):
    """'<host>/<physical id>' of the GPU a rank computes on (uuid, else PCI address, of the torch device)."""
    import socket
    host = socket.gethostname()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type != 'cuda':
        return '%s/cpu' % host
    props = torch.cuda.get_device_properties(device)
    uuid = getattr(props, 'uuid', None)
    if uuid is not None and str(uuid).strip('0-'):
        return '%s/%s' % (host, uuid)
    pci = [getattr(props, k, None) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')]
    if any(x is not None for x in pci):
        return '%s/pci-%s' % (host, ':'.join(str(x) for x in pci))
    vis = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('CUDA_VISIBLE_DEVICES') or ''
    return '%s/visible[%s]#%d' % (host, vis, torch.device(device).index or 0)


def exchange_idents(store, rank, world, ident):
    """Every rank's ``ident`` through the rendezvous store (no collective, no communicator needed)."""
    store.set('upamd/device/%d' % rank, ident)
    return [store.get('upamd/device/%d' % r).decode() for r in range(world)]


def assert_one_rank_per_device(idents, rank=0):
    """RCCL needs one GPU per rank: raise when two ranks resolved to the same physical device."""
    seen = {}
    for r, ident in enumerate(idents):
        if ident in seen:
            raise RuntimeError(
                'ranks %d and %d both resolved to GPU %s: RCCL needs one device per rank.  The reference pins every '
                'process to --gpu_index (urban_planning/train.py:50,54); launch through `python -m torch.distributed.run '
                '... -m drl_urban_planning_amd.launch -m urban_planning.train ...`, which maps LOCAL_RANK to a device '
                'before the reference picks one (or set UPAMD_DIST_BACKEND=gloo to let test ranks share a GPU)'
                % (seen[ident], r, ident))
        seen[ident] = r


def shard_rows(rows, rank, world):
    """This rank's contiguous share of a global minibatch's row list (len(rows) % world == 0)."""
    n = len(rows)
    if n % world != 0:
        raise ValueError('global minibatch of %d rows is not divisible by %d ranks' % (n, world))
    per = n // world
    return rows[rank * per:(rank + 1) * per]


def split_minibatch(rows, weights, rank, world, balance='edges'):
    """Rank ``rank``'s ``len(rows) / world`` rows of one global minibatch (``rows`` = state ids in the permuted
    order the reference would process them).

    ``balance='none'``: the contiguous slice ``[rank*B/G, (rank+1)*B/G)``.
    ``balance='edges'``: rows sorted by ``weights`` (live edge count; stable, descending) are dealt to the ranks in
    serpentine order, so that every rank gets the same number of rows and (nearly) the same total weight -- what
    matters for mixed HLG + DHM minibatches, where a contiguous slice can be 15 % heavier than its neighbour and every
    step waits for the slowest rank.  Within a rank the rows keep their order in ``rows``.  Every rank computes the
    same partition from the same inputs; the union over ranks is exactly ``rows``."""
    rows = np.asarray(rows)
    B = rows.size
    if B % world != 0:
        raise ValueError('global minibatch of %d rows is not divisible by %d ranks' % (B, world))
    if world == 1:
        return rows
    if balance == 'none':
        return shard_rows(rows, rank, world)
    if balance != 'edges':
        raise ValueError("balance must be 'none' or 'edges'")
    w = np.asarray(weights)
    by_weight = np.argsort(-w, kind='stable')
    pos = np.arange(B)
    rnd, slot = pos // world, pos % world
    owner = np.where(rnd % 2 == 0, slot, world - 1 - slot)
    mine = np.sort(by_weight[owner == rank])
    return rows[mine]


def order_fingerprint(order):
    """crc32 of a permutation (fits float64 exactly): ranks compare it to prove they drew the same shuffle."""
    return zlib.crc32(np.ascontiguousarray(order, dtype=np.int64).tobytes())


def batch_fingerprint(batch):
    """(T, crc32 of actions / rewards / masks / exps) of a replay batch: identical batches on every rank <=> equal."""
    crc = 0
    for a in (batch.actions, batch.rewards, batch.masks, batch.exps):
        crc = zlib.crc32(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes(), crc)
    return len(batch.states), crc


def states_fingerprint(packed):
    """crc32 of what the packer extracted from the STATES of a replay: per-state live node / edge / candidate counts,
    stage, action slot and node-mask count (meta columns 0-6; the pad sizes are left out because a compact record and
    the padded tuple it was made from differ there) and every state's numerical feature row.  Two replays with the
    same per-row scalars but different graphs (synthetic replays, equal action seeds) differ here, so 'auto' cannot
    mistake rank-local shards for one shared batch."""
    meta = np.ascontiguousarray(packed.meta[:, :7])
    crc = zlib.crc32(meta.tobytes())
    L = packed.layout
    num = packed.host_buf.numpy()[L.off_numerical:L.off_numerical + 4 * int(L.T) * int(L.numerical_dim)]
    return zlib.crc32(num.tobytes(), crc)


def global_counts(ctx, lists, device):
    """Element-wise sum over ranks of several equal-length integer lists (per-minibatch row counts);
    one tiny all-reduce per epoch, not per step."""
    if not ctx.active:
        return [list(x) for x in lists]
    t = torch.tensor([list(x) for x in lists], dtype=torch.float64, device=device)
    ctx.all_reduce_sum(t)
    return [row.tolist() for row in t]


def broadcast_batch(ctx, batch, src=0, device=None):
    """Hand rank ``src``'s replay batch to every rank (``global`` mode when only one rank sampled).

    States travel as compact wire records (``packer.compact_state``: ~2.5x smaller than the padded tuples,
    SURVEY.md section 8f row 1) concatenated into ONE uint8 tensor, i.e. one broadcast for the states and one for
    the per-row arrays; the receiving ranks get records that ``pack_replay`` consumes as zero-copy views."""
    from . import packer, synth
    if not ctx.active or ctx.world == 1:
        return batch
    use_dev = device if (ctx.backend == 'nccl' and device is not None) else 'cpu'
    if ctx.rank == src:
        recs = [s if packer.is_record(s) else packer.compact_state(s) for s in batch.states]
        sizes = np.array([r.size for r in recs], dtype=np.int64)
        blob = np.concatenate(recs) if recs else np.zeros(0, np.uint8)
        rows = np.concatenate([np.asarray(batch.actions, dtype=np.float64).reshape(len(recs), 2),
                               np.asarray(batch.masks, dtype=np.float64).reshape(-1, 1),
                               np.asarray(batch.rewards, dtype=np.float64).reshape(-1, 1),
                               np.asarray(batch.exps, dtype=np.float64).reshape(-1, 1)], axis=1)
        head = torch.tensor([len(recs), blob.size], dtype=torch.int64, device=use_dev)
    else:
        head = torch.zeros(2, dtype=torch.int64, device=use_dev)
    ctx.broadcast(head, src)
    T, nbytes = int(head[0]), int(head[1])
    if ctx.rank == src:
        t_sizes = torch.from_numpy(sizes).to(use_dev)
        t_blob = torch.from_numpy(blob).to(use_dev)
        t_rows = torch.from_numpy(rows).to(use_dev)
    else:
        t_sizes = torch.empty(T, dtype=torch.int64, device=use_dev)
        t_blob = torch.empty(nbytes, dtype=torch.uint8, device=use_dev)
        t_rows = torch.empty(T, 5, dtype=torch.float64, device=use_dev)
    for t in (t_sizes, t_blob, t_rows):
        ctx.broadcast(t, src)
    if ctx.rank == src:
        return batch
    sizes = t_sizes.cpu().numpy()
    blob = t_blob.cpu().numpy()
    rows = t_rows.cpu().numpy()
    ends = np.cumsum(sizes)
    states = [blob[e - s:e] for s, e in zip(sizes, ends)]
    return synth.Replay(states, rows[:, :2].astype(np.float32), rows[:, 2].copy(), rows[:, 3].copy(), rows[:, 4].copy())
