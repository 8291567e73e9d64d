"""Reference-side binding: what makes ``urban_planning.train`` run its PPO update on the HIP engine.

``UrbanPlanningAgent`` defines ``update_params`` in its own class body
(urban_planning/agents/urban_planning_agent.py:248-271), and a class's own attribute beats every base class -- adding a
mixin to the BASES of that class changes nothing.  The binding therefore derives a class FROM it::

    UrbanPlanningAgent = bind_reference_agent(UrbanPlanningAgent)      # last line of urban_planning_agent.py

``type(agent).update_params is HipUpdateMixin.update_params`` then holds for every agent ``urban_planning.train`` /
``urban_planning.eval`` construct, the reference class body stays untouched (its ``update_params`` remains reachable as
``super().update_params`` but is never called), and ``tests/test_reference_binding.py`` pins exactly this against the
REAL class.  ``patch_reference_module`` applies the same two changes (model factories + rebinding) to the imported
module object, which is what ``python -m drl_urban_planning_amd.launch`` does for a reference tree that is not edited
at all.

``TorchrunPolicyMixin`` is the part of the binding that only matters under ``torch.distributed.run`` (one process per
GPU; the reference is single-process, urban_planning/train.py:49-55, so all of this is policy the north-star adds):

* **who samples** -- ``UPAMD_DP_SAMPLING=rank0`` (default with ``UPAMD_DP_MODE`` auto / global): rank 0 runs the
  reference's ``Agent.sample`` (khrylib/rl/agents/agent.py:75-100) with all its env workers, the replay reaches the other
  ranks as compact records in one broadcast (``dist.broadcast_batch``) and the update runs in ``global`` mode -- the
  1-GPU / reference minibatch sequence, each rank computing ``mini_batch_size / world`` rows of every minibatch.
  ``per_rank`` (default with ``UPAMD_DP_MODE=local``): every rank samples ``num_samples / world`` steps with its own
  seed (``seed + 7919 * rank`` for numpy and torch, set once) and the update runs on the rank-local shards;
  Two constraints of the rank-0 form: ranks > 0 wait INSIDE a collective for the whole of rank 0's sampling, so the process group
  is created with a long watchdog time-out (``UPAMD_DIST_TIMEOUT_S``, 4 h by default; RCCL's own default of 10 minutes would abort a
  long sampling phase); and rank 0 forks its env workers AFTER RCCL has been initialised -- safe as long as the children never touch
  the GPU runtime or the process group (the reference's workers run CPU modules only; the ``UPAMD_ROLLOUT=server`` children talk to
  the learner through shared memory and eventfd doorbells only), which is why ``rollout_binding`` empties ``sample_modules`` in the child.
* **who evaluates** -- rank 0 runs ``eval_agent`` (:402-467) and the log object is broadcast, so ``best_rewards`` /
  ``best_plans`` / ``save_best_flag`` (:373-381) are the same on every rank;
* **who writes** -- rank 0 only: TensorBoard scalars and checkpoints (:172-194) have a single writer, the other ranks get
  a null writer, text logs of rank r > 0 go to ``<log_dir>/rank<r>``, and ``save_checkpoint`` ends in a barrier because
  ``freeze_land_use`` (:215-222) re-loads ``best.p`` on every rank.
"""
import math
import os

import numpy as np
import torch

from .agent import HipUpdateMixin
from .dist import DistContext, broadcast_batch
from .rollout_binding import CheckpointMixin, RolloutMixin

_RANK_SEED_STRIDE = 7919


class NullWriter:
    """Stands in for ``SummaryWriter`` on ranks that do not write: every method is a no-op."""

    def __getattr__(self, name):
        return lambda *a, **k: None


class TorchrunPolicyMixin:
    """Sampling / evaluation / single-writer policy of the drop-in under ``torch.distributed.run`` (module docstring).
    With one process (``WORLD_SIZE`` unset or 1) every method falls straight through to the reference's."""

    # -- context ----------------------------------------------------------------------------------------------------
    def _upamd_ctx(self):
        ctx = getattr(self, 'dist_ctx', None)
        if ctx is None:
            dev = getattr(self, 'device', None)
            dev = dev if (dev is not None and torch.device(dev).type == 'cuda') else None
            ctx = self.dist_ctx = DistContext.from_env(device=dev)
        return ctx

    @staticmethod
    def _upamd_world():
        return int(os.environ.get('WORLD_SIZE', '1'))

    @staticmethod
    def _upamd_rank():
        return int(os.environ.get('RANK', '0'))

    def _upamd_sampling(self):
        mode = os.environ.get('UPAMD_DP_SAMPLING')
        if mode is None:
            mode = 'per_rank' if os.environ.get('UPAMD_DP_MODE', 'auto') == 'local' else 'rank0'
        if mode not in ('rank0', 'per_rank'):
            raise ValueError("UPAMD_DP_SAMPLING must be 'rank0' or 'per_rank'")
        return mode

    def _upamd_device(self):
        dev = getattr(self, 'device', None)
        return dev if (dev is not None and torch.device(dev).type == 'cuda') else 'cpu'

    # -- logging: one writer ------------------------------------------------------------------------------------------
    def setup_logger(self, num_threads):
        rank = self._upamd_rank() if self._upamd_world() > 1 else 0
        if rank == 0:
            return super().setup_logger(num_threads)
        cfg = self.cfg
        saved = {k: getattr(cfg, k) for k in ('log_dir', 'tb_dir') if hasattr(cfg, k)}
        for k, v in saved.items():
            sub = os.path.join(v, 'rank%d' % rank)
            os.makedirs(sub, exist_ok=True)
            setattr(cfg, k, sub)
        try:
            out = super().setup_logger(num_threads)
        finally:
            for k, v in saved.items():
                setattr(cfg, k, v)
        tb = getattr(self, 'tb_logger', None)
        if tb is not None and hasattr(tb, 'close'):
            tb.close()
        self.tb_logger = NullWriter() if tb is not None else None
        return out

    def save_checkpoint(self, iteration):
        if self._upamd_world() == 1:
            return super().save_checkpoint(iteration)
        ctx = self._upamd_ctx()
        if ctx.rank == 0:
            super().save_checkpoint(iteration)
        ctx.barrier()               # freeze_land_use re-loads best.p on every rank (:215-222)

    # -- sampling -----------------------------------------------------------------------------------------------------
    def sample(self, num_samples, mean_action=False, nthreads=None):
        if self._upamd_world() == 1:
            return super().sample(num_samples, mean_action, nthreads)
        ctx = self._upamd_ctx()
        dev = self._upamd_device()
        if self._upamd_sampling() == 'rank0':
            batch = log = None
            if ctx.rank == 0:
                batch, log = super().sample(num_samples, mean_action, nthreads)
            batch = broadcast_batch(ctx, batch, src=0, device=dev if dev != 'cpu' else None)
            log = ctx.broadcast_object(log, dev)
            return batch, log
        if not getattr(self, '_upamd_reseeded', False):
            # train.py:55-56 seeds every process identically; a rank-local shard needs its own stream
            base = int(getattr(self.cfg, 'seed', 0) or 0)
            np.random.seed((base + _RANK_SEED_STRIDE * ctx.rank) % (2 ** 32))
            torch.manual_seed(base + _RANK_SEED_STRIDE * ctx.rank)
            self._upamd_reseeded = True
        batch, log = super().sample(int(math.ceil(num_samples / ctx.world)), mean_action, nthreads)
        logs = ctx.gather_objects(log)
        merged = self.logger_cls.merge(logs, **self.logger_kwargs)
        merged.sample_time = max(getattr(x, 'sample_time', 0.0) for x in logs)
        return batch, merged

    # -- evaluation ---------------------------------------------------------------------------------------------------
    def eval_agent(self, *args, **kwargs):
        if self._upamd_world() == 1 or not getattr(self, 'training', True):
            return super().eval_agent(*args, **kwargs)
        ctx = self._upamd_ctx()
        log = super().eval_agent(*args, **kwargs) if ctx.rank == 0 else None
        return ctx.broadcast_object(log, self._upamd_device())


def bind_reference_agent(cls):
    """``cls`` = the reference's ``UrbanPlanningAgent``; returns the class that replaces it (same name, same module,
    same constructor): ``update_params`` runs on the HIP engine, the torchrun policies sit in front of ``sample`` /
    ``eval_agent`` / ``save_checkpoint`` / ``setup_logger``, behind them the rollout side of SURVEY section 8f
    (``rollout_binding``: ``UPAMD_ROLLOUT=server`` -- batched GPU action serving + shared-memory replay arenas in
    ``sample``, evaluation behind a client; optimizer state in the checkpoint files), everything else is inherited."""
    if getattr(cls, '_upamd_bound', False):
        return cls
    if 'update_params' not in cls.__dict__ and not any('update_params' in b.__dict__ for b in cls.__mro__[1:]):
        raise TypeError('%r has no update_params to replace' % (cls,))
    bound = type(cls.__name__, (HipUpdateMixin, TorchrunPolicyMixin, RolloutMixin, CheckpointMixin, cls),
                 {'__module__': cls.__module__, '__doc__': cls.__doc__, '_upamd_bound': True,
                  '_upamd_reference_class': cls})
    bound.__qualname__ = cls.__qualname__
    return bound


def patch_reference_module(module):
    """The documented two-line patch of ``urban_planning/agents/urban_planning_agent.py``
    applied to the IMPORTED module object: model factories swapped (``setup_model`` looks them up as module globals,
    :119-126), ``UrbanPlanningAgent`` rebound.  Idempotent."""
    from . import models
    module.create_sgnn_model = models.create_sgnn_model
    module.create_mlp_model = models.create_mlp_model
    module.ActorCritic = models.ActorCritic
    module.UrbanPlanningAgent = bind_reference_agent(module.UrbanPlanningAgent)
    return module.UrbanPlanningAgent
