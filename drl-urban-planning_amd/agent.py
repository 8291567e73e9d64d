"""PPO update surface: ``update_params(batch, iteration) -> seconds`` with the semantics of
``UrbanPlanningAgent.update_params`` / ``update_policy``
(urban_planning/agents/urban_planning_agent.py:248-361) running on the HIP engine.

What is reproduced exactly:
  * value + old-log-prob pre-pass (one fused sweep instead of the reference's two), GAE without
    normalisation (khrylib/rl/core/common.py:5-26);
  * ``num_optim_epoch`` epochs, per epoch one ``np.random.shuffle`` on the numpy GLOBAL RNG whose
    permutations compose across epochs (:306-312), optional regrouping by stage (:314-319),
    ``floor(T / mini_batch_size)`` minibatches with the tail dropped (:321);
  * value loss over all rows, surrogate + entropy over rows with ``exps != 0`` (:326-333, 363-371);
  * gradient clipping that is effective on the first optimizer step of the process only, applied
    to the policy set and then the value set (:46 + khrylib/rl/agents/agent_ppo.py:43-46);
  * torch.optim.Adam semantics incl. coupled weight decay, and "a head that saw no rows has no
    gradient, so Adam skips it" (torch >= 2.0 zero_grad(set_to_none=True));
  * ``loss_iter`` + the per-minibatch / per-epoch / per-iteration TensorBoard scalars (:338-361).

What differs by design: the shared encoder is evaluated once per step (same gradients); the
replay is packed and uploaded once per iteration; the four loss scalars stay on the GPU until the
iteration ends (no host sync inside the minibatch loop); optimizer state lives in flat device
buffers owned by the updater (the reference never checkpoints optimizer state either).

Data parallelism (the reference has none): see ``dist.py``.  ``mini_batch_size`` keeps
the reference's meaning in ``dp_mode='global'`` -- the GLOBAL minibatch, of which every rank processes
``mini_batch_size / world`` rows -- and is the per-rank minibatch in ``dp_mode='local'``.

``zero_grad`` semantics: under torch >= 2.0 ``zero_grad()`` sets the gradients of a head that saw no row to ``None``
and Adam skips that head.  The reference pins torch <= 1.13 (requirements.txt:3), whose ``zero_grad()`` zero-fills:
there, a head without rows still takes an Adam step with g = 0 once it has had a gradient (step count + moment
decay + a momentum move).  ``legacy_zero_grad=True`` reproduces the pinned behaviour, ``False`` the newer one, and the
default ``None`` follows the torch that is INSTALLED -- i.e. what the reference's own ``self.optimizer.zero_grad()``
would do in this process (both are pinned by goldens generated with the real reference: tests/golden/case_s.npz).
"""
import inspect
import math
import os
import time

import numpy as np
import torch

from . import native, packer
from .dist import DistContext, batch_fingerprint, global_counts, split_minibatch, states_fingerprint
from .models import backend_of


def torch_zero_grad_zero_fills():
    """True when the installed torch's ``Optimizer.zero_grad()`` zero-fills gradients (torch < 2.0) instead of setting
    them to None."""
    try:
        return inspect.signature(torch.optim.Optimizer.zero_grad).parameters['set_to_none'].default is False
    except (KeyError, ValueError, TypeError):
        return False


class IterationState:
    """Everything one PPO iteration keeps resident in HBM: packed replay + per-row RL tensors."""

    def __init__(self, packed, T, exps_np, exps, values, old_logp, adv, ret):
        self.packed, self.T = packed, T
        self.exps_np, self.exps = exps_np, exps
        self.values, self.old_logp, self.adv, self.ret = values, old_logp, adv, ret
        self.order = np.arange(T)


class Epoch:
    """One epoch's minibatch schedule (device index arrays uploaded once) + global row counts."""

    def __init__(self, sched, order_dev, nb, rows_glob, ind_glob, land_glob, road_glob):
        self.sched, self.order_dev, self.nb = sched, order_dev, nb
        self.rows_glob, self.ind_glob, self.land_glob, self.road_glob = rows_glob, ind_glob, land_glob, road_glob


class PPOUpdater:
    """Owns the flat parameter / Adam buffers of one (policy_net, value_net) pair on one GPU."""

    def __init__(self, policy_net, value_net, lr=4e-4, eps=1e-5, weight_decay=0.0, betas=(0.9, 0.999), gamma=1.0,
                 tau=0.0, clip_epsilon=0.2, value_pred_coef=0.5, entropy_coef=0.01, num_optim_epoch=4,
                 mini_batch_size=256, batch_stage=False, max_grad_norm=1.0, dist_ctx=None, pack_threads=0,
                 dp_mode='auto', dp_balance='edges', legacy_zero_grad=None):
        self.policy_net, self.value_net = policy_net, value_net
        self.backend = backend_of(policy_net)
        self.lr, self.eps, self.weight_decay, self.betas = lr, eps, weight_decay, betas
        self.gamma, self.tau = gamma, tau
        self.clip_epsilon = clip_epsilon
        self.value_pred_coef, self.entropy_coef = value_pred_coef, entropy_coef
        self.num_optim_epoch, self.mini_batch_size = num_optim_epoch, mini_batch_size
        self.batch_stage = batch_stage
        self.max_grad_norm = max_grad_norm
        self.dist = dist_ctx or DistContext()
        if dp_mode not in ('auto', 'global', 'local'):
            raise ValueError("dp_mode must be 'auto', 'global' or 'local'")
        self.dp_mode, self.dp_balance = dp_mode, dp_balance
        self._mode = 'single'                 # resolved per update_params: 'single' | 'global' | 'local'
        self.legacy_zero_grad = torch_zero_grad_zero_fills() if legacy_zero_grad is None else bool(legacy_zero_grad)
        self._group_seen = [False, False, False]
        self.pack_threads = pack_threads
        self.loss_iter = 0
        self.clip_pending = True              # the generator lists are live until the first call
        self.group_steps = [0, 0, 0]          # Adam step counts: encoder+value / land head / road head
        self.flat = self.m = self.v = self.grads = None
        self.engine = None
        self.fused_small = True               # small models: forward + loss + backward as one launch (csrc/tiny.hip)
        self.last_losses = None               # np [steps, 4] of the last update_params call
        self.last_timing = {}
        self.collective_events = None         # a list: record a HIP-event pair around every step's gradient all-reduce (bench.py)
        # data parallelism: all-reduce the gradient buffer bucket by bucket on a communication stream while the backward of the
        # layers below is still running (engine.grad_buckets); False = ONE collective behind the whole backward.  Same bits.
        # (UPAMD_GRAD_BUCKETS: 1 default | 0 single collective | force = also with ONE rank in an initialised process group, which is
        # how the RCCL route -- async work objects on the process group's own stream -- is exercised on a one-GPU box)
        self.bucketed_allreduce = os.environ.get('UPAMD_GRAD_BUCKETS', '1') != '0'
        self._buckets_forced = os.environ.get('UPAMD_GRAD_BUCKETS') == 'force'
        # UPAMD_GRAD_BUCKETS=check (any world size, also without a process group): a FINALITY check of the engine's bucket events -- the
        # communication stream snapshots every range the moment its event has fired, and behind the whole backward the snapshots
        # are compared with the final gradient bit for bit (bucket_check_mismatches counts differing floats; tests assert 0).  A range
        # the backward marks final too early is invisible to a one-rank RCCL group (its sum is the identity) and caught under gloo
        # only by chance of timing; this sees it on one GPU.
        self._buckets_check = os.environ.get('UPAMD_GRAD_BUCKETS') == 'check'
        self._bucket_check_selftest = False   # tests only: every snapshot gets one float changed -- the comparison must count exactly those
        self.bucket_check_mismatches = None   # device int64 scalar (check mode)
        self.bucket_checks = 0                # ranges compared so far (check mode)
        # lab knobs of the bucketed route, read once (profiles/r05_lab_bucket_overhead.md)
        self._bucket_lab = os.environ.get('UPAMD_BUCKET_LAB', '')
        self._bucket_tail_main = os.environ.get('UPAMD_BUCKET_TAIL', 'main') == 'main'
        self._comm_priority = int(os.environ.get('UPAMD_COMM_PRIORITY', '0'))
        self.last_buckets = None              # [(begin, end)] of the last bucketed step (tests, bench.py)
        # prepare(): the replay is packed / uploaded / swept in about this many chunks (1 = no pipeline)
        # 0 = auto: 8 where the pre-pass is worth hiding (gcn_node_dim > 32), 2 for the small models (round 6: with the fill three times
        # faster the H2D of one half hides under the fill of the other: 49.3 vs 52.5 ms per call at hlg_ref, 42.3 vs 44.5 from records,
        # profiles/r06_bench_hlg_ref_chunks{1,2,4}.json; round 5's 8 chunks cost 8 % there, profiles/r05_lab_prepare.md)
        self.pipeline_chunks = int(os.environ.get('UPAMD_PREPARE_CHUNKS', '0'))
        self._copy_stream = None
        self.pending_state = None             # a checkpoint's optimizer state waiting for the GPU buffers (load_state_dict at the next update_params)
        self._comm = None
        # UPAMD_LANES=2 (opt-in, round 6): every minibatch is dealt into two edge-balanced halves (dist.split_minibatch, the deal a
        # 2-rank data-parallel run uses) whose forward + loss + backward run on TWO streams with their own workspaces and gradient
        # buffers; the halves' gradients are added before clip / Adam.  Graphs are independent, so one half's GEMMs (matrix pipe) can
        # meet the other half's message passing (vector ALU) wherever the hardware dispatcher lets them share the chip -- the
        # stream-level form of "software-pipeline two half-minibatches" (DESIGN section 5a has the measurement).  Same arithmetic as
        # a 2-rank 'global' run: deterministic, inside every tolerance, NOT the bits of the one-lane step (another summation order).
        self.lanes = int(os.environ.get('UPAMD_LANES', '1'))
        if self.lanes not in (1, 2):
            raise ValueError('UPAMD_LANES must be 1 or 2')
        self._lane_streams = self._lane_bufs = None

    # ------------------------------------------------------------------ buffers
    def _device(self):
        dev = next(self.policy_net.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('update_params runs on the HIP engine and needs the networks on a GPU device '
                               '(they are on %s); there is no CPU fallback for the update path' % dev)
        return dev

    def attach(self):
        """Bind to the engine of the networks' device and (re)load the flat parameter buffer from them."""
        dev = self._device()
        engine = self.backend.engine(dev)
        if self.flat is None or self.flat.device != engine.device or self.flat.numel() != engine.n_floats:
            self.flat = engine.new_flat()
            self.m = engine.new_flat()
            self.v = engine.new_flat()
            # the gradient buffer carries the 4 loss scalars at its tail: one all-reduce moves both
            self.grads = torch.zeros(engine.n_floats + 4, dtype=torch.float32, device=engine.device)
            self.scratch = torch.zeros(4096, dtype=torch.float32, device=engine.device)
        self.engine = engine
        self._named = self.backend.named_params()
        engine.flatten(self._named, out=self.flat)
        return engine

    def _ensure_rowbufs(self, rows):
        if getattr(self, '_rowbuf_B', None) != rows:
            dev = self.engine.device
            self._rows = [torch.empty(rows, device=dev) for _ in range(6)]   # value, logp, ent, dvalue, dlogp, dent
            self._rowbuf_B = rows

    # ------------------------------------------------------------------ data-parallel mode
    def _resolve_mode(self, batch, packed):
        """'single' (no collectives) | 'global' (same batch everywhere, slices of one global permutation) |
        'local' (own shard per rank).  'auto' picks 'global' exactly when every rank was handed the same batch."""
        d = self.dist
        if not d.active:
            return 'single'
        if d.world == 1 or self.dp_mode == 'local':
            return 'local'
        same = d.same_everywhere(batch_fingerprint(batch) + (states_fingerprint(packed),), self.engine.device)
        if self.dp_mode == 'global' and not same:
            raise RuntimeError("dp_mode='global' needs the same replay batch on every rank (sample once and "
                               "dist.broadcast_batch it, or use dp_mode='local' for per-rank shards)")
        return 'global' if same else 'local'

    def local_rows(self):
        """Rows THIS rank processes per optimizer step."""
        if self._mode == 'global':
            if self.mini_batch_size % self.dist.world:
                raise ValueError('mini_batch_size %d is not divisible by %d ranks' % (self.mini_batch_size, self.dist.world))
            return self.mini_batch_size // self.dist.world
        return self.mini_batch_size

    def global_rows(self):
        """Rows of one GLOBAL minibatch (what one optimizer step consumes over all ranks)."""
        return self.mini_batch_size if self._mode != 'local' else self.mini_batch_size * self.dist.world

    def detach(self):
        """Write the flat parameters back into the nn.Module parameters."""
        self.engine.unflatten(self.flat, self._named)

    # ------------------------------------------------------------------ optimizer checkpoint
    def state_dict(self):
        """Optimizer state the reference's checkpoint omits (urban_planning_agent.py:172-194 saves only the networks):
        Adam moments per named parameter (CPU tensors), the per-group step counts, the loss-curve position and whether
        the first-step clipping quirk is still pending.  `load_state_dict` restores it, so a resumed run continues the
        same trajectory instead of restarting Adam."""
        out = {'group_steps': list(self.group_steps), 'loss_iter': int(self.loss_iter),
               'clip_pending': bool(self.clip_pending), 'group_seen': list(self._group_seen), 'exp_avg': {},
               'exp_avg_sq': {}}
        if self.m is not None:
            eng = self.engine if self.engine is not None else self.backend.engine(self._device())
            for name, off, rows, cols, _ in eng.table:
                shape = tuple(self.backend.named_params()[name].shape)
                out['exp_avg'][name] = self.m[off:off + rows * cols].detach().cpu().clone().view(shape)
                out['exp_avg_sq'][name] = self.v[off:off + rows * cols].detach().cpu().clone().view(shape)
        return out

    def load_state_dict(self, state):
        self.group_steps = [int(x) for x in state['group_steps']]
        self.loss_iter = int(state['loss_iter'])
        self.clip_pending = bool(state['clip_pending'])
        self._group_seen = [bool(x) for x in state.get('group_seen', [n > 0 for n in self.group_steps])]
        if state['exp_avg']:
            eng = self.attach()
            for name, off, rows, cols, _ in eng.table:
                self.m[off:off + rows * cols].copy_(state['exp_avg'][name].reshape(-1))
                self.v[off:off + rows * cols].copy_(state['exp_avg_sq'][name].reshape(-1))

    @staticmethod
    def _to_f32(x, device):
        return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(device)

    # ------------------------------------------------------------------ iteration set-up
    def prepare(self, batch, exact_plan=False):
        """Pack + upload the replay, value / old-log-prob pre-pass (:256-264, :283-292), GAE (:267) -- as a three-stage
        pipeline over chunks of the replay: the host threads pack chunk k + 1 (``PackedReplay.fill``) while chunk k travels
        to HBM on a copy stream (page-locked source, one contiguous range per section) and the no-grad forward of chunk
        k - 1 runs on the caller's stream; GAE follows the last chunk.  ``pipeline_chunks = 1`` is the unpipelined form (pack
        everything, one upload, then the sweep): the same bytes in HBM, the same launches in the same order, so values, log-probs
        and advantages are bit-identical either way (tests/test_gpu_update_branches.py)."""
        if not hasattr(self, '_pack_cache'):
            self._pack_cache = {}
        try:
            return self._prepare(batch, exact_plan)
        except packer.NeedsExactPlan:           # (states no extractor emits: a live edge on a node outside the masks)
            torch.cuda.synchronize(self.engine.device)
            return self._prepare(batch, True)

    def _prepare(self, batch, exact_plan):
        engine, dev = self.engine, self.engine.device
        agent = self.policy_net.agent
        T = len(batch.states)
        packed = packer.plan_replay(batch.states, np.asarray(batch.actions), agent.node_dim,
                                    agent.numerical_feature_size, n_threads=self.pack_threads,
                                    reuse=self._pack_cache, exact=exact_plan,
                                    mlp_fields=int(engine.desc.encoder) == native.ENCODER_MLP)
        # the small per-row arrays first, through the recycled page-locked ring: a pageable upload later on would make the host
        # wait for every kernel queued before it
        exps_np = np.asarray(batch.exps, dtype=np.float32)
        rewards, _ = packer.upload_pinned(np.asarray(batch.rewards, dtype=np.float32), dev)
        masks, _ = packer.upload_pinned(np.asarray(batch.masks, dtype=np.float32), dev)
        exps, _ = packer.upload_pinned(exps_np, dev)
        if self.dist.active and self.dist.world > 1 and self.dp_mode != 'local':
            packed.fill(0, T)        # 'auto' / 'global' compare a fingerprint of the PACKED replay across the ranks first
        self._mode = self._resolve_mode(batch, packed)
        R = self.local_rows()
        self._ensure_rowbufs(R)
        shared = self._mode == 'global'
        # 'global': every rank holds all T rows, so the no-grad sweep is shared out chunk by chunk and the two
        # T-float results are summed over the ranks (rows a rank did not compute are zero)
        values = torch.zeros(T, device=dev) if shared else torch.empty(T, device=dev)
        logp = torch.zeros(T, device=dev) if shared else torch.empty(T, device=dev)
        ent = torch.empty(T, device=dev)
        row_lists = [np.arange(i, min(i + R, T)) for i in range(0, T, R)]
        sched = packer.Schedule(packed, row_lists, dev)       # (needs the meta table only: known since the plan)
        # pack-chunks: whole pre-pass minibatches, about `pipeline_chunks` of them over the replay
        chunks = int(self.pipeline_chunks) or (8 if int(engine.desc.D) > 32 else 2)
        per = max(1, -(-len(row_lists) // max(1, chunks)))
        main = torch.cuda.current_stream(dev)
        if self._copy_stream is None or self._copy_stream.device != dev:
            self._copy_stream = torch.cuda.Stream(device=dev)
        packed.alloc_device(dev)
        self._copy_stream.wait_stream(main)     # the device buffer may be the previous iteration's: its last readers are on `main`
        for first in range(0, len(row_lists), per):
            last = min(first + per, len(row_lists))
            t0, t1 = int(row_lists[first][0]), int(row_lists[last - 1][-1]) + 1
            if packed.filled < t1:
                packed.fill(t0, t1)
            with torch.cuda.stream(self._copy_stream):
                packed.upload(t0, t1)
                landed = torch.cuda.Event()
                landed.record(self._copy_stream)
            main.wait_event(landed)
            for k in range(first, last):
                if shared and k % self.dist.world != self.dist.rank:
                    continue
                rows = row_lists[k]
                mb, _ = sched.minibatch(k)
                lo, hi = int(rows[0]), int(rows[-1]) + 1
                # no backward is pending on slot 0 here: the pre-pass shares the training arena (no second multi-GB slot)
                engine.forward(packed, mb, self.flat, values[lo:hi], logp[lo:hi], ent[lo:hi], keep=False, slot=0)
        self._copy_stream.synchronize()           # the pinned staging buffer may be refilled next iteration
        if shared:
            self.dist.all_reduce_sum(values)
            self.dist.all_reduce_sum(logp)
        adv = torch.empty(T, device=dev)
        ret = torch.empty(T, device=dev)
        engine.gae(rewards, masks, values, self.gamma, self.tau, adv, ret)
        return IterationState(packed, T, exps_np, exps, values, logp, adv, ret)

    def draw_permutations(self, it, n_epochs):
        """'global' data parallelism: the iteration's ``n_epochs`` shuffles drawn up front -- one ``np.random.shuffle`` per
        epoch on the numpy GLOBAL RNG, exactly the draws (and the order of draws) of the reference's epoch loop (:306-312),
        nothing else touches that RNG in between -- and rank 0's draws handed to every rank in ONE broadcast per iteration
        (a broadcast per epoch drained the GPU pipeline once per epoch).  Every rank draws, so identically seeded ranks stay
        aligned; rank 0's stream is the one all of them use, so no rank-local np.random use (env code, another thread) can
        make the ranks slice different global minibatches."""
        perms = np.empty((n_epochs, it.T), dtype=np.int64)
        for e in range(n_epochs):
            perm = np.arange(it.T)
            np.random.shuffle(perm)
            perms[e] = perm
        it.perms = self.dist.broadcast_array(perms, self.engine.device) if n_epochs else perms
        it.perm_next = 0

    def make_epoch(self, it):
        """Next epoch's schedule: numpy-global-RNG shuffle composed onto the running order (:306-319)."""
        T, dev, d = it.T, self.engine.device, self.dist
        if getattr(it, 'perms', None) is not None and it.perm_next < len(it.perms):
            perm = it.perms[it.perm_next]
            it.perm_next += 1
        else:
            perm = np.arange(T)
            np.random.shuffle(perm)
            if self._mode == 'global':
                perm = d.broadcast_array(perm, dev)
        it.order = it.order[perm]
        meta = it.packed.meta
        stage_np = meta[:, packer.M_STAGE]
        if self.batch_stage:
            # get_perm_batch_stage (:273-279): land-use rows first, then road rows, both in the shuffled order; the
            # reference indexes a two-entry list with stage.argmax(), so a row of any other stage is an IndexError
            st = stage_np[it.order]
            if ((st != 0) & (st != 1)).any():
                raise IndexError('batch_stage: row %d is neither a land-use nor a road row (list index out of range in '
                                 'the reference\'s get_perm_batch_stage)' % int(it.order[np.flatnonzero((st != 0) & (st != 1))[0]]))
            it.order = np.concatenate([it.order[st == 0], it.order[st == 1]])
        count = lambda rows: [len(rows), int((it.exps_np[rows] != 0).sum()), int((stage_np[rows] == 0).sum()),
                              int((stage_np[rows] == 1).sum())]
        if self._mode == 'global':
            B = self.mini_batch_size
            nb = int(math.floor(T / B))
            glob = [it.order[i * B:(i + 1) * B] for i in range(nb)]
            row_lists = [split_minibatch(g, meta[g, packer.M_E], d.rank, d.world, self.dp_balance) for g in glob]
            counts = [list(c) for c in zip(*[count(g) for g in glob])] if nb else [[], [], [], []]
        else:
            B = self.mini_batch_size
            nb = d.agree_min(int(math.floor(T / B)), dev)       # ranks may hold different numbers of rows
            row_lists = [it.order[i * B:(i + 1) * B] for i in range(nb)]
            # per-minibatch global counts (rows, rows with exps != 0, land-use rows, road rows): known on the
            # host, exchanged with ONE tiny all-reduce per epoch when data-parallel
            local = [list(c) for c in zip(*[count(r) for r in row_lists])] if nb else [[], [], [], []]
            counts = global_counts(d, local, dev) if nb else local
        rows_glob, ind_glob, land_glob, road_glob = counts
        lanes = self.lanes if (nb and self.lanes > 1 and int(self.engine.desc.D) > 32
                               and all(len(r) % self.lanes == 0 and len(r) >= 2 * self.lanes for r in row_lists)) else 1
        if lanes > 1:       # minibatch k -> schedule entries k * lanes + j: the j-th edge-balanced share of its rows
            row_lists = [split_minibatch(r, meta[r, packer.M_E], j, lanes, self.dp_balance) for r in row_lists for j in range(lanes)]
        sched = packer.Schedule(it.packed, row_lists, dev) if nb else None
        flat_rows = (np.concatenate(row_lists) if nb else np.zeros(0)).astype(np.int64)
        order_dev, _ = packer.upload_pinned(flat_rows, dev)      # (recycled page-locked staging buffer, asynchronous copy)
        ep = Epoch(sched, order_dev, nb, rows_glob, ind_glob, land_glob, road_glob)
        ep.lanes = lanes
        return ep

    # ------------------------------------------------------------------ one optimizer step
    def step(self, it, ep, k, loss_out=None):
        """forward + loss + backward (+ all-reduce) + first-step clip + Adam on minibatch k of the epoch.
        Everything is enqueued on the current stream; nothing synchronises with the host."""
        engine, B = self.engine, self.local_rows()
        dev = engine.device
        value_b, logp_b, ent_b, dvalue, dlogp, dent = self._rows
        nflt = engine.n_floats
        inv_rows = 1.0 / ep.rows_glob[k]
        inv_ind = 1.0 / ep.ind_glob[k] if ep.ind_glob[k] > 0 else float('nan')
        if getattr(ep, 'lanes', 1) > 1:
            return self._step_lanes(it, ep, k, loss_out, inv_rows, inv_ind)
        mb, _ = ep.sched.minibatch(k)
        idx = ep.order_dev[k * B:(k + 1) * B]
        if self.fused_small and engine.step_fused_ok(mb):
            # small models (gcn_node_dim <= 32, the shipped YAML dims): gathers + forward + loss + backward of the whole
            # minibatch in ONE launch, the per-workgroup gradient slabs added by a second one (csrc/tiny.hip)
            engine.step_fused(it.packed, mb, self.flat, idx, it.adv, it.ret, it.old_logp, it.exps, self.clip_epsilon,
                              self.value_pred_coef, self.entropy_coef, inv_rows, inv_ind, value_b, logp_b, ent_b,
                              self.grads[:nflt], self.grads[nflt:])
            return self._finish_step(ep, k, loss_out, buckets=False)
        engine.forward(it.packed, mb, self.flat, value_b, logp_b, ent_b, keep=True)
        # gathers of the minibatch rows, the loss and zero_grad in one launch
        engine.ppo_loss_rows(B, value_b, logp_b, ent_b, idx, it.adv, it.ret, it.old_logp, it.exps,
                             self.clip_epsilon, self.value_pred_coef, self.entropy_coef, inv_rows, inv_ind,
                             dvalue, dlogp, dent, self.grads[nflt:], zero=self.grads[:nflt])
        engine.backward(it.packed, mb, self.flat, dvalue, dlogp, dent, self.grads)
        self._finish_step(ep, k, loss_out)

    def _step_lanes(self, it, ep, k, loss_out, inv_rows, inv_ind):
        """One optimizer step as `ep.lanes` half-minibatch steps on their own streams (UPAMD_LANES); see __init__."""
        engine, L = self.engine, ep.lanes
        dev, nflt = engine.device, engine.n_floats
        Bh = self.local_rows() // L
        if self._lane_streams is None or self._lane_streams[0].device != dev:
            self._lane_streams = [torch.cuda.Stream(device=dev) for _ in range(L)]
            self._lane_bufs = None
        if self._lane_bufs is None or self._lane_bufs[0][0][0].numel() != Bh:
            self._lane_bufs = [([torch.empty(Bh, device=dev) for _ in range(6)], torch.zeros(nflt + 4, device=dev)) for _ in range(L)]
        main = torch.cuda.current_stream(dev)
        for j, st in enumerate(self._lane_streams):
            st.wait_stream(main)                # parameters (Adam of the previous step), the schedule upload
            (value_b, logp_b, ent_b, dvalue, dlogp, dent), g = self._lane_bufs[j]
            q = k * L + j
            with torch.cuda.stream(st):
                mb, _ = ep.sched.minibatch(q)
                idx = ep.order_dev[q * Bh:(q + 1) * Bh]
                engine.forward(it.packed, mb, self.flat, value_b, logp_b, ent_b, keep=True, slot=('lane', j))
                # losses scaled by the counts of the WHOLE minibatch, as a data-parallel rank scales its share
                engine.ppo_loss_rows(Bh, value_b, logp_b, ent_b, idx, it.adv, it.ret, it.old_logp, it.exps,
                                     self.clip_epsilon, self.value_pred_coef, self.entropy_coef, inv_rows, inv_ind,
                                     dvalue, dlogp, dent, g[nflt:], zero=g[:nflt])
                engine.backward(it.packed, mb, self.flat, dvalue, dlogp, dent, g, slot=('lane', j))
        for st in self._lane_streams:
            main.wait_stream(st)
        torch.add(self._lane_bufs[0][1], self._lane_bufs[1][1], out=self.grads)      # gradients + the 4 loss scalars
        self._finish_step(ep, k, loss_out, buckets=False)

    def _reduce_gradients(self, buckets):
        """The step's gradient all-reduce.  Bucketed: the engine's backward has recorded one event per
        range of the flat buffer, in the order the ranges became final; each range is reduced on the communication stream as
        soon as ITS event has fired, i.e. underneath the backward of the layers below it, and only the last range -- numerical
        + node encoder and the first GCN layer, final with the backward's last launch -- is exposed.  The ranges are disjoint
        and a sum over ranks is element-wise: bucketed and single-collective steps give the same bits."""
        d, nflt = self.dist, self.engine.n_floats
        # Bucketed by default under RCCL only.  gloo moves HOST memory: every bucket is a blocking D2H + host reduce + H2D of the calling
        # thread, nothing overlaps on the wire, and with several test ranks time-slicing ONE GPU the four blocking hand-overs per step
        # cost whole scheduling rounds (4 ranks: 18.6 ms per step with one collective, 243 ms bucketed -- profiles/r05_lab_bucket_overhead.md);
        # UPAMD_GRAD_BUCKETS=force keeps the route testable there (and with one rank).
        if buckets and self._buckets_check:
            self._check_bucket_finality()
        on = buckets and self.bucketed_allreduce and d.active and ((d.world > 1 and d.backend == 'nccl') or self._buckets_forced)
        ranges = self.engine.grad_buckets() if on else []
        if len(ranges) <= 1:
            self.last_buckets = None
            d.all_reduce_sum(self.grads)                  # ONE collective per optimizer step (no-op for one rank)
            return
        dev = self.engine.device
        if self._comm is None or self._comm.device != dev:
            # NORMAL priority: the engine's side streams are high-priority, and a high-priority communication stream (which only ever
            # holds event waits) shared their hardware queue -- its waits then stood in front of side-stream kernels: +0.16 ms per
            # 256-row step under a one-rank RCCL group against +0.07 at normal priority (profiles/r05_lab_bucket_overhead.md)
            self._comm = torch.cuda.Stream(device=dev, priority=self._comm_priority)
        ranges = [(b, e + 4 if e == nflt else e) for b, e in ranges]      # the 4 loss scalars ride behind the last parameter
        works = []
        lab = self._bucket_lab                              # lab only: 'late' = the same collectives, all behind the backward;
        if lab == 'two':                                    # 'two' = the first range early, everything else as one late collective
            lo = min(b for b, e in ranges[1:])
            hi = max(e for b, e in ranges[1:])
            self.engine.grad_bucket_wait(0, self._comm)
            with torch.cuda.stream(self._comm):
                works.append(d.all_reduce_sum_async(self.grads[ranges[0][0]:ranges[0][1]]))
            self.engine.grad_bucket_wait(len(ranges) - 1, self._comm)
            with torch.cuda.stream(self._comm):
                works.append(d.all_reduce_sum_async(self.grads[lo:hi]))
            ranges = []
        for j, (b, e) in enumerate(ranges):
            if j == len(ranges) - 1 and lab != 'late' and self._bucket_tail_main:
                # the last range becomes final with the backward's last launch on the CALLER's stream: issue its collective from
                # there (one stream hand-over less than through the communication stream)
                works.append(d.all_reduce_sum_async(self.grads[b:e]))
                continue
            self.engine.grad_bucket_wait(len(ranges) - 1 if lab == 'late' else j, self._comm)
            with torch.cuda.stream(self._comm):
                works.append(d.all_reduce_sum_async(self.grads[b:e]))
        for w in works:
            w.wait()                                      # the caller's stream continues behind every bucket
        self.last_buckets = ranges

    def _check_bucket_finality(self):
        """UPAMD_GRAD_BUCKETS=check: snapshot every range behind ITS event on the communication stream, compare with the gradient
        buffer behind the whole backward (the caller's stream).  Called right behind engine.backward, i.e. where the collectives of
        the bucketed route are issued."""
        ranges = self.engine.grad_buckets()
        if len(ranges) <= 1:
            return
        dev = self.engine.device
        if self._comm is None or self._comm.device != dev:
            self._comm = torch.cuda.Stream(device=dev, priority=self._comm_priority)
        if self.bucket_check_mismatches is None:
            self.bucket_check_mismatches = torch.zeros((), dtype=torch.int64, device=dev)
        snaps = []
        for j, (b, e) in enumerate(ranges[:-1]):          # (the last range is final with the backward's last launch by definition)
            self.engine.grad_bucket_wait(j, self._comm)
            with torch.cuda.stream(self._comm):
                snap = self.grads[b:e].clone()
                if self._bucket_check_selftest:
                    snap[0] += 1.0                        # a float that differs from the final gradient for certain
                snaps.append((b, e, snap))
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(self._comm)
        for b, e, snap in snaps:
            snap.record_stream(cur)                       # allocated on the communication stream, read on the caller's
            # bit comparison (NaNs included): a float that changed after its range's event is a finality violation
            self.bucket_check_mismatches += (snap.view(torch.int32) != self.grads[b:e].view(torch.int32)).sum()
            self.bucket_checks += 1

    def _finish_step(self, ep, k, loss_out, buckets=True):
        """all-reduce, first-step clip, Adam -- everything behind the backward of a step"""
        engine, nflt = self.engine, self.engine.n_floats
        timed = self.collective_events is not None and self.dist.world > 1
        if timed:       # on the stream the backward was launched on, right behind its last launch: the pair measures how long that
            # stream WAITS for the collective(s) -- all of a single collective, the exposed tail of the bucketed form (plus rank skew)
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
        self._reduce_gradients(buckets)
        if timed:
            pair[1].record()
            self.collective_events.append(pair)
        if self.clip_pending:
            engine.clip_first_step(self.grads, self.max_grad_norm, self.scratch)
            self.clip_pending = False
        has_rows = (True, ep.land_glob[k] > 0, ep.road_glob[k] > 0)
        steps = [0, 0, 0]
        for g in range(3):
            # torch >= 2.0: a head without rows has grad None and Adam skips it; legacy_zero_grad (torch <= 1.13): once
            # a head has had a gradient it keeps stepping with g = 0
            if has_rows[g] or (self.legacy_zero_grad and self._group_seen[g]):
                self._group_seen[g] = True
                self.group_steps[g] += 1
                steps[g] = self.group_steps[g]
        # every group's Adam step (and the copy-out of the step's loss scalars) in one launch
        engine.adam_groups(steps, self.flat, self.grads, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                           self.weight_decay, loss_src=self.grads[nflt:] if loss_out is not None else None,
                           loss_dst=loss_out)

    def collective_ms(self):
        """[ms per recorded step] of the gradient all-reduce (HIP events; synchronises); clears the record"""
        pairs, self.collective_events = self.collective_events or [], []
        if pairs:
            pairs[-1][1].synchronize()
        return [a.elapsed_time(b) for a, b in pairs]

    # ------------------------------------------------------------------ the reference's entry point
    def update_params(self, batch, iteration=0, tb_logger=None, max_steps=None):
        t0 = time.time()
        engine = self.attach()
        dev = engine.device
        self.policy_net.train(True)
        self.value_net.train(True)
        if self.pending_state is not None:
            state, self.pending_state = self.pending_state, None
            self.load_state_dict(state)
        it = self.prepare(batch)
        t_loop = time.time()
        cap = self.num_optim_epoch * int(math.floor(it.T / self.mini_batch_size))      # upper bound on the steps
        steps_total = cap if max_steps is None else min(max_steps, cap)
        loss_log = torch.zeros(max(steps_total, 1), 4, device=dev)
        step = 0
        epoch_ranges = []
        if self._mode == 'global':
            per_epoch = int(math.floor(it.T / self.mini_batch_size))
            self.draw_permutations(it, -(-steps_total // per_epoch) if per_epoch else 0)
        for _ in range(self.num_optim_epoch):
            if step >= steps_total:
                break
            ep = self.make_epoch(it)
            first = step
            for k in range(ep.nb):
                if step >= steps_total:
                    break
                self.step(it, ep, k, loss_out=loss_log[step])
                step += 1
            epoch_ranges.append((first, step))
        losses = loss_log[:step].cpu().numpy() if step > 0 else np.zeros((0, 4), dtype=np.float32)
        self.detach()
        torch.cuda.synchronize(dev)
        t_end = time.time()

        # ---- logging exactly as the reference emits it (:338-361), after the fact
        if tb_logger is not None:
            tags = ('loss/loss', 'loss/value_loss', 'loss/surr_loss', 'loss/entropy_loss')
            for i in range(step):
                for j, tag in enumerate(tags):
                    tb_logger.add_scalar(tag, float(losses[i, j]), self.loss_iter + i)
            totals = np.zeros(4)
            for e, (a, b) in enumerate(epoch_ranges):
                ep_sum = losses[a:b].astype(np.float64).sum(0) if b > a else np.zeros(4)
                totals += ep_sum
                ge = iteration * self.num_optim_epoch + e
                for j, tag in enumerate(('loss/epoch_loss', 'loss/epoch_value_loss', 'loss/epoch_surr_loss',
                                         'loss/epoch_entropy_loss')):
                    tb_logger.add_scalar(tag, float(ep_sum[j]), ge)
            for j, tag in enumerate(('loss/total_loss', 'loss/total_value_loss', 'loss/total_surr_loss',
                                     'loss/total_entropy_loss')):
                tb_logger.add_scalar(tag, float(totals[j] / self.num_optim_epoch), iteration)
        self.loss_iter += step
        self.last_losses = losses
        self.last_timing = dict(prepare=t_loop - t0, loop=t_end - t_loop, total=t_end - t0, steps=step,
                                rows_per_step=self.global_rows(), dp_mode=self._mode)
        return t_end - t0


class HipUpdateMixin:
    """Mix into the reference agent to route ``update_params`` through the HIP engine:

        class Agent(HipUpdateMixin, UrbanPlanningAgent): pass

    ``setup_model`` must create the networks with ``drl_urban_planning_amd.create_sgnn_model``
    (see INTEGRATION.md).  Hyper-parameters are read from the attributes the reference's
    ``AgentPPO.__init__`` sets and from ``self.optimizer.param_groups[0]``.
    """

    def _hip_updater(self):
        up = getattr(self, '_upamd_updater', None)
        if up is None:
            pg = self.optimizer.param_groups[0]
            dev = next(self.policy_net.parameters()).device
            # under torchrun (WORLD_SIZE > 1) the update is data-parallel: one process per GPU, gradients all-reduced
            # over RCCL; UPAMD_DP_MODE = auto | global | local (dist.py), UPAMD_DIST_BACKEND overrides the backend
            ctx = getattr(self, 'dist_ctx', None) or DistContext.from_env(device=dev if dev.type == 'cuda' else None)
            specs = getattr(getattr(self, 'cfg', None), 'agent_specs', None) or {}
            up = PPOUpdater(self.policy_net, self.value_net, lr=pg['lr'], eps=pg['eps'],
                            weight_decay=pg['weight_decay'], betas=tuple(pg['betas']), gamma=self.gamma, tau=self.tau,
                            clip_epsilon=self.clip_epsilon, value_pred_coef=self.value_pred_coef,
                            entropy_coef=self.entropy_coef, num_optim_epoch=self.opt_num_epochs,
                            mini_batch_size=self.mini_batch_size, batch_stage=bool(specs.get('batch_stage', False)),
                            dist_ctx=ctx, dp_mode=os.environ.get('UPAMD_DP_MODE', 'auto'),
                            legacy_zero_grad=self._legacy_zero_grad())
            up.loss_iter = self.loss_iter
            up.pending_state = getattr(self, '_upamd_pending_opt', None)       # a checkpoint loaded before the updater existed
            self._upamd_pending_opt = None
            self._upamd_updater = up
        return up

    def _legacy_zero_grad(self):
        """What ``self.optimizer.zero_grad()`` (urban_planning_agent.py:334) does in THIS process: zero-fill (the
        reference's pinned torch <= 1.13) or set to None (torch >= 2.0).  UPAMD_LEGACY_ZERO_GRAD = 0 | 1 overrides."""
        env = os.environ.get('UPAMD_LEGACY_ZERO_GRAD')
        if env in ('0', '1'):
            return env == '1'
        try:
            return inspect.signature(self.optimizer.zero_grad).parameters['set_to_none'].default is False
        except (AttributeError, KeyError, ValueError, TypeError):
            return torch_zero_grad_zero_fills()

    def update_params(self, batch, iteration):
        up = self._hip_updater()
        up.loss_iter = self.loss_iter
        elapsed = up.update_params(batch, iteration, tb_logger=getattr(self, 'tb_logger', None))
        self.loss_iter = up.loss_iter
        return elapsed


def install(agent):
    """Patch an existing reference agent INSTANCE (its networks must come from our create_sgnn_model)."""
    import types
    agent._hip_updater = types.MethodType(HipUpdateMixin._hip_updater, agent)
    agent._legacy_zero_grad = types.MethodType(HipUpdateMixin._legacy_zero_grad, agent)
    agent.update_params = types.MethodType(HipUpdateMixin.update_params, agent)
    return agent
