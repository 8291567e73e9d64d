"""The reference-side binding, pinned against the REAL reference class (build container only: /root/reference does not
travel) and, for the torchrun policies, on a stand-in under a world-2 gloo group.

* the diff INTEGRATION.md documents is applied TEXTUALLY (``patch -p1``) to a copy of
  urban_planning/agents/urban_planning_agent.py, imported with the geometry stack stubbed, and the agent is built the way
  the reference builds it (its own ``setup_model`` / ``setup_optimizer`` / ``AgentPPO.__init__``, :28-47);
* ``type(agent).update_params is HipUpdateMixin.update_params`` -- and NOT under the form this repo documented until
  round 3 (mixin added to the class's bases: the class's own ``update_params``, :248, shadows it);
* the launcher's in-memory patch gives the same class to ``from ... import UrbanPlanningAgent`` (subprocess);
* LOCAL_RANK -> device mapping, the duplicate-device error, who samples / evaluates / writes under WORLD_SIZE = 2.
"""
import importlib.util
import os
import pickle
import re
import shutil
import socket
import subprocess
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import ref_import

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 'urban_planning/agents/urban_planning_agent.py'
needs_reference = pytest.mark.skipif(not ref_import.available(), reason='/root/reference is only in the build container')

HYPER = dict(lr=3e-4, eps=2e-5, weightdecay=1e-3, gamma=0.97, tau=0.9, clip_epsilon=0.15, value_pred_coef=0.4,
             entropy_coef=0.02, num_optim_epoch=3, mini_batch_size=8)


def documented_diff():
    """The ```diff block of INTEGRATION.md that patches urban_planning_agent.py."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```diff\n(.*?)```', text, flags=re.S)
    hits = [b for b in blocks if ('+++ b/' + REL) in b]
    assert len(hits) == 1, 'INTEGRATION.md must hold exactly one diff of %s' % REL
    return hits[0]


def patched_module(tmp_path, name='urban_planning.agents._upamd_patched_agent'):
    """Copy of the real file with the documented diff applied, imported next to the real package."""
    ref_import.load_reference()                      # stubs + sys.path
    tree = tmp_path / 'tree'
    dst = tree / REL
    dst.parent.mkdir(parents=True)
    shutil.copy(os.path.join(ref_import.REFERENCE_ROOT, REL), dst)
    (tmp_path / 'doc.diff').write_text(documented_diff())
    res = subprocess.run(['patch', '-p1', '--no-backup-if-mismatch', '-i', str(tmp_path / 'doc.diff')], cwd=str(tree),
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    spec = importlib.util.spec_from_file_location(name, str(dst))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.modules.pop(name, None)
    return mod


def reference_cfg(tmp_path, **over):
    cfg = ref_import.DuckCfg(D=16, L=2, max_nodes=40, max_edges=120)
    cfg.agent = 'rl-sgnn'
    for k, v in dict(HYPER, **over).items():
        setattr(cfg, k, v)
    cfg.model_dir = str(tmp_path / 'models')
    os.makedirs(cfg.model_dir, exist_ok=True)
    cfg.save_model_interval = 1
    return cfg


def build_like_the_reference(cls, ref, cfg):
    """What ``UrbanPlanningAgent.__init__`` (:28-47) does, minus the env / logger set-up that needs the geometry stack:
    the class's own setup_model + setup_optimizer, then the REAL ``AgentPPO.__init__`` with the reference's arguments."""
    ag = object.__new__(cls)
    ag.cfg, ag.training, ag.device, ag.loss_iter = cfg, True, torch.device('cpu'), 0
    ag.node_dim, ag.numerical_feature_size = 23, 52                      # setup_env (:93-96)
    ag.tb_logger = ref_import.ScalarLog()
    ag.logger = types.SimpleNamespace(info=lambda *a, **k: None)
    ag.best_rewards, ag.best_plans, ag.current_rewards, ag.current_plans = -1000.0, [], -1000.0, []
    ag.save_best_flag = False
    ag.setup_model()
    ag.setup_optimizer()
    ref.AgentPPO.__init__(ag, env=None, dtype=torch.float32, device=ag.device, logger_cls=object, traj_cls=object,
                          num_threads=1, policy_net=ag.policy_net, value_net=ag.value_net, optimizer=ag.optimizer,
                          opt_num_epochs=cfg.num_optim_epoch, gamma=cfg.gamma, tau=cfg.tau,
                          clip_epsilon=cfg.clip_epsilon, value_pred_coef=cfg.value_pred_coef,
                          entropy_coef=cfg.entropy_coef,
                          policy_grad_clip=[(ag.policy_net.parameters(), 1), (ag.value_net.parameters(), 1)],
                          mini_batch_size=cfg.mini_batch_size)
    return ag


@needs_reference
def test_documented_patch_binds_update_params_on_the_real_class(tmp_path):
    from drl_urban_planning_amd import HipUpdateMixin, models, synth
    from drl_urban_planning_amd.binding import TorchrunPolicyMixin
    ref = ref_import.load_reference()
    mod = patched_module(tmp_path)
    cls = mod.UrbanPlanningAgent
    # the class train.py / eval.py import: HIP update in front, reference class behind, reference body untouched
    assert cls.__name__ == 'UrbanPlanningAgent' and cls._upamd_bound
    assert cls.__mro__[:3] == (cls, HipUpdateMixin, TorchrunPolicyMixin)
    assert 'update_params' in cls._upamd_reference_class.__dict__            # (:248 is still there, and shadowed)
    assert cls.update_params is HipUpdateMixin.update_params
    assert mod.create_sgnn_model is models.create_sgnn_model and mod.ActorCritic is models.ActorCritic

    cfg = reference_cfg(tmp_path)
    ag = build_like_the_reference(cls, ref, cfg)
    assert type(ag).update_params is HipUpdateMixin.update_params
    assert isinstance(ag.policy_net, models.UrbanPlanningPolicy) and ag.policy_net.shared_net is ag.value_net.shared_net

    # every hyper-parameter comes off the object the real AgentPPO.__init__ / setup_optimizer initialised
    up = ag._hip_updater()
    assert (up.lr, up.eps, up.weight_decay, tuple(up.betas)) == (cfg.lr, cfg.eps, cfg.weightdecay, (0.9, 0.999))
    assert (up.gamma, up.tau, up.clip_epsilon) == (cfg.gamma, cfg.tau, cfg.clip_epsilon)
    assert (up.value_pred_coef, up.entropy_coef) == (cfg.value_pred_coef, cfg.entropy_coef)
    assert (up.num_optim_epoch, up.mini_batch_size, up.batch_stage) == (cfg.num_optim_epoch, cfg.mini_batch_size, False)
    assert up.policy_net is ag.policy_net and up.value_net is ag.value_net and not up.dist.active
    cfg2 = reference_cfg(tmp_path)
    cfg2.agent_specs = {'batch_stage': True}
    assert build_like_the_reference(cls, ref, cfg2)._hip_updater().batch_stage is True

    # no silent CPU route: the networks are on the CPU here, the update must refuse
    rep = synth.make_replay(16, 'hlg', max_nodes=40, max_edges=120, seed=1, n_range=(12, 30))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ag.update_params(rep, 0)


@needs_reference
def test_the_form_documented_until_round_3_does_not_bind():
    """`class UrbanPlanningAgent(HipUpdateMixin, AgentPPO)` with the reference body: its own update_params wins."""
    from drl_urban_planning_amd import HipUpdateMixin
    ref = ref_import.load_reference()
    real = ref.UrbanPlanningAgent
    old_form = type('UrbanPlanningAgent', (HipUpdateMixin, ref.AgentPPO), dict(real.__dict__))
    assert old_form.update_params is not HipUpdateMixin.update_params
    assert old_form.update_params is real.__dict__['update_params']


@needs_reference
def test_reference_checkpoints_round_trip_through_the_bound_agent(tmp_path):
    """The REAL save_checkpoint / load_checkpoint (:153-194) between a bound agent (this package's modules) and an
    untouched reference agent (the reference's modules): 52 keys, same tensors, both directions."""
    ref = ref_import.load_reference()
    mod = patched_module(tmp_path)
    cfg_a, cfg_b = reference_cfg(tmp_path), reference_cfg(tmp_path)
    torch.manual_seed(3)
    ours = build_like_the_reference(mod.UrbanPlanningAgent, ref, cfg_a)
    torch.manual_seed(4)
    theirs = build_like_the_reference(ref.UrbanPlanningAgent, ref, cfg_b)
    assert type(theirs).update_params is ref.UrbanPlanningAgent.__dict__['update_params']
    sd_ours = {k: v.clone() for k, v in ours.actor_critic_net.state_dict().items()}
    sd_theirs = {k: v.clone() for k, v in theirs.actor_critic_net.state_dict().items()}
    assert list(sd_ours) == list(sd_theirs) and len(sd_ours) == 52
    assert any(not torch.equal(sd_ours[k], sd_theirs[k]) for k in sd_ours)

    ours.loss_iter = 17
    ours.save_checkpoint(0)                                   # -> iteration_0001.p, written by the reference's code
    assert theirs.load_checkpoint(1, True) == 1 and theirs.loss_iter == 17
    for k, v in theirs.actor_critic_net.state_dict().items():
        assert torch.equal(v, sd_ours[k]), k

    theirs.actor_critic_net.load_state_dict(sd_theirs)
    theirs.loss_iter = 5
    theirs.save_checkpoint(1)                                 # -> iteration_0002.p
    assert ours.load_checkpoint(2, True) == 2 and ours.loss_iter == 5
    for k, v in ours.actor_critic_net.state_dict().items():
        assert torch.equal(v, sd_theirs[k]), k
    cp = pickle.load(open(os.path.join(cfg_b.model_dir, 'iteration_0002.p'), 'rb'))
    assert set(cp) >= {'actor_critic_dict', 'loss_iter', 'iteration'}


_SITECUSTOMIZE = '''
import sys
from unittest.mock import MagicMock
for m in %r:
    sys.modules.setdefault(m, MagicMock())
sys.path.insert(0, %r)
sys.path.insert(0, %r)
'''

_ENTRY = '''
import sys
from urban_planning.agents.urban_planning_agent import UrbanPlanningAgent     # what urban_planning/train.py:11 does
import urban_planning.agents.urban_planning_agent as m
from drl_urban_planning_amd import HipUpdateMixin, models
assert UrbanPlanningAgent.update_params is HipUpdateMixin.update_params
assert m.create_sgnn_model is models.create_sgnn_model
assert __name__ == '__main__' and sys.argv[1:] == ['--cfg', 'hlg']
print('BOUND', UrbanPlanningAgent.__name__, [c.__name__ for c in UrbanPlanningAgent.__mro__[1:6]])
'''


@needs_reference
def test_launcher_binds_an_unedited_reference_tree(tmp_path):
    """`python -m drl_urban_planning_amd.launch ENTRY args`: the entry point's own import gets the bound class."""
    (tmp_path / 'sitecustomize.py').write_text(_SITECUSTOMIZE % (ref_import._STUBS, ref_import.REFERENCE_ROOT, ROOT))
    (tmp_path / 'entry.py').write_text(_ENTRY)
    env = dict(os.environ, PYTHONPATH=str(tmp_path))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, '-m', 'drl_urban_planning_amd.launch', str(tmp_path / 'entry.py'), '--cfg', 'hlg'],
                         env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert ("BOUND UrbanPlanningAgent ['HipUpdateMixin', 'TorchrunPolicyMixin', 'RolloutMixin', 'CheckpointMixin', "
            "'UrbanPlanningAgent']") in res.stdout


# --------------------------------------------------------------------------------------------- rank -> device mapping
def test_local_rank_is_mapped_to_one_visible_device():
    from drl_urban_planning_amd.launch import configure_rank
    env = {'LOCAL_RANK': '3', 'LOCAL_WORLD_SIZE': '8', 'HIP_VISIBLE_DEVICES': '0,1,2,3,4,5,6,7'}
    assert configure_rank(env) == '3' and env['HIP_VISIBLE_DEVICES'] == '3' and env['UPAMD_RANK_DEVICE'] == '3'
    env = {'LOCAL_RANK': '1', 'LOCAL_WORLD_SIZE': '2', 'CUDA_VISIBLE_DEVICES': '4,6'}       # an outer mask is honoured
    assert configure_rank(env) == '6' and env['HIP_VISIBLE_DEVICES'] == '6' and 'CUDA_VISIBLE_DEVICES' not in env
    env = {'HIP_VISIBLE_DEVICES': '0,1'}                                                     # single process: untouched
    assert configure_rank(env) is None and env == {'HIP_VISIBLE_DEVICES': '0,1'}
    env = {'LOCAL_RANK': '0', 'LOCAL_WORLD_SIZE': '1', 'HIP_VISIBLE_DEVICES': '0,1'}
    assert configure_rank(env) is None
    with pytest.raises(RuntimeError, match='one device per rank'):                           # 2 ranks, 1 GPU, RCCL
        configure_rank({'LOCAL_RANK': '1', 'LOCAL_WORLD_SIZE': '2', 'HIP_VISIBLE_DEVICES': '0'})
    # a ROCr-level mask (scheduler / cgroup): HIP indices are relative to what it leaves
    env = {'LOCAL_RANK': '1', 'LOCAL_WORLD_SIZE': '2', 'ROCR_VISIBLE_DEVICES': '4,6'}
    assert configure_rank(env) == '1' and env['HIP_VISIBLE_DEVICES'] == '1' and env['ROCR_VISIBLE_DEVICES'] == '4,6'
    with pytest.raises(RuntimeError, match='one device per rank'):
        configure_rank({'LOCAL_RANK': '2', 'LOCAL_WORLD_SIZE': '3', 'ROCR_VISIBLE_DEVICES': 'GPU-abc,GPU-def'})
    env = {'LOCAL_RANK': '1', 'LOCAL_WORLD_SIZE': '2', 'HIP_VISIBLE_DEVICES': '0', 'UPAMD_DIST_BACKEND': 'gloo'}
    assert configure_rank(env) == '0'                                                        # test ranks may share


def test_two_ranks_on_one_device_are_refused():
    from drl_urban_planning_amd.dist import assert_one_rank_per_device
    assert_one_rank_per_device(['box/GPU-a', 'box/GPU-b', 'other/GPU-a'])
    with pytest.raises(RuntimeError, match='ranks 0 and 2 both resolved to GPU box/GPU-a'):
        assert_one_rank_per_device(['box/GPU-a', 'box/GPU-b', 'box/GPU-a'])


# ------------------------------------------------------------------------------ torchrun policies, world 2 over gloo
class _Log:
    def __init__(self, **kw):
        self.num_steps, self.sample_time, self.tag = 0, 0.0, None

    @classmethod
    def merge(cls, logs, **kw):
        out = cls()
        out.num_steps = sum(x.num_steps for x in logs)
        out.tag = [x.tag for x in logs]
        return out


class _Writer:
    def __init__(self, path):
        self.path, self.rows = path, []
        open(os.path.join(path, 'events.%d' % os.getpid()), 'w').close()

    def add_scalar(self, *a):
        self.rows.append(a)

    def flush(self):
        pass

    def close(self):
        pass


class _ReferenceLikeAgent:
    """The five methods of UrbanPlanningAgent / Agent the policies wrap, with the reference's signatures."""
    logger_cls, logger_kwargs = _Log, {}

    def __init__(self, cfg):
        self.cfg, self.training, self.device, self.calls = cfg, True, torch.device('cpu'), []
        self.setup_logger(1)

    def setup_logger(self, num_threads):
        self.tb_logger = _Writer(self.cfg.tb_dir)
        self.log_path = os.path.join(self.cfg.log_dir, 'log_train.txt')
        open(self.log_path, 'a').write('x')

    def sample(self, num_samples, mean_action=False, nthreads=None):
        from drl_urban_planning_amd import synth
        self.calls.append(('sample', num_samples))
        seed = int(np.random.randint(1 << 30))                 # env code consumes the process's numpy stream
        log = _Log()
        log.num_steps, log.tag = num_samples, 'sampled by rank %s' % os.environ['RANK']
        return synth.make_replay(num_samples, 'hlg', max_nodes=60, max_edges=200, seed=seed, n_range=(20, 50)), log

    def eval_agent(self, num_samples=1, mean_action=True, visualize=False):
        self.calls.append(('eval', num_samples))
        log = _Log()
        log.tag = 'evaluated by rank %s' % os.environ['RANK']
        return log

    def save_checkpoint(self, iteration):
        self.calls.append(('save', iteration))
        with open(os.path.join(self.cfg.model_dir, 'iteration_%04d.p' % (iteration + 1)), 'a') as f:
            f.write('rank %s\n' % os.environ['RANK'])

    def update_params(self, batch, iteration):
        raise AssertionError('the reference update ran')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _policy_worker(rank, world, port, root):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), UPAMD_DIST_BACKEND='gloo')
    import torch.distributed as dist
    from drl_urban_planning_amd import HipUpdateMixin
    from drl_urban_planning_amd.binding import NullWriter, bind_reference_agent
    from drl_urban_planning_amd.dist import (DistContext, assert_one_rank_per_device, batch_fingerprint,
                                             exchange_idents)
    cfg = types.SimpleNamespace(seed=11, **{k: os.path.join(root, k) for k in ('tb_dir', 'log_dir', 'model_dir')})
    cls = bind_reference_agent(_ReferenceLikeAgent)
    assert bind_reference_agent(cls) is cls
    np.random.seed(cfg.seed)                                   # train.py:55: every process the same
    ag = cls(cfg)
    assert type(ag).update_params is HipUpdateMixin.update_params
    ctx = ag._upamd_ctx()
    assert isinstance(ctx, DistContext) and ctx.world == 2 and ctx.backend == 'gloo' and ag._hip_updater.__self__ is ag

    # single writer: rank 0 owns <tb_dir> and <log_dir>; rank 1 logs under rank1/ and holds a null TensorBoard writer
    if rank == 0:
        assert isinstance(ag.tb_logger, _Writer) and ag.tb_logger.path == cfg.tb_dir
        assert ag.log_path == os.path.join(cfg.log_dir, 'log_train.txt')
    else:
        assert isinstance(ag.tb_logger, NullWriter) and ag.tb_logger.add_scalar('a', 1.0, 0) is None
        assert ag.log_path == os.path.join(cfg.log_dir, 'rank1', 'log_train.txt') and cfg.log_dir.endswith('log_dir')

    # rank0 sampling: one rollout, one broadcast, the same replay (and log) everywhere
    batch, log = ag.sample(12)
    assert ag.calls == ([('sample', 12)] if rank == 0 else [])
    assert len(batch.states) == 12 and log.tag == 'sampled by rank 0'
    assert ctx.same_everywhere(batch_fingerprint(batch))
    # rank 0's numpy stream has moved on (env code), rank 1's has not: the permutation every rank uses is rank 0's
    perm = np.arange(100)
    np.random.shuffle(perm)
    mine = perm.copy()
    perm = ctx.broadcast_array(perm)
    everyone = ctx.gather_objects(mine.tolist())
    assert everyone[0] != everyone[1] and perm.tolist() == everyone[0]

    # evaluation on rank 0, log broadcast
    ev = ag.eval_agent(num_samples=1, mean_action=True)
    assert ev.tag == 'evaluated by rank 0' and (('eval', 1) in ag.calls) == (rank == 0)

    # checkpoints: one writer, and the file is there for everyone when save_checkpoint returns
    ag.save_checkpoint(0)
    assert open(os.path.join(cfg.model_dir, 'iteration_0001.p')).read() == 'rank 0\n'

    # per-rank sampling: own seed, own shard of num_samples / world steps, merged rollout log
    os.environ['UPAMD_DP_SAMPLING'] = 'per_rank'
    ag.calls.clear()
    batch, log = ag.sample(13)
    assert ag.calls == [('sample', 7)] and len(batch.states) == 7
    assert log.num_steps == 14 and log.tag == ['sampled by rank 0', 'sampled by rank 1']
    assert not ctx.same_everywhere(batch_fingerprint(batch))

    # the store-side device check RCCL launches go through: both ranks claim one GPU -> refused on every rank
    store = dist.distributed_c10d._get_default_store()
    idents = exchange_idents(store, rank, world, 'box/GPU-0')
    assert idents == ['box/GPU-0', 'box/GPU-0']
    with pytest.raises(RuntimeError, match='both resolved to GPU'):
        assert_one_rank_per_device(idents, rank)
    ctx.barrier()
    if rank == 1:
        open(os.path.join(root, 'ok'), 'w').close()
    ctx.close()


def test_torchrun_policies_world_2(tmp_path):
    for k in ('tb_dir', 'log_dir', 'model_dir'):
        os.makedirs(str(tmp_path / k))
    port = _free_port()
    mp.spawn(_policy_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(str(tmp_path / 'ok'))
    assert len(os.listdir(str(tmp_path / 'tb_dir'))) == 2          # rank 0's event file + the rank1/ directory


# --------------------------------------------------- SURVEY section 8f rows 1-3 bound into the REAL class (rollout_binding)
def _rollout_agent(tmp_path, num_threads=2, seed=3, cls=None, ref=None):
    """The REAL bound class built the way the reference builds it, with the stub env and the reference's own LoggerRL."""
    import logging
    ref = ref or ref_import.load_reference()                # (stubs + sys.path before the reference imports below)
    from khrylib.rl.core import LoggerRL
    from urban_planning.utils.tools import TrajBatchDisc
    from stub_env import StubCityEnv
    cls = cls or patched_module(tmp_path).UrbanPlanningAgent
    cfg = reference_cfg(tmp_path)
    torch.manual_seed(seed)
    ag = build_like_the_reference(cls, ref, cfg)
    ag.env = StubCityEnv()
    ag.logger_cls, ag.logger_kwargs, ag.traj_cls, ag.num_threads = LoggerRL, {}, TrajBatchDisc, num_threads
    ag.thread_loggers = [logging.getLogger('upamd-test-%d' % i) for i in range(num_threads)]
    ag.logger = logging.getLogger('upamd-test')
    return ag, ref


@needs_reference
def test_server_rollout_through_the_real_sample_worker(tmp_path, monkeypatch):
    """UPAMD_ROLLOUT=server: ``sample()`` of the REAL bound class runs the reference's own ``sample_worker`` (:49-91) in
    forked children behind action clients and shared-memory arenas.  Episodes come back as a RecordBatch whose rows are,
    bit for bit, the env's states in worker order; greedy sampling gives exactly the actions of the module's own
    ``select_action`` (policy.py:67-85); the default (reference) mode still returns the reference's TrajBatchDisc."""
    from drl_urban_planning_amd import packer, rollout
    from stub_env import StubCityEnv
    # the reference's learner runs with OMP_NUM_THREADS=1 (khrylib/rl/agents/agent.py:12): its workers fork() and run torch
    # forwards, which deadlocks in a child of a process whose OpenMP pool has already started
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    request_finalizer = lambda: torch.set_num_threads(threads)
    ag, ref = _rollout_agent(tmp_path)
    assert [c.__name__ for c in type(ag).__mro__[1:6]] == ['HipUpdateMixin', 'TorchrunPolicyMixin', 'RolloutMixin',
                                                           'CheckpointMixin', 'UrbanPlanningAgent']
    monkeypatch.setenv('UPAMD_ROLLOUT', 'server')
    np.random.seed(1)
    torch.manual_seed(1)
    batch, log = ag.sample(20, mean_action=True)            # 2 workers x 2 episodes x 5 steps, greedy
    assert isinstance(batch, rollout.RecordBatch) and len(batch) == 20 and log.num_steps == 20 and log.num_episodes == 4
    assert all(packer.is_record(s) for s in batch.states)
    assert ag._upamd_server_stats['requests'] == 20 and ag._upamd_server_stats['rows'] == 20
    env = StubCityEnv()
    rows = []
    for worker in range(2):                                 # every worker starts from its own fork of the same env
        e = StubCityEnv()
        for ep in range(2):
            s = e.reset()
            for t in range(5):
                a = ag.policy_net.select_action(ref.tensorfy([s]), True).numpy().squeeze(0)
                rows.append((s, a))
                s, r, done, info = e.step(a)
    for (want_s, want_a), rec, got_a in zip(rows, batch.states, batch.actions):
        for x, y in zip(packer.expand_state(rec, padded=True), want_s):
            assert np.array_equal(x, y)
        assert np.array_equal(got_a, want_a)
    assert batch.masks.tolist() == [1, 1, 1, 1, 0] * 4 and set(batch.exps.tolist()) == {0.0}     # exp = 1 - use_mean_action (:73)
    assert batch.rewards[4] == pytest.approx(1.0 + 0.01 * float(batch.actions[:5].sum()))
    assert log.avg_episode_reward == pytest.approx(float(batch.rewards.sum()) / 4)
    # stochastic sampling: valid actions, exps = 1 (noise_rate = 1), a different draw on the next call (the learner's streams advance)
    b1, _ = ag.sample(20)
    a1 = b1.actions.copy()
    b2, _ = ag.sample(20)
    assert set(b1.exps.tolist()) == {1.0} and not np.array_equal(a1, b2.actions)
    for rec, a in zip(b2.states, b2.actions):
        st = packer.expand_state(rec, padded=True)
        stage = int(np.argmax(st[8]))
        assert (st[6] if stage == 0 else st[7])[int(a[stage])] and a[1 - stage] == 0
    # the records pack exactly like the padded tuples the reference's queue would have carried
    padded = [packer.expand_state(r, padded=True) for r in b2.states]
    pk_a = packer.pack_replay(b2.states, b2.actions, 23, 52, pin=False)
    pk_b = packer.pack_replay(padded, b2.actions, 23, 52, pin=False)
    from test_packer import _sections
    sa, sb = _sections(pk_a, 52), _sections(pk_b, 52)
    assert all(np.array_equal(sa[k], sb[k]) for k in sa if k != 'meta')
    # switched off: the reference's own sample() (CPU forwards in the workers, Memory + queue) and its TrajBatchDisc
    monkeypatch.setenv('UPAMD_ROLLOUT', 'reference')
    b3, log3 = ag.sample(20, mean_action=True)
    assert type(b3).__name__ == 'TrajBatchDisc' and len(b3.states) == 20
    assert np.array_equal(np.stack(b3.actions), batch.actions)
    ag._upamd_release_arenas()
    request_finalizer()


@needs_reference
def test_server_evaluation_serial_and_overlapped(tmp_path, monkeypatch):
    """``eval_agent`` (:402-467) behind a client gives the reference's own evaluation (same greedy episode, same reward);
    UPAMD_EVAL=overlap runs that episode as one more client of the sampling phase and hands its log to the ``eval_agent``
    call ``optimize_policy`` (:225-246) makes after the update, with the evaluated weights kept for ``best.p``."""
    ag, ref = _rollout_agent(tmp_path)
    want = ag._upamd_eval_reference(1, True)                # the reference's body in this process (CPU modules)
    assert ag.env.mode == 'train' and want.num_episodes == 1 and len(want.plans) == 1
    monkeypatch.setenv('UPAMD_ROLLOUT', 'server')
    ag.env.episode = -1                                     # the same first episode again
    got = ag.eval_agent(num_samples=1, mean_action=True)
    assert got.avg_episode_reward == want.avg_episode_reward and got.plans == want.plans and got.sample_time > 0
    assert ag._upamd_server_stats['requests'] == 5
    monkeypatch.setenv('UPAMD_EVAL', 'overlap')
    ag.env.episode = -1
    sd0 = {k: v.clone() for k, v in ag.actor_critic_net.state_dict().items()}
    batch, log = ag.sample(20)
    assert ag._upamd_server_stats['requests'] == 25         # 20 sampling steps + the 5 of the evaluation episode, one serving phase
    with torch.no_grad():                                   # "update_params": the weights move on
        for p in ag.actor_critic_net.parameters():
            p.add_(0.01)
    ahead = ag.eval_agent(num_samples=1, mean_action=True)  # no second serving phase: the log of the overlapped episode
    assert ag._upamd_server_stats['requests'] == 25 and ahead.avg_episode_reward == want.avg_episode_reward
    assert all(torch.equal(ag._upamd_eval_sd[k], v) for k, v in sd0.items())
    # best.p is written from the EVALUATED weights, the periodic file from the current ones
    ag.save_best_flag, ag.best_rewards = True, 1.25
    ag.save_checkpoint(0)
    best = pickle.load(open(os.path.join(ag.cfg.model_dir, 'best.p'), 'rb'))
    periodic = pickle.load(open(os.path.join(ag.cfg.model_dir, 'iteration_0001.p'), 'rb'))
    k = 'actor_net.shared_net.node_encoder.weight'
    assert torch.equal(best['actor_critic_dict'][k], sd0[k]) and torch.equal(periodic['actor_critic_dict'][k], sd0[k] + 0.01)
    assert torch.equal(ag.actor_critic_net.state_dict()[k], sd0[k] + 0.01)
    ag._upamd_release_arenas()


@needs_reference
def test_checkpoints_carry_the_optimizer_state_through_the_real_save_and_load(tmp_path):
    """``save_checkpoint`` (:172-194) of the bound class adds 'hip_optimizer' to the files the reference's code wrote;
    a FRESH agent's ``load_checkpoint`` (:153-170, called from __init__ before AgentPPO.__init__ has run) parks it and the
    updater created later picks it up; an untouched reference agent still loads the same file; a load in the middle of a
    run leaves Adam alone (the reference keeps its optimizer object across freeze_land_use, :215-222)."""
    ref = ref_import.load_reference()
    mod = patched_module(tmp_path)
    ag, _ = _rollout_agent(tmp_path, cls=mod.UrbanPlanningAgent, ref=ref)
    up = ag._hip_updater()
    state = {'group_steps': [7, 7, 0], 'loss_iter': 21, 'clip_pending': False, 'group_seen': [True, True, False],
             'exp_avg': {'x': torch.arange(3.0)}, 'exp_avg_sq': {'x': torch.ones(3)}}
    up.m, up.state_dict = object(), lambda: state           # (the real buffers live on a GPU: tests/test_gpu_parity.py)
    ag.loss_iter = 21
    ag.save_checkpoint(0)
    cp = pickle.load(open(os.path.join(ag.cfg.model_dir, 'iteration_0001.p'), 'rb'))
    assert cp['hip_optimizer']['group_steps'] == [7, 7, 0] and torch.equal(cp['hip_optimizer']['exp_avg']['x'], torch.arange(3.0))
    fresh, _ = _rollout_agent(tmp_path, cls=mod.UrbanPlanningAgent, ref=ref, seed=9)
    assert getattr(fresh, '_upamd_updater', None) is None
    assert fresh.load_checkpoint(1, True) == 1 and fresh.loss_iter == 21
    assert fresh._upamd_pending_opt['group_steps'] == [7, 7, 0]
    up2 = fresh._hip_updater()
    assert up2.pending_state['loss_iter'] == 21 and fresh._upamd_pending_opt is None
    theirs = build_like_the_reference(ref.UrbanPlanningAgent, ref, reference_cfg(tmp_path))
    assert theirs.load_checkpoint(1, True) == 1             # the extra key does not disturb the reference's loader
    up.pending_state = None
    ag.load_checkpoint(1, True)                             # mid-run (the updater has stepped): Adam is left alone
    assert up.pending_state is None
    # the extended file was renamed into place (no temporary left behind), and only the files the reference's conditions name were
    # touched: a best save adds best.p and best_reward..., both with the key
    assert not [f for f in os.listdir(ag.cfg.model_dir) if f.endswith('.upamd_tmp')]
    ag.save_best_flag, ag.best_rewards = True, 1.25
    ag.save_checkpoint(1)
    names = sorted(f for f in os.listdir(ag.cfg.model_dir) if f.endswith('.p'))
    assert names == ['best.p', 'best_reward1.25_iteration_0002.p', 'iteration_0001.p', 'iteration_0002.p'], names
    assert all('hip_optimizer' in pickle.load(open(os.path.join(ag.cfg.model_dir, f), 'rb')) for f in names)


@needs_reference
def test_checkpoint_optimizer_state_can_be_switched_off(tmp_path, monkeypatch):
    """UPAMD_CKPT_OPTIMIZER=0: the files are exactly what the reference writes (no extra key, written once), and a resume from a file
    that HAS the key ignores it -- Adam restarts, as the reference's resumed run does."""
    ref = ref_import.load_reference()
    mod = patched_module(tmp_path)
    ag, _ = _rollout_agent(tmp_path, cls=mod.UrbanPlanningAgent, ref=ref)
    up = ag._hip_updater()
    state = {'group_steps': [3, 3, 0], 'loss_iter': 9, 'clip_pending': False, 'group_seen': [True, True, False],
             'exp_avg': {'x': torch.arange(3.0)}, 'exp_avg_sq': {'x': torch.ones(3)}}
    up.m, up.state_dict = object(), lambda: state
    ag.save_checkpoint(0)                                   # with the key
    monkeypatch.setenv('UPAMD_CKPT_OPTIMIZER', '0')
    ag.save_checkpoint(1)                                   # without
    assert 'hip_optimizer' in pickle.load(open(os.path.join(ag.cfg.model_dir, 'iteration_0001.p'), 'rb'))
    assert 'hip_optimizer' not in pickle.load(open(os.path.join(ag.cfg.model_dir, 'iteration_0002.p'), 'rb'))
    fresh, _ = _rollout_agent(tmp_path, cls=mod.UrbanPlanningAgent, ref=ref, seed=9)
    assert fresh.load_checkpoint(1, True) == 1 and fresh._upamd_pending_opt is None


@needs_reference
def test_server_rollout_reports_a_failing_worker_instead_of_hanging(tmp_path, monkeypatch):
    """An env worker that raises (here: the env's ``step`` in worker 1) must surface in the learner as a RuntimeError carrying the
    worker's traceback within seconds -- not as a queue wait without end -- and leave the agent usable for the next ``sample()``."""
    ag, ref = _rollout_agent(tmp_path)
    monkeypatch.setenv('UPAMD_ROLLOUT', 'server')
    good_step = ag.env.step
    calls = {'n': 0}

    def step(action, logger=None):
        calls['n'] += 1
        if os.getpid() != learner and calls['n'] == 3 and int(os.environ.get('UPAMD_TEST_FAIL', '0')):
            raise ValueError('geometry blew up in the worker')
        return good_step(action, logger)
    learner = os.getpid()
    ag.env.step = step
    monkeypatch.setenv('UPAMD_TEST_FAIL', '1')
    with pytest.raises(RuntimeError, match='geometry blew up in the worker'):
        ag.sample(20)
    monkeypatch.setenv('UPAMD_TEST_FAIL', '0')
    batch, log = ag.sample(20)
    assert len(batch) == 20 and log.num_episodes == 4
    ag._upamd_release_arenas()


@needs_reference
def test_server_rollout_post_mortem_names_the_signal_and_quotes_the_workers_stderr(tmp_path, monkeypatch):
    """A worker that DIES (here: SIGSEGV inside the env's ``step``) cannot report.  The learner's RuntimeError must carry what is
    needed to find the cause: which workers reported, every exit code (-11 = killed by SIGSEGV), the tail of the dead worker's
    stderr / faulthandler file (the Python stack at the fault), the serving thread's last error."""
    import signal
    ag, ref = _rollout_agent(tmp_path)
    monkeypatch.setenv('UPAMD_ROLLOUT', 'server')
    good_step = ag.env.step
    learner = os.getpid()
    calls = {'n': 0}

    def step(action, logger=None):
        calls['n'] += 1
        if os.getpid() != learner and calls['n'] == 3 and os.environ.get('UPAMD_TEST_FAIL') == 'segv':
            os.kill(os.getpid(), signal.SIGSEGV)
        return good_step(action, logger)
    ag.env.step = step
    monkeypatch.setenv('UPAMD_TEST_FAIL', 'segv')
    with pytest.raises(RuntimeError) as err:
        ag.sample(20)
    msg = str(err.value)
    assert 'without reporting' in msg and 'post-mortem' in msg
    assert 'exitcode=-11' in msg, msg
    assert 'Segmentation fault' in msg and 'in step' in msg, msg           # faulthandler's dump of the worker's Python stack
    assert 'action server: last_error=' in msg
    monkeypatch.setenv('UPAMD_TEST_FAIL', '0')
    batch, log = ag.sample(20)                                            # the agent is usable afterwards
    assert len(batch) == 20
    batch.close()


@needs_reference
def test_a_record_batch_owns_its_arenas(tmp_path, monkeypatch):
    """The batch ``sample()`` returns holds zero-copy record views and raw record addresses into shared-memory arenas.  Those
    arenas live exactly as long as the batch: another ``sample()`` does not unmap them (it used to: a batch kept across the next
    call -- a second sample before update_params, debugging, replay reuse -- then read freed memory: SIGSEGV in the interpreter or
    in the C packer threads); ``close()`` / garbage collection unmaps AND unlinks them; a closed batch is empty."""
    import gc
    from drl_urban_planning_amd import packer
    ag, ref = _rollout_agent(tmp_path)
    monkeypatch.setenv('UPAMD_ROLLOUT', 'server')
    b1, _ = ag.sample(20, mean_action=True)
    names1 = [a.name for a in b1._arenas]
    want = [bytes(r) for r in b1.states]
    b2, _ = ag.sample(20, mean_action=True)
    assert [bytes(r) for r in b1.states] == want                          # still mapped, still the same bytes
    pk = packer.pack_replay(b1.states, b1.actions, 23, 52, pin=False)     # the C packer reads b1's raw addresses
    assert pk.T == 20
    assert all(os.path.exists('/dev/shm/' + n.lstrip('/')) for n in names1)
    names2 = [a.name for a in b2._arenas]
    b2.close()
    assert b2.closed and len(b2) == 0 and not any(os.path.exists('/dev/shm/' + n.lstrip('/')) for n in names2)
    with pytest.raises(ValueError, match='empty replay'):
        packer.pack_replay(b2.states, b2.actions, 23, 52, pin=False)
    del b1, pk
    gc.collect()
    assert not any(os.path.exists('/dev/shm/' + n.lstrip('/')) for n in names1)      # collected -> unmapped and unlinked
