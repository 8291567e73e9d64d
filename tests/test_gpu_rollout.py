"""Batched action serving on the HIP modules: forked env workers send compact state
records, the learner process answers every pending request with one GPU forward and on-device sampling; a greedy
"evaluation" client runs concurrently with the sampling clients (no CPU<->GPU round trip of the model, no serial eval
phase).  Greedy answers are compared with the oracle's arg-max; the collected arenas feed update_params directly."""
import json
import multiprocessing as mp
import os
import time

import numpy as np
import pytest
import torch

import helpers
from oracle import sgnn_oracle as orc

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'
PADS = dict(max_nodes=64, max_edges=200)


def _states(T, seed):
    from drl_urban_planning_amd import synth
    return synth.make_replay(T, 'hlg', seed=seed, road_fraction=0.4, n_range=(20, 55), **PADS)


def _worker(client, pid, n_steps, arena_name, greedy_only, q):
    from drl_urban_planning_amd import rollout
    arena = rollout.SharedArena(64, 1 << 20, name=arena_name)
    rep = _states(n_steps, 200 + pid)
    got = []
    for t, s in enumerate(rep.states):
        mean = greedy_only or (t % 2 == 0)
        a = client.select_action([s], mean).numpy().squeeze(0)
        got.append((t, bool(mean), a.copy()))
        arena.append(s, a, 0 if t == n_steps - 1 else 1, 0.1 * t, 1 - int(mean))
    q.put((pid, got))
    client.close()
    arena.close(unlink=False)


def test_gpu_action_server_with_sampling_and_eval_clients():
    from drl_urban_planning_amd import PPOUpdater, rollout
    cfg = helpers.make_cfg(D=32, L=2, heads=2, **PADS)
    policy_net, value_net, ac = helpers.build_product(cfg, seed=6)
    sd = helpers.perturbed_state_dict(ac, 7, scale=0.2)
    ac.load_state_dict(sd)
    ac.to(DEV)
    with torch.no_grad():               # the HIP library is initialised BEFORE the fork; the children never touch it
        policy_net.select_action(_states(2, 1).states, True)
    n_workers, n_steps = 4, 8
    server = rollout.ActionServer(policy_net, n_workers + 1, slot_bytes=1 << 18)
    arenas = [rollout.SharedArena(64, 1 << 20) for _ in range(n_workers + 1)]
    ctx = mp.get_context('fork')
    q = ctx.Queue()
    t0 = time.time()
    # launch() forks the workers BEFORE the serving thread exists (a fork next to a live runtime thread is unsafe);
    # the last worker is the greedy evaluation episode
    procs = server.launch(_worker, [(i, n_steps, arenas[i].name, i == n_workers, q) for i in range(n_workers + 1)], ctx)
    results = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    elapsed = time.time() - t0
    server.stop()
    st = server.stats
    assert st['requests'] == (n_workers + 1) * n_steps
    assert st['max_rows'] >= 2, 'requests of different workers were never batched: %s' % st
    P = helpers.oracle_params(sd, requires_grad=False)
    for pid, got in results.items():
        rep = _states(n_steps, 200 + pid)
        with torch.no_grad():
            land0, road0, stage0 = orc.policy_forward(P, orc.tensorfy(rep.states), 2)
        want = torch.zeros(n_steps, 2)
        want[stage0[:, 0].bool(), 0] = land0.probs.argmax(1).float()
        want[stage0[:, 1].bool(), 1] = road0.probs.argmax(1).float()
        for t, mean, a in got:
            stage = int(np.argmax(rep.states[t][8]))
            assert (rep.states[t][6] if stage == 0 else rep.states[t][7])[int(a[stage])]
            if mean:
                assert np.array_equal(a, want[t].numpy()), (pid, t, a, want[t])
    # the arenas go straight into the update (records are consumed in place)
    for a in arenas[:n_workers]:                # page-locked in the learner (hipHostRegister): the packer reads pinned memory
        assert a.pin()
        assert torch.frombuffer(a.shm.buf, dtype=torch.uint8).is_pinned()
    batch = rollout.RecordBatch([rollout.ArenaMemory(a) for a in arenas[:n_workers]])
    assert len(batch) == n_workers * n_steps
    before = {k: v.detach().clone() for k, v in ac.state_dict().items()}
    up = PPOUpdater(policy_net, value_net, num_optim_epoch=1, mini_batch_size=8)
    np.random.seed(3)
    up.update_params(batch, 0)
    assert up.last_losses.shape == (4, 4) and np.isfinite(up.last_losses).all()
    after_arena = {k: v.detach().cpu().numpy().copy() for k, v in ac.state_dict().items()}
    # the same rows as PADDED 9-field tuples (what the reference's queue would have carried) through a fresh updater
    # on the same initial weights: the losses and the updated parameters must be bit-identical -- the compact records
    # lose nothing, and the packer reads them exactly as it reads the padded form
    from drl_urban_planning_amd import packer, synth
    padded = synth.Replay([packer.expand_state(r, padded=True) for r in batch.states], batch.actions.copy(),
                          batch.masks.copy(), batch.rewards.copy(), batch.exps.copy())
    for t, s in enumerate(padded.states):
        assert s[1].shape == (PADS['max_nodes'], 23) and s[2].shape == (PADS['max_edges'], 2)
    ac.load_state_dict(before)
    up2 = PPOUpdater(policy_net, value_net, num_optim_epoch=1, mini_batch_size=8)
    np.random.seed(3)
    up2.update_params(padded, 0)
    assert np.array_equal(up2.last_losses, up.last_losses)
    for k, v in ac.state_dict().items():
        assert np.array_equal(v.detach().cpu().numpy(), after_arena[k]), k
    # ... and they are the ORACLE's losses on those padded rows (first step; the sampled actions came from the server)
    P0 = helpers.oracle_params({k: v.cpu() for k, v in before.items()})
    ou = orc.OracleUpdater(P0, num_optim_epoch=1, mini_batch_size=8, num_heads=2)
    np.random.seed(3)
    ou.update_params(padded)
    np.testing.assert_allclose(up.last_losses, np.array(ou.loss_log), rtol=2e-4, atol=5e-6)
    print('served %d requests in %d batches (largest %d rows) in %.2f s' % (st['requests'], st['batches'], st['max_rows'], elapsed))
    server.close()
    for a in arenas:
        a.close()


def test_select_actions_kernel_is_the_reference_categorical(monkeypatch):
    """upamd_select_actions (one launch: arg-max or inverse-CDF draw per row over the row's RAGGED candidates) against the
    reference's semantics (policy.py:67-85): greedy rows equal the oracle's arg-max over the PADDED Categorical and round 5's
    torch route exactly; sampled rows are always a live candidate, and their frequencies over 4096 draws of one state follow the
    oracle's probabilities (every slot within 5 standard deviations)."""
    cfg = helpers.make_cfg(D=32, L=2, heads=2, **PADS)
    policy_net, value_net, ac = helpers.build_product(cfg, seed=3)
    sd = helpers.perturbed_state_dict(ac, 4, scale=0.3)
    ac.load_state_dict(sd)
    ac.to(DEV)
    backend = policy_net._backend[0]
    rep = _states(48, 77)
    P = helpers.oracle_params(sd, requires_grad=False)
    with torch.no_grad():
        land0, road0, stage0 = orc.policy_forward(P, orc.tensorfy(rep.states), 2)
    want = torch.zeros(48, 2)
    want[stage0[:, 0].bool(), 0] = land0.probs.argmax(1).float()
    want[stage0[:, 1].bool(), 1] = road0.probs.argmax(1).float()
    greedy = backend.serve_actions(rep.states, np.ones(48, dtype=bool))
    assert np.array_equal(greedy, want.numpy())
    monkeypatch.setenv('UPAMD_SERVE_SELECT', 'torch')
    assert np.array_equal(backend.serve_actions(rep.states, np.ones(48, dtype=bool)), greedy)
    monkeypatch.setenv('UPAMD_SERVE_SELECT', 'hip')
    # mixed request: even rows greedy, odd rows sampled
    flags = np.arange(48) % 2 == 0
    mixed = backend.serve_actions(rep.states, flags)
    assert np.array_equal(mixed[flags], greedy[flags])
    for s, a in zip(rep.states, mixed):
        stage = int(np.argmax(s[8]))
        assert (s[6] if stage == 0 else s[7])[int(a[stage])] and a[1 - stage] == 0
    # the distribution of the draws: one land-use state and one road state, 4096 draws each
    li = int(np.flatnonzero(stage0[:, 0].numpy())[0])
    ri = int(np.flatnonzero(stage0[:, 1].numpy())[0])
    lrow = int(stage0[:li + 1, 0].sum()) - 1            # row of that state inside land0 / road0
    rrow = int(stage0[:ri + 1, 1].sum()) - 1
    N = 4096
    torch.manual_seed(11)
    for src, col, probs in ((li, 0, land0.probs[lrow]), (ri, 1, road0.probs[rrow])):
        draws = np.concatenate([backend.serve_actions([rep.states[src]] * 512, np.zeros(512, dtype=bool))[:, col] for _ in range(N // 512)])
        freq = np.bincount(draws.astype(np.int64), minlength=probs.numel())[:probs.numel()] / N
        p = probs.numpy().astype(np.float64)
        assert freq[p == 0].sum() == 0                   # a masked slot is never drawn
        sigma = np.sqrt(np.maximum(p * (1 - p), 1e-12) / N)
        assert (np.abs(freq - p) <= 5 * sigma + 1e-3).all(), float(np.abs(freq - p).max())
        assert len(np.unique(draws)) > 1


# ------------------------------------------------------------------------------------------------ bound into the agent class
class _ReferenceLikeAgent:
    """The methods of ``UrbanPlanningAgent`` / ``Agent`` the rollout + checkpoint binding wraps, restated with the reference's
    signatures and file format (urban_planning_agent.py:49-91 sample_worker, :153-194 load / save_checkpoint, :402-467
    eval_agent; khrylib/rl/agents/agent.py:75-100 sample is never reached in server mode) -- the REAL class is exercised on
    the CPU in tests/test_reference_binding.py, where /root/reference exists; this stand-in carries the same contract to
    the GPU box."""

    def __init__(self, cfg, env, policy_net, value_net, actor_critic, hy, num_optim_epoch, mini_batch_size, num_threads):
        import logging
        from oracle.ref_import import ScalarLog
        self.cfg, self.env, self.training, self.loss_iter = cfg, env, True, 0
        self.node_dim, self.numerical_feature_size = 23, 52
        self.policy_net, self.value_net, self.actor_critic_net = policy_net, value_net, actor_critic
        self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=hy['lr'], eps=hy['eps'], weight_decay=hy['weight_decay'])
        self.gamma, self.tau, self.clip_epsilon = hy['gamma'], hy['tau'], hy['clip_epsilon']
        self.value_pred_coef, self.entropy_coef = hy['value_pred_coef'], hy['entropy_coef']
        self.opt_num_epochs, self.mini_batch_size, self.num_threads = num_optim_epoch, mini_batch_size, num_threads
        self.noise_rate, self.sample_modules, self.logger_kwargs = 1.0, [policy_net], {}
        self.logger_cls = _EpisodeLog
        self.tb_logger, self.logger = ScalarLog(), logging.getLogger('upamd-gpu-test')
        self.thread_loggers = [self.logger] * num_threads
        self.best_rewards, self.best_plans, self.current_rewards, self.current_plans, self.save_best_flag = -1000.0, [], -1000.0, [], False

    def seed_worker(self, pid):
        if pid > 0:
            torch.manual_seed(torch.randint(0, 5000, (1,)) * pid)
            np.random.seed(np.random.randint(5000) * pid)

    def push_memory(self, memory, state, action, mask, next_state, reward, exp):
        memory.push(state, action, mask, next_state, reward, exp)

    def sample_worker(self, pid, queue, num_samples, mean_action):
        self.seed_worker(pid)
        memory = Memory()                       # noqa: F821 -- a module global in the reference too (the binding swaps it in the child)
        logger = self.logger_cls(**self.logger_kwargs)
        while logger.num_steps < num_samples:
            state = self.env.reset()
            messages = []
            for t in range(10000):
                state_var = [[torch.tensor(x) for x in state]]
                use_mean_action = mean_action or torch.bernoulli(torch.tensor([1 - self.noise_rate])).item()
                action = self.policy_net.select_action(state_var, use_mean_action).numpy().squeeze(0)
                next_state, reward, done, info = self.env.step(action, self.thread_loggers[pid])
                messages.append([state, action, 0 if done else 1, next_state, reward, 1 - use_mean_action])
                if done:
                    break
                state = next_state
            logger.num_episodes += 1
            for m in messages:
                logger.num_steps += 1
                logger.total_reward += m[4]
                self.push_memory(memory, *m)
        if queue is not None:
            queue.put([pid, memory, logger])
        else:
            return memory, logger

    def eval_agent(self, num_samples=1, mean_action=True, visualize=False):
        self.env.eval()
        logger = self.logger_cls()
        while logger.num_steps < num_samples:
            state = self.env.reset()
            for t in range(1, 10000):
                action = self.policy_net.select_action([[torch.tensor(x) for x in state]], mean_action).numpy().squeeze(0)
                state, reward, done, info = self.env.step(action, self.logger)
                logger.num_steps += 1
                logger.total_reward += reward
                if done:
                    break
            logger.num_episodes += 1
        self.env.train()
        return self.logger_cls.merge([logger])

    def save_checkpoint(self, iteration):
        import pickle
        cp = {'actor_critic_dict': {k: v.detach().cpu() for k, v in self.actor_critic_net.state_dict().items()},
              'loss_iter': self.loss_iter, 'best_rewards': self.best_rewards, 'iteration': iteration}
        if self.cfg.save_model_interval > 0 and (iteration + 1) % self.cfg.save_model_interval == 0:       # (:186)
            with open('%s/iteration_%04d.p' % (self.cfg.model_dir, iteration + 1), 'wb') as fh:
                pickle.dump(cp, fh)

    def load_checkpoint(self, checkpoint, restore_best_rewards):
        import pickle
        cp = pickle.load(open('%s/iteration_%04d.p' % (self.cfg.model_dir, checkpoint), 'rb'))
        self.actor_critic_net.load_state_dict(cp['actor_critic_dict'])
        self.loss_iter = cp['loss_iter']
        return cp['iteration'] + 1

    def update_params(self, batch, iteration):
        raise AssertionError('the reference update ran')


class _EpisodeLog:
    def __init__(self, **kw):
        self.num_steps, self.num_episodes, self.total_reward, self.sample_time = 0, 0, 0.0, 0.0

    @classmethod
    def merge(cls, logs, **kw):
        out = cls()
        out.num_steps, out.num_episodes = sum(x.num_steps for x in logs), sum(x.num_episodes for x in logs)
        out.total_reward = sum(x.total_reward for x in logs)
        out.avg_episode_reward = out.total_reward / max(out.num_episodes, 1)
        return out


def _bound_agent(tmp_path, name, num_threads=3, seed=None, load_sd=True):
    import types
    from drl_urban_planning_amd.binding import bind_reference_agent
    from stub_env import StubCityEnv
    from test_oracle_golden import CASE_B, CASE_EPOCHS, CASE_HYPER
    z, sd, states = helpers.load_case(name)
    kw = helpers.CASE_MODEL[name]
    cfg = helpers.make_cfg(**kw)
    cfg.model_dir = str(tmp_path)
    cfg.save_model_interval = 1
    policy_net, value_net, ac = helpers.build_product(cfg, seed=0 if seed is None else seed)
    if load_sd:
        ac.load_state_dict(sd)
    ac.to(DEV)
    cls = bind_reference_agent(_ReferenceLikeAgent)
    env = StubCityEnv(max_nodes=kw['max_nodes'], max_edges=kw['max_edges'], episode_len=6, pool=48, seed=8)
    return cls(cfg, env, policy_net, value_net, ac, CASE_HYPER[name], CASE_EPOCHS[name], CASE_B[name], num_threads), z, states


def test_bound_agent_samples_through_the_server_and_updates_on_the_records(tmp_path, monkeypatch):
    """UPAMD_ROLLOUT=server on the bound class with the HIP modules: ``sample()`` -> forked ``sample_worker`` children behind
    action clients (ONE batched GPU forward per serving round) -> RecordBatch -> ``update_params`` on the arenas; greedy
    evaluation behind a client equals the in-process greedy episode; UPAMD_EVAL=overlap runs it inside the sampling phase."""
    from drl_urban_planning_amd import rollout
    from stub_env import StubCityEnv
    ag, z, _ = _bound_agent(tmp_path, 'case_a')
    with torch.no_grad():                   # the HIP library is initialised before the first fork
        ag.policy_net.select_action(StubCityEnv(max_nodes=40, max_edges=96).states[:2], True)
    monkeypatch.setenv('UPAMD_ROLLOUT', 'server')
    monkeypatch.setenv('UPAMD_ROLLOUT_TIMEOUT_S', '240')       # a stuck serving phase fails WITH its post-mortem instead of hanging the suite
    # (no retry: a failure of the serving phase raises with every worker's exit code, stderr / faulthandler tail and the serving
    # thread's traceback in the message -- rollout_binding._upamd_failure_report)
    batch, log = ag.sample(36)              # 3 workers x 12 steps = 2 episodes of 6 each
    assert isinstance(batch, rollout.RecordBatch) and len(batch) == 36 and log.num_episodes == 6
    st = ag._upamd_server_stats
    assert st['requests'] == 36 and st['rows'] == 36, st        # (how many requests share a round is a matter of timing)
    if os.environ.get('UPAMD_TEST_STATS'):      # flake loop (tools/r06/flake_loop.sh): would round 5's `max_rows >= 2` have held in THIS run?
        with open(os.environ['UPAMD_TEST_STATS'], 'a') as fh:
            fh.write(json.dumps({'test': 'bound_agent_first_sample', 'requests': int(st['requests']), 'batches': int(st['batches']),
                                 'max_rows': int(st['max_rows'])}) + '\n')
    for rec, a in zip(batch.states, batch.actions):
        s = __import__('drl_urban_planning_amd').packer.expand_state(rec, padded=True)
        stage = int(np.argmax(s[8]))
        assert (s[6] if stage == 0 else s[7])[int(a[stage])] and a[1 - stage] == 0
    # a batch owns its arenas: a SECOND sample() before the update must leave the first batch readable (the arenas used to be
    # unmapped by the next sample() call underneath the zero-copy record views and the raw addresses the C packer reads)
    first_bytes = [bytes(r[:64]) for r in batch.states]
    ag.env.episode = -1
    batch2, _ = ag.sample(36)
    assert [bytes(r[:64]) for r in batch.states] == first_bytes and not batch.closed
    batch2.close()
    assert len(batch2) == 0 and batch2.closed
    del batch2
    np.random.seed(5)
    ag.update_params(batch, 0)
    losses = ag._hip_updater().last_losses
    assert losses.shape[0] > 0 and np.isfinite(losses).all()
    # greedy evaluation behind a client == the ORACLE's arg-max on the weights the update has just produced.  The stub environment's
    # states do not depend on the actions, so the evaluated episode is states[0 .. episode_len) and its reward is a function of the
    # six greedy actions.  (Two fp32 soft-max routes may break a near-tie differently: the comparison is exact wherever the
    # oracle's top-two probabilities are further apart than 1e-5, and such a tie is reported, not hidden.)
    kw = helpers.CASE_MODEL['case_a']
    P = helpers.oracle_params({k: v.detach().cpu() for k, v in ag.actor_critic_net.state_dict().items()}, requires_grad=False)
    ep_states = [ag.env.states[t % len(ag.env.states)] for t in range(ag.env.episode_len)]
    with torch.no_grad():
        land0, road0, stage0 = orc.policy_forward(P, orc.tensorfy(ep_states), kw['heads'])
    acc, margin, seen = 0.0, float('inf'), [0, 0]
    for t in range(len(ep_states)):
        col = 0 if bool(stage0[t, 0]) else 1
        probs = (land0 if col == 0 else road0).probs[seen[col]]         # (a head's Categorical holds the rows of ITS stage only)
        seen[col] += 1
        top = torch.topk(probs, 2).values
        margin = min(margin, float(top[0] - top[1]))
        a = np.zeros(2)
        a[col] = float(probs.argmax())
        acc += 0.01 * float(a.sum())
    ag.env.episode = -1
    got = ag.eval_agent(num_samples=1, mean_action=True)
    assert ag._upamd_server_stats['requests'] == 6 and got.num_episodes == 1
    assert margin > 1e-5, 'the evaluated episode has a near-tie (top-2 margin %.2e): pick another seed for this test' % margin
    assert got.total_reward == 1.0 + acc, (got.total_reward, 1.0 + acc)
    monkeypatch.setenv('UPAMD_EVAL', 'overlap')
    ag.env.episode = -1
    batch, log = ag.sample(36)
    assert ag._upamd_server_stats['requests'] == 42 and ag._upamd_eval_ahead is not None
    ahead = ag.eval_agent(num_samples=1, mean_action=True)
    assert ahead.num_episodes == 1 and ag._upamd_server_stats['requests'] == 42 and ag._upamd_eval_ahead is None
    ag._upamd_release_arenas()


@pytest.mark.parametrize('name', ['case_a'])
def test_bound_agent_checkpoint_resumes_adam_on_the_reference_trajectory(name, tmp_path):
    """save_checkpoint / load_checkpoint of the bound class carry 'hip_optimizer': first update_params -> save -> a FRESH
    agent (other initial weights, no optimizer history) -> load -> second update_params must land on the parameters the
    uninterrupted REFERENCE run reaches after its second call (golden upd2_sd), which restarting Adam does not."""
    import pickle
    from drl_urban_planning_amd import synth
    from test_oracle_golden import CASE_SEED
    ag, z, states = _bound_agent(tmp_path, name)
    replay = synth.Replay(states, z['actions'], z['masks'], z['rewards'], z['exps'])
    np.random.seed(CASE_SEED[name] + 11)
    ag.update_params(replay, 0)
    ag.save_checkpoint(0)
    cp = pickle.load(open('%s/iteration_0001.p' % tmp_path, 'rb'))
    assert cp['hip_optimizer']['clip_pending'] is False and sum(cp['hip_optimizer']['group_steps']) > 0
    assert cp['loss_iter'] == int(z['upd/loss_iter'])

    def second_call(with_optimizer):
        fresh, _, _ = _bound_agent(tmp_path, name, seed=99, load_sd=False)
        if not with_optimizer:              # what the reference's own checkpoint holds
            stripped = dict(cp)
            stripped.pop('hip_optimizer')
            pickle.dump(stripped, open('%s/iteration_0002.p' % tmp_path, 'wb'))
        assert fresh.load_checkpoint(1 if with_optimizer else 2, True) == 1 and fresh.loss_iter == cp['loss_iter']
        np.random.seed(CASE_SEED[name] + 12)
        fresh.update_params(replay, 1)
        mine = {k: v.detach().cpu().numpy() for k, v in fresh.actor_critic_net.state_dict().items()}
        return max(float(np.linalg.norm(mine[k] - z['upd2_sd/' + k]) / max(np.linalg.norm(z['upd2_sd/' + k]), 1e-30)) for k in mine)

    assert second_call(True) <= 2e-4
    assert second_call(False) > 5e-4        # Adam restarted (and the first-step clip re-applied): a different trajectory
