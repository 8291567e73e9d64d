"""The rl-mlp ablation encoder (MLPStateEncoder, urban_planning/models/state_encoder.py:217-308; ``--agent rl-mlp``,
urban_planning/train.py:18) on the HIP engine: forward rows, the four loss terms, every gradient and two whole
update_params calls against vectors produced by the REAL reference (``create_mlp_model``), plus a wider random model
against the oracle.  Needs a real MI355X (-m gpu)."""
import numpy as np
import pytest
import torch

import cases
import helpers
from oracle import sgnn_oracle as orc
from test_gpu_parity import DEV, _check_grads, _forward, _rel_l2

pytestmark = pytest.mark.gpu


def _setup(cfg, sd, states, actions):
    from drl_urban_planning_amd import packer
    from drl_urban_planning_amd.models import backend_of
    policy_net, value_net, ac = helpers.build_product(cfg, mlp=True)
    ac.load_state_dict(sd)
    ac.to(DEV)
    backend = backend_of(policy_net)
    eng = backend.engine(torch.device(DEV))
    flat = eng.flatten(backend.named_params())
    pk = packer.pack_replay(states, actions, 23, 52).to(DEV)
    sched = packer.Schedule(pk, [np.arange(len(states))], DEV)
    mb, _ = sched.minibatch(0)
    mb._keepalive = sched            # the Minibatch struct holds raw device pointers into the schedule tensor
    return policy_net, value_net, ac, eng, flat, pk, mb


def test_mlp_forward_loss_and_gradients_match_reference():
    from test_oracle_golden import CASE_HYPER
    name = 'case_m'
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    hy = CASE_HYPER[name]
    B = z['fwd/value'].shape[0]
    _, _, _, eng, flat, pk, mb = _setup(cfg, sd, states[:B], z['actions'][:B])
    value, logp, ent = _forward(eng, pk, mb, flat)
    ns = pk.meta[:B, 0]
    offs = np.concatenate([[0], np.cumsum(ns)])
    H0 = eng.ws_tensor(mb, 'H0').cpu().numpy()
    for b in range(B):
        assert np.abs(H0[offs[b]:offs[b + 1]] - z['fwd/h_nodes_last'][b, :ns[b]]).max() < 2e-5
    sv = eng.ws_tensor(mb, 'SV').cpu().numpy()
    Wd = z['fwd/state_value'].shape[1]
    assert np.abs(sv[:, :Wd] - z['fwd/state_value']).max() < 2e-5 and not sv[:, Wd:].any()
    np.testing.assert_allclose(value.cpu().numpy(), z['fwd/value'][:, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), z['fwd/logp'][:, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), z['fwd/entropy'][:, 0], rtol=1e-4, atol=1e-5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    dvalue, dlogp, dent = (torch.empty(B, device=DEV) for _ in range(3))
    losses = torch.zeros(4, device=DEV)
    nind = int((z['exps'][:B] != 0).sum())
    eng.ppo_loss(B, value, logp, ent, t(z['mb/adv'][:, 0]), t(z['mb/ret'][:, 0]), t(z['mb/old_logp'][:, 0]), t(z['exps'][:B]),
                 hy['clip_epsilon'], hy['value_pred_coef'], hy['entropy_coef'], 1.0 / B, 1.0 / nind, dvalue, dlogp, dent, losses)
    np.testing.assert_allclose(losses.cpu().numpy(), z['mb/losses'], rtol=2e-5, atol=2e-6)
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dvalue, dlogp, dent, grads)
    torch.cuda.synchronize()
    _check_grads(eng, grads, lambda nm: z[helpers.golden_key('grad/', nm)])


def test_mlp_update_params_matches_reference():
    from drl_urban_planning_amd import PPOUpdater, synth
    from test_oracle_golden import CASE_B, CASE_EPOCHS, CASE_HYPER, CASE_SEED
    name = 'case_m'
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    hy = CASE_HYPER[name]
    policy_net, value_net, ac = helpers.build_product(cfg, mlp=True)
    ac.load_state_dict(sd)
    ac.to(DEV)
    up = PPOUpdater(policy_net, value_net, lr=hy['lr'], eps=hy['eps'], weight_decay=hy['weight_decay'], gamma=hy['gamma'],
                    tau=hy['tau'], clip_epsilon=hy['clip_epsilon'], value_pred_coef=hy['value_pred_coef'],
                    entropy_coef=hy['entropy_coef'], num_optim_epoch=CASE_EPOCHS[name], mini_batch_size=CASE_B[name])
    replay = synth.Replay(states, z['actions'], z['masks'], z['rewards'], z['exps'])
    np.random.seed(CASE_SEED[name] + 11)
    up.update_params(replay, 0)
    np.testing.assert_allclose(up.last_losses, z['upd/scalars'], rtol=1e-4, atol=2e-6)
    for k, v in ac.state_dict().items():
        assert _rel_l2(v.detach().cpu().numpy(), z['upd_sd/' + k]) <= 1e-4, k
    np.random.seed(CASE_SEED[name] + 12)
    up.update_params(replay, 1)
    for k, v in ac.state_dict().items():
        assert _rel_l2(v.detach().cpu().numpy(), z['upd2_sd/' + k]) <= 2e-4, k


@pytest.mark.parametrize('D,T,road_fraction', [(64, 12, 0.3), (16, 7, 1.0), (256, 6, 0.0)])
def test_mlp_wide_model_matches_oracle(D, T, road_fraction):
    """Random weights, mixed stages, dead candidates, self-loop / duplicate edges: values, losses, every gradient."""
    cfg = helpers.make_cfg(D=D, L=0, max_nodes=60, max_edges=150)
    _, _, ac = helpers.build_product(cfg, seed=31, mlp=True)
    sd = helpers.perturbed_state_dict(ac, 32, scale=0.1)
    replay = cases.quirky_replay(T, 60, 150, seed=17, road_fraction=road_fraction, dead_candidate=True, full_row=False)
    _, _, _, eng, flat, pk, mb = _setup(cfg, sd, replay.states, replay.actions)
    value, logp, ent = _forward(eng, pk, mb, flat)
    P = helpers.oracle_params(sd)
    assert orc.is_mlp_params(P)
    xs = orc.tensorfy(replay.states)
    act_t = torch.from_numpy(replay.actions).float()
    g = torch.Generator().manual_seed(5)
    adv, ret = torch.randn(T, 1, generator=g), torch.randn(T, 1, generator=g)
    with torch.no_grad():
        lp0, _ = orc.get_log_prob_entropy(P, xs, act_t)
    old = lp0 + 0.3 * torch.randn(T, 1, generator=g)
    exps = torch.ones(T)
    loss, vl, sl, el = orc.ppo_losses(P, xs, act_t, adv, ret, old, exps, 0.2, 0.5, 0.01)
    loss.backward()
    np.testing.assert_allclose(logp.cpu().numpy(), lp0[:, 0].numpy(), rtol=1e-4, atol=1e-5)
    dvalue, dlogp, dent = (torch.empty(T, device=DEV) for _ in range(3))
    losses = torch.zeros(4, device=DEV)
    eng.ppo_loss(T, value, logp, ent, adv[:, 0].to(DEV), ret[:, 0].to(DEV), old[:, 0].to(DEV), exps.to(DEV), 0.2, 0.5, 0.01,
                 1.0 / T, 1.0 / T, dvalue, dlogp, dent, losses)
    np.testing.assert_allclose(losses.cpu().numpy(), [loss.item(), vl.item(), sl.item(), el.item()], rtol=1e-4, atol=1e-5)
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dvalue, dlogp, dent, grads)
    torch.cuda.synchronize()
    _check_grads(eng, grads, lambda nm: (P[nm].grad if P[nm].grad is not None else torch.zeros_like(P[nm])).numpy())
