"""Pins the oracle (oracle/sgnn_oracle.py) against golden vectors produced by the REAL
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import sgnn_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASE_HEADS = {'case_a': 1, 'case_b': 2, 'case_c': 1, 'case_m': 1, 'case_k': 2, 'case_s': 1}
CASE_B = {'case_a': 8, 'case_b': 5, 'case_c': 6, 'case_m': 8, 'case_k': 8, 'case_s': 6}
CASE_EPOCHS = {'case_a': 2, 'case_b': 2, 'case_c': 1, 'case_m': 2, 'case_k': 2, 'case_s': 2}
CASE_SEED = {'case_a': 3, 'case_b': 5, 'case_c': 9, 'case_m': 13, 'case_k': 17, 'case_s': 23}
CASE_HYPER = {
    'case_a': dict(lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=1.0, tau=0.0, clip_epsilon=0.2,
                   value_pred_coef=0.5, entropy_coef=0.01),
    'case_b': dict(lr=1e-3, eps=1e-5, weight_decay=1e-3, gamma=0.97, tau=0.9, clip_epsilon=0.1,
                   value_pred_coef=0.5, entropy_coef=0.02),
    'case_c': dict(lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=1.0, tau=0.0, clip_epsilon=0.2,
                   value_pred_coef=0.5, entropy_coef=0.01),
    # the rl-mlp ablation encoder (state_encoder.py:217-308), generated with the reference's create_mlp_model
    'case_m': dict(lr=4e-4, eps=1e-5, weight_decay=1e-4, gamma=0.99, tau=0.95, clip_epsilon=0.2,
                   value_pred_coef=0.5, entropy_coef=0.01),
    # num_edge_fc_layers = 2 (state_encoder.py:59-82)
    'case_k': dict(lr=4e-4, eps=1e-5, weight_decay=1e-4, gamma=0.99, tau=0.95, clip_epsilon=0.2,
                   value_pred_coef=0.5, entropy_coef=0.01),
    # cfg.agent_specs['batch_stage'] = True (urban_planning_agent.py:273-279, 314-319), both zero_grad semantics
    'case_s': dict(lr=1e-3, eps=1e-5, weight_decay=1e-4, gamma=0.99, tau=0.95, clip_epsilon=0.2,
                   value_pred_coef=0.5, entropy_coef=0.01),
}


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    states = cases.unstack_states({k[3:]: z[k] for k in z.files if k.startswith('st/')})
    return z, sd, states


@pytest.mark.parametrize('name', ['case_a', 'case_b', 'case_c', 'case_m', 'case_k'])
def test_forward_matches_reference(name):
    z, sd, states = load_case(name)
    P = orc.leaf_params(orc.split_actor_critic_state_dict(sd), requires_grad=False)
    B = CASE_B[name]
    H = CASE_HEADS[name]
    xs = orc.tensorfy(states[:B])
    actions = torch.from_numpy(z['actions'][:B]).float()
    keep = {}
    value = orc.value_forward(P, xs, H, keep)
    logp, ent = orc.get_log_prob_entropy(P, xs, actions, H)
    np.testing.assert_allclose(value.numpy(), z['fwd/value'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(logp.numpy(), z['fwd/logp'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ent.numpy(), z['fwd/entropy'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(keep['state_value'].numpy(), z['fwd/state_value'], rtol=1e-6, atol=1e-6)
    L = max(int(k.split('_')[-1]) for k in keep if k.startswith('h_nodes_') and k[-1].isdigit())
    np.testing.assert_allclose(keep['h_nodes_%d' % L].numpy(), z['fwd/h_nodes_last'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(keep['h_edges_%d' % L].numpy(), z['fwd/h_edges_last'], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name', ['case_a', 'case_b', 'case_c', 'case_m', 'case_k'])
def test_minibatch_losses_and_grads_match_reference(name):
    z, sd, states = load_case(name)
    P = orc.leaf_params(orc.split_actor_critic_state_dict(sd))
    B = CASE_B[name]
    hy = CASE_HYPER[name]
    loss, vl, sl, el = orc.ppo_losses(
        P, orc.tensorfy(states[:B]), torch.from_numpy(z['actions'][:B]).float(), torch.from_numpy(z['mb/adv']),
        torch.from_numpy(z['mb/ret']), torch.from_numpy(z['mb/old_logp']), torch.from_numpy(z['exps'][:B]).float(),
        hy['clip_epsilon'], hy['value_pred_coef'], hy['entropy_coef'], CASE_HEADS[name])
    np.testing.assert_allclose([loss.item(), vl.item(), sl.item(), el.item()], z['mb/losses'], rtol=2e-6, atol=1e-7)
    loss.backward()
    for k, p in P.items():
        ref_key = 'grad/actor_net.' + k if not k.startswith('value_head.') else 'grad/value_net.' + k
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        np.testing.assert_allclose(g.numpy(), z[ref_key], rtol=1e-5, atol=2e-7, err_msg=k)


@pytest.mark.parametrize('name', ['case_a', 'case_b'])
def test_gae_matches_reference(name):
    z, _, _ = load_case(name)
    hy = CASE_HYPER[name]
    rewards = torch.from_numpy(z['rewards']).float()
    masks = torch.from_numpy(z['masks']).float()
    values = torch.from_numpy(z['gae/values'])
    for tag, (ga, ta) in dict(cfg=(hy['gamma'], hy['tau']), g95=(0.99, 0.95)).items():
        adv, ret = orc.estimate_advantages(rewards, masks, values, ga, ta)
        assert np.array_equal(adv.numpy(), z['gae/%s_adv' % tag])
        assert np.array_equal(ret.numpy(), z['gae/%s_ret' % tag])


@pytest.mark.parametrize('name', ['case_a', 'case_b', 'case_c', 'case_m', 'case_k', 'case_s'])
def test_update_params_matches_reference(name):
    """Full update_params: permutation schedule, tail drop, first-step double clip, Adam."""
    z, sd, states = load_case(name)
    hy = CASE_HYPER[name]
    P = orc.leaf_params(orc.split_actor_critic_state_dict(sd))
    up = orc.OracleUpdater(P, num_optim_epoch=CASE_EPOCHS[name], mini_batch_size=CASE_B[name],
                           num_heads=CASE_HEADS[name], **hy)
    from drl_urban_planning_amd import synth
    replay = synth.Replay(states, z['actions'], z['masks'], z['rewards'], z['exps'])
    np.random.seed(CASE_SEED[name] + 11)
    up.update_params(replay)
    got = np.array(up.loss_log)
    np.testing.assert_allclose(got, z['upd/scalars'], rtol=2e-5, atol=1e-6)
    assert got.shape[0] == int(z['upd/loss_iter'])
    for k, p in P.items():
        ref_key = 'upd_sd/actor_net.' + k if not k.startswith('value_head.') else 'upd_sd/value_net.' + k
        np.testing.assert_allclose(p.detach().numpy(), z[ref_key], rtol=1e-5, atol=1e-6, err_msg=k)
    np.random.seed(CASE_SEED[name] + 12)
    up.update_params(replay)
    for k, p in P.items():
        ref_key = 'upd2_sd/actor_net.' + k if not k.startswith('value_head.') else 'upd2_sd/value_net.' + k
        np.testing.assert_allclose(p.detach().numpy(), z[ref_key], rtol=1e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('tag,legacy', [('upds', False), ('updz', True)])
def test_batch_stage_update_matches_reference(tag, legacy):
    """Stage-regrouped minibatches (urban_planning_agent.py:273-279, 314-319) under both zero_grad semantics: the
    installed torch's (grad None: Adam skips the pointer head that saw no row) and the reference's pinned
    torch <= 1.13 (zero-filled: the idle head still steps).  Two update_params calls on the reference itself."""
    name = 'case_s'
    z, sd, states = load_case(name)
    P = orc.leaf_params(orc.split_actor_critic_state_dict(sd))
    up = orc.OracleUpdater(P, num_optim_epoch=CASE_EPOCHS[name], mini_batch_size=CASE_B[name], num_heads=1,
                           batch_stage=True, legacy_zero_grad=legacy, **CASE_HYPER[name])
    from drl_urban_planning_amd import synth
    replay = synth.Replay(states, z['actions'], z['masks'], z['rewards'], z['exps'])
    for call, sdkey in ((0, tag + '_sd/'), (1, tag + '2_sd/')):
        np.random.seed(CASE_SEED[name] + 11 + call)
        up.update_params(replay)
        for k, p in P.items():
            ref_key = sdkey + ('actor_net.' if not k.startswith('value_head.') else 'value_net.') + k
            np.testing.assert_allclose(p.detach().numpy(), z[ref_key], rtol=1e-5, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(np.array(up.loss_log), z[tag + '/scalars'], rtol=2e-5, atol=1e-6)
    # the two semantics really differ on this case (else the test above proves nothing about the flag)
    other = 'updz' if tag == 'upds' else 'upds'
    assert np.abs(z[tag + '/scalars'] - z[other + '/scalars']).max() > 1e-3


def test_batch_stage_rejects_rows_of_other_stages():
    """get_perm_batch_stage indexes a two-entry list with stage.argmax(): a 'done' row raises IndexError."""
    z, sd, states = load_case('case_s')
    P = orc.leaf_params(orc.split_actor_critic_state_dict(sd))
    up = orc.OracleUpdater(P, num_optim_epoch=1, mini_batch_size=6, batch_stage=True)
    from drl_urban_planning_amd import synth
    states = [list(s) for s in states[:12]]
    states[5][8] = np.array([0, 0, 1], dtype=np.float32)
    replay = synth.Replay(states, z['actions'][:12], z['masks'][:12], z['rewards'][:12], z['exps'][:12])
    with pytest.raises(IndexError):
        up.update_params(replay)
