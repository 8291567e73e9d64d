"""Generate the committed golden vectors by running the REAL reference (build container only).

    python tests/golden/make_golden.py [case ...]

Imports /root/reference's own hot-path modules (oracle/ref_import.py recipe), builds the
reference's ``policy_net``/``value_net`` with seeded weights, feeds them seeded quirky
replays (tests/cases.py) and records:

  * per-row value / log-prob / entropy and a few encoder intermediates,
  * the four loss terms and EVERY parameter gradient of one minibatch,
  * GAE advantages/returns from ``khrylib.rl.core.estimate_advantages``,
  * a full ``UrbanPlanningAgent.update_params`` run (T rows, several epochs): the per-minibatch
    TensorBoard scalars and the parameters after it -- this exercises the cumulative numpy
    permutations, tail-drop, the first-step-only double gradient clip and torch Adam.

Outputs: tests/golden/case_*.npz (small; committed).  The GPU box has no /root/reference,
so tests there read these files only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import ref_import  # noqa: E402
import cases  # noqa: E402

CASES = {
    # name: model dims, pads, replay shape, PPO hyper-parameters
    'case_a': dict(model=dict(D=16, L=2, S=(64, 16), heads=1, land_head=(32, 1), road_head=(32, 1),
                              value_head=(32, 32, 1)),
                   max_nodes=40, max_edges=96, T=24, B=8, epochs=2, seed=3, road_fraction=0.35,
                   hyper=dict(lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=1.0, tau=0.0, clip_epsilon=0.2,
                              value_pred_coef=0.5, entropy_coef=0.01)),
    'case_b': dict(model=dict(D=32, L=3, S=(32, 16), heads=2, land_head=(16, 1), road_head=(16, 1),
                              value_head=(16, 16, 1)),
                   max_nodes=36, max_edges=80, T=20, B=5, epochs=2, seed=5, road_fraction=0.0,
                   hyper=dict(lr=1e-3, eps=1e-5, weight_decay=1e-3, gamma=0.97, tau=0.9, clip_epsilon=0.1,
                              value_pred_coef=0.5, entropy_coef=0.02)),
    'case_c': dict(model=dict(D=16, L=2, S=(64, 16), heads=1, land_head=(32, 1), road_head=(32, 1),
                              value_head=(32, 32, 1)),
                   max_nodes=30, max_edges=64, T=12, B=6, epochs=1, seed=9, road_fraction=1.0,
                   hyper=dict(lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=1.0, tau=0.0, clip_epsilon=0.2,
                              value_pred_coef=0.5, entropy_coef=0.01)),
    # the rl-mlp ablation encoder (urban_planning/models/state_encoder.py:217-308, model.py:22-33)
    'case_m': dict(model=dict(D=32, L=0, S=(64, 16), heads=1, land_head=(32, 1), road_head=(16, 1),
                              value_head=(32, 32, 1)), encoder='mlp', dead_candidate=True,
                   max_nodes=40, max_edges=96, T=24, B=8, epochs=2, seed=13, road_fraction=0.3,
                   hyper=dict(lr=4e-4, eps=1e-5, weight_decay=1e-4, gamma=0.99, tau=0.95, clip_epsilon=0.2,
                              value_pred_coef=0.5, entropy_coef=0.01)),
    # num_edge_fc_layers = 2 (urban_planning/models/state_encoder.py:59-82): a second Linear + tanh behind the first one
    'case_k': dict(model=dict(D=32, L=2, K=2, S=(64, 16), heads=2, land_head=(32, 1), road_head=(16, 1),
                              value_head=(32, 32, 1)), dead_candidate=True,
                   max_nodes=40, max_edges=96, T=24, B=8, epochs=2, seed=17, road_fraction=0.3,
                   hyper=dict(lr=4e-4, eps=1e-5, weight_decay=1e-4, gamma=0.99, tau=0.95, clip_epsilon=0.2,
                              value_pred_coef=0.5, entropy_coef=0.01)),
    # cfg.agent_specs['batch_stage'] = True (urban_planning_agent.py:273-279, 314-319): minibatches regrouped by stage, so
    # most of them hold one stage only and the other pointer head gets no gradient -- the case where torch <= 1.13
    # (zero_grad zero-fills: the idle head still takes Adam steps) and torch >= 2.0 (grad None: Adam skips it) differ.
    # `upd*` = the installed torch's semantics, `updz*` = the reference's pinned ones
    'case_s': dict(model=dict(D=16, L=2, S=(64, 16), heads=1, land_head=(32, 1), road_head=(32, 1),
                              value_head=(32, 32, 1)), batch_stage=True,
                   max_nodes=40, max_edges=96, T=28, B=6, epochs=2, seed=23, road_fraction=0.5,
                   hyper=dict(lr=1e-3, eps=1e-5, weight_decay=1e-4, gamma=0.99, tau=0.95, clip_epsilon=0.2,
                              value_pred_coef=0.5, entropy_coef=0.01)),
}


def build(ref, spec):
    m = spec['model']
    cfg = ref_import.DuckCfg(D=m['D'], L=m['L'], K=m.get('K', 1), S=m['S'], heads=m['heads'], max_nodes=spec['max_nodes'],
                             max_edges=spec['max_edges'], land_head=m['land_head'], road_head=m['road_head'],
                             value_head=m['value_head'])
    torch.manual_seed(spec['seed'])
    create = ref.create_mlp_model if spec.get('encoder') == 'mlp' else ref.create_sgnn_model
    policy_net, value_net = create(cfg, ref_import.DuckAgent())
    # default inits give tiny logits spreads; perturb so softmaxes/ratios are non-trivial
    with torch.no_grad():
        for p in ref.ActorCritic(policy_net, value_net).parameters():
            p.add_(0.15 * torch.randn_like(p))
    return cfg, policy_net, value_net


def main():
    ref = ref_import.load_reference()
    only = sys.argv[1:]                          # optional: the cases to (re)generate; default all
    for name, spec in CASES.items():
        if only and name not in only:
            continue
        cfg, policy_net, value_net = build(ref, spec)
        ac = ref.ActorCritic(policy_net, value_net)
        out = {}
        for k, v in ac.state_dict().items():
            out['sd/' + k] = v.detach().numpy().copy()
        replay = cases.quirky_replay(spec['T'], spec['max_nodes'], spec['max_edges'], seed=spec['seed'],
                                     road_fraction=spec['road_fraction'], **(dict(dead_candidate=True) if spec.get('dead_candidate') else {}))
        for k, v in cases.stack_states(replay.states).items():
            out['st/' + k] = v
        out['actions'] = replay.actions
        out['masks'] = replay.masks
        out['rewards'] = replay.rewards
        out['exps'] = replay.exps

        # ---- forward on the first B rows, with grads of one minibatch loss
        B = spec['B']
        hy = spec['hyper']
        xs = ref.tensorfy(replay.states[:B])
        actions_b = torch.from_numpy(replay.actions[:B]).float()
        value = value_net(xs)
        logp, ent = policy_net.get_log_prob_entropy(xs, actions_b)
        out['fwd/value'] = value.detach().numpy()
        out['fwd/logp'] = logp.detach().numpy()
        out['fwd/entropy'] = ent.detach().numpy()
        s_land, s_road, s_value, _, _, _ = policy_net.shared_net(xs)
        out['fwd/state_value'] = s_value.detach().numpy()
        out['fwd/h_nodes_last'] = s_road.detach().numpy()
        D = spec['model']['D']
        out['fwd/h_edges_last'] = s_land[:, :, :D].detach().numpy()

        g = torch.Generator().manual_seed(spec['seed'] + 1)
        adv_b = torch.randn(B, 1, generator=g)
        ret_b = torch.randn(B, 1, generator=g)
        old_b = logp.detach() + 0.3 * torch.randn(B, 1, generator=g)     # some ratios outside the clip range
        exps_b = torch.from_numpy(replay.exps[:B]).float()
        ag = ref_import.make_reference_agent(ref, cfg, policy_net, value_net, num_optim_epoch=spec['epochs'],
                                             mini_batch_size=B, **hy)
        ind = exps_b.nonzero(as_tuple=False).squeeze(1)
        value_loss = ag.value_loss(xs, ret_b)
        surr_loss, entropy_loss = ag.ppo_entropy_loss(xs, actions_b, adv_b, old_b, ind)
        loss = surr_loss + ag.value_pred_coef * value_loss + ag.entropy_coef * entropy_loss
        ag.optimizer.zero_grad()
        loss.backward()
        out['mb/adv'], out['mb/ret'], out['mb/old_logp'] = adv_b.numpy(), ret_b.numpy(), old_b.numpy()
        out['mb/losses'] = np.array([loss.item(), value_loss.item(), surr_loss.item(), entropy_loss.item()])
        for k, p in ac.named_parameters():
            out['grad/' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            out['gradnone/' + k] = np.array(p.grad is None)
        ag.optimizer.zero_grad()

        # ---- GAE on the reference's own function, with non-trivial gamma/tau as well
        with torch.no_grad():
            vals = torch.cat([value_net(ref.tensorfy(replay.states[i:i + B])) for i in range(0, spec['T'], B)])
        rewards = torch.from_numpy(replay.rewards).float()
        masks = torch.from_numpy(replay.masks).float()
        for tag, (ga, ta) in dict(cfg=(hy['gamma'], hy['tau']), g95=(0.99, 0.95)).items():
            adv, ret = ref.estimate_advantages(rewards, masks, vals, ga, ta)
            out['gae/%s_adv' % tag], out['gae/%s_ret' % tag] = adv.numpy(), ret.numpy()
        out['gae/values'] = vals.numpy()

        # ---- the whole update_params on fresh (re-built, identical) weights
        cfg2, policy2, value2 = build(ref, spec)
        ag2 = ref_import.make_reference_agent(ref, cfg2, policy2, value2, num_optim_epoch=spec['epochs'],
                                              mini_batch_size=B, **hy)
        np.random.seed(spec['seed'] + 11)
        ag2.update_params(replay, 0)
        out['upd/scalars'] = np.array([v for (t, v, s) in ag2.tb_logger.scalars if t in (
            'loss/loss', 'loss/value_loss', 'loss/surr_loss', 'loss/entropy_loss')]).reshape(-1, 4)
        out['upd/loss_iter'] = np.array(ag2.loss_iter)
        for k, v in ref.ActorCritic(policy2, value2).state_dict().items():
            out['upd_sd/' + k] = v.detach().numpy().copy()
        # second call on the same agent: clipping is a no-op now (generators exhausted)
        np.random.seed(spec['seed'] + 12)
        ag2.update_params(replay, 1)
        for k, v in ref.ActorCritic(policy2, value2).state_dict().items():
            out['upd2_sd/' + k] = v.detach().numpy().copy()
        out['upd2/scalars'] = np.array([v for (t, v, s) in ag2.tb_logger.scalars if t in (
            'loss/loss', 'loss/value_loss', 'loss/surr_loss', 'loss/entropy_loss')]).reshape(-1, 4)

        if spec.get('batch_stage'):
            # the same two calls with the stage regrouping on, under both zero_grad semantics
            for tag, legacy in (('upds', False), ('updz', True)):
                cfg3, policy3, value3 = build(ref, spec)
                ag3 = ref_import.make_reference_agent(ref, cfg3, policy3, value3, num_optim_epoch=spec['epochs'],
                                                      mini_batch_size=B, batch_stage=True, legacy_zero_grad=legacy, **hy)
                for call in (0, 1):
                    np.random.seed(spec['seed'] + 11 + call)
                    ag3.update_params(replay, call)
                    for k, v in ref.ActorCritic(policy3, value3).state_dict().items():
                        out['%s%s_sd/%s' % (tag, '' if call == 0 else '2', k)] = v.detach().numpy().copy()
                out[tag + '/scalars'] = np.array([v for (t, v, s) in ag3.tb_logger.scalars if t in (
                    'loss/loss', 'loss/value_loss', 'loss/surr_loss', 'loss/entropy_loss')]).reshape(-1, 4)
            st = np.array([int(np.argmax(s[8])) for s in replay.states])
            out['upds/stage_of_row'] = st

        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        print(name, 'written', os.path.getsize(path), 'bytes;', 'steps:', out['upd/scalars'].shape[0],
              'losses:', out['mb/losses'])


if __name__ == '__main__':
    main()
