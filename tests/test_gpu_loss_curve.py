"""Loss-curve parity over >= 200 optimizer steps: the same replay, the same injected
numpy permutation sequence, HIP `PPOUpdater.update_params` vs the oracle's restatement of the reference's
`update_params`; per-step loss terms within 1 %, final parameters close.  T is not a multiple of the minibatch
size, so the tail-drop rule (urban_planning_agent.py:321) is exercised on every epoch; the first step is the
only clipped one.  Needs a GPU."""
import numpy as np
import pytest
import torch

import cases
import helpers
from oracle import sgnn_oracle as orc

pytestmark = pytest.mark.gpu


def test_loss_curve_224_steps():
    from drl_urban_planning_amd import PPOUpdater
    cfg = helpers.make_cfg(D=16, L=2, max_nodes=28, max_edges=60)
    policy_net, value_net, ac = helpers.build_product(cfg, seed=4)
    sd = helpers.perturbed_state_dict(ac, 5, scale=0.1)
    ac.load_state_dict(sd)
    ac.to('cuda:0')
    T, B, epochs, iters = 70, 8, 4, 7          # floor(70/8) = 8 minibatches per epoch, 6 rows dropped each time
    replay = cases.quirky_replay(T, 28, 60, seed=21, road_fraction=0.3, n_lo=10, full_row=False)
    hy = dict(lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=0.99, tau=0.95, clip_epsilon=0.2, value_pred_coef=0.5,
              entropy_coef=0.01)
    up = PPOUpdater(policy_net, value_net, num_optim_epoch=epochs, mini_batch_size=B, **hy)
    P = helpers.oracle_params(sd)
    ou = orc.OracleUpdater(P, num_optim_epoch=epochs, mini_batch_size=B, **hy)
    mine = []
    for it in range(iters):
        np.random.seed(100 + it)
        up.update_params(replay, it)
        mine.append(up.last_losses.copy())
        np.random.seed(100 + it)
        ou.update_params(replay)
    mine = np.concatenate(mine)
    ref = np.array(ou.loss_log)
    assert mine.shape == ref.shape == (iters * epochs * (T // B), 4)
    assert mine.shape[0] >= 200
    scale = np.maximum(np.abs(ref), 0.05 * np.abs(ref).max(axis=0, keepdims=True))
    rel = np.abs(mine - ref) / scale
    assert rel.max() < 0.01, 'worst per-step relative loss deviation %.4f at %s' % (rel.max(), np.unravel_index(rel.argmax(), rel.shape))
    assert up.loss_iter == ref.shape[0]
    flat = orc.split_actor_critic_state_dict({k: v.detach().cpu() for k, v in ac.state_dict().items()})
    for k, p in P.items():
        a, b = flat[k].numpy(), p.detach().numpy()
        assert np.linalg.norm(a - b) <= 5e-3 * max(np.linalg.norm(b), 1e-6), k
