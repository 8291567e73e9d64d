"""The fused per-step launches against the single-purpose entry points they replace (needs a real MI355X: -m gpu).

* ``upamd_ppo_loss_rows`` (row gathers + loss + zero_grad in one launch) == ``upamd_ppo_loss`` on gathered inputs, bit
  for bit, and the zero range is cleared (urban_planning_agent.py:316-335).
* ``upamd_adam_groups`` (all optimizer groups + the loss copy-out in one launch) == one ``upamd_adam_step`` per group,
  bit for bit, including a skipped group (a head without rows: grad None in the reference).
"""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _engine():
    from drl_urban_planning_amd.models import backend_of
    cfg = helpers.make_cfg(**helpers.CASE_MODEL['case_a'])
    policy_net, value_net, ac = helpers.build_product(cfg, seed=3)
    ac.to(DEV)
    backend = backend_of(policy_net)
    eng = backend.engine(torch.device(DEV))
    return eng, eng.flatten(backend.named_params())


def test_loss_rows_equals_gathered_loss():
    eng, _ = _engine()
    g = torch.Generator(device='cpu').manual_seed(11)
    T, B = 500, 96
    adv, ret, old = (torch.randn(T, generator=g).to(DEV) for _ in range(3))
    exps = (torch.rand(T, generator=g) > 0.2).float().to(DEV)
    rows = torch.randperm(T, generator=g)[:B].to(DEV)
    value, logp, ent = (torch.randn(B, generator=g).to(DEV) for _ in range(3))
    logp = old[rows] + 0.3 * logp                       # ratios on both sides of the clip range
    n_ind = float((exps[rows] != 0).sum())
    outs = []
    for fused in (False, True):
        dv, dl, de = (torch.full((B,), 7.0, device=DEV) for _ in range(3))
        buf = torch.full((1000 + 4,), 3.0, device=DEV)
        args = (0.2, 0.5, 0.01, 1.0 / B, 1.0 / n_ind, dv, dl, de, buf[1000:])
        if fused:
            eng.ppo_loss_rows(B, value, logp, ent, rows, adv, ret, old, exps, *args, zero=buf[:1000])
            assert not buf[:1000].any()
        else:
            eng.ppo_loss(B, value, logp, ent, adv[rows], ret[rows], old[rows], exps[rows], *args)
            assert bool((buf[:1000] == 3.0).all())
        outs.append([t.cpu().numpy().copy() for t in (dv, dl, de, buf[1000:])])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert np.isfinite(outs[0][3]).all() and np.abs(outs[0][1]).max() > 0


@pytest.mark.parametrize('steps', [(5, 5, 5), (9, 0, 4), (1, 1, 0)])
def test_adam_groups_equals_per_group_steps(steps):
    eng, flat = _engine()
    n = eng.n_floats
    g = torch.Generator(device='cpu').manual_seed(5)
    grads = torch.cat([torch.randn(n, generator=g), torch.tensor([1.0, 2.0, 3.0, 4.0])]).to(DEV)
    m0, v0 = torch.randn(n, generator=g).to(DEV) * 0.1, torch.rand(n, generator=g).to(DEV) * 0.01
    hy = (4e-4, 0.9, 0.999, 1e-5, 1e-3)
    pa, ma, va = flat.clone(), m0.clone(), v0.clone()
    for grp, st in enumerate(steps):
        if st > 0:
            eng.adam_step(grp, pa, grads, ma, va, st, *hy)
    pb, mb, vb = flat.clone(), m0.clone(), v0.clone()
    loss_out = torch.zeros(4, device=DEV)
    eng.adam_groups(steps, pb, grads, mb, vb, *hy, loss_src=grads[n:], loss_dst=loss_out)
    for a, b in ((pa, pb), (ma, mb), (va, vb)):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    np.testing.assert_array_equal(loss_out.cpu().numpy(), [1.0, 2.0, 3.0, 4.0])
    for grp, st in enumerate(steps):                     # a skipped group is untouched
        b, e = eng.groups[grp]
        changed = bool((pb[b:e] != flat[b:e]).any())
        assert changed == (st > 0)
