"""Oracle comparisons at the BASELINE.json model size (SGNN 3 layers x 256) for every graph family the configs
name -- DHM-shaped graphs (cfg-3: up to 397 nodes, the one-workgroup-per-CU size class of the message-passing
kernels), the concept pads 1500 / 4000 (cfg-4), a mixed HLG + DHM minibatch (cfg-5) -- and a row subsample of the
full 2048-row minibatch (cfg-2).  Values, log-probs, entropies, the four loss terms and EVERY parameter gradient are
compared with the oracle (tolerances of SURVEY.md section 8c); the oracle runs the padded dense batch in chunks of a
few rows (rows are independent; its autograd graph needs ~0.2 GB per padded row at D = 256).

Needs a real MI355X (-m gpu); everything goes through the C ABI; nothing reads /root/reference.
"""
import numpy as np
import pytest
import torch

import helpers
from oracle import sgnn_oracle as orc
from test_gpu_parity import DEV, _check_grads, _engine_setup, _forward

pytestmark = pytest.mark.gpu

HEADS = 1
CLIP, CV, CE = 0.2, 0.5, 0.01


def _model(D=256, L=3, max_nodes=1000, max_edges=3000, seed=0, noise=0.03):
    cfg = helpers.make_cfg(D=D, L=L, heads=HEADS, max_nodes=max_nodes, max_edges=max_edges)
    _, _, ac = helpers.build_product(cfg, seed=seed)
    sd = helpers.perturbed_state_dict(ac, seed + 1, scale=noise)
    return cfg, sd


def _oracle_rows(P, states, actions, chunk=8):
    """value / logp / entropy of every row, no gradient, chunked."""
    v, lp, en = [], [], []
    with torch.no_grad():
        for i in range(0, len(states), chunk):
            xs = orc.tensorfy(states[i:i + chunk])
            a = torch.from_numpy(actions[i:i + chunk]).float()
            v.append(orc.value_forward(P, xs, HEADS))
            l, e = orc.get_log_prob_entropy(P, xs, a, HEADS)
            lp.append(l)
            en.append(e)
    return torch.cat(v)[:, 0], torch.cat(lp)[:, 0], torch.cat(en)[:, 0]


def _oracle_ppo_chunked(P, states, actions, adv, ret, old, chunk=8):
    """The minibatch's four loss terms (urban_planning_agent.py:326-333,363-371; exps == 1) with the gradients left in
    P[*].grad, accumulated chunk by chunk: every term is a mean over the T rows, so chunk sums / T add up exactly."""
    T = len(states)
    tot = np.zeros(3)
    for i in range(0, T, chunk):
        xs = orc.tensorfy(states[i:i + chunk])
        a = torch.from_numpy(actions[i:i + chunk]).float()
        value = orc.value_forward(P, xs, HEADS)
        logp, ent = orc.get_log_prob_entropy(P, xs, a, HEADS)
        ratio = torch.exp(logp - old[i:i + chunk])
        surr = torch.min(ratio * adv[i:i + chunk], torch.clamp(ratio, 1.0 - CLIP, 1.0 + CLIP) * adv[i:i + chunk])
        vl, sl, el = (value - ret[i:i + chunk]).pow(2).sum() / T, -surr.sum() / T, -ent.sum() / T
        (sl + CV * vl + CE * el).backward()
        tot += [vl.item(), sl.item(), el.item()]
    return np.array([tot[1] + CV * tot[0] + CE * tot[2], tot[0], tot[1], tot[2]])


def _full_comparison(cfg, sd, replay, chunk=8):
    states, actions = replay.states, replay.actions
    T = len(states)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, states, actions)
    value, logp, ent = _forward(eng, pk, mb, flat)
    P = helpers.oracle_params(sd)
    v0, lp0, en0 = _oracle_rows(P, states, actions, chunk)
    np.testing.assert_allclose(value.cpu().numpy(), v0.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), lp0.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), en0.numpy(), rtol=1e-4, atol=1e-5)
    g = torch.Generator().manual_seed(5)
    adv, ret = torch.randn(T, 1, generator=g), torch.randn(T, 1, generator=g)
    old = lp0[:, None] + 0.3 * torch.randn(T, 1, generator=g)          # ratios on both sides of the clip range
    ref_losses = _oracle_ppo_chunked(P, states, actions, adv, ret, old, chunk)
    dvalue, dlogp, dent = (torch.empty(T, device=DEV) for _ in range(3))
    losses = torch.zeros(4, device=DEV)
    eng.ppo_loss(T, value, logp, ent, adv[:, 0].to(DEV), ret[:, 0].to(DEV), old[:, 0].to(DEV), torch.ones(T, device=DEV),
                 CLIP, CV, CE, 1.0 / T, 1.0 / T, dvalue, dlogp, dent, losses)
    np.testing.assert_allclose(losses.cpu().numpy(), ref_losses, rtol=1e-4, atol=1e-5)
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dvalue, dlogp, dent, grads)
    torch.cuda.synchronize()
    _check_grads(eng, grads, lambda nm: (P[nm].grad if P[nm].grad is not None else torch.zeros_like(P[nm])).numpy())
    return pk


@pytest.mark.parametrize('family', ['dhm', 'hlg_concept_pads', 'mixed'])
def test_baseline_graph_families_match_oracle_at_d256(family):
    from drl_urban_planning_amd import synth
    T = 32
    if family == 'dhm':                 # BASELINE cfg-3: 240..397 nodes, up to 2203 edges
        cfg, sd = _model()
        replay = synth.make_replay(T, 'dhm', max_nodes=1000, max_edges=3000, seed=31)
    elif family == 'hlg_concept_pads':  # BASELINE cfg-4: hlg_concept.yaml pads (actions / pad fill index into 1500 / 4000)
        cfg, sd = _model(max_nodes=1500, max_edges=4000)
        replay = synth.make_replay(T, 'hlg', max_nodes=1500, max_edges=4000, seed=32)
    else:                               # BASELINE cfg-5: heterogeneous HLG + DHM graphs in one minibatch
        cfg, sd = _model()
        replay = synth.make_replay(T, 'mixed', max_nodes=1000, max_edges=3000, seed=33)
    pk = _full_comparison(cfg, sd, replay)
    n = pk.meta[:, 0]
    if family == 'dhm':
        assert n.max() > 350            # graphs beyond the two-workgroups-per-CU LDS class are present
    if family == 'mixed':
        assert n.min() < 240 and n.max() > 350


def test_dhm_graphs_with_h_staged_in_lds_match_oracle():
    """The large size class of the forward normally leaves H in HBM (two workgroups per CU for the > 350-node graphs);
    tune knob fwd_h_hbm = 0 is the older path that stages H with the whole LDS (one workgroup per CU)."""
    from drl_urban_planning_amd import native, synth
    native.check(native.lib().upamd_tune(b'fwd_h_hbm', 0), 'upamd_tune')
    try:
        cfg, sd = _model()
        pk = _full_comparison(cfg, sd, synth.make_replay(16, 'dhm', max_nodes=1000, max_edges=3000, seed=35))
        assert pk.meta[:, 0].max() > 350
    finally:
        native.check(native.lib().upamd_tune(b'fwd_h_hbm', 1), 'upamd_tune')


@pytest.mark.parametrize('family', ['dhm', 'mixed'])
def test_two_per_cu_forms_of_the_large_size_class_are_bit_identical_to_the_one_per_cu_forms(family):
    """The > 350-node graphs run the message-passing kernels with H left in HBM (forward; the folded first layer writes its
    H_0 tiles straight to the output slice) and the neighbour ids walked from global memory (backward): the same arithmetic in
    the same order as the one-workgroup-per-CU forms behind fwd_h_hbm = 0 / bwd_nb_global = 0 -- every output and every
    parameter gradient must agree to the bit."""
    from drl_urban_planning_amd import native, synth
    cfg, sd = _model()
    replay = synth.make_replay(24, family, max_nodes=1000, max_edges=3000, seed=37)
    T = len(replay.states)
    g = torch.Generator().manual_seed(9)
    dv, dl, de = (torch.randn(T, generator=g).to(DEV) for _ in range(3))

    def run(hbm, nbg):
        native.check(native.lib().upamd_tune(b'fwd_h_hbm', hbm), 'upamd_tune')
        native.check(native.lib().upamd_tune(b'bwd_nb_global', nbg), 'upamd_tune')
        _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, replay.states, replay.actions)
        assert pk.meta[:, 0].max() > 350
        outs = [t.clone() for t in _forward(eng, pk, mb, flat)]
        grads = torch.zeros(eng.n_floats, device=DEV)
        eng.backward(pk, mb, flat, dv, dl, de, grads)
        torch.cuda.synchronize()
        return outs + [grads]

    try:
        new, old = run(1, 1), run(0, 0)
    finally:
        native.check(native.lib().upamd_tune(b'fwd_h_hbm', 1), 'upamd_tune')
        native.check(native.lib().upamd_tune(b'bwd_nb_global', 1), 'upamd_tune')
    for a, b in zip(new, old):
        assert torch.equal(a, b)
    assert float(new[-1].abs().max()) > 0


@pytest.mark.parametrize('family', ['hlg', 'mixed_stage'])
def test_gradient_buckets_cover_the_buffer_and_leave_the_bits_alone(family):
    """upamd_grad_buckets (data parallelism): at the BASELINE model size the backward finalises the flat
    gradient buffer in >= 3 ranges -- attention + value head + pointer heads first, each upper GCN layer behind its weight-gradient
    GEMM, the front of the buffer last -- which are disjoint and cover it; moving the reductions there (tune knob grad_buckets = 1,
    the default) must not change a single bit of any gradient against the single final flush (0).  'mixed_stage' has road
    rows, i.e. the pointer-head chain on the caller's stream instead of the side stream."""
    from drl_urban_planning_amd import native, synth
    cfg, sd = _model()
    replay = synth.make_replay(24, 'hlg', max_nodes=1000, max_edges=3000, seed=41,
                               road_fraction=0.3 if family == 'mixed_stage' else 0.0)
    T = len(replay.states)
    g = torch.Generator().manual_seed(10)
    dv, dl, de = (torch.randn(T, generator=g).to(DEV) for _ in range(3))

    def run(on):
        native.check(native.lib().upamd_tune(b'grad_buckets', on), 'upamd_tune')
        _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, replay.states, replay.actions)
        _forward(eng, pk, mb, flat)
        grads = torch.zeros(eng.n_floats, device=DEV)
        eng.backward(pk, mb, flat, dv, dl, de, grads)
        buckets = eng.grad_buckets()
        # every bucket's event can be waited for from another stream
        side = torch.cuda.Stream()
        for k in range(len(buckets)):
            eng.grad_bucket_wait(k, side)
        side.synchronize()
        torch.cuda.synchronize()
        return grads, buckets, eng

    try:
        (g1, b1, eng), (g0, b0, _) = run(1), run(0)
    finally:
        native.check(native.lib().upamd_tune(b'grad_buckets', 1), 'upamd_tune')
    assert b0 == [(0, eng.n_floats)]
    assert len(b1) >= 3 and b1[-1][0] == 0, b1
    ordered = sorted(b1)
    assert ordered[0][0] == 0 and ordered[-1][1] == eng.n_floats
    assert all(a[1] == b[0] for a, b in zip(ordered, ordered[1:])), b1
    first = dict((name, off) for name, off, _, _, _ in eng.table)
    assert b1[0] == (first['shared_net.attention_layer.in_proj_weight'], eng.n_floats)
    assert b1[1][0] == first['shared_net.edge_fc_layers.2.linear_0.weight']
    assert b1[-1][1] == first['shared_net.edge_fc_layers.1.linear_0.weight']
    assert torch.equal(g1, g0)
    assert float(g1.abs().max()) > 0


def test_b2048_minibatch_rows_match_oracle_on_a_subsample():
    """BASELINE cfg-2 at full size (2048 HLG-shaped graphs, D = 256, L = 3): 64 of its rows are compared with the
    oracle evaluated on those rows alone -- forward rows directly; the backward through seeds that are zero outside the
    64 rows (rows are independent, so the gradient of such a loss is the gradient of the 64-row loss)."""
    from drl_urban_planning_amd import synth
    B, K = 2048, 64
    cfg, sd = _model(seed=4)
    replay = synth.make_replay(B, 'hlg', seed=11, unique=512)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, replay.states, replay.actions)
    value, logp, ent = _forward(eng, pk, mb, flat)
    rows = np.sort(np.random.default_rng(3).choice(B, size=K, replace=False))
    sub_states = [replay.states[i] for i in rows]
    sub_actions = replay.actions[rows]
    P = helpers.oracle_params(sd)
    v0, lp0, en0 = _oracle_rows(P, sub_states, sub_actions)
    np.testing.assert_allclose(value.cpu().numpy()[rows], v0.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy()[rows], lp0.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy()[rows], en0.numpy(), rtol=1e-4, atol=1e-5)
    g = torch.Generator().manual_seed(8)
    wv, wl, we = (torch.randn(K, generator=g) / K for _ in range(3))
    for i in range(0, K, 8):
        xs = orc.tensorfy(sub_states[i:i + 8])
        a = torch.from_numpy(sub_actions[i:i + 8]).float()
        v = orc.value_forward(P, xs, HEADS)[:, 0]
        l, e = orc.get_log_prob_entropy(P, xs, a, HEADS)
        ((v * wv[i:i + 8]).sum() + (l[:, 0] * wl[i:i + 8]).sum() + (e[:, 0] * we[i:i + 8]).sum()).backward()
    seeds = [torch.zeros(B) for _ in range(3)]
    for s, w in zip(seeds, (wv, wl, we)):
        s[torch.from_numpy(rows)] = w
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, seeds[0].to(DEV), seeds[1].to(DEV), seeds[2].to(DEV), grads)
    torch.cuda.synchronize()
    _check_grads(eng, grads, lambda nm: (P[nm].grad if P[nm].grad is not None else torch.zeros_like(P[nm])).numpy())


# ------------------------------------------------------------------------------------------------ module surface
def _surface_case(T=10, D=32, heads=2, seed=41, road_fraction=0.4, max_nodes=64, max_edges=200):
    from drl_urban_planning_amd import synth
    cfg = helpers.make_cfg(D=D, L=2, heads=heads, max_nodes=max_nodes, max_edges=max_edges)
    policy_net, value_net, ac = helpers.build_product(cfg, seed=seed)
    sd = helpers.perturbed_state_dict(ac, seed + 1, scale=0.2)
    ac.load_state_dict(sd)
    ac.to(DEV)
    replay = synth.make_replay(T, 'hlg', max_nodes=max_nodes, max_edges=max_edges, seed=seed, road_fraction=road_fraction,
                               n_range=(20, 55))
    xs = [[torch.tensor(f).to(DEV) for f in s] for s in replay.states]
    return policy_net, value_net, ac, sd, replay, xs, heads


def test_gpu_policy_forward_and_select_action_match_oracle():
    """policy_net.forward(x) on a GPU module returns the reference's objects (policy.py:45-65): Categorical over the
    PADDED slots of the rows in each stage, masked slots at the pad constant; select_action(mean_action=True) is the
    oracle's arg-max (policy.py:67-85); sampled actions follow those distributions."""
    policy_net, value_net, ac, sd, replay, xs, heads = _surface_case()
    P = helpers.oracle_params(sd, requires_grad=False)
    keep = {}
    with torch.no_grad():
        land0, road0, stage0 = orc.policy_forward(P, orc.tensorfy(replay.states), heads, keep)
        land, road, stage = policy_net(xs)
    assert torch.equal(stage.cpu(), stage0)
    assert land is not None and road is not None
    for mine, ref in ((land, keep['land_logits']), (road, keep['road_logits'])):
        assert mine.logits.shape == ref.shape
        np.testing.assert_allclose(mine.probs.cpu().numpy(), torch.softmax(ref, -1).numpy(), rtol=1e-4, atol=1e-6)
        masked = ref == orc.PAD_LOGIT
        assert float(mine.probs.cpu()[masked].max()) == 0.0
    greedy = policy_net.select_action(xs, mean_action=True).cpu()
    want = torch.zeros_like(greedy)
    want[stage0[:, 0].bool(), 0] = land0.probs.argmax(1).float()
    want[stage0[:, 1].bool(), 1] = road0.probs.argmax(1).float()
    assert torch.equal(greedy, want)
    # sampling: valid slots only, and the empirical frequencies of one row follow its distribution
    torch.manual_seed(0)
    draws = torch.stack([policy_net.select_action(xs).cpu() for _ in range(200)])       # [200, B, 2]
    for b, s in enumerate(replay.states):
        col = 0 if stage0[b, 0] else 1
        mask = s[6] if col == 0 else s[7]
        assert mask[draws[:, b, col].long().numpy()].all()
        assert (draws[:, b, 1 - col] == 0).all()
    b = int(torch.nonzero(stage0[:, 0])[0])
    p_ref = land0.probs[int(stage0[:b, 0].sum())].numpy()
    freq = np.bincount(draws[:, b, 0].long().numpy(), minlength=p_ref.size) / draws.shape[0]
    assert np.abs(freq - p_ref).max() < 0.15


def test_interleaved_forwards_keep_their_own_activations():
    """value_net(x1), then get_log_prob_entropy(x2, a), a no-grad select_action in between, then ONE backward: every
    forward owns its workspace, so the value branch back-propagates through x1's activations (round 1 shared one
    workspace and silently used x2's).  A second backward through the same forward raises instead of re-reading."""
    policy_net, value_net, ac, sd, replay, xs, heads = _surface_case(T=12, seed=51)
    x1, x2 = xs[:6], xs[6:]
    a2 = torch.from_numpy(replay.actions[6:]).float().to(DEV)
    v1 = value_net(x1)
    with torch.no_grad():
        policy_net.select_action(x2)
        value_net(x2)
    lp2, en2 = policy_net.get_log_prob_entropy(x2, a2)
    w = torch.linspace(-1.0, 1.0, 6, device=DEV)[:, None]
    loss = (v1 * w).sum() + (lp2 * w.flip(0)).sum() + 0.3 * en2.sum()
    loss.backward(retain_graph=True)
    P = helpers.oracle_params(sd)
    wc = w.cpu()
    v1o = orc.value_forward(P, orc.tensorfy(replay.states[:6]), heads)
    lp2o, en2o = orc.get_log_prob_entropy(P, orc.tensorfy(replay.states[6:]), a2.cpu(), heads)
    ((v1o * wc).sum() + (lp2o * wc.flip(0)).sum() + 0.3 * en2o.sum()).backward()
    np.testing.assert_allclose(v1.detach().cpu().numpy(), v1o.detach().numpy(), rtol=1e-4, atol=1e-5)
    flat = orc.split_actor_critic_state_dict({k: v for k, v in ac.named_parameters()})
    scale = max(float(p.grad.abs().max()) for p in P.values() if p.grad is not None)
    for k, p in P.items():
        mine = flat[k].grad
        mine = np.zeros(tuple(p.shape), np.float32) if mine is None else mine.cpu().numpy()
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        rel = np.linalg.norm(mine - ref) / max(np.linalg.norm(ref), 1e-30)
        assert np.abs(mine - ref).max() <= 1e-5 * scale or rel <= 1e-4, (k, rel)
    with pytest.raises(RuntimeError, match='already been differentiated'):
        loss.backward()


@pytest.mark.parametrize('how', ['mixin', 'install'])
def test_drop_in_mixin_and_install_reproduce_the_reference_update(how):
    """The patch INTEGRATION.md documents: ``class Agent(HipUpdateMixin, <reference agent>)`` / ``install(agent)`` read
    the hyper-parameters off the agent (``optimizer.param_groups[0]``, gamma, tau, ...), shadow ``update_params`` and
    keep ``loss_iter`` / the TensorBoard scalars; two calls must land on the real reference's trajectory."""
    from duck_agent import make_duck_agent
    from drl_urban_planning_amd import synth
    from test_oracle_golden import CASE_B, CASE_EPOCHS, CASE_HYPER, CASE_SEED
    name = 'case_b'                      # weight decay, gamma / tau != defaults: every hyper-parameter matters
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    policy_net, value_net, ac = helpers.build_product(cfg)
    ac.load_state_dict(sd)
    ac.to(DEV)
    agent = make_duck_agent(cfg, policy_net, value_net, ac, CASE_HYPER[name], CASE_EPOCHS[name], CASE_B[name],
                            mixin=(how == 'mixin'))
    replay = synth.Replay(states, z['actions'], z['masks'], z['rewards'], z['exps'])
    np.random.seed(CASE_SEED[name] + 11)
    elapsed = agent.update_params(replay, 0)
    assert elapsed > 0 and agent.loss_iter == int(z['upd/loss_iter'])
    up = agent._hip_updater()
    assert up.dist.world == 1 and up.last_timing['dp_mode'] == 'single'
    np.testing.assert_allclose(up.last_losses, z['upd/scalars'], rtol=1e-4, atol=2e-6)
    tags = [t for (t, v, s) in agent.tb_logger.scalars]
    assert tags.count('loss/loss') == agent.loss_iter and tags.count('loss/total_loss') == 1
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    for k, v in ac.state_dict().items():
        assert rel(v.detach().cpu().numpy(), z['upd_sd/' + k]) <= 1e-4, k
    np.random.seed(CASE_SEED[name] + 12)
    agent.update_params(replay, 1)
    assert agent.loss_iter == z['upd2/scalars'].shape[0]
    for k, v in ac.state_dict().items():
        assert rel(v.detach().cpu().numpy(), z['upd2_sd/' + k]) <= 2e-4, k
