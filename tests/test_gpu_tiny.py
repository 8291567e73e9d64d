"""The fused small-model path (csrc/tiny.hip: one workgroup per graph, forward + PPO loss + backward in one launch) on the GPU.

Its per-graph program is checked on the CPU against the reference-generated goldens by tests/test_tiny_emul.py; what only
the GPU can show is the parallel execution -- barriers, ownership of every element, the slab reduction -- so besides the
golden / oracle comparisons this file leans on BIT-IDENTITY: the program's sums are serial inside one iteration, so 512
and 1024 threads per workgroup, one or several graphs per workgroup and repeated runs must all give the same bits.
"""
import numpy as np
import pytest
import torch

import helpers
from oracle import sgnn_oracle as orc
from test_gpu_parity import _check_grads, _engine_setup, _forward, tune  # noqa: F401  (tune: fixture)
from test_oracle_golden import CASE_HYPER

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step_fused(eng, pk, mb, flat, adv, ret, old, exps, hy, inv_rows, inv_ind):
    B = mb.B
    value, logp, ent = (torch.empty(B, device=DEV) for _ in range(3))
    grads = torch.full((eng.n_floats + 4,), float('nan'), device=DEV)       # overwritten, not accumulated: NaNs must vanish
    eng.step_fused(pk, mb, flat, None, adv, ret, old, exps, hy['clip_epsilon'], hy['value_pred_coef'], hy['entropy_coef'],
                   inv_rows, inv_ind, value, logp, ent, grads[:eng.n_floats], grads[eng.n_floats:])
    torch.cuda.synchronize()
    return value, logp, ent, grads


@pytest.mark.parametrize('name', ['case_a', 'case_b', 'case_c', 'case_s'])
def test_fused_step_matches_the_reference_goldens(name, tune):
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    hy = CASE_HYPER[name]
    B = z['fwd/value'].shape[0]
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, states[:B], z['actions'][:B])
    assert eng.step_fused_ok(mb)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    adv, ret, old, exps = t(z['mb/adv'][:, 0]), t(z['mb/ret'][:, 0]), t(z['mb/old_logp'][:, 0]), t(z['exps'][:B])
    nind = int((z['exps'][:B] != 0).sum())
    value, logp, ent, grads = _step_fused(eng, pk, mb, flat, adv, ret, old, exps, hy, 1.0 / B, 1.0 / nind)
    np.testing.assert_allclose(value.cpu().numpy(), z['fwd/value'][:, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), z['fwd/logp'][:, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), z['fwd/entropy'][:, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(grads[eng.n_floats:].cpu().numpy(), z['mb/losses'], rtol=2e-5, atol=2e-6)
    _check_grads(eng, grads[:eng.n_floats], lambda nm: z[helpers.golden_key('grad/', nm)])
    # the three-call form of the same step (forward, loss kernel, backward from seeds) lands on the same gradients
    v2, l2, e2 = _forward(eng, pk, mb, flat)
    assert torch.equal(v2, value) and torch.equal(l2, logp) and torch.equal(e2, ent)
    dvalue, dlogp, dent = (torch.empty(B, device=DEV) for _ in range(3))
    losses = torch.zeros(4, device=DEV)
    eng.ppo_loss(B, v2, l2, e2, adv, ret, old, exps, hy['clip_epsilon'], hy['value_pred_coef'], hy['entropy_coef'], 1.0 / B,
                 1.0 / nind, dvalue, dlogp, dent, losses)
    g2 = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dvalue, dlogp, dent, g2)
    torch.cuda.synchronize()
    scale = float(g2.abs().max())
    assert float((g2 - grads[:eng.n_floats]).abs().max()) <= 2e-6 * scale
    # 512 threads per workgroup: the same bits (every sum is serial inside one iteration of the program)
    tune('tiny_threads', 512)
    v3, l3, e3, g3 = _step_fused(eng, pk, mb, flat, adv, ret, old, exps, hy, 1.0 / B, 1.0 / nind)
    assert torch.equal(v3, value) and torch.equal(l3, logp) and torch.equal(e3, ent)
    assert torch.equal(g3, grads), float((g3 - grads).abs().max())


def _hlg_case(T, seed, road_fraction, D=16, L=2, heads=1, community='hlg'):
    from drl_urban_planning_amd import synth
    cfg = helpers.make_cfg(D=D, L=L, heads=heads, max_nodes=400, max_edges=2300)
    _, _, ac = helpers.build_product(cfg, seed=seed)
    sd = helpers.perturbed_state_dict(ac, seed + 1)
    rep = synth.make_replay(T, community, max_nodes=400, max_edges=2300, seed=seed + 2, road_fraction=road_fraction)
    return cfg, sd, rep


@pytest.mark.parametrize('community,road_fraction', [('hlg', 0.0), ('dhm', 0.3), ('mixed', 0.5), ('grid', 0.5)])
def test_reference_dims_on_full_size_graphs_match_the_oracle(community, road_fraction):
    """hlg.yaml / dhm.yaml dims (D = 16, L = 2) on HLG- and DHM-sized graphs (up to 397 nodes / 2216 edges: the largest LDS
    plan the path takes) and on BASELINE cfg-1's grid-shaped graphs (grid.yaml:16-33: 120-250 nodes, land-use and road rows
    mixed): rows, loss and every gradient against the oracle's autograd."""
    T = 12
    cfg, sd, rep = _hlg_case(T, 30, road_fraction, community=community)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, rep.states, rep.actions)
    assert eng.step_fused_ok(mb), 'full-size graphs must stay on the fused path'
    P = helpers.oracle_params(sd)
    xs = orc.tensorfy(rep.states)
    act_t = torch.from_numpy(np.asarray(rep.actions, dtype=np.float32))
    g = torch.Generator().manual_seed(5)
    adv, ret = torch.randn(T, 1, generator=g), torch.randn(T, 1, generator=g)
    with torch.no_grad():
        lp0, _ = orc.get_log_prob_entropy(P, xs, act_t, 1)
    old = lp0 + 0.3 * torch.randn(T, 1, generator=g)
    exps = torch.ones(T)
    exps[1] = 0.0
    hy = dict(clip_epsilon=0.2, value_pred_coef=0.5, entropy_coef=0.01)
    loss, vl, sl, el = orc.ppo_losses(P, xs, act_t, adv, ret, old, exps, 0.2, 0.5, 0.01, 1)
    loss.backward()
    value, logp, ent, grads = _step_fused(eng, pk, mb, flat, adv[:, 0].to(DEV), ret[:, 0].to(DEV), old[:, 0].to(DEV),
                                          exps.to(DEV), hy, 1.0 / T, 1.0 / (T - 1))
    np.testing.assert_allclose(grads[eng.n_floats:].cpu().numpy(), [loss.item(), vl.item(), sl.item(), el.item()], rtol=1e-4,
                               atol=1e-5)
    _check_grads(eng, grads[:eng.n_floats],
                 lambda nm: (P[nm].grad if P[nm].grad is not None else torch.zeros_like(P[nm])).numpy())
    # run-to-run: bit-identical (fixed summation order everywhere, no atomics)
    for _ in range(2):
        v2, l2, e2, g2 = _step_fused(eng, pk, mb, flat, adv[:, 0].to(DEV), ret[:, 0].to(DEV), old[:, 0].to(DEV), exps.to(DEV),
                                     hy, 1.0 / T, 1.0 / (T - 1))
        assert torch.equal(g2, grads) and torch.equal(v2, value)


def test_many_graphs_per_workgroup_and_row_order_invariance():
    """600 rows on 256 persistent workgroups (two or three graphs each, slabs accumulated across them): every row's outputs
    equal those of the same graph in a small minibatch, bit for bit, and the gradient equals the sum over three sub-batches."""
    from drl_urban_planning_amd import packer
    T = 600
    cfg, sd, rep = _hlg_case(T, 40, 0.25)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, rep.states, rep.actions)
    assert eng.step_fused_ok(mb)
    g = torch.Generator().manual_seed(9)
    seeds = [torch.randn(T, generator=g).to(DEV) for _ in range(3)]
    value, logp, ent = _forward(eng, pk, mb, flat)
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, seeds[0], seeds[1], seeds[2], grads)
    torch.cuda.synchronize()
    assert torch.isfinite(grads).all()
    parts = [np.arange(0, 200), np.arange(200, 400), np.arange(400, 600)]
    sched3 = packer.Schedule(pk, parts, DEV)
    acc = torch.zeros(eng.n_floats, device=DEV, dtype=torch.float64)
    for k, rows in enumerate(parts):
        mbk, _ = sched3.minibatch(k)
        vk, lk, ek = (torch.empty(rows.size, device=DEV) for _ in range(3))
        eng.forward(pk, mbk, flat, vk, lk, ek, keep=True)
        gk = torch.zeros(eng.n_floats, device=DEV)
        sl = torch.from_numpy(rows).to(DEV)
        eng.backward(pk, mbk, flat, seeds[0][sl].contiguous(), seeds[1][sl].contiguous(), seeds[2][sl].contiguous(), gk)
        torch.cuda.synchronize()
        assert torch.equal(vk, value[sl]) and torch.equal(lk, logp[sl]) and torch.equal(ek, ent[sl])
        acc += gk.double()
    scale = float(grads.abs().max())
    assert float((acc - grads.double()).abs().max()) <= 1e-5 * scale


def test_fused_and_general_paths_agree_and_fall_back_where_they_must(tune):
    """The two implementations of a small model agree to fp32 noise; graphs beyond a workgroup's LDS, one-layer models and
    deep edge MLPs stay on the general path (upamd_step_fused_ok == 0) and still run."""
    from drl_urban_planning_amd import synth
    T = 10
    cfg, sd, rep = _hlg_case(T, 50, 0.3, D=32, L=3, heads=2)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, rep.states, rep.actions)
    g = torch.Generator().manual_seed(3)
    seeds = [torch.randn(T, generator=g).to(DEV) for _ in range(3)]

    def run():
        value, logp, ent = _forward(eng, pk, mb, flat)
        grads = torch.zeros(eng.n_floats, device=DEV)
        eng.backward(pk, mb, flat, seeds[0], seeds[1], seeds[2], grads)
        torch.cuda.synchronize()
        return value, logp, ent, grads
    fused_ok = eng.step_fused_ok(mb)
    a = run()
    tune('tiny_fused', 0)
    assert not eng.step_fused_ok(mb)
    b = run()
    tune('tiny_fused', 1)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-4, atol=1e-5)
    scale = float(b[3].abs().max())
    assert float((a[3] - b[3]).abs().max()) <= 2e-5 * scale
    assert fused_ok in (True, False)      # (D = 32 at full size: whether it fits a workgroup's LDS is the plan's call)
    # beyond the LDS of a workgroup -> general path
    big = synth.make_replay(3, 'hlg', max_nodes=1000, max_edges=3000, seed=8, n_range=(900, 1000))
    cfg_b = helpers.make_cfg(D=16, L=2, max_nodes=1000, max_edges=3000)
    _, _, ac = helpers.build_product(cfg_b, seed=1)
    _, _, _, eng_b, flat_b, pk_b, _, mb_b = _engine_setup(cfg_b, ac.state_dict(), big.states, big.actions)
    assert not eng_b.step_fused_ok(mb_b)
    v, l, e = _forward(eng_b, pk_b, mb_b, flat_b)
    assert torch.isfinite(v).all() and torch.isfinite(l).all()
    # one GCN layer -> general path
    cfg_1 = helpers.make_cfg(D=16, L=1, max_nodes=400, max_edges=2300)
    _, _, ac1 = helpers.build_product(cfg_1, seed=1)
    _, _, _, eng_1, _, _, _, mb_1 = _engine_setup(cfg_1, ac1.state_dict(), rep.states, rep.actions)
    assert not eng_1.step_fused_ok(mb_1)


@pytest.mark.parametrize('D,L,heads,S,value_head', [(16, 2, 1, (48, 32, 24, 16), (32, 1)), (16, 4, 4, (16,), (64, 32, 16, 1))])
def test_other_model_shapes_fused_and_general_agree(D, L, heads, S, value_head, tune):
    """More per-sample chain layers than graph phases to host them (the rest run as phases of their own), and fewer: the fused kernel
    against the general kernels (the emulator checks the same shapes against the oracle on the CPU, tests/test_tiny_emul.py)."""
    from drl_urban_planning_amd import synth
    T = 9
    cfg = helpers.make_cfg(D=D, L=L, heads=heads, S=S, value_head=value_head, max_nodes=400, max_edges=2300)
    _, _, ac = helpers.build_product(cfg, seed=61)
    sd = helpers.perturbed_state_dict(ac, 62)
    rep = synth.make_replay(T, 'hlg', max_nodes=400, max_edges=2300, seed=63, road_fraction=0.3,
                            n_range=None if L <= 2 else (60, 150))       # (four layers' H slots: smaller graphs fit the LDS)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, rep.states, rep.actions)
    assert eng.step_fused_ok(mb)
    g = torch.Generator().manual_seed(5)
    seeds = [torch.randn(T, generator=g).to(DEV) for _ in range(3)]

    def run():
        value, logp, ent = _forward(eng, pk, mb, flat)
        grads = torch.zeros(eng.n_floats, device=DEV)
        eng.backward(pk, mb, flat, seeds[0], seeds[1], seeds[2], grads)
        torch.cuda.synchronize()
        return value, logp, ent, grads
    a = run()
    tune('tiny_fused', 0)
    b = run()
    tune('tiny_fused', 1)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-4, atol=1e-5)
    scale = float(b[3].abs().max())
    assert float((a[3] - b[3]).abs().max()) <= 2e-5 * scale
    # a forward on the fused path followed by a GENERAL backward (the knob flipped in between): the general backward would read
    # activations nobody wrote -- refused; the other order is fine (the fused backward recomputes its forward)
    _forward(eng, pk, mb, flat)
    tune('tiny_fused', 0)
    with pytest.raises(RuntimeError, match='did not run the general kernels'):
        eng.backward(pk, mb, flat, seeds[0], seeds[1], seeds[2], torch.zeros(eng.n_floats, device=DEV))
    _forward(eng, pk, mb, flat)
    tune('tiny_fused', 1)
    g2 = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, seeds[0], seeds[1], seeds[2], g2)
    torch.cuda.synchronize()
    assert torch.equal(g2, a[3])


def test_action_heads_read_the_fused_forwards_logits():
    """policy_net.forward / select_action on a GPU module read the candidate logits back (upamd_ws_tensor 'z_he' / 'z_rn'):
    greedy actions of the fused path == the oracle's arg-max."""
    T = 8
    cfg, sd, rep = _hlg_case(T, 60, 0.5)
    policy_net, value_net, ac = helpers.build_product(cfg)
    ac.load_state_dict(sd)
    ac.to(DEV)
    xs = [[torch.tensor(f).to(DEV) for f in s] for s in rep.states]
    with torch.no_grad():
        act = policy_net.select_action(xs, mean_action=True).cpu().numpy()
    P = helpers.oracle_params(sd, requires_grad=False)
    with torch.no_grad():
        land, road, stage = orc.policy_forward(P, orc.tensorfy(rep.states), 1)
    il, ir = 0, 0
    for b in range(T):
        st = int(np.argmax(rep.states[b][8]))
        if st == 0:
            assert int(act[b, 0]) == int(land.logits[il].argmax())
            il += 1
        elif st == 1:
            assert int(act[b, 1]) == int(road.logits[ir].argmax())
            ir += 1
    # the fused path materialises nothing else: a stage tensor of the general path must be refused, not read from stale memory
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, rep.states, rep.actions)
    assert eng.step_fused_ok(mb)
    _forward(eng, pk, mb, flat)
    assert eng.ws_tensor(mb, 'z_he').numel() + eng.ws_tensor(mb, 'z_rn').numel() > 0
    for name in ('H1', 'hbarV', 'SV', 'att'):
        with pytest.raises(RuntimeError, match='fused small-model path'):
            eng.ws_tensor(mb, name)


@pytest.mark.parametrize('gain,bias', [(40.0, 0.0), (1.0, 3.0), (12.0, 0.0)])
def test_saturating_edge_mlp_on_the_fused_path(gain, bias):
    """P | Q outside the exp-form range: the layer's flag goes up (an LDS atomic OR) and that graph walks in the linear form;
    graphs of the same minibatch that stay in range keep the exp form."""
    from test_gpu_parity import _check_against_oracle, _random_case
    D, L, heads, T, n_range = 16, 2, 1, 6, (30, 60)
    cfg, sd, replay = _random_case(D, L, heads, (64, 16), (32, 1), (32, 1), (32, 32, 1), T, n_range[1] + 5,
                                   int(5.55 * n_range[1]) + 10, seed=33, road_fraction=0.3, n_range=n_range)
    sd = dict(sd)
    for k in list(sd):
        if 'edge_fc_layers' in k:
            sd[k] = sd[k] * gain if k.endswith('weight') else sd[k] + bias
    _check_against_oracle(cfg, sd, replay, heads, T, tol=3e-4 if gain > 1 else 1e-4)


def test_two_engines_in_one_process_keep_their_own_knobs():
    """``engine.set_tune``: engine A runs the general kernels (tiny_fused = 0 for itself), engine B -- same model, same process -- stays
    on the fused path, call by call interleaved; both match each other to the usual tolerance and the process default is untouched."""
    from drl_urban_planning_amd import native
    from drl_urban_planning_amd.engine import NativeEngine
    T = 10
    cfg, sd, rep = _hlg_case(T, 70, 0.3)
    _, _, _, engB, flat, pk, sched, mb = _engine_setup(cfg, sd, rep.states, rep.actions)
    engA = NativeEngine(engB.desc, torch.device(DEV))
    engA.set_tune('tiny_fused', 0)
    assert engB.step_fused_ok(mb) and not engA.step_fused_ok(mb) and engB.step_fused_ok(mb)
    g = torch.Generator().manual_seed(4)
    seeds = [torch.randn(T, generator=g).to(DEV) for _ in range(3)]
    outs = {}
    for name, eng in (('A', engA), ('B', engB), ('A2', engA), ('B2', engB)):
        v, l, e = _forward(eng, pk, mb, flat)
        grads = torch.zeros(eng.n_floats, device=DEV)
        eng.backward(pk, mb, flat, seeds[0], seeds[1], seeds[2], grads)
        torch.cuda.synchronize()
        outs[name] = [t.clone() for t in (v, l, e, grads)]
    assert engA.ws_tensor(mb, 'H1').numel() > 0                 # A's workspace holds the general path's tensors ...
    with pytest.raises(RuntimeError, match='fused small-model path'):
        engB.ws_tensor(mb, 'H1')                                # ... B's does not
    for a, b in zip(outs['A'], outs['A2']):
        assert torch.equal(a, b)
    for a, b in zip(outs['B'], outs['B2']):
        assert torch.equal(a, b)
    for a, b in zip(outs['A'][:3], outs['B'][:3]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5)
    scale = float(outs['A'][3].abs().max())
    assert float((outs['A'][3] - outs['B'][3]).abs().max()) <= 2e-5 * scale
    assert native.lib().upamd_step_fused_ok(engB.handle, __import__('ctypes').byref(mb)) == 1      # the process default is back
