"""Parity of the HIP path against the oracle / golden vectors (needs a real MI355X: -m gpu).

Everything goes through the C ABI (ctypes -> libupamd.so); nothing here touches /root/reference.
Tolerances (fp32): per-row value / log-prob / entropy abs <= 1e-5 or rel <= 1e-4;
loss scalars rel <= 1e-5 (abs 1e-6); parameter gradients rel-L2 <= 1e-4 and max-abs <= 1e-5 * scale;
parameters after the update rel-L2 <= 1e-4; GAE bit-exact.
"""
import os

import numpy as np
import pytest
import torch

import csr_model
import helpers
from oracle import sgnn_oracle as orc

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _engine_setup(cfg, sd_actor_critic, states, actions):
    from drl_urban_planning_amd import packer
    from drl_urban_planning_amd.models import backend_of
    policy_net, value_net, ac = helpers.build_product(cfg)
    ac.load_state_dict(sd_actor_critic)
    ac.to(DEV)
    backend = backend_of(policy_net)
    eng = backend.engine(torch.device(DEV))
    flat = eng.flatten(backend.named_params())
    pk = packer.pack_replay(states, actions, 23, 52).to(DEV)
    sched = packer.Schedule(pk, [np.arange(len(states))], DEV)
    mb, item = sched.minibatch(0)
    return policy_net, value_net, ac, eng, flat, pk, sched, mb


def _forward(eng, pk, mb, flat):
    B = mb.B
    value, logp, ent = (torch.empty(B, device=DEV) for _ in range(3))
    eng.forward(pk, mb, flat, value, logp, ent, keep=True)
    torch.cuda.synchronize()
    return value, logp, ent


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _check_grads(eng, grads_flat, ref_of, atol_scale=1e-5, rtol_l2=1e-4):
    g = grads_flat.detach().cpu().numpy()
    refs = {name: ref_of(name) for name, *_ in eng.table}
    scale = max(max(float(np.abs(r).max()) for r in refs.values()), 1e-12)
    report = []
    bad = []
    for name, off, rows, cols, _ in eng.table:
        mine = g[off:off + rows * cols].reshape(refs[name].shape)
        err = float(np.abs(mine - refs[name]).max())
        rl2 = _rel_l2(mine, refs[name])
        report.append('%-60s max|d|=%.3e relL2=%.3e |ref|max=%.3e' % (name, err, rl2, float(np.abs(refs[name]).max())))
        # tensors whose reference gradient is pure rounding noise (key-side attention biases) only get the abs test
        if err > atol_scale * scale and (rl2 > rtol_l2):
            bad.append(name)
        elif err > 50 * atol_scale * scale:
            bad.append(name)
    assert not bad, 'gradient mismatch in %s\n%s' % (bad, '\n'.join(report))


@pytest.fixture
def tune():
    """set native tune knobs for one test; every knob is put back to its default afterwards"""
    from drl_urban_planning_amd import native
    defaults = {'fold_layer1': 1, 'gemm_split': 0, 'he_fused': 1, 'side_stream': 1, 'fe_half': 1, 'pq_exp': 1, 'nt_min_wgs': 128, 'bwd_nb_global': 1, 'side_heads': 1, 'side_wgrad': 1, 'tiny_fused': 1,
                'tiny_threads': int(os.environ.get('UPAMD_TEST_TINY_THREADS', '1024'))}      # (the variable: run the suite on the 512-thread variant)
    touched = []

    def _set(name, value):
        assert name in defaults
        native.check(native.lib().upamd_tune(name.encode(), int(value)), 'upamd_tune')
        touched.append(name)
    yield _set
    for name in touched:
        native.check(native.lib().upamd_tune(name.encode(), defaults[name]), 'upamd_tune')


@pytest.fixture(params=['fused', 'general'])
def small_path(request, tune):
    """Models with gcn_node_dim <= 32 run the fused one-workgroup-per-graph kernels (csrc/tiny.hip) by default; the general
    kernels remain the path of larger graphs / one-layer models / deep edge MLPs at those dims: tests of small models
    run on both."""
    tune('tiny_fused', 1 if request.param == 'fused' else 0)
    return request.param


@pytest.mark.parametrize('fold', [1, 0])
@pytest.mark.parametrize('name', ['case_a', 'case_b', 'case_c'])
def test_forward_stages_match_oracle(name, fold, tune):
    tune('tiny_fused', 0)          # (the stage tensors of the GENERAL path; the fused kernels keep theirs in LDS)
    # fold = 1 (default): the first GCN layer is computed inside the message-passing kernels, H0 / PQ1 never exist in
    # HBM (asking for them fails); fold = 0: two K = 32 GEMMs materialise them
    tune('fold_layer1', fold)
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    B = z['fwd/value'].shape[0]
    heads = helpers.CASE_MODEL[name]['heads']
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, states[:B], z['actions'][:B])
    value, logp, ent = _forward(eng, pk, mb, flat)
    P = helpers.oracle_params(sd, requires_grad=False)
    keep = {}
    with torch.no_grad():
        orc.value_forward(P, orc.tensorfy(states[:B]), heads, keep)
    L = helpers.CASE_MODEL[name]['L']
    ns = pk.meta[:B, 0]
    offs = np.concatenate([[0], np.cumsum(ns)])
    msgs = []
    if fold:
        with pytest.raises(RuntimeError):
            eng.ws_tensor(mb, 'H0')
    for l in range(1 if fold else 0, L + 1):
        mine = eng.ws_tensor(mb, 'H%d' % l).cpu().numpy()
        ref = keep['h_nodes_%d' % l].numpy()
        err = max(float(np.abs(mine[offs[b]:offs[b + 1]] - ref[b, :ns[b]]).max()) for b in range(B))
        msgs.append('H%d err %.3e' % (l, err))
        assert err < 2e-5, msgs
    for nm, key in (('hbarV', 'h_nodes_mean'), ('hbarE', 'h_edges_mean'), ('C', 'h_cur'), ('att', 'h_att'),
                    ('SV', 'state_value')):
        ref = keep[key].numpy()
        mine = eng.ws_tensor(mb, nm).cpu().numpy()
        if nm == 'SV':                      # rows are padded to a multiple of 16 columns with zeros
            assert not mine[:, ref.shape[1]:].any()
            mine = mine[:, :ref.shape[1]]
        err = float(np.abs(mine - ref).max())
        msgs.append('%s err %.3e' % (nm, err))
        assert err < 2e-5, msgs
    np.testing.assert_allclose(value.cpu().numpy(), z['fwd/value'][:, 0], rtol=1e-4, atol=1e-5, err_msg=str(msgs))
    np.testing.assert_allclose(logp.cpu().numpy(), z['fwd/logp'][:, 0], rtol=1e-4, atol=1e-5, err_msg=str(msgs))
    np.testing.assert_allclose(ent.cpu().numpy(), z['fwd/entropy'][:, 0], rtol=1e-4, atol=1e-5, err_msg=str(msgs))


@pytest.mark.parametrize('fold', [1, 0])
@pytest.mark.parametrize('name', ['case_a', 'case_b', 'case_c'])
def test_loss_and_gradients_match_reference(name, fold, tune, small_path):
    if small_path == 'fused' and fold == 0:
        pytest.skip('fold_layer1 is a knob of the general path')
    tune('fold_layer1', fold)
    from test_oracle_golden import CASE_HYPER
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    hy = CASE_HYPER[name]
    B = z['fwd/value'].shape[0]
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, states[:B], z['actions'][:B])
    value, logp, ent = _forward(eng, pk, mb, flat)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    adv, ret, old = t(z['mb/adv'][:, 0]), t(z['mb/ret'][:, 0]), t(z['mb/old_logp'][:, 0])
    exps = t(z['exps'][:B])
    dvalue, dlogp, dent = (torch.empty(B, device=DEV) for _ in range(3))
    losses = torch.zeros(4, device=DEV)
    nind = int((z['exps'][:B] != 0).sum())
    eng.ppo_loss(B, value, logp, ent, adv, ret, old, exps, hy['clip_epsilon'], hy['value_pred_coef'],
                 hy['entropy_coef'], 1.0 / B, 1.0 / nind, dvalue, dlogp, dent, losses)
    np.testing.assert_allclose(losses.cpu().numpy(), z['mb/losses'], rtol=2e-5, atol=2e-6)
    # seeds agree with the spec
    _, dv, dl, de = csr_model.ppo_seeds(value.cpu().numpy().astype(np.float64), logp.cpu().numpy().astype(np.float64),
                                        ent.cpu().numpy().astype(np.float64), z['mb/adv'][:, 0], z['mb/ret'][:, 0],
                                        z['mb/old_logp'][:, 0], z['exps'][:B], hy['clip_epsilon'],
                                        hy['value_pred_coef'], hy['entropy_coef'])
    np.testing.assert_allclose(dvalue.cpu().numpy(), dv, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dlogp.cpu().numpy(), dl, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dent.cpu().numpy(), de, rtol=1e-5, atol=1e-9)
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dvalue, dlogp, dent, grads)
    torch.cuda.synchronize()
    _check_grads(eng, grads, lambda nm: z[helpers.golden_key('grad/', nm)])


@pytest.mark.parametrize('name', ['case_a', 'case_c'])
def test_module_surface_autograd(name, small_path):
    """The reference's own call pattern: value_net(x), policy_net.get_log_prob_entropy(x, a), loss.backward()."""
    from test_oracle_golden import CASE_HYPER
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    hy = CASE_HYPER[name]
    B = z['fwd/value'].shape[0]
    policy_net, value_net, ac = helpers.build_product(cfg)
    ac.load_state_dict(sd)
    ac.to(DEV)
    xs = [[torch.tensor(f).to(DEV) for f in s] for s in states[:B]]
    actions = torch.from_numpy(z['actions'][:B]).float().to(DEV)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    adv, ret, old, exps = t(z['mb/adv']), t(z['mb/ret']), t(z['mb/old_logp']), t(z['exps'][:B])
    ind = exps.nonzero(as_tuple=False).squeeze(1)
    values_pred = value_net(xs)
    value_loss = (values_pred - ret).pow(2).mean()
    log_probs, entropy = policy_net.get_log_prob_entropy(xs, actions)
    ratio = torch.exp(log_probs[ind] - old[ind])
    surr1 = ratio * adv[ind]
    surr2 = torch.clamp(ratio, 1.0 - hy['clip_epsilon'], 1.0 + hy['clip_epsilon']) * adv[ind]
    surr_loss = -torch.min(surr1, surr2).mean()
    entropy_loss = -entropy[ind].mean()
    loss = surr_loss + hy['value_pred_coef'] * value_loss + hy['entropy_coef'] * entropy_loss
    np.testing.assert_allclose([loss.item(), value_loss.item(), surr_loss.item(), entropy_loss.item()], z['mb/losses'],
                               rtol=2e-5, atol=2e-6)
    loss.backward()
    scale = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith('grad/'))
    for k, p in ac.named_parameters():
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)
        ref = z['grad/' + k]
        assert np.abs(g - ref).max() <= 1e-5 * scale or _rel_l2(g, ref) <= 1e-4, k
    with torch.no_grad():
        act = policy_net.select_action(xs, mean_action=True).cpu().numpy()
    for b in range(B):
        st = int(np.argmax(states[b][8]))
        if st == 0:
            assert states[b][6][int(act[b, 0])]
        elif st == 1:
            assert states[b][7][int(act[b, 1])]


@pytest.mark.parametrize('name', ['case_a', 'case_b'])
def test_gae_bit_exact(name):
    from test_oracle_golden import CASE_HYPER
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    hy = CASE_HYPER[name]
    _, _, _, eng, *_ = _engine_setup(cfg, sd, states[:4], z['actions'][:4])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    rewards, masks, values = t(z['rewards']), t(z['masks']), t(z['gae/values'][:, 0])
    for tag, (ga, ta) in dict(cfg=(hy['gamma'], hy['tau']), g95=(0.99, 0.95)).items():
        adv, ret = torch.empty_like(values), torch.empty_like(values)
        eng.gae(rewards, masks, values, ga, ta, adv, ret)
        assert np.array_equal(adv.cpu().numpy(), z['gae/%s_adv' % tag][:, 0]), tag
        assert np.array_equal(ret.cpu().numpy(), z['gae/%s_ret' % tag][:, 0]), tag


@pytest.mark.parametrize('name', ['case_a', 'case_b', 'case_c'])
def test_update_params_matches_reference(name, small_path):
    """The whole update_params (two calls): loss curve within 1e-4 rel per step, parameters rel-L2 <= 1e-4,
    loss_iter bookkeeping, TB scalar tags; clipping active on the very first step only."""
    from drl_urban_planning_amd import PPOUpdater, synth
    from oracle.ref_import import ScalarLog
    from test_oracle_golden import CASE_HYPER, CASE_EPOCHS, CASE_SEED, CASE_B
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    hy = CASE_HYPER[name]
    policy_net, value_net, ac = helpers.build_product(cfg)
    ac.load_state_dict(sd)
    ac.to(DEV)
    up = PPOUpdater(policy_net, value_net, lr=hy['lr'], eps=hy['eps'], weight_decay=hy['weight_decay'],
                    gamma=hy['gamma'], tau=hy['tau'], clip_epsilon=hy['clip_epsilon'],
                    value_pred_coef=hy['value_pred_coef'], entropy_coef=hy['entropy_coef'],
                    num_optim_epoch=CASE_EPOCHS[name], mini_batch_size=CASE_B[name])
    replay = synth.Replay(states, z['actions'], z['masks'], z['rewards'], z['exps'])
    log = ScalarLog()
    np.random.seed(CASE_SEED[name] + 11)
    elapsed = up.update_params(replay, 0, tb_logger=log)
    assert elapsed > 0
    np.testing.assert_allclose(up.last_losses, z['upd/scalars'], rtol=1e-4, atol=2e-6)
    assert up.loss_iter == int(z['upd/loss_iter'])
    per_step = [v for (tag, v, s) in log.scalars if tag == 'loss/loss']
    assert len(per_step) == up.loss_iter
    assert [s for (tag, v, s) in log.scalars if tag == 'loss/loss'] == list(range(up.loss_iter))
    assert sum(1 for (tag, v, s) in log.scalars if tag == 'loss/epoch_loss') == CASE_EPOCHS[name]
    mine = {k: v.detach().cpu().numpy() for k, v in ac.state_dict().items()}
    for k in mine:
        assert _rel_l2(mine[k], z['upd_sd/' + k]) <= 1e-4, (k, _rel_l2(mine[k], z['upd_sd/' + k]))
    np.random.seed(CASE_SEED[name] + 12)
    up.update_params(replay, 1, tb_logger=log)
    mine = {k: v.detach().cpu().numpy() for k, v in ac.state_dict().items()}
    for k in mine:
        assert _rel_l2(mine[k], z['upd2_sd/' + k]) <= 2e-4, (k, _rel_l2(mine[k], z['upd2_sd/' + k]))
    n1 = z['upd/scalars'].shape[0]
    np.testing.assert_allclose(up.last_losses, z['upd2/scalars'][n1:], rtol=2e-4, atol=5e-6)


@pytest.mark.parametrize('name', ['case_a', 'case_c'])
def test_resume_from_checkpoint_continues_the_run(name):
    """Networks + PPOUpdater.state_dict() saved after the first update_params, loaded into fresh objects: the second
    update_params must land where the uninterrupted reference run does (Adam moments, step counts and the
    already-consumed first-step clipping all restored)."""
    import io
    from drl_urban_planning_amd import PPOUpdater, synth
    from test_oracle_golden import CASE_HYPER, CASE_EPOCHS, CASE_SEED, CASE_B
    z, sd, states = helpers.load_case(name)
    cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
    hy = CASE_HYPER[name]
    kw = dict(lr=hy['lr'], eps=hy['eps'], weight_decay=hy['weight_decay'], gamma=hy['gamma'], tau=hy['tau'],
              clip_epsilon=hy['clip_epsilon'], value_pred_coef=hy['value_pred_coef'], entropy_coef=hy['entropy_coef'],
              num_optim_epoch=CASE_EPOCHS[name], mini_batch_size=CASE_B[name])
    replay = synth.Replay(states, z['actions'], z['masks'], z['rewards'], z['exps'])
    policy_net, value_net, ac = helpers.build_product(cfg)
    ac.load_state_dict(sd)
    ac.to(DEV)
    up = PPOUpdater(policy_net, value_net, **kw)
    np.random.seed(CASE_SEED[name] + 11)
    up.update_params(replay, 0)
    blob = io.BytesIO()
    torch.save({'model': ac.state_dict(), 'optim': up.state_dict()}, blob)      # through a real serialisation
    blob.seek(0)
    ckpt = torch.load(blob, map_location='cpu', weights_only=False)
    assert ckpt['optim']['clip_pending'] is False and sum(ckpt['optim']['group_steps']) > 0
    policy2, value2, ac2 = helpers.build_product(cfg, seed=99)                   # different initial weights
    ac2.load_state_dict(ckpt['model'])
    ac2.to(DEV)
    up2 = PPOUpdater(policy2, value2, **kw)
    up2.load_state_dict(ckpt['optim'])
    assert up2.loss_iter == int(z['upd/loss_iter'])
    np.random.seed(CASE_SEED[name] + 12)
    up2.update_params(replay, 1)
    mine = {k: v.detach().cpu().numpy() for k, v in ac2.state_dict().items()}
    for k in mine:
        assert _rel_l2(mine[k], z['upd2_sd/' + k]) <= 2e-4, (k, _rel_l2(mine[k], z['upd2_sd/' + k]))


def _random_case(D, L, heads, S, land, road, value, T, max_nodes, max_edges, seed, road_fraction, n_range):
    from drl_urban_planning_amd import synth
    cfg = helpers.make_cfg(D=D, L=L, S=S, heads=heads, land_head=land, road_head=road, value_head=value,
                           max_nodes=max_nodes, max_edges=max_edges)
    _, _, ac = helpers.build_product(cfg, seed=seed)
    sd = helpers.perturbed_state_dict(ac, seed + 1, scale=0.05)
    replay = synth.make_replay(T, 'hlg', max_nodes=max_nodes, max_edges=max_edges, seed=seed,
                               road_fraction=road_fraction, n_range=n_range)
    return cfg, sd, replay


@pytest.mark.parametrize('D,L,heads,n_range,T', [(256, 3, 1, (200, 345), 5), (128, 2, 4, (40, 90), 6),
                                                    (64, 2, 2, (30, 60), 6),
                                                    # 520..700 nodes: staged in LDS (one workgroup per CU) but too
                                                    # large for the single-round-trip stage-in -> the looped stage-in
                                                    (64, 2, 2, (520, 700), 3)])
def test_wide_model_matches_oracle(D, L, heads, n_range, T):
    """BASELINE cfg-2 dims (3 layers x 256) on HLG-shaped graphs: exercises the MFMA GEMM tiles."""
    cfg, sd, replay = _random_case(D, L, heads, (64, 16), (32, 1), (32, 1), (32, 32, 1), T, n_range[1] + 5,
                                   int(5.55 * n_range[1]) + 10, seed=21, road_fraction=0.3, n_range=n_range)
    _check_against_oracle(cfg, sd, replay, heads, T)


@pytest.mark.parametrize('D,L,heads,n_range,T,wgrad', [(256, 3, 1, (200, 345), 24, 1), (64, 2, 2, (30, 60), 16, 1),
                                                          (256, 3, 1, (200, 345), 24, 3), (128, 3, 4, (40, 90), 12, 2)])
def test_forked_step_is_bit_identical_to_the_single_stream_step(D, L, heads, n_range, T, wgrad, tune):
    """The per-sample chains and the grouped weight-gradient launch run on the engine's side stream underneath the GCN
    layers (tune knob side_stream, default on).  Same kernels, same reduction orders: values and every gradient must equal
    the single-stream step BIT FOR BIT, on every repetition (a missing fork / join dependency would show up as a mismatch)."""
    cfg, sd, replay = _random_case(D, L, heads, (64, 16), (32, 1), (32, 1), (32, 32, 1), T, n_range[1] + 5,
                                   int(5.55 * n_range[1]) + 10, seed=23, road_fraction=0.3, n_range=n_range)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, replay.states, replay.actions)
    g = torch.Generator().manual_seed(9)
    seeds = [torch.randn(T, generator=g).to(DEV) for _ in range(3)]

    def run():
        value, logp, ent = _forward(eng, pk, mb, flat)
        grads = torch.zeros(eng.n_floats, device=DEV)
        eng.backward(pk, mb, flat, seeds[0], seeds[1], seeds[2], grads)
        torch.cuda.synchronize()
        return [value.clone(), logp.clone(), ent.clone(), grads]
    # side_wgrad: 1 = the default (weight-gradient GEMMs on the side stream for minibatches of <= 98 k nodes: all of these),
    # 3 = always next to the dgrad, 2 = behind it; side_heads (default on) puts the pointer-head chain there as well
    tune('side_wgrad', wgrad)
    tune('side_stream', 0)
    ref = run()
    tune('side_stream', 1)
    for rep in range(6):
        out = run()
        for a, b in zip(ref, out):
            assert torch.equal(a, b), 'repetition %d differs from the single-stream step' % rep


@pytest.mark.parametrize('D,L,heads,n_range,T,fold', [(256, 3, 1, (200, 345), 24, 1), (64, 3, 2, (30, 60), 16, 1),
                                                         (64, 2, 2, (30, 60), 16, 0), (128, 3, 4, (380, 520), 6, 1), (128, 3, 4, (360, 400), 8, 1),
                                                         (64, 3, 2, (520, 700), 3, 1)])
def test_dma_stage_in_is_bit_identical_to_the_register_stage_in(D, L, heads, n_range, T, fold, tune):
    """Round 3: the P/Q GEMM of layers 2..L stores 2^(C2 x) block by block (tune knob pq_exp, default on) and the
    message-passing kernels copy their slices HBM -> LDS by LDS-DMA instead of loading, exponentiating and ds_writing
    them.  Same arithmetic (the exponential moved into the GEMM epilogue, everything else untouched): values and every
    gradient must equal the register-staged path BIT FOR BIT.  nt_min_wgs = 1 makes the small cases take the 128-wide
    LDS-DMA GEMM tile that carries the exp-form store; fold = 0 also runs layer 1 through the plain kernels; the
    380..520- and 520..700-node cases are the large size classes (H in HBM / one workgroup per CU, looped DMA), 360..400
    the DHM size class whose backward walks the neighbour ids from global memory (two workgroups per CU)."""
    tune('nt_min_wgs', 1)
    tune('fold_layer1', fold)
    cfg, sd, replay = _random_case(D, L, heads, (64, 16), (32, 1), (32, 1), (32, 32, 1), T, n_range[1] + 5,
                                   int(5.55 * n_range[1]) + 10, seed=29, road_fraction=0.3, n_range=n_range)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, replay.states, replay.actions)
    g = torch.Generator().manual_seed(9)
    seeds = [torch.randn(T, generator=g).to(DEV) for _ in range(3)]

    def run():
        value, logp, ent = _forward(eng, pk, mb, flat)
        grads = torch.zeros(eng.n_floats, device=DEV)
        eng.backward(pk, mb, flat, seeds[0], seeds[1], seeds[2], grads)
        torch.cuda.synchronize()
        return [value.clone(), logp.clone(), ent.clone(), grads]
    tune('pq_exp', 0)
    ref = run()
    tune('pq_exp', 1)
    for rep in range(3):
        out = run()
        for name, a, b in zip(('value', 'logp', 'entropy', 'grads'), ref, out):
            assert torch.equal(a, b), '%s differs from the register-staged path (repetition %d, max |diff| %.3e)' % (
                name, rep, (a - b).abs().max().item())


@pytest.mark.parametrize('gain,bias', [(40.0, 0.0), (1.0, 3.0), (400.0, 0.5), (12.0, 0.0)])
def test_saturating_edge_mlp_on_the_dma_path_matches_oracle(gain, bias, tune):
    """The saturation cases with the exp-form store in play (3 layers, nt_min_wgs = 1): GEMM output blocks beyond the
    exp-form limit stay linear and raise their flag, the workgroups that meet one convert their slice in LDS (log2 of the
    exp-form blocks; back to exp form when their own slice allows it, else the linear walk) -- gain 12 leaves a mix of
    flagged and unflagged blocks."""
    tune('nt_min_wgs', 1)
    D, L, heads, T, n_range = 64, 3, 2, 6, (30, 60)
    cfg, sd, replay = _random_case(D, L, heads, (64, 16), (32, 1), (32, 1), (32, 32, 1), T, n_range[1] + 5,
                                   int(5.55 * n_range[1]) + 10, seed=33, road_fraction=0.3, n_range=n_range)
    sd = dict(sd)
    for k in list(sd):
        if 'edge_fc_layers' in k:
            sd[k] = sd[k] * gain if k.endswith('weight') else sd[k] + bias
    # gain 400 over THREE layers leaves every tanh saturated to the last ulp: the node-encoder gradient is then the sum of a
    # few O(1) terms whose rounding differs between the exp-form and torch's tanh (rel-L2 3.6e-4 with or without the DMA
    # path, profiles/archive/r03_diag_saturating.log) -- the tolerance is widened for that case only
    _check_against_oracle(cfg, sd, replay, heads, T, tol=1e-3 if gain >= 100 else (3e-4 if gain > 1 else 1e-4))


def test_wide_model_with_full_head_input_tensor_matches_oracle(tune):
    """fe_half = 0: the land-use head's input tensor FE = [m ; m*c] is materialised in full and its first Linear / weight
    gradient run as plain GEMMs (the path of head shapes head.hip does not cover) instead of the default m-only form with
    per-graph effective weights."""
    tune('fe_half', 0)
    cfg, sd, replay = _random_case(128, 2, 4, (64, 16), (32, 1), (32, 1), (32, 32, 1), 6, 95, int(5.55 * 90) + 10, seed=21,
                                   road_fraction=0.3, n_range=(40, 90))
    _check_against_oracle(cfg, sd, replay, 4, 6)


def test_wide_model_with_unfused_head_backward_matches_oracle(tune):
    """The land-use head's feature backward as two launches (K = 32 GEMM writing dFE + he_feat_bwd) -- the path of head
    widths other than 32 -- instead of the default fused kernel (tune knob he_fused)."""
    tune('he_fused', 0)
    cfg, sd, replay = _random_case(128, 2, 4, (64, 16), (32, 1), (32, 1), (32, 32, 1), 6, 95, int(5.55 * 90) + 10, seed=21,
                                   road_fraction=0.3, n_range=(40, 90))
    _check_against_oracle(cfg, sd, replay, 4, 6)


@pytest.mark.parametrize('nprod', [6, 9])
def test_wide_model_with_split_gemm_matches_oracle(nprod, tune):
    """The opt-in split-bf16 node GEMMs (tune knob gemm_split, csrc/gemm_split.hip) under the SAME oracle tolerances as the
    exact-fp32 default: values, log-probs, entropies, losses and every gradient at the BASELINE dims."""
    tune('gemm_split', nprod)
    cfg, sd, replay = _random_case(256, 3, 1, (64, 16), (32, 1), (32, 1), (32, 32, 1), 5, 350, int(5.55 * 345) + 10, seed=21,
                                   road_fraction=0.3, n_range=(200, 345))
    _check_against_oracle(cfg, sd, replay, 1, 5)


@pytest.mark.parametrize('gain,bias', [(40.0, 0.0), (1.0, 3.0), (400.0, 0.5)])
def test_saturating_edge_mlp_matches_oracle(gain, bias):
    """Edge-MLP pre-activations far outside the exp-form range of the message-passing kernels (|P|, |Q| > 19 or
    |b| > 2): those workgroups must take the linear-form walk and still match the reference (tanh saturates; some
    graphs of the minibatch stay in range, so both walks run in one launch)."""
    D, L, heads, T, n_range = 64, 2, 2, 6, (30, 60)
    cfg, sd, replay = _random_case(D, L, heads, (64, 16), (32, 1), (32, 1), (32, 32, 1), T, n_range[1] + 5,
                                   int(5.55 * n_range[1]) + 10, seed=33, road_fraction=0.3, n_range=n_range)
    sd = dict(sd)
    hit = 0
    for k in list(sd):
        if 'edge_fc_layers' in k:
            if k.endswith('weight'):
                sd[k] = sd[k] * gain
            else:
                sd[k] = sd[k] + bias
            hit += 1
    assert hit >= 4           # actor and critic share the encoder: L weights + L biases at least
    _check_against_oracle(cfg, sd, replay, heads, T, tol=3e-4 if gain > 1 else 1e-4)


def _check_against_oracle(cfg, sd, replay, heads, T, tol=1e-4):
    states, actions = replay.states, replay.actions
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, states, actions)
    value, logp, ent = _forward(eng, pk, mb, flat)
    P = helpers.oracle_params(sd)
    xs = orc.tensorfy(states)
    act_t = torch.from_numpy(actions).float()
    g = torch.Generator().manual_seed(5)
    adv, ret = torch.randn(T, 1, generator=g), torch.randn(T, 1, generator=g)
    with torch.no_grad():
        v0 = orc.value_forward(P, xs, heads)
        lp0, en0 = orc.get_log_prob_entropy(P, xs, act_t, heads)
    np.testing.assert_allclose(value.cpu().numpy(), v0[:, 0].numpy(), rtol=tol, atol=tol / 10)
    np.testing.assert_allclose(logp.cpu().numpy(), lp0[:, 0].numpy(), rtol=tol, atol=tol / 10)
    np.testing.assert_allclose(ent.cpu().numpy(), en0[:, 0].numpy(), rtol=tol, atol=tol / 10)
    old = lp0 + 0.3 * torch.randn(T, 1, generator=g)
    exps = torch.ones(T)
    loss, vl, sl, el = orc.ppo_losses(P, xs, act_t, adv, ret, old, exps, 0.2, 0.5, 0.01, heads)
    loss.backward()
    dvalue, dlogp, dent = (torch.empty(T, device=DEV) for _ in range(3))
    losses = torch.zeros(4, device=DEV)
    eng.ppo_loss(T, value, logp, ent, adv[:, 0].to(DEV), ret[:, 0].to(DEV), old[:, 0].to(DEV), exps.to(DEV), 0.2, 0.5,
                 0.01, 1.0 / T, 1.0 / T, dvalue, dlogp, dent, losses)
    np.testing.assert_allclose(losses.cpu().numpy(), [loss.item(), vl.item(), sl.item(), el.item()], rtol=1e-4, atol=1e-5)
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dvalue, dlogp, dent, grads)
    torch.cuda.synchronize()
    _check_grads(eng, grads, lambda nm: (P[nm].grad if P[nm].grad is not None else torch.zeros_like(P[nm])).numpy(),
                 rtol_l2=tol)


def test_degenerate_rows_match_reference_semantics(small_path):
    """A row without any valid candidate (all logits = the pad constant: in fp32 the reference's normalised
    logits are all 0, policy.py:50-52), a row whose stored action points at a masked slot, a stage-2 row."""
    from drl_urban_planning_amd import synth
    cfg = helpers.make_cfg(D=16, L=2, max_nodes=24, max_edges=50)
    _, _, ac = helpers.build_product(cfg, seed=2)
    sd = helpers.perturbed_state_dict(ac, 3, scale=0.2)
    states, actions = [], []
    for i, stage in enumerate([0, 0, 1, 2, 0]):
        rng = np.random.default_rng(50 + i)
        s, a = synth.make_state(rng, 18, 40, 24, 50, min(stage, 1))
        if stage == 2:
            s[8] = np.array([0, 0, 1], dtype=np.float32)
        states.append(s)
        actions.append(a)
    states[0][6][:] = False                                   # no land-use candidate at all
    bad = int(np.flatnonzero(~states[1][6])[0])
    actions[1][0] = bad                                       # action outside the candidate set
    actions = np.stack(actions)
    T = len(states)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, states, actions)
    value, logp, ent = _forward(eng, pk, mb, flat)
    P = helpers.oracle_params(sd)
    xs = orc.tensorfy(states)
    act_t = torch.from_numpy(actions).float()
    v0 = orc.value_forward(P, xs, 1)
    lp0, en0 = orc.get_log_prob_entropy(P, xs, act_t, 1)
    np.testing.assert_allclose(value.cpu().numpy(), v0[:, 0].detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), en0[:, 0].detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), lp0[:, 0].detach().numpy(), rtol=1e-6, atol=1e-5)
    assert float(logp[0]) == 0.0 and float(logp[1]) < -4e9 and float(logp[3]) == 0.0
    # gradients of a loss that touches every row
    w = torch.tensor([0.3, -0.2, 0.5, 0.1, -0.4])
    (v0[:, 0] * w).sum().backward(retain_graph=True)
    ((lp0[:, 0] * w)[[0, 2, 3, 4]].sum() + (en0[:, 0] * w).sum()).backward()      # row 1's logp is -4e9: leave it out
    dl = w.clone()
    dl[1] = 0.0
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, w.to(DEV), dl.to(DEV), w.to(DEV), grads)
    torch.cuda.synchronize()
    _check_grads(eng, grads, lambda nm: (P[nm].grad if P[nm].grad is not None else torch.zeros_like(P[nm])).numpy())


def test_maximum_size_graphs_and_mixed_pads():
    """Graphs at the padding limits of the shipped configs (1000 nodes / 3000 edges, no padding left: the
    un-staged fallback of the edge kernels) mixed with small graphs that use different pad sizes."""
    from drl_urban_planning_amd import synth
    cfg = helpers.make_cfg(D=32, L=2, heads=2, S=(64, 16), max_nodes=1000, max_edges=3000)
    _, _, ac = helpers.build_product(cfg, seed=11)
    sd = helpers.perturbed_state_dict(ac, 12, scale=0.05)
    states, actions = [], []
    for i, (n, e, N, E, stage) in enumerate([(1000, 3000, 1000, 3000, 0), (40, 100, 64, 128, 0), (1000, 3000, 1000, 3000, 1),
                                              (700, 2800, 1500, 4000, 0), (25, 60, 1000, 3000, 1)]):
        rng = np.random.default_rng(1000 + i)
        s, a = synth.make_state(rng, n, e, N, E, stage)
        states.append(s)
        actions.append(a)
    actions = np.stack(actions)
    T = len(states)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, states, actions)
    value, logp, ent = _forward(eng, pk, mb, flat)
    P = helpers.oracle_params(sd)
    g = torch.Generator().manual_seed(5)
    adv, ret = torch.randn(T, 1, generator=g), torch.randn(T, 1, generator=g)
    # the oracle needs equal pads inside one call: evaluate row by row (rows are independent)
    v0, lp0, en0 = [], [], []
    with torch.no_grad():
        for b in range(T):
            xs = orc.tensorfy(states[b:b + 1])
            v0.append(orc.value_forward(P, xs, 2))
            lp, en = orc.get_log_prob_entropy(P, xs, torch.from_numpy(actions[b:b + 1]).float(), 2)
            lp0.append(lp)
            en0.append(en)
    v0, lp0, en0 = torch.cat(v0), torch.cat(lp0), torch.cat(en0)
    np.testing.assert_allclose(value.cpu().numpy(), v0[:, 0].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), lp0[:, 0].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), en0[:, 0].numpy(), rtol=1e-4, atol=1e-5)
    old = lp0 + 0.3 * torch.randn(T, 1, generator=g)
    loss = 0
    for b in range(T):
        xs = orc.tensorfy(states[b:b + 1])
        vb = orc.value_forward(P, xs, 2)
        lpb, enb = orc.get_log_prob_entropy(P, xs, torch.from_numpy(actions[b:b + 1]).float(), 2)
        ratio = torch.exp(lpb - old[b:b + 1])
        surr = torch.min(ratio * adv[b:b + 1], torch.clamp(ratio, 0.8, 1.2) * adv[b:b + 1])
        loss = loss + (-surr.sum() + 0.5 * (vb - ret[b:b + 1]).pow(2).sum() - 0.01 * enb.sum()) / T
    loss.backward()
    dvalue, dlogp, dent = (torch.empty(T, device=DEV) for _ in range(3))
    losses = torch.zeros(4, device=DEV)
    eng.ppo_loss(T, value, logp, ent, adv[:, 0].to(DEV), ret[:, 0].to(DEV), old[:, 0].to(DEV), torch.ones(T, device=DEV), 0.2,
                 0.5, 0.01, 1.0 / T, 1.0 / T, dvalue, dlogp, dent, losses)
    assert abs(float(losses[0]) - float(loss)) <= 1e-4 * max(1.0, abs(float(loss)))
    grads = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dvalue, dlogp, dent, grads)
    torch.cuda.synchronize()
    _check_grads(eng, grads, lambda nm: (P[nm].grad if P[nm].grad is not None else torch.zeros_like(P[nm])).numpy())


def test_full_size_properties():
    """BASELINE-size minibatch (2048 HLG-shaped graphs, D=256, L=3) through size-independent properties:
    padding invariance, row-order invariance (bit-exact), run-to-run determinism of the gradients."""
    from drl_urban_planning_amd import packer, synth
    B = 2048
    cfg = helpers.make_cfg(D=256, L=3)
    _, _, ac = helpers.build_product(cfg, seed=3)
    sd = ac.state_dict()
    replay = synth.make_replay(B, 'hlg', seed=9, unique=256)
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, replay.states, replay.actions)
    value, logp, ent = _forward(eng, pk, mb, flat)
    assert torch.isfinite(value).all() and torch.isfinite(logp).all() and torch.isfinite(ent).all()
    assert (ent > 0).all() and (logp < 0).all()
    # states are tiled from 256 unique ones: identical graphs must give identical rows (bit-exact)
    v = value.cpu().numpy().reshape(8, 256)
    assert np.array_equal(v, np.broadcast_to(v[0], v.shape))
    lp = logp.cpu().numpy().reshape(8, 256)
    assert np.array_equal(lp, np.broadcast_to(lp[0], lp.shape))
    # row order invariance
    perm = np.random.default_rng(0).permutation(B)
    sched2 = packer.Schedule(pk, [perm], DEV)
    mb2, _ = sched2.minibatch(0)
    value2, logp2, ent2 = _forward(eng, pk, mb2, flat)
    assert np.array_equal(value2.cpu().numpy(), value.cpu().numpy()[perm])
    assert np.array_equal(ent2.cpu().numpy(), ent.cpu().numpy()[perm])
    # padding invariance: same graphs re-padded tightly
    sub = 64
    tight = []
    for s in replay.states[:sub]:
        n, e = int(s[4].sum()), int(s[5].sum())
        ei = np.full((e + 3, 2), n + 1, dtype=np.int64)
        ei[:e] = s[2][:e]
        tight.append([s[0], np.concatenate([s[1][:n], np.zeros((2, 23), np.float32)]), ei, s[3],
                      np.concatenate([s[4][:n], [False, False]]), np.concatenate([s[5][:e], [False] * 3]),
                      np.concatenate([s[6][:e], [False] * 3]), np.concatenate([s[7][:n], [False, False]]), s[8]])
    pk3 = packer.pack_replay(tight, replay.actions[:sub], 23, 52).to(DEV)
    sched3 = packer.Schedule(pk3, [np.arange(sub)], DEV)
    mb3, _ = sched3.minibatch(0)
    value3, logp3, ent3 = _forward(eng, pk3, mb3, flat)
    assert np.array_equal(value3.cpu().numpy(), value.cpu().numpy()[:sub])
    assert np.array_equal(logp3.cpu().numpy(), logp.cpu().numpy()[:sub])
    # determinism of the backward
    g = torch.Generator().manual_seed(1)
    dv, dl, de = (torch.randn(B, generator=g).to(DEV) / B for _ in range(3))
    value, logp, ent = _forward(eng, pk, mb, flat)
    g1 = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dv, dl, de, g1)
    value, logp, ent = _forward(eng, pk, mb, flat)
    g2 = torch.zeros(eng.n_floats, device=DEV)
    eng.backward(pk, mb, flat, dv, dl, de, g2)
    torch.cuda.synchronize()
    assert torch.isfinite(g1).all()
    assert torch.equal(g1, g2)
    assert float(g1.abs().max()) > 0


def test_native_failure_modes():
    from drl_urban_planning_amd import native
    cfg = helpers.make_cfg(**helpers.CASE_MODEL['case_a'])
    z, sd, states = helpers.load_case('case_a')
    _, _, _, eng, flat, pk, sched, mb = _engine_setup(cfg, sd, states[:4], z['actions'][:4])
    eng.ensure_workspace(mb)
    import ctypes as C
    v = torch.empty(4, device=DEV)
    rc = eng.lib.upamd_forward(eng.handle, C.c_void_p(pk.dev_buf.data_ptr()), C.byref(pk.layout), C.byref(mb),
                               C.c_void_p(flat.data_ptr()), C.c_void_p((eng.ws.data_ptr() + 255) // 256 * 256),
                               C.c_int64(1024), C.c_void_p(v.data_ptr()), C.c_void_p(v.data_ptr()),
                               C.c_void_p(v.data_ptr()), 0, None)
    assert rc == -4 and b'workspace too small' in eng.lib.upamd_last_error()


@pytest.mark.parametrize('seed', list(range(10)))
def test_random_small_configurations(seed, small_path):
    """Fuzz over model shapes and graph sizes (tiny graphs below one 8-node chunk, one-layer models, several heads,
    road-only and land-only minibatches): forward values, losses and every gradient against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.choice([16, 32, 64]))
    heads = int(rng.choice([h for h in (1, 2, 4) if D % h == 0]))
    L = int(rng.integers(1, 4))
    lo = int(rng.choice([3, 6, 12, 30]))
    n_range = (lo, lo + int(rng.integers(0, 40)))
    T = int(rng.integers(2, 7))
    road_fraction = float(rng.choice([0.0, 0.3, 1.0]))
    cfg, sd, replay = _random_case(D, L, heads, (64, 16), (32, 1), (32, 1), (32, 32, 1), T, n_range[1] + 3,
                                   int(5.55 * n_range[1]) + 12, seed=50 + seed, road_fraction=road_fraction,
                                   n_range=n_range)
    _check_against_oracle(cfg, sd, replay, heads, T)
