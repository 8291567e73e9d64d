"""Data-parallel protocol on CPU (world size 2, gloo): global row counts once per epoch, loss scaled by
GLOBAL counts on every rank, ONE all-reduce(sum) of the flat gradient buffer (+4 loss scalars) per step.
The per-shard compute is done by the oracle here (the HIP engine needs a GPU); what is under test is
`drl_urban_planning_amd.dist` and the scaling rule `PPOUpdater.step` applies."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import helpers
from oracle import sgnn_oracle as orc
from test_oracle_golden import CASE_HYPER


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _flat(P, names):
    return torch.cat([(P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])).reshape(-1) for k in names])


def _shard_loss(P, states, actions, adv, ret, old, exps, hy, inv_rows, inv_ind, heads):
    """Sum-form loss of one shard with global scaling (what upamd_ppo_loss computes per rank)."""
    xs = orc.tensorfy(states)
    value = orc.value_forward(P, xs, heads)
    logp, ent = orc.get_log_prob_entropy(P, xs, actions, heads)
    ind = exps.nonzero(as_tuple=False).squeeze(1)
    vl = (value - ret).pow(2).sum() * inv_rows
    ratio = torch.exp(logp[ind] - old[ind])
    s1 = ratio * adv[ind]
    s2 = torch.clamp(ratio, 1 - hy['clip_epsilon'], 1 + hy['clip_epsilon']) * adv[ind]
    sl = -torch.min(s1, s2).sum() * inv_ind
    el = -ent[ind].sum() * inv_ind
    loss = sl + hy['value_pred_coef'] * vl + hy['entropy_coef'] * el
    return loss, torch.stack([loss.detach(), vl.detach(), sl.detach(), el.detach()])


def _worker(rank, world, port, name, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'] = str(rank)
    os.environ['WORLD_SIZE'] = str(world)
    torch.set_num_threads(1)
    from drl_urban_planning_amd.dist import DistContext, global_counts, shard_rows
    ctx = DistContext.from_env(backend='gloo')
    assert ctx.world == world and ctx.rank == rank
    z, sd, states = helpers.load_case(name)
    hy = CASE_HYPER[name]
    heads = helpers.CASE_MODEL[name]['heads']
    B = z['mb/adv'].shape[0]
    rows = shard_rows(list(range(B)), rank, world)
    P = helpers.oracle_params(sd)
    names = list(P.keys())
    exps_all = torch.from_numpy(z['exps'][:B]).float()
    counts = global_counts(ctx, [[len(rows)], [int((exps_all[rows] != 0).sum())]], 'cpu')
    assert counts[0][0] == B and counts[1][0] == int((exps_all != 0).sum())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[:B][rows], dtype=np.float32))
    loss, scal = _shard_loss(P, [states[i] for i in rows], t(z['actions']), t(z['mb/adv']), t(z['mb/ret']),
                             t(z['mb/old_logp']), exps_all[rows], hy, 1.0 / counts[0][0], 1.0 / counts[1][0], heads)
    loss.backward()
    buf = torch.cat([_flat(P, names), scal.float()])
    ctx.all_reduce_sum(buf)                       # the one collective of a step
    ctx.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, 'reduced.npy'), buf.numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('name', ['case_a'])
def test_two_rank_gradient_allreduce_matches_full_batch(name, tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), 'reduced.npy'))
    z, sd, states = helpers.load_case(name)
    P = helpers.oracle_params(sd)
    names = list(P.keys())
    ref = np.concatenate([z[helpers.golden_key('grad/', k)].reshape(-1) for k in names])
    scale = np.abs(ref).max()
    np.testing.assert_allclose(got[:-4], ref, rtol=1e-4, atol=2e-6 * scale)
    np.testing.assert_allclose(got[-4:], z['mb/losses'], rtol=1e-5, atol=1e-6)


def test_shard_rows_and_single_rank_counts():
    from drl_urban_planning_amd.dist import DistContext, global_counts, shard_rows
    assert shard_rows(list(range(8)), 1, 2) == [4, 5, 6, 7]
    with pytest.raises(ValueError):
        shard_rows(list(range(7)), 0, 2)
    assert global_counts(DistContext(), [[3, 4], [1, 2]], 'cpu') == [[3, 4], [1, 2]]


def test_split_minibatch_is_a_partition_and_balances_edge_counts():
    """Every rank takes B / G rows of the global minibatch; the union is the minibatch; with balance='edges' the ranks'
    edge totals differ by far less than with contiguous slices on a mixed (bimodal) minibatch."""
    from drl_urban_planning_amd.dist import split_minibatch
    rng = np.random.default_rng(0)
    B, G = 2048, 8
    rows = rng.permutation(50000)[:B]
    e = np.where(rng.random(B) < 0.5, rng.integers(1150, 1920, B), rng.integers(1320, 2216, B))     # HLG / DHM mix
    for balance in ('none', 'edges'):
        parts = [split_minibatch(rows, e, r, G, balance) for r in range(G)]
        assert all(p.size == B // G for p in parts)
        assert np.array_equal(np.sort(np.concatenate(parts)), np.sort(rows))
        pos = {int(v): i for i, v in enumerate(rows)}
        for p in parts:            # rows keep the order they have in the global minibatch
            assert all(pos[int(a)] < pos[int(b)] for a, b in zip(p[:-1], p[1:]))
    lookup = dict(zip(rows.tolist(), e.tolist()))
    tot = lambda bal: np.array([sum(lookup[int(v)] for v in split_minibatch(rows, e, r, G, bal)) for r in range(G)])
    spread_none, spread_edges = np.ptp(tot('none')) / tot('none').mean(), np.ptp(tot('edges')) / tot('edges').mean()
    assert spread_edges < 0.002 and spread_edges < spread_none / 5, (spread_none, spread_edges)
    assert np.array_equal(split_minibatch(rows, e, 0, 1), rows)
    with pytest.raises(ValueError):
        split_minibatch(rows[:7], e[:7], 0, 2)


def _agree_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from drl_urban_planning_amd import packer, synth
    from drl_urban_planning_amd.dist import DistContext, batch_fingerprint, broadcast_batch, order_fingerprint
    ctx = DistContext.from_env(backend='gloo')
    assert ctx.backend == 'gloo'
    assert ctx.agree_min(5 + rank) == 5
    assert ctx.same_everywhere([3, 2 ** 40 + 1]) and not ctx.same_everywhere([rank])
    np.random.seed(123)
    perm = np.random.permutation(1000)
    assert ctx.same_everywhere([order_fingerprint(perm)])
    np.random.seed(123 + rank)
    assert not ctx.same_everywhere([order_fingerprint(np.random.permutation(1000))])
    # rank 0's batch reaches rank 1 as compact records and packs to the same bytes
    rep = synth.make_replay(6, 'hlg', max_nodes=60, max_edges=200, seed=5, n_range=(20, 50)) if rank == 0 else None
    got = broadcast_batch(ctx, rep, src=0)
    ref = synth.make_replay(6, 'hlg', max_nodes=60, max_edges=200, seed=5, n_range=(20, 50))
    assert batch_fingerprint(got) == batch_fingerprint(ref)
    if rank == 1:
        assert all(packer.is_record(s) for s in got.states)
        for s, r in zip(got.states, ref.states):
            for a, b in zip(packer.expand_state(s, padded=True), r):
                assert np.array_equal(a, b)
        pk = packer.pack_replay(got.states, got.actions, 23, 52, pin=False)
        pk0 = packer.pack_replay(ref.states, ref.actions, 23, 52, pin=False)
        # same live content; the pad-size columns differ (records are trimmed to the last used row)
        cols = [c for c in range(13) if c not in (packer.M_PADN, packer.M_PADE)]
        assert np.array_equal(pk.meta[:, cols], pk0.meta[:, cols])
        np.save(os.path.join(out_dir, 'ok.npy'), np.ones(1))
    ctx.barrier()
    ctx.close()


def test_two_rank_agreements_and_batch_broadcast(tmp_path):
    port = _free_port()
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), 'ok.npy'))


def test_auto_mode_fingerprint_tells_replays_with_equal_scalars_apart():
    """dp_mode='auto' calls a replay 'the same on every rank' only if the per-row scalars AND what the packer extracted
    from the states agree: two synthetic replays with the same actions / rewards / masks / exps but different graphs
    (equal action seeds on rank-local shards) must not be mistaken for one shared batch; a replay of compact wire
    records and the padded tuples it was made from are the same replay."""
    from drl_urban_planning_amd import dist, packer, synth
    a = synth.make_replay(12, 'hlg', max_nodes=64, max_edges=200, seed=3, road_fraction=0.4, n_range=(20, 55))
    b = synth.make_replay(12, 'hlg', max_nodes=64, max_edges=200, seed=4, road_fraction=0.4, n_range=(20, 55))
    b = synth.Replay(b.states, a.actions, a.masks, a.rewards, a.exps)          # other graphs, the same per-row scalars
    assert dist.batch_fingerprint(a) == dist.batch_fingerprint(b)
    nd, num = a.states[0][1].shape[1], a.states[0][0].shape[0]

    def fp(states, actions):
        return dist.states_fingerprint(packer.pack_replay(states, actions, nd, num))
    assert fp(a.states, a.actions) != fp(b.states, b.actions)
    assert fp(a.states, a.actions) == fp(a.states, a.actions)
    records = [packer.compact_state(s) for s in a.states]
    assert fp(records, a.actions) == fp(a.states, a.actions)


def _bucket_worker(rank, world, port, out_dir):
    """The host side of the bucketed all-reduce on CPU tensors: ranges reduced one by one through ``all_reduce_sum_async``
    equal ONE collective over the whole buffer bit for bit; the iteration's permutations travel in one 2-D broadcast."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from drl_urban_planning_amd.dist import DistContext
    ctx = DistContext.from_env(backend='gloo')
    g = torch.Generator().manual_seed(100 + rank)
    n = 10_000
    grads = torch.randn(n + 4, generator=g)
    whole = grads.clone()
    ctx.all_reduce_sum(whole)
    ranges = [(7000, n + 4), (4000, 7000), (1500, 4000), (0, 1500)]          # readiness order: back to front, as the engine reports them
    works = [ctx.all_reduce_sum_async(grads[b:e]) for b, e in ranges]
    for w in works:
        w.wait()
    assert torch.equal(grads, whole)
    assert ctx.all_reduce_sum_async(torch.ones(3)).wait() is not None
    # rank 0's permutations for the whole iteration in ONE broadcast (agent.PPOUpdater.draw_permutations)
    np.random.seed(5 + rank)                                                  # the ranks' own streams differ on purpose
    perms = np.stack([np.random.permutation(50) for _ in range(3)])
    got = ctx.broadcast_array(perms)
    np.random.seed(5)
    want = np.stack([np.random.permutation(50) for _ in range(3)])
    assert got.shape == (3, 50) and np.array_equal(got, want)
    ctx.barrier()
    if rank == 1:
        open(os.path.join(out_dir, 'ok'), 'w').close()
    ctx.close()


def test_bucketed_ranges_equal_one_collective_and_permutations_travel_once(tmp_path):
    port = _free_port()
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(str(tmp_path / 'ok'))
