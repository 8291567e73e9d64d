"""Rollout transport and batched action serving on the CPU path (forked workers, shared memory, eventfd doorbells): the same
server object serves the HIP modules on a GPU box (tests/test_gpu_rollout.py)."""
import multiprocessing as mp

import numpy as np
import pytest
import torch

import helpers
from drl_urban_planning_amd import packer, rollout, synth


def _policy(seed=3, D=16):
    cfg = helpers.make_cfg(D=D, L=2, max_nodes=64, max_edges=200)
    policy_net, value_net, ac = helpers.build_product(cfg, seed=seed)
    ac.load_state_dict(helpers.perturbed_state_dict(ac, seed + 1, scale=0.2))
    return policy_net, value_net, ac


def _states(T, seed, road_fraction=0.4):
    return synth.make_replay(T, 'hlg', max_nodes=64, max_edges=200, seed=seed, road_fraction=road_fraction, n_range=(20, 55))


def _worker(client, pid, n_steps, arena_name, cap_rows, cap_bytes, out_q):
    """What a patched ``sample_worker`` does per env step: ask the server for an action, push the transition."""
    arena = rollout.SharedArena(cap_rows, cap_bytes, name=arena_name)
    memory = rollout.ArenaMemory(arena)
    rep = _states(n_steps, 100 + pid)
    greedy = []
    for t, s in enumerate(rep.states):
        mean = (t % 3 == 0)
        a = client.select_action([[torch.from_numpy(f) for f in s]], mean).numpy().squeeze(0)
        stage = int(np.argmax(s[8]))
        assert (s[6] if stage == 0 else s[7])[int(a[stage])], 'action outside the mask'
        assert a[1 - stage] == 0
        if mean:
            greedy.append((t, a.copy()))
        memory.push(s, a, 0 if t == n_steps - 1 else 1, None, 0.5 * t, 1 - int(mean))
    out_q.put((pid, len(memory), greedy))
    client.close()
    arena.close(unlink=False)


def test_action_server_and_arena_transport_with_forked_workers():
    policy_net, value_net, ac = _policy()
    n_workers, n_steps = 3, 9
    server = rollout.ActionServer(policy_net, n_workers, slot_bytes=1 << 18).start()
    arenas = [rollout.SharedArena(64, 1 << 20) for _ in range(n_workers)]
    ctx = mp.get_context('fork')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(server.client(i), i, n_steps, arenas[i].name, 64, 1 << 20, q)) for i in range(n_workers)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        pid, n, greedy = q.get(timeout=120)
        results[pid] = (n, greedy)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    server.stop()
    assert server.stats['requests'] == n_workers * n_steps and server.stats['batches'] <= server.stats['requests']
    # greedy answers == the module's own select_action(mean_action=True) on the same states
    for pid, (n, greedy) in results.items():
        assert n == n_steps
        rep = _states(n_steps, 100 + pid)
        for t, a in greedy:
            want = policy_net.select_action([[torch.from_numpy(f) for f in rep.states[t]]], True).numpy().squeeze(0)
            assert np.array_equal(a, want)
    # the learner-side batch: records in place, per-row arrays in worker order; packs identically to the padded states
    batch = rollout.RecordBatch([rollout.ArenaMemory(a) for a in arenas])
    assert len(batch) == n_workers * n_steps and all(packer.is_record(s) for s in batch.states)
    ref_states = sum((_states(n_steps, 100 + pid).states for pid in range(n_workers)), [])
    for rec, ref in zip(batch.states, ref_states):
        for a, b in zip(packer.expand_state(rec, padded=True), ref):
            assert np.array_equal(a, b)
    assert batch.masks.tolist() == ([1] * (n_steps - 1) + [0]) * n_workers
    np.testing.assert_array_equal(batch.rewards[:n_steps], 0.5 * np.arange(n_steps))
    assert set(batch.exps.tolist()) == {0.0, 1.0}
    pk = packer.pack_replay(batch.states, batch.actions, 23, 52, pin=False)
    pk0 = packer.pack_replay(ref_states, batch.actions, 23, 52, pin=False)
    cols = [c for c in range(13) if c not in (packer.M_PADN, packer.M_PADE)]
    assert np.array_equal(pk.meta[:, cols], pk0.meta[:, cols])
    server.close()
    for a in arenas:
        a.close()


def test_server_batches_concurrent_requests_and_reports_errors():
    policy_net, _, _ = _policy(seed=5)
    server = rollout.ActionServer(policy_net, 4, slot_bytes=1 << 18, linger_s=0.0)
    rep = _states(4, 7)
    import threading
    out = [None] * 4

    def ask(i):
        out[i] = server.client(i).select_action([rep.states[i]], True)
    threads = [threading.Thread(target=ask, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    import time
    time.sleep(0.3)                       # all four requests are pending before the server looks
    served = server.serve_once(timeout=1.0)
    for t in threads:
        t.join()
    assert served == 4 and server.stats['batches'] == 1 and server.stats['max_rows'] == 4
    want = policy_net.select_action([[torch.from_numpy(f) for f in s] for s in rep.states], True)
    assert torch.equal(torch.cat(out), want)
    # a request the networks cannot evaluate comes back as an error, the server keeps running
    # (two states of one stage with different pad sizes cannot form one padded Categorical)
    other = synth.make_replay(1, 'hlg', max_nodes=80, max_edges=240, seed=7, road_fraction=0.0, n_range=(20, 55)).states[0]
    same_stage = [s for s in rep.states if s[8][0] == 1][:1] or [_states(1, 8, 0.0).states[0]]
    err = []

    def ask_bad():
        try:
            server.client(0).select_action([same_stage[0], other], True)
        except RuntimeError as exc:
            err.append(str(exc))
    t = threading.Thread(target=ask_bad)
    t.start()
    for _ in range(20):
        if server.serve_once(timeout=0.5):
            break
    t.join(timeout=10)
    assert err and 'action server' in err[0]
    server.close()


def test_a_silent_server_times_the_client_out_and_a_dead_client_does_not_stall_the_others():
    policy_net, _, _ = _policy(seed=6)
    server = rollout.ActionServer(policy_net, 4, slot_bytes=1 << 18, linger_s=0.0)
    rep = _states(3, 11)
    import threading
    import time
    # nobody serves: the client gives up instead of blocking the sampling phase for ever
    slow = server.client(0)
    slow.timeout_s = 0.3
    t0 = time.perf_counter()
    with pytest.raises(TimeoutError):
        slow.select_action([rep.states[0]], True)
    assert time.perf_counter() - t0 < 5.0
    # the timed-out request is still in flight on the shared slot: the client is finished (a retry would overwrite the slot
    # under the server and take the stale 'ok' for its own answer), and the server survives answering a client that no longer listens
    with pytest.raises(RuntimeError, match='closed'):
        slow.select_action([rep.states[0]], True)
    assert server.serve_once(timeout=0.5) in (0, 1)
    # client 1 dies after posting its request; client 2 must still get its answer
    dead = server.client(1)
    rec = packer.compact_state(rep.states[1])
    dead._slot()[rollout.ActionClient.REC:][:rec.size] = rec
    dead.post([int(rec.size)], True)            # ... and nobody will ever wait for the answer
    dead.close()
    out = []
    t = threading.Thread(target=lambda: out.append(server.client(2).select_action([rep.states[2]], True)))
    t.start()
    time.sleep(0.2)
    served = 0
    for _ in range(20):
        served += server.serve_once(timeout=0.5)
        if out:
            break
    t.join(timeout=10)
    want = policy_net.select_action([[torch.from_numpy(f) for f in rep.states[2]]], True)
    assert out and torch.equal(out[0], want)
    # a request whose sizes do not fit the slot is refused with an error string, not an exception in the serving thread
    bad = server.client(3)
    bad.post([1 << 30], True)
    assert server.serve_once(timeout=0.5) == 1
    hdr = bad._slot()
    assert int(hdr[rollout._H_STATUS:rollout._H_STATUS + 4].view(np.uint32)[0]) == 1 and 'does not fit' in server.last_error
    assert int(hdr[rollout._H_RESP:rollout._H_RESP + 8].view(np.uint64)[0]) == 1          # answered: the worker would wake up with the error
    server.close()


def test_arena_overflow_is_loud():
    arena = rollout.SharedArena(2, 1 << 16)
    rep = _states(3, 9)
    for s, a in zip(rep.states[:2], rep.actions[:2]):
        arena.append(s, a, 1, 0.0, 1)
    with pytest.raises(MemoryError):
        arena.append(rep.states[2], rep.actions[2], 1, 0.0, 1)
    arena.close()


def test_ragged_actions_equal_the_padded_categorical_route():
    """The action server's lean route picks from the RAGGED pointer-head logits (one entry per candidate, pack order); the
    reference -- and ``policy_net.forward`` -- build a Categorical over the PADDED slots with the pad constant on the masked
    ones (policy.py:45-65).  Fed with the CPU module's own logits, the ragged route must give the same greedy action for
    every row (mixed stages, records and tuples mixed, a row without any candidate), and samples inside each row's mask
    with the right frequencies."""
    from drl_urban_planning_amd.models import ragged_actions
    policy_net, value_net, ac = _policy(seed=5)
    rep = _states(14, 21, road_fraction=0.5)
    states = [[np.array(f, copy=True) for f in s] for s in rep.states]
    states[3][6][:] = False                                   # a land-use row without a single candidate
    states[3][8][:] = (1, 0, 0)
    x = [packer.compact_state(s) if i % 2 else s for i, s in enumerate(states)]
    with torch.no_grad():
        land, road, stage = policy_net.forward([[torch.from_numpy(np.asarray(f)) for f in s] for s in states])
    pk = packer.pack_replay(x, np.zeros((len(x), 2), np.float32), 23, 52, pin=False)
    # the ragged logits in pack order, cut out of the padded ones
    z_he, z_rn, il, ir = [], [], 0, 0
    want = np.zeros((len(x), 2), dtype=np.float32)
    for b, s in enumerate(states):
        st = int(np.argmax(s[8]))
        if st == 0:
            z_he.append(land.logits[il][torch.from_numpy(s[6])] + land.logits[il].exp().sum().log() * 0)    # (log-softmaxed: a per-row shift)
            want[b, 0] = float(land.probs[il].argmax())
            il += 1
        elif st == 1:
            z_rn.append(road.logits[ir][torch.from_numpy(s[7])])
            want[b, 1] = float(road.probs[ir].argmax())
            ir += 1
    z_he, z_rn = torch.cat(z_he), torch.cat(z_rn)
    got = ragged_actions(pk, x, z_he, z_rn, np.ones(len(x), dtype=bool), 'cpu')
    assert np.array_equal(got, want), (got, want)
    # sampling: inside the mask, and the empirical frequencies of one row follow its probabilities
    torch.manual_seed(0)
    b = next(i for i, s in enumerate(states) if int(np.argmax(s[8])) == 0 and s[6].sum() >= 3)
    draws = np.stack([ragged_actions(pk, x, z_he, z_rn, np.zeros(len(x), dtype=bool), 'cpu') for _ in range(400)])
    for i, s in enumerate(states):
        st = int(np.argmax(s[8]))
        if st in (0, 1) and (s[6] if st == 0 else s[7]).any():
            assert (s[6] if st == 0 else s[7])[draws[:, i, st].astype(int)].all() and not draws[:, i, 1 - st].any()
    row = sum(1 for s in states[:b] if int(np.argmax(s[8])) == 0)
    p = land.probs[row].numpy()
    freq = np.bincount(draws[:, b, 0].astype(int), minlength=p.size) / draws.shape[0]
    assert np.abs(freq - p).max() < 0.12
    assert (draws[:, 3, 0] >= 0).all() and len(set(draws[:, 3, 0].tolist())) > 3       # the candidate-less row: uniform over the pads


def test_record_views_are_built_on_access_and_pack_like_a_list():
    """``packer.RecordViews`` (what ``RecordBatch.states`` is over arenas): a sequence that carries the records' addresses and sizes
    and builds a record's zero-copy view only when it is asked for -- ``plan_replay`` never touches an element.  It must index,
    slice, iterate and pack exactly like the list of views it replaces."""
    rep = _states(12, 21)
    arena = rollout.SharedArena(16, 1 << 20)
    for t, s in enumerate(rep.states):
        arena.append(s, rep.actions[t], 1, 0.0, 1)
    batch = rollout.RecordBatch([rollout.ArenaMemory(arena)])
    views = batch.states
    assert isinstance(views, packer.RecordViews) and len(views) == 12 and views.addr.shape == (12,) and views.size.shape == (12,)
    eager, _, _, _, _ = arena.rows()
    assert all(np.array_equal(a, b) for a, b in zip(views, eager))
    assert np.array_equal(views[-1], eager[-1]) and [bytes(v) for v in views[3:6]] == [bytes(v) for v in eager[3:6]]
    with pytest.raises(IndexError):
        views[12]
    a = packer.pack_replay(views, batch.actions, 23, 52, pin=False)
    b = packer.pack_replay(eager, batch.actions, 23, 52, pin=False)
    from test_packer import _sections           # (section by section: the alignment gaps of the buffer are never written)
    sa, sb = _sections(a, 52), _sections(b, 52)
    assert all(np.array_equal(sa[k], sb[k]) for k in sa) and np.array_equal(a.meta, b.meta)
    del views, eager, a, b
    batch.close()
    arena.close()
