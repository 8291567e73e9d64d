"""Host packer (csrc/packer.cpp) against the python packer spec (tests/csr_model.pack_state). CPU only."""
import numpy as np
import pytest
import torch

import cases
import csr_model
from drl_urban_planning_amd import native, packer, synth


def _check(replay):
    T = len(replay.states)
    pk = packer.pack_replay(replay.states, replay.actions, synth.NODE_DIM, synth.NUMERICAL_DIM, n_threads=2, pin=False)
    L = pk.layout
    meta = pk.meta
    X = pk.section('x', np.float32, L.total_nodes * native.NODE_PAD).reshape(-1, native.NODE_PAD)
    nmask = pk.section('nmask', np.uint8, L.total_nodes)
    rowptr = pk.section('rowptr', np.int32, L.total_nodes + T)
    inc_nbr = pk.section('inc_nbr', np.uint16, 2 * L.total_edges)
    he_src = pk.section('he_src', np.uint16, L.total_he)
    he_dst = pk.section('he_dst', np.uint16, L.total_he)
    he_live = pk.section('he_live', np.uint8, L.total_he)
    he_slot = pk.section('he_slot', np.int32, L.total_he)
    rn_node = pk.section('rn_node', np.uint16, L.total_rn)
    order = pk.section('order', np.uint16, L.total_nodes)
    hinc_ptr = pk.section('hinc_ptr', np.int32, L.total_nodes + T)
    hinc_nbr = pk.section('hinc_nbr', np.uint16, 2 * L.total_he)
    hinc_he = pk.section('hinc_he', np.uint16, 2 * L.total_he)
    numerical = pk.section('numerical', np.float32, T * synth.NUMERICAL_DIM).reshape(T, -1)
    cur = pk.section('cur', np.float32, T * native.NODE_PAD).reshape(T, -1)
    meta_dev = pk.section('meta', np.int32, T * native.META_STRIDE).reshape(T, -1)
    assert np.array_equal(meta_dev, meta)
    for t in range(T):
        g = csr_model.pack_state(replay.states[t], replay.actions[t])
        m = meta[t]
        assert (m[0], m[1], m[4], m[5], m[6]) == (g['n'], g['e'], g['stage'], g['act'], g['n_mask']), t
        assert m[2] == g['he_src'].size and m[3] == g['rn_node'].size
        assert m[7] == g['pad_n'] and m[8] == g['pad_e']
        n, e = g['n'], g['e']
        no, eo, ho, ro, rp = m[9], m[10], m[11], m[12], m[13]
        np.testing.assert_array_equal(X[no:no + n, :synth.NODE_DIM], g['X'].astype(np.float32))
        assert not X[no:no + n, synth.NODE_DIM:].any()
        np.testing.assert_array_equal(nmask[no:no + n].astype(bool), g['nmask'])
        np.testing.assert_array_equal(rowptr[rp:rp + n + 1], g['row_ptr'])
        np.testing.assert_array_equal(inc_nbr[2 * eo:2 * eo + 2 * e], g['inc_nbr'])
        np.testing.assert_array_equal(he_src[ho:ho + m[2]], g['he_src'])
        np.testing.assert_array_equal(he_dst[ho:ho + m[2]], g['he_dst'])
        np.testing.assert_array_equal(he_live[ho:ho + m[2]], g['he_live'])
        np.testing.assert_array_equal(he_slot[ho:ho + m[2]], g['he_slot'])
        np.testing.assert_array_equal(rn_node[ro:ro + m[3]], g['rn_node'])
        np.testing.assert_array_equal(order[no:no + n], g['order'])
        np.testing.assert_array_equal(hinc_ptr[rp:rp + n + 1], g['hinc_ptr'])
        nhi = int(g['hinc_ptr'][-1])
        np.testing.assert_array_equal(hinc_nbr[2 * ho:2 * ho + nhi], g['hinc_nbr'])
        np.testing.assert_array_equal(hinc_he[2 * ho:2 * ho + nhi], g['hinc_he'])
        np.testing.assert_array_equal(numerical[t], replay.states[t][0])
        np.testing.assert_array_equal(cur[t, :synth.NODE_DIM], replay.states[t][3])
    return pk


def test_packer_quirky_cases():
    _check(cases.quirky_replay(24, 40, 96, seed=3))
    _check(cases.quirky_replay(12, 30, 64, seed=9, road_fraction=1.0))


def test_packer_hlg_shape_and_schedule():
    replay = synth.make_replay(16, 'hlg', seed=4, road_fraction=0.3)
    pk = _check(replay)
    sched = packer.Schedule(pk, [np.arange(0, 8), np.array([15, 3, 9])], 'cpu')
    mb, it = sched.minibatch(1)
    assert mb.B == 3 and it['n_nodes'] == int(pk.meta[[15, 3, 9], 0].sum())
    flat = sched.dev.numpy()
    base = it['base']
    np.testing.assert_array_equal(flat[base:base + 3], [15, 3, 9])
    np.testing.assert_array_equal(flat[base + 3:base + 7], np.concatenate([[0], np.cumsum(pk.meta[[15, 3, 9], 0])]))
    # incidence (edge-direction) prefix sums, read by models with num_edge_fc_layers > 1: the last of the four offset arrays
    ninc = 2 * pk.meta[[15, 3, 9], 1]
    assert mb.n_inc == it['n_inc'] == int(ninc.sum()) and it['max_inc'] == int(ninc.max())
    np.testing.assert_array_equal(flat[base + 4 * 3 + 3:base + 5 * 3 + 4], np.concatenate([[0], np.cumsum(ninc)]))
    assert mb.inc_off_dev == sched.dev.data_ptr() + 4 * (base + 4 * 3 + 3)


def test_packer_rejects_bad_input():
    replay = cases.quirky_replay(4, 20, 40, seed=1, full_row=False)
    replay.states[1][2][0, 0] = 25          # live edge endpoint outside [0, pad_n)
    with pytest.raises(RuntimeError, match='endpoint'):
        packer.pack_replay(replay.states, replay.actions, synth.NODE_DIM, synth.NUMERICAL_DIM, pin=False)


@pytest.mark.parametrize('field,new_len', [(0, 40), (3, 20), (8, 2)])
def test_packer_rejects_short_fixed_width_fields(field, new_len):
    """numerical / current-node / stage are copied with fixed widths by the C packer: a shorter array (config / state
    mismatch) must raise instead of reading past the buffer -- on the C fast path and on the Python path alike."""
    replay = cases.quirky_replay(4, 20, 40, seed=1, full_row=False)
    replay.states[2][field] = np.ascontiguousarray(replay.states[2][field][:new_len])
    with pytest.raises(ValueError, match='state 2'):
        packer.pack_replay(replay.states, replay.actions, synth.NODE_DIM, synth.NUMERICAL_DIM, pin=False)


def test_action_outside_candidates_maps_to_minus_one():
    replay = cases.quirky_replay(6, 20, 40, seed=2, road_fraction=0.0, full_row=False)
    s = replay.states[4]
    bad = int(np.flatnonzero(~s[6])[0])
    replay.actions[4, 0] = bad
    pk = packer.pack_replay(replay.states, replay.actions, synth.NODE_DIM, synth.NUMERICAL_DIM, pin=False)
    assert pk.meta[4, 5] == -1


def _sections(pk, Fn, mlp_fields=True):
    """Every section of a pack as arrays (the alignment gaps between sections are not written by the packer)."""
    L, T = pk.layout, pk.T
    N, E, H, R = int(L.total_nodes), int(L.total_edges), int(L.total_he), int(L.total_rn)
    spec = [('meta', np.int32, T * native.META_STRIDE), ('x', np.float32, N * native.NODE_PAD), ('nmask', np.uint8, N),
            ('rowptr', np.int32, N + T), ('inc_nbr', np.uint16, 2 * E), ('he_src', np.uint16, H), ('he_dst', np.uint16, H),
            ('he_live', np.uint8, H), ('he_slot', np.int32, H), ('rn_node', np.uint16, R), ('numerical', np.float32, T * Fn),
            ('cur', np.float32, T * native.NODE_PAD), ('order', np.uint16, N), ('hinc_ptr', np.int32, N + T),
            ('hinc_nbr', np.uint16, 2 * H), ('hinc_he', np.uint16, 2 * H)]
    if mlp_fields:
        spec += [('he_sel', np.uint16, H), ('xbar', np.float32, T * native.NODE_PAD)]
    return {name: pk.section(name, dt, cnt).copy() for name, dt, cnt in spec}


def _same_pack(a, b, Fn):
    sa, sb = _sections(a, Fn), _sections(b, Fn)
    return all(np.array_equal(sa[k], sb[k]) for k in sa) and np.array_equal(a.meta, b.meta)


def test_host_helper_matches_python_path():
    """csrc/_upamd_host.so (pointer tables extracted in C) must give byte-identical packs to the Python loop, and
    step aside for inputs it does not recognise (tensors, wrong dtypes) instead of guessing."""
    import torch
    if packer._host_helper() is None:
        pytest.skip('_upamd_host.so not built')
    rep = synth.make_replay(24, 'hlg', max_nodes=60, max_edges=200, seed=5, n_range=(20, 50))
    fast = packer.pack_replay(rep.states, rep.actions, 23, rep.states[0][0].shape[-1], pin=False)
    saved = packer._host_mod
    try:
        packer._host_mod = None
        slow = packer.pack_replay(rep.states, rep.actions, 23, rep.states[0][0].shape[-1], pin=False)
    finally:
        packer._host_mod = saved
    Fn = rep.states[0][0].shape[-1]
    assert _same_pack(fast, slow, Fn)
    # states given as torch tensors / float64 arrays: the helper reports them, the validating Python path converts
    odd = [list(s) for s in rep.states]
    odd[3] = [torch.from_numpy(a) for a in odd[3]]
    odd[7][1] = odd[7][1].astype(np.float64)
    conv = packer.pack_replay(odd, rep.actions, 23, rep.states[0][0].shape[-1], pin=False)
    assert _same_pack(conv, slow, Fn)
    # and a malformed state is still rejected
    broken = [list(s) for s in rep.states]
    broken[2] = broken[2][:8]
    with pytest.raises(ValueError):
        packer.pack_replay(broken, rep.actions, 23, rep.states[0][0].shape[-1], pin=False)


def test_compact_wire_record_is_lossless_and_packs_identically():
    """the compact record of a state (trailing empty node / edge rows dropped) expands back to
    the exact padded tuple, and a replay given as records packs to the same graph data as the padded states."""
    rep = synth.make_replay(20, 'mixed', max_nodes=120, max_edges=400, seed=9, road_fraction=0.3, n_range=(20, 60))
    quirky = cases.quirky_replay(6, 28, 60, seed=4, road_fraction=0.5, n_lo=10, full_row=True)
    for states in (rep.states, quirky.states):
        recs = [packer.compact_state(s) for s in states]
        for s, rec in zip(states, recs):
            assert packer.is_record(rec) and rec.dtype == np.uint8 and rec.ndim == 1
            back = packer.expand_state(rec, padded=True)
            assert all(np.array_equal(a, b) and np.asarray(a).dtype == b.dtype and np.asarray(a).shape == b.shape
                       for a, b in zip(s, back))
    recs = [packer.compact_state(s) for s in rep.states]
    assert sum(r.size for r in recs) < 0.6 * sum(sum(np.asarray(a).nbytes for a in s) for s in rep.states)
    Fn = rep.states[0][0].shape[-1]
    a = packer.pack_replay(rep.states, rep.actions, 23, Fn, pin=False)
    mixed = [recs[i] if i % 2 else rep.states[i] for i in range(len(recs))]        # records and padded states together
    b = packer.pack_replay(mixed, rep.actions, 23, Fn, pin=False)
    ma, mb = a.meta.copy(), b.meta.copy()
    ma[:, packer.M_PADN:packer.M_PADE + 1] = 0                                      # the records carry trimmed pads
    mb[:, packer.M_PADN:packer.M_PADE + 1] = 0
    assert np.array_equal(ma, mb)
    sa, sb = _sections(a, Fn), _sections(b, Fn)
    assert all(np.array_equal(sa[k], sb[k]) for k in sa if k != 'meta')
    with pytest.raises(ValueError):
        packer.expand_state(recs[0][:100])


def test_mlp_encoder_inputs_selected_endpoint_and_mean_features():
    """rl-mlp encoder inputs (state_encoder.py:262-282): per candidate the endpoint that represents the edge (the second
    one iff its type is FEASIBLE), per state the mean of those endpoints' raw features over the live edges."""
    replay = cases.quirky_replay(10, 30, 70, seed=4, road_fraction=0.3, dead_candidate=True)
    pk = packer.pack_replay(replay.states, replay.actions, synth.NODE_DIM, synth.NUMERICAL_DIM, pin=False)
    L, T = pk.layout, pk.T
    he_sel = pk.section('he_sel', np.uint16, max(int(L.total_he), 1))
    he_live = pk.section('he_live', np.uint8, max(int(L.total_he), 1))
    xbar = pk.section('xbar', np.float32, T * native.NODE_PAD).reshape(T, native.NODE_PAD)
    saw_second = False
    for t, s in enumerate(replay.states):
        feat, ei, emask, lmask = s[1], s[2], s[5], s[6]
        sel = np.where(feat[ei[:, 1], :14].argmax(1) == 1, ei[:, 1], ei[:, 0])
        saw_second |= bool((sel[emask] == ei[emask, 1]).any() and (ei[emask, 0] != ei[emask, 1]).any())
        want = feat[sel[emask]].astype(np.float64).mean(0)
        np.testing.assert_allclose(xbar[t, :23], want, rtol=1e-6, atol=1e-7)
        assert not xbar[t, 23:].any()
        if pk.meta[t, packer.M_STAGE] == 0:
            o = pk.meta[t, packer.M_HE_OFF]
            slots = np.flatnonzero(lmask)
            for q, k in enumerate(slots):
                assert he_live[o + q] == emask[k]
                if emask[k]:
                    assert he_sel[o + q] == sel[k]
    assert saw_second


def test_compact_record_straight_from_the_unpadded_arrays():
    """`compact_from_arrays` (what a patched ObservationExtractor.get_obs calls, observation_extractor.py:207-228) builds the
    record from the UNPADDED node / edge arrays and masks; it must equal `compact_state(padded tuple)` byte for byte --
    incl. a graph with a self-loop and duplicate edges, isolated nodes, the unpadded row (n == N, e == E) and land-use
    candidates on slots beyond the live edges -- and raise the extractor's errors when a limit is exceeded."""
    import cases
    N, E = 40, 96
    rep = cases.quirky_replay(24, N, E, seed=3, road_fraction=0.35, dead_candidate=True)
    for t, s in enumerate(rep.states):
        n, e = int(s[4].sum()), int(s[5].sum())
        lm_rows = max(e, int(np.flatnonzero(s[6])[-1]) + 1 if s[6].any() else 0)      # a candidate may sit behind the live edges
        rec = packer.compact_from_arrays(s[0], s[1][:n], s[2][:e], s[3], s[6][:lm_rows], s[7][:n], s[8], N, E)
        assert np.array_equal(rec, packer.compact_state(s)), t
        back = packer.expand_state(rec, padded=True)
        for f in range(9):
            assert np.array_equal(back[f], s[f]), (t, f)
    s = rep.states[0]
    with pytest.raises(ValueError, match='number of nodes exceeds'):
        packer.compact_from_arrays(s[0], s[1], s[2][:5], s[3], s[6][:5], s[7], s[8], N - 1, E)
    with pytest.raises(ValueError, match='number of edges exceeds'):
        packer.compact_from_arrays(s[0], s[1][:10], s[2], s[3], s[6], s[7][:10], s[8], N, E - 1)


def test_range_fill_and_byte_ranges_reproduce_the_whole_pack():
    """The streamed form of the packer (PPOUpdater.prepare packs / uploads / sweeps the replay chunk by chunk): filling the
    states range by range gives the sections ``pack_replay`` writes, and the per-section byte ranges of a chunk
    (``byte_ranges``: what one chunk's H2D copies move) are disjoint and, over all chunks, carry every section entirely."""
    rep = synth.make_replay(37, 'mixed', max_nodes=120, max_edges=400, seed=12, road_fraction=0.35, n_range=(20, 60))
    Fn = rep.states[0][0].shape[-1]
    whole = packer.pack_replay(rep.states, rep.actions, 23, Fn, pin=False)
    pk = packer.plan_replay(rep.states, rep.actions, 23, Fn, pin=False)
    assert pk.filled == 0 and np.array_equal(pk.meta, whole.meta)
    pk.host_buf.zero_()
    image = np.zeros(pk.host_buf.numel(), dtype=np.uint8)          # what HBM would hold after the chunks' copies
    cuts = [0, 5, 6, 20, 37]
    seen = []
    for t0, t1 in zip(cuts[:-1], cuts[1:]):
        pk.fill(t0, t1)
        for lo, hi in pk.byte_ranges(t0, t1):
            assert all(hi <= a or lo >= b for a, b in seen), 'chunks must not share bytes'
            seen.append((lo, hi))
            image[lo:hi] = pk.host_buf.numpy()[lo:hi]
    assert pk.filled == 37 and pk._ptrs is None
    with pytest.raises(RuntimeError):
        pk.fill(0, 1)
    assert _same_pack(pk, whole, Fn)
    staged = packer.PackedReplay(pk.meta, pk.layout, torch.from_numpy(image))
    assert _same_pack(staged, whole, Fn)
    he_sel = lambda p: p.section('he_sel', np.uint16, int(p.layout.total_he))
    xbar = lambda p: p.section('xbar', np.float32, 37 * native.NODE_PAD)
    he_slot = lambda p: p.section('he_slot', np.int32, int(p.layout.total_he))
    for sec in (he_sel, xbar, he_slot):
        assert np.array_equal(sec(staged), sec(whole))


def test_masks_only_plan_equals_the_exact_plan_and_falls_back_when_it_must():
    """The counting pass of the streamed pack reads the masks only (``exact=False``: the int64 edge lists are not touched):
    same meta table, same pack as the exact pass for states whose edges join masked nodes -- every state the extractor emits
    (observation_extractor.py:84-132) -- and ``fill`` asks for the exact pass (``NeedsExactPlan``; ``pack_replay`` and
    ``PPOUpdater.prepare`` then re-plan) when a live edge touches a node beyond the extent the masks give."""
    rep = cases.quirky_replay(12, 30, 70, seed=6, road_fraction=0.4)
    Fn = rep.states[0][0].shape[-1]
    exact = packer.plan_replay(rep.states, rep.actions, 23, Fn, pin=False, exact=True)
    light = packer.plan_replay(rep.states, rep.actions, 23, Fn, pin=False, exact=False)
    assert np.array_equal(exact.meta, light.meta) and bytes(exact.layout) == bytes(light.layout)
    exact.fill(0, exact.T)
    light.fill(0, 5)
    light.fill(5, light.T)
    assert _same_pack(exact, light, Fn)
    # a live edge onto node n (one past the last masked node): only the exact pass sees that the graph has n + 1 nodes
    states = [[np.array(f, copy=True) for f in s] for s in rep.states]
    s = states[4]
    n = int(np.flatnonzero(s[4])[-1]) + 1
    if int(np.argmax(s[8])) == 1:
        n = max(n, int(np.flatnonzero(s[7])[-1]) + 1)
    assert n < s[1].shape[0]
    k = int(np.flatnonzero(s[5])[0])
    s[2][k, 1] = n
    light = packer.plan_replay(states, rep.actions, 23, Fn, pin=False, exact=False)
    light.fill(0, 4)
    with pytest.raises(packer.NeedsExactPlan):
        light.fill(4, 8)
    want = packer.plan_replay(states, rep.actions, 23, Fn, pin=False, exact=True)
    want.fill(0, want.T)
    assert want.meta[4, packer.M_N] == n + 1
    assert _same_pack(packer.pack_replay(states, rep.actions, 23, Fn, pin=False), want, Fn)


def test_replay_without_the_rl_mlp_fields_is_the_same_everywhere_else():
    """``plan_replay(mlp_fields=False)`` (what an SGNN model asks for): the two sections only the rl-mlp encoder reads -- ``he_sel``,
    ``xbar`` -- are not planned (offset -1) and not computed (two thirds of a state's fill time: the per-edge feature mean); every
    other section holds the same bytes."""
    rep = synth.make_replay(24, 'hlg', max_nodes=64, max_edges=200, seed=5, road_fraction=0.4, n_range=(20, 55))
    full = packer.pack_replay(rep.states, rep.actions, 23, 52, pin=False)
    lean = packer.pack_replay(rep.states, rep.actions, 23, 52, pin=False, mlp_fields=False)
    assert lean.layout.off_xbar == -1 and lean.layout.off_he_sel == -1 and lean.layout.total_bytes < full.layout.total_bytes
    a, b = _sections(full, 52), _sections(lean, 52, mlp_fields=False)
    assert set(a) - set(b) == {'he_sel', 'xbar'}
    for k in b:
        assert np.array_equal(a[k], b[k]), k
    for t0, t1 in ((0, 24), (5, 17)):
        for lo, hi in lean.byte_ranges(t0, t1):
            assert 0 <= lo < hi <= lean.layout.total_bytes
