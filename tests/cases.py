"""Small seeded replay sets with the edge cases SURVEY.md section 8(c) lists:
mixed stages, a road-only / land-use-only set, self-loop and duplicate edges, isolated nodes,
a graph with n == N and e == E (no padding)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from drl_urban_planning_amd import synth  # noqa: E402


def quirky_replay(T, max_nodes, max_edges, seed=0, road_fraction=0.35, n_lo=8, full_row=True, dead_candidate=False):
    """T small states; row 0 has a self-loop and a duplicated edge, row 1 has isolated nodes,
    row 2 (if ``full_row``) is unpadded (n == max_nodes, e == max_edges)."""
    states, actions = [], []
    for i in range(T):
        rng = np.random.default_rng(seed * 7919 + i)
        if full_row and i == 2:
            n, e = max_nodes, max_edges
        else:
            n = int(rng.integers(n_lo, max_nodes))
            e = int(min(max_edges - 2, max(3, round(rng.uniform(1.2, 2.6) * n))))
            e = min(e, n * (n - 1) // 2)
        stage = 1 if rng.random() < road_fraction else 0
        s, a = synth.make_state(rng, n, e, max_nodes, max_edges, stage)
        if i == 0:
            ei = s[2]
            live = int(s[5].sum())
            ei[1] = (ei[0, 0], ei[0, 0])          # self loop
            ei[2] = ei[0]                          # duplicate of edge 0
            if live + 1 <= max_edges:              # one more live edge that duplicates edge 3 reversed
                ei[live] = (ei[3, 1], ei[3, 0])
                s[5][live] = True
        if i == 1:
            ei = s[2]
            live = int(s[5].sum())
            n_live = int(s[4].sum())
            victim = n_live - 1                    # make the last live node isolated
            for k in range(live):
                if ei[k, 0] == victim:
                    ei[k, 0] = 0
                if ei[k, 1] == victim:
                    ei[k, 1] = 1
            s[7][victim] = False
            if stage == 1 and not s[7].any():
                s[7][0] = True
                a[1] = 0.0
            if stage == 1 and not s[7][int(a[1])]:
                a[1] = float(np.flatnonzero(s[7])[0])
        if dead_candidate and i >= 3 and stage == 0 and int(s[5].sum()) < max_edges:
            # a land-use candidate on a slot that is NOT a live edge (edge_mask False): the reference still scores it
            # (sgnn: zero message; rl-mlp: the embedding of zero features = the encoder bias)
            s[6][int(s[5].sum())] = True
        states.append(s)
        actions.append(a)
    rng = np.random.default_rng(seed * 7919 + 100003)
    actions = np.stack(actions).astype(np.float32)
    masks = np.ones(T)
    rewards = np.zeros(T)
    t = 0
    while t < T:
        end = min(T, t + int(rng.integers(3, 7))) - 1
        masks[end] = 0.0
        rewards[end] = rng.uniform(0.0, 5.0)
        if rng.random() < 0.5 and end - 1 >= t:
            rewards[end - 1] = rng.uniform(-1.0, 1.0)   # non-terminal shaped reward
        t = end + 1
    exps = np.ones(T)
    if T > 4:
        exps[3] = 0.0                              # one row excluded from surrogate/entropy means
    return synth.Replay(states, actions, masks, rewards, exps)


def stack_states(states):
    """list[T] of list[9] -> dict of 9 stacked arrays (compact fixture form)."""
    names = ['numerical', 'node_features', 'edge_index', 'current_node', 'node_mask', 'edge_mask',
             'land_use_mask', 'road_mask', 'stage']
    return {n: np.stack([s[i] for s in states]) for i, n in enumerate(names)}


def unstack_states(d):
    names = ['numerical', 'node_features', 'edge_index', 'current_node', 'node_mask', 'edge_mask',
             'land_use_mask', 'road_mask', 'stage']
    T = d['stage'].shape[0]
    return [[np.ascontiguousarray(d[n][t]) for n in names] for t in range(T)]
