"""Seeded synthetic replay generator in the reference's state wire format.

Every state is the 9-field list that ``ObservationExtractor.get_obs`` emits
(reference: urban_planning/envs/observation_extractor.py:207-228):

    [0] numerical      f32[52]
    [1] node_features  f32[N, 23]   one-hot type(14) + 9 continuous cols in [-1, 1], zero padded
    [2] edge_index     i64[E, 2]    undirected unique pairs, padded with N-1   (:84-97)
    [3] current_node   f32[23]
    [4] node_mask      bool[N]
    [5] edge_mask      bool[E]
    [6] land_use_mask  bool[E]
    [7] road_mask      bool[N]
    [8] stage          f32[3] one-hot

and a replay is what ``TrajBatchDisc`` hands to ``update_params``
(reference: urban_planning/utils/tools.py:4-16): ``states`` (python list),
``actions f32[T, 2]``, ``masks``, ``rewards``, ``exps`` (1-D arrays of len T).

Graph shapes follow SURVEY.md section 8(d): HLG n_c=345, DHM n_c=397, e = round(5.55 n),
heavy-tailed degrees with a few hubs of degree 40-50, ~50 % "line", 32 % "point",
18 % "polygon" node classes.  The real env (geopandas/shapely) cannot run in the build
image, so this generator is the only source of replay data for tests and the bench.
"""
import numpy as np

NODE_DIM = 23          # city_config.NUM_TYPES + 1 (=14) one-hot + 9 continuous (observation_extractor.py:112-121)
NUM_TYPES_P1 = 14
NUMERICAL_DIM = 52     # 2 x (13 ratios + 13 counts)        (observation_extractor.py:38-50)

COMMUNITY_NODES = {'hlg': 345, 'dhm': 397, 'grid': 250}


class Replay:
    """Duck-typed ``TrajBatchDisc`` (reference: urban_planning/utils/tools.py:4-16)."""

    def __init__(self, states, actions, masks, rewards, exps):
        self.states = states
        self.actions = actions
        self.masks = masks
        self.rewards = rewards
        self.exps = exps
        self.next_states = None

    def __len__(self):
        return len(self.states)


def _sample_edges(rng, n, e, hubs=3):
    """e unique undirected pairs (i<j) with a heavy-tailed degree profile."""
    w = rng.gamma(2.0, 1.0, size=n)                    # skewed degrees: mean ~11, 90th pct ~22
    hub = rng.choice(n, size=min(hubs, n), replace=False)
    w[hub] = 5.0 + 2.0 * rng.random(hub.size)          # a few hubs of degree ~40-50
    p = w / w.sum()
    max_e = n * (n - 1) // 2
    e = min(e, max_e)
    got = np.zeros(0, dtype=np.int64)
    tries = 0
    while got.size < e:
        m = int((e - got.size) * 1.6) + 16
        a = rng.choice(n, size=m, p=p)
        b = rng.choice(n, size=m, p=p)
        keep = a != b
        lo = np.minimum(a[keep], b[keep])
        hi = np.maximum(a[keep], b[keep])
        key = lo * n + hi
        allk = np.concatenate([got, key])
        _, first = np.unique(allk, return_index=True)
        got = allk[np.sort(first)]
        tries += 1
        if tries > 64:      # pathological tiny graphs: fill deterministically
            full = np.array([i * n + j for i in range(n) for j in range(i + 1, n)], dtype=np.int64)
            allk = np.concatenate([got, full])
            _, first = np.unique(allk, return_index=True)
            got = allk[np.sort(first)]
    got = got[:e]
    return np.stack([got // n, got % n], axis=1).astype(np.int64)


def make_state(rng, n, e, max_nodes, max_edges, stage):
    """One padded state with ``n`` live nodes and ``e`` live edges."""
    assert n <= max_nodes and e <= max_edges
    cls = rng.choice(3, size=n, p=[0.50, 0.32, 0.18])         # 0 line, 1 point, 2 polygon
    # node type ids: polygons get land-use types 1..12, lines type 13 (road/boundary), points 0
    typ = np.where(cls == 2, rng.integers(1, 13, size=n), np.where(cls == 0, 13, 0))
    feat = np.zeros((max_nodes, NODE_DIM), dtype=np.float32)
    feat[np.arange(n), typ] = 1.0
    feat[:n, NUM_TYPES_P1:] = rng.uniform(-1.0, 1.0, size=(n, NODE_DIM - NUM_TYPES_P1)).astype(np.float32)

    edges = _sample_edges(rng, n, e)
    e = edges.shape[0]
    edge_index = np.full((max_edges, 2), max_nodes - 1, dtype=np.int64)
    edge_index[:e] = edges

    numerical = rng.uniform(0.0, 1.0, size=NUMERICAL_DIM).astype(np.float32)
    cur = np.zeros(NODE_DIM, dtype=np.float32)
    cur[rng.integers(4, 13)] = 1.0
    # fixed geometry columns of the "to be placed" node (plan_client.py:337-345 scaled per
    # observation_extractor.py:144-156): centred, unit-less placeholders inside [-1, 1]
    cur[NUM_TYPES_P1:] = np.array([0.0, 0.0, -0.9, -0.8, -0.8, -0.8, 1.0, 1.0, 1.0], dtype=np.float32)

    node_mask = np.zeros(max_nodes, dtype=bool)
    node_mask[:n] = True
    edge_mask = np.zeros(max_edges, dtype=bool)
    edge_mask[:e] = True

    land_use_mask = np.zeros(max_edges, dtype=bool)
    k = max(1, int(round(e * rng.uniform(0.05, 0.20))))
    land_use_mask[rng.choice(e, size=min(k, e), replace=False)] = True
    road_mask = np.zeros(max_nodes, dtype=bool)
    lines = np.flatnonzero(cls == 0)
    if lines.size == 0:
        lines = np.arange(n)
    k = max(1, int(round(lines.size * 0.30)))
    road_mask[rng.choice(lines, size=k, replace=False)] = True

    st = np.zeros(3, dtype=np.float32)
    st[stage] = 1.0
    action = np.zeros(2, dtype=np.float32)
    if stage == 0:
        action[0] = float(rng.choice(np.flatnonzero(land_use_mask)))
    elif stage == 1:
        action[1] = float(rng.choice(np.flatnonzero(road_mask)))
    state = [numerical, feat, edge_index, cur, node_mask, edge_mask, land_use_mask, road_mask, st]
    return state, action


def make_replay(T, community='hlg', max_nodes=1000, max_edges=3000, seed=0, road_fraction=0.0,
                unique=None, episode_len=50, n_range=None):
    """Seeded replay of ``T`` states.

    ``community``: 'hlg' | 'dhm' | 'grid' | 'mixed' (50/50 hlg+dhm).  ``unique`` < T tiles a
    pool of ``unique`` distinct states cyclically (python references; the packer still packs
    every row, so device-side data is full size) -- used by bench.py to keep host generation
    time bounded.  Seeds: state i uses ``default_rng(seed*100003 + 1000 + i)``.
    """
    pool = T if unique is None else min(unique, T)
    states, actions = [], []
    for i in range(pool):
        rng = np.random.default_rng(seed * 100003 + 1000 + i)
        comm = community if community != 'mixed' else ('hlg', 'dhm')[i % 2]
        n_c = COMMUNITY_NODES[comm]
        if n_range is not None:
            lo, hi = n_range
        elif comm == 'grid':
            lo, hi = 120, 250
        else:
            lo, hi = int(np.ceil(0.6 * n_c)), n_c
        n = int(rng.integers(lo, hi + 1))
        e = int(round(5.55 * n))
        n = min(n, max_nodes)
        e = min(e, max_edges)
        stage = 1 if rng.random() < road_fraction else 0
        s, a = make_state(rng, n, e, max_nodes, max_edges, stage)
        states.append(s)
        actions.append(a)
    if pool < T:
        states = [states[i % pool] for i in range(T)]
        actions = [actions[i % pool] for i in range(T)]
    rng = np.random.default_rng(seed * 100003 + 7)
    actions = np.stack(actions).astype(np.float32)
    masks = np.ones(T, dtype=np.float64)
    rewards = np.zeros(T, dtype=np.float64)
    t = 0
    while t < T:
        ln = int(rng.integers(max(2, episode_len - 10), episode_len + 11))
        end = min(T, t + ln) - 1
        masks[end] = 0.0
        rewards[end] = rng.uniform(0.0, 5.0)
        t = end + 1
    exps = np.ones(T, dtype=np.float64)
    return Replay(states, actions, masks, rewards, exps)


def live_counts(states):
    """(sum of live nodes, sum of live edges) of a state list."""
    nn = sum(int(s[4].sum()) for s in states)
    ee = sum(int(s[5].sum()) for s in states)
    return nn, ee
