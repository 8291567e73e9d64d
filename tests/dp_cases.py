"""Deterministic inputs shared by tests/dp_worker.py (one process per rank) and the tests that check the ranks' result
against the oracle: every process rebuilds the same model weights and the same replay from seeds."""
import numpy as np

import helpers

BIG = dict(D=256, L=3, max_nodes=400, max_edges=2300, T=36, B=16, epochs=2, seed=31,
           hyper=dict(lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=0.99, tau=0.95, clip_epsilon=0.2, value_pred_coef=0.5,
                      entropy_coef=0.01))


def big_mixed():
    """BASELINE model size (SGNN 3 x 256) on a mixed HLG + DHM replay (cfg-5's graph mix; pads cut to the data so the
    oracle's dense padded batch stays small): (cfg, state dict, replay)."""
    from drl_urban_planning_amd import synth
    c = BIG
    cfg = helpers.make_cfg(D=c['D'], L=c['L'], max_nodes=c['max_nodes'], max_edges=c['max_edges'])
    _, _, ac = helpers.build_product(cfg, seed=c['seed'])
    sd = helpers.perturbed_state_dict(ac, c['seed'] + 1, scale=0.03)
    replay = synth.make_replay(c['T'], 'mixed', max_nodes=c['max_nodes'], max_edges=c['max_edges'], seed=c['seed'],
                               road_fraction=0.25, episode_len=9)
    replay.exps[5] = 0.0
    return cfg, sd, replay


def shard(replay, rank, world):
    """Rank `rank`'s contiguous 1 / world of a replay ('local' mode: every rank rolled out its own episodes)."""
    from drl_urban_planning_amd import synth
    T = len(replay.states) // world
    sl = slice(rank * T, (rank + 1) * T)
    return synth.Replay(replay.states[sl], np.asarray(replay.actions)[sl], np.asarray(replay.masks)[sl],
                        np.asarray(replay.rewards)[sl], np.asarray(replay.exps)[sl])
