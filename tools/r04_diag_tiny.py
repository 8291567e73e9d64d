"""Where a reference-dims (D = 16) optimizer step spends its wall time: host enqueue per engine call vs GPU time.
Usage (GPU box): python tools/r04_diag_tiny.py [workload]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from drl_urban_planning_amd import PPOUpdater, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'hlg_ref'
w = bench.WORKLOADS[name]
dev = torch.device('cuda', 0)
cfg = bench.model_cfg(w)
policy_net, value_net, ac = bench.build_networks(cfg, seed=0)
ac.to(dev)
up = PPOUpdater(policy_net, value_net, num_optim_epoch=4, mini_batch_size=w['B'])
T = max(w['T'], 2 * w['B'])
replay = synth.make_replay(T, w['community'], max_nodes=w['max_nodes'], max_edges=w['max_edges'], seed=100, unique=w['unique'],
                           road_fraction=w.get('road_fraction', 0.0))
np.random.seed(7)
engine = up.attach()
it = up.prepare(replay)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    ep = up.make_epoch(it)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(ep.nb):
        up.step(it, ep, k)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print('%s epoch %d: make_epoch %.2f ms, %d steps: host enqueue %.3f ms/step, wall %.3f ms/step' % (
        name, rep, 1e3 * (t1 - t0), ep.nb, 1e3 * (t2 - t1) / ep.nb, 1e3 * (t3 - t1) / ep.nb))
# per-call host cost inside one step
import cProfile
import pstats
pr = cProfile.Profile()
ep = up.make_epoch(it)
torch.cuda.synchronize()
pr.enable()
for k in range(ep.nb):
    up.step(it, ep, k)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
