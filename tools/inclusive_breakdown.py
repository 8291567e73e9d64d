"""Where one whole ``update_params(batch)`` call goes (bench.py's update_params_inclusive leg), phase by phase with a device
synchronisation between phases (so the phases do NOT overlap here -- the pipelined call is timed next to it): address tables +
counting pass, host fill, H2D, pre-pass forward sweep, GAE, per-epoch schedule + upload, the optimizer steps, read-back.

    python tools/inclusive_breakdown.py --workload hlg_ref [--unique]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='hlg_ref')
    ap.add_argument('--unique', action='store_true', help='T distinct host states (default: the bench pool of 1024)')
    args = ap.parse_args()
    torch.set_num_threads(1)
    from drl_urban_planning_amd import PPOUpdater, packer, synth
    w = dict(bench.WORKLOADS[args.workload])
    dev = torch.device('cuda', 0)
    cfg = bench.model_cfg(w)
    policy_net, value_net, ac = bench.build_networks(cfg)
    ac.to(dev)
    up = PPOUpdater(policy_net, value_net, num_optim_epoch=4, mini_batch_size=w['B'])
    T = w['T']
    replay = synth.make_replay(T, w['community'], max_nodes=w['max_nodes'], max_edges=w['max_edges'], seed=100,
                               unique=None if args.unique else w['unique'], road_fraction=w.get('road_fraction', 0.0))
    import gc
    gc.collect(); gc.freeze()
    up.attach()
    sync = lambda: torch.cuda.synchronize(dev)
    out = {'workload': args.workload, 'T': T, 'unique': bool(args.unique)}
    for rep in range(3):
        # ---- the pipelined call (what update_params does) and the unpipelined one
        for chunks in (8, 1):
            up.pipeline_chunks = chunks
            np.random.seed(1)
            sync(); t0 = time.perf_counter()
            up.update_params(replay, 0)
            sync()
            out['call_chunks%d_ms' % chunks] = 1e3 * (time.perf_counter() - t0)
            out['call_chunks%d_prepare_ms' % chunks] = 1e3 * up.last_timing['prepare']
            out['call_chunks%d_loop_ms' % chunks] = 1e3 * up.last_timing['loop']
        # ---- the phases one by one
        agent = policy_net.agent
        ph = {}
        sync(); t = time.perf_counter()
        pk = packer.plan_replay(replay.states, np.asarray(replay.actions), agent.node_dim, agent.numerical_feature_size, reuse=up._pack_cache,
                                 mlp_fields=up.backend.needs_mlp_fields())
        ph['plan'] = time.perf_counter() - t; t = time.perf_counter()
        pk.fill(0, T)
        ph['fill'] = time.perf_counter() - t; t = time.perf_counter()
        pk.alloc_device(dev); pk.upload(0, T); sync()
        ph['h2d'] = time.perf_counter() - t; t = time.perf_counter()
        R = w['B']
        rows = [np.arange(i, min(i + R, T)) for i in range(0, T, R)]
        sched = packer.Schedule(pk, rows, dev); sync()
        ph['prepass_schedule'] = time.perf_counter() - t; t = time.perf_counter()
        values, logp, ent = (torch.empty(T, device=dev) for _ in range(3))
        up._ensure_rowbufs(R)
        for k, r in enumerate(rows):
            mb, _ = sched.minibatch(k)
            up.engine.forward(pk, mb, up.flat, values[r[0]:r[-1] + 1], logp[r[0]:r[-1] + 1], ent[r[0]:r[-1] + 1], keep=False, slot=0)
        sync()
        ph['prepass_forward'] = time.perf_counter() - t
        out['bytes_packed'] = int(pk.layout.total_bytes)
        it = up.prepare(replay); sync()
        t = time.perf_counter()
        ep = up.make_epoch(it); sync()
        ph['make_epoch_x1'] = time.perf_counter() - t; t = time.perf_counter()
        for k in range(ep.nb):
            up.step(it, ep, k)
        sync()
        ph['steps_one_epoch'] = time.perf_counter() - t; t = time.perf_counter()
        up.detach(); sync()
        ph['detach'] = time.perf_counter() - t
        out['phases_ms'] = {k: 1e3 * v for k, v in ph.items()}
        out['steps_per_epoch'] = ep.nb
    print(json.dumps(out))


if __name__ == '__main__':
    main()
