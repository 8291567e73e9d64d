"""Rollout serving: closed-loop throughput and latency of ``rollout.ActionServer`` on the HIP
modules -- N forked clients each send one HLG-shaped state per request (as a compact wire record, the form the patched
extractor emits, or as the padded 9-field tuple the reference's workers hold) and wait for the action -- next to what the
reference does instead: ``policy_net.select_action`` on ONE padded state on the CPU inside every worker
(urban_planning/agents/urban_planning_agent.py:60-61, policy.py:67-85; torch threads = 1, khrylib/rl/agents/agent.py:12),
timed in 1 and in N concurrent processes on this host.

    python tools/rollout_bench.py [--D 16 --L 2] [--clients 8 16 32 64] [--requests 200] [--tuples]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


WARM = 30        # untimed requests per client in front of the timed ones: workspaces, page-locked buffers, clocks, every worker up


def _client(client, pid, pool, n_req, q, ready, go):
    for i in range(WARM):
        client.select_action([pool[(pid * 7 + i) % len(pool)]], False)
    ready.put(pid)
    go.wait()                    # all clients start their timed requests together
    lat = np.empty(n_req)
    t_first = time.time()
    for i in range(n_req):
        s = pool[(pid * 7 + i) % len(pool)]
        t = time.perf_counter()
        client.select_action([s], False)
        lat[i] = time.perf_counter() - t
    q.put((pid, lat, t_first, time.time()))
    client.close()


def _cpu_worker(policy_cpu, pool, n_req, pid, q, go):
    torch.set_num_threads(1)
    go.wait()
    t = time.perf_counter()
    with torch.no_grad():
        for i in range(n_req):
            s = pool[(pid * 7 + i) % len(pool)]
            policy_cpu.select_action([[torch.tensor(x) for x in s]], False)
    q.put((pid, time.perf_counter() - t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--D', type=int, default=16)
    ap.add_argument('--L', type=int, default=2)
    ap.add_argument('--clients', type=int, nargs='+', default=[8, 16, 32, 64])
    ap.add_argument('--requests', type=int, default=400, help='timed requests per client (after the warm-up ones)')
    ap.add_argument('--tuples', action='store_true', help='clients hold padded 9-field tuples (compacted per request) instead of records')
    ap.add_argument('--cpu-procs', type=int, nargs='+', default=[1, 8, 16])
    ap.add_argument('--cpu-requests', type=int, default=40)
    ap.add_argument('--profile', type=int, default=0, help='cProfile N in-process batched select_action calls of 64 records (stderr)')
    args = ap.parse_args()
    torch.set_num_threads(1)
    from drl_urban_planning_amd import packer, rollout, synth
    w = dict(community='hlg', D=args.D, L=args.L, max_nodes=1000, max_edges=3000)
    cfg = bench.model_cfg(w)
    pool_t = synth.make_replay(64, 'hlg', max_nodes=1000, max_edges=3000, seed=300).states
    pool_r = [packer.compact_state(s) for s in pool_t]
    pool = pool_t if args.tuples else pool_r
    out = {'model': 'SGNN %d x %d' % (args.L, args.D), 'state': 'HLG-shaped, pads 1000/3000, %s' % ('padded tuples' if args.tuples else 'compact records'),
           'record_bytes_mean': float(np.mean([r.size for r in pool_r])), 'host_cpus': os.cpu_count(), 'serving': [], 'cpu_select_action': []}
    ctx = mp.get_context('fork')
    # ---- the reference's way: CPU forward of one padded state per env step, inside every worker (before any HIP work: plain fork)
    policy_cpu, _, _ = bench.build_networks(cfg, seed=0)
    for procs in args.cpu_procs:
        q, go = ctx.Queue(), ctx.Event()
        ps = [ctx.Process(target=_cpu_worker, args=(policy_cpu, pool_t, args.cpu_requests, i, q, go)) for i in range(procs)]
        for p in ps:
            p.start()
        time.sleep(0.5)
        t0 = time.perf_counter()
        go.set()
        times = [q.get()[1] for _ in ps]
        wall = time.perf_counter() - t0
        for p in ps:
            p.join()
        out['cpu_select_action'].append({'processes': procs, 'actions_per_s': procs * args.cpu_requests / wall,
                                         'ms_per_action_per_process': 1e3 * float(np.mean(times)) / args.cpu_requests})
    # ---- the action server on the GPU modules
    policy_net, value_net, ac = bench.build_networks(cfg, seed=0)
    ac.to('cuda:0')
    with torch.no_grad():
        policy_net.select_action(pool_t[:4], True)              # HIP initialised before the first fork
    if args.profile:
        import cProfile, pstats
        # what one serving round runs for 64 single-state requests (rollout.ActionServer._actions -> serve_actions), in process
        batch = packer.RecordList(pool_r[:64], np.array([r.ctypes.data for r in pool_r[:64]], dtype=np.uint64),
                                  np.array([r.size for r in pool_r[:64]], dtype=np.int64))
        flags = np.zeros(64, dtype=bool)
        backend = policy_net._backend[0]
        for _ in range(5):
            backend.serve_actions(batch, flags)
        t0 = time.perf_counter()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.profile):
            backend.serve_actions(batch, flags)
        pr.disable()
        out['inprocess_ms_per_64_row_batch'] = 1e3 * (time.perf_counter() - t0) / args.profile
        pstats.Stats(pr, stream=sys.stderr).sort_stats('cumulative').print_stats(28)
    for n in args.clients:
        server = rollout.ActionServer(policy_net, n, slot_bytes=1 << 18, mp_context=ctx)
        q, ready, go = ctx.Queue(), ctx.Queue(), ctx.Event()
        procs = server.launch(_client, [(i, pool, args.requests, q, ready, go) for i in range(n)], ctx)
        for _ in procs:
            ready.get(timeout=120)              # every client has finished its warm-up requests
        st0 = dict(server.stats)
        go.set()
        res = [q.get(timeout=180) for _ in procs]
        for p in procs:
            p.join()
        server.stop()
        st = {k: server.stats[k] - st0.get(k, 0) for k in ('batches', 'requests', 'rows', 'busy_s')}
        server.close()
        lat = np.concatenate([r[1] for r in res]) * 1e3
        wall = max(r[3] for r in res) - min(r[2] for r in res)          # first timed request sent .. last timed answer received
        out['serving'].append({'clients': n, 'requests': int(st['requests']), 'actions_per_s': n * args.requests / wall,
                               'batches': int(st['batches']), 'mean_rows_per_batch': st['rows'] / max(st['batches'], 1),
                               'latency_ms_p50': float(np.percentile(lat, 50)),
                               'latency_ms_p99': float(np.percentile(lat, 99)), 'server_busy_fraction': st['busy_s'] / wall,
                               'server_ms_per_batch': 1e3 * st['busy_s'] / max(st['batches'], 1),
                               'window': 'steady state: %d warm-up requests per client first, then %d timed ones started together' % (WARM, args.requests)})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
