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
from drl_urban_planning_amd import native as _nat
for kv in os.environ.get('UPAMD_TUNE', '').split(','):
    if kv:
        _nat.check(_nat.lib().upamd_tune(kv.split('=')[0].encode(), int(kv.split('=')[1])), 'upamd_tune')
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
# the slowest host-side steps of three more epochs, and what happened in them (workspace re-allocation? device allocations?)
slow = []
for rep in range(3):
    ep = up.make_epoch(it)
    for k in range(ep.nb):
        ws0 = engine.ws.data_ptr() if engine.ws is not None else 0
        na0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
        t0 = time.perf_counter()
        up.step(it, ep, k)
        dt = time.perf_counter() - t0
        slow.append((dt, rep, k, engine.ws.data_ptr() != ws0, torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - na0))
torch.cuda.synchronize()
slow.sort(reverse=True)
print('slowest host-side steps (ms, epoch, step, workspace moved, device allocations):')
for dt, rep, k, moved, na in slow[:6]:
    print('   %.3f  epoch %d step %d  ws moved %s  hipMalloc calls %d' % (1e3 * dt, rep, k, moved, na))
import gc
_gc_log = []
def _gc_cb(phase, info):
    if phase == 'start':
        _gc_log.append([time.perf_counter(), info['generation'], None])
    else:
        _gc_log[-1][2] = time.perf_counter() - _gc_log[-1][0]
gc.callbacks.append(_gc_cb)
import cProfile as _cp
_pr = _cp.Profile()
# eight epochs back to back, NO synchronisation in between (what bench.py and update_params do): host time of every make_epoch
# and every step
torch.cuda.synchronize()
ev = []
t_all = time.perf_counter()
for rep in range(8):
    t0 = time.perf_counter()
    _pr.enable()
    ep = up.make_epoch(it)
    _pr.disable()
    ev.append((time.perf_counter() - t0, 'make_epoch %d' % rep))
    for k in range(ep.nb):
        t0 = time.perf_counter()
        up.step(it, ep, k)
        ev.append((time.perf_counter() - t0, 'epoch %d step %d' % (rep, k)))
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
print('8 epochs unsynchronised: enqueue %.1f ms, until done %.1f ms (%.3f ms/step)' % (1e3 * t_enq, 1e3 * t_all, 1e3 * t_all / (8 * ep.nb)))
ev.sort(reverse=True)
for dt, what in ev[:10]:
    print('   %8.3f ms  %s' % (1e3 * dt, what))
print('garbage collections during the run:', [(g, round(1e3 * (dur or 0), 1)) for _, g, dur in _gc_log])
import pstats as _ps
_ps.Stats(_pr).sort_stats('tottime').print_stats(8)
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

# section profile of the fused kernel (workgroup 0, its first graph): 100 MHz stamps -> microseconds
from drl_urban_planning_amd import native
import ctypes as C
buf = torch.zeros(48, dtype=torch.int64, device=dev)
native.check(native.lib().upamd_tiny_profile(C.c_void_p(buf.data_ptr())))
ep = up.make_epoch(it)
up.step(it, ep, 0)
torch.cuda.synchronize()
native.check(native.lib().upamd_tiny_profile(None))
st = buf.cpu().numpy()
names = {0: 'lists+num encoder', 1: 'C, q chain', 2: 'encode nodes', 3: 'GCN forward', 4: 'means', 5: 'attention fwd', 6: 'SV + value head',
         7: 'pointer head fwd', 8: 'loss seeds', 9: 'value head bwd', 10: 'num encoder bwd', 11: 'attention dense bwd',
         12: 'attention core bwd', 13: 'pointer head bwd: zero', 14: 'G^L', 15: 'q chain bwd', 16: 'GCN bwd layer L: dS', 17: 'GCN bwd lower layers',
         19: 'node encoder grads', 20: 'end', 21: '  head: first chunk inputs', 22: '  head: first chunk hidden', 23: '  head: rest of the chunks',
         24: '  head: softmax stats', 25: '  head bwd: chunk 0 inputs + hidden', 26: '  head bwd: chunk 0 dpre', 27: '  head bwd: chunk 0 sums + dm',
         28: '  head bwd: other chunks', 29: '  head bwd: W1 grads', 30: '  gcn bwd L half 0: P|Q', 31: '  gcn bwd L half 0: walk',
         32: '  gcn bwd L half 0: partials + dgrad', 33: '  gcn bwd L half 0: combine', 34: '  gcn bwd L half 1'}
keys = sorted([k for k in names if st[k] > 0], key=lambda k: st[k])
print('fused kernel sections (graph 0: n=%d e=%d candidates=%d), us:' % tuple(int(it.packed.meta[int(ep.sched._host[0]), q]) for q in (0, 1, 2)))
for a, b2 in zip(keys[:-1], keys[1:]):
    print('  %-44s %8.2f' % (names[a], (st[b2] - st[a]) / 100.0))
print('  %-44s %8.2f' % ('total', (st[keys[-1]] - st[keys[0]]) / 100.0))
