"""Barrier-by-barrier timeline of the fused small-model kernel (lab build: tools/lab_trace/build.py).  GPU box:
    python tools/lab_trace/trace_tiny.py [workload]
Prints every phase (span between two barriers) of workgroup 0's graph with the source line of the barrier that ends it, the
phases aggregated by source line, and the code at the heaviest lines."""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_num_threads(1)
import bench  # noqa: E402
from drl_urban_planning_amd import native  # noqa: E402

native.LIB_PATH = os.path.join(ROOT, 'tools', 'lab_trace', 'csrc', 'libupamd.so')
from drl_urban_planning_amd import PPOUpdater, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'hlg_ref'
w = bench.WORKLOADS[name]
dev = torch.device('cuda', 0)
cfg = bench.model_cfg(w)
policy_net, value_net, ac = bench.build_networks(cfg, seed=0)
ac.to(dev)
up = PPOUpdater(policy_net, value_net, num_optim_epoch=4, mini_batch_size=w['B'])
T = int(os.environ.get('TRACE_T', max(w['T'], 2 * w['B'])))
replay = synth.make_replay(T, w['community'], max_nodes=w['max_nodes'], max_edges=w['max_edges'], seed=100, unique=w['unique'],
                           road_fraction=w.get('road_fraction', 0.0))
np.random.seed(7)
engine = up.attach()
it = up.prepare(replay)
for rep in range(8):            # warm (clocks, caches)
    ep = up.make_epoch(it)
    for k in range(ep.nb):
        up.step(it, ep, k)
torch.cuda.synchronize()
buf = torch.zeros(64 + 2000, dtype=torch.int64, device=dev)
native.check(native.lib().upamd_tiny_profile(C.c_void_p(buf.data_ptr())))
ep = up.make_epoch(it)
up.step(it, ep, 0)
torch.cuda.synchronize()
native.check(native.lib().upamd_tiny_profile(None))
marks = buf.cpu().numpy()[:64]
tr = buf.cpu().numpy()[64:].reshape(-1, 2)
print('section marks relative to the first barrier stamp (us): ' + ', '.join('%d:%.2f' % (k, (marks[k] - tr[0, 0]) / 100.0) for k in range(40) if marks[k] > 0))
tr = tr[tr[:, 0] > 0]
meta = it.packed.meta[int(ep.sched._host[0])]
print('graph 0: n=%d e=%d candidates=%d; %d barriers, %.2f us from the first to the last' % (
    int(meta[0]), int(meta[1]), int(meta[2]), len(tr), (tr[-1, 0] - tr[0, 0]) / 100.0))
src = open(os.path.join(ROOT, 'tools', 'lab_trace', 'csrc', 'tiny_body.h')).read().splitlines()
by_line = defaultdict(lambda: [0, 0.0])
print('phase timeline (us, line of the closing barrier):')
for a, b in zip(tr[:-1], tr[1:]):
    d = (b[0] - a[0]) / 100.0
    rel = (b[0] - tr[0, 0]) / 100.0
    if int(b[1]) < 1000000:
        by_line[int(b[1])][0] += 1
        by_line[int(b[1])][1] += d
    print('  %7.2f  L%d   (t = %.2f)' % (d, int(b[1]), rel))
print('by source line (total us, count, line, code two lines above the barrier .. barrier):')
for line, (cnt, tot) in sorted(by_line.items(), key=lambda kv: -kv[1][1])[:40]:
    ctx = ' | '.join(x.strip()[:70] for x in src[max(0, line - 4):line])
    print('  %7.2f  x%-3d L%-5d %s' % (tot, cnt, line, ctx))
