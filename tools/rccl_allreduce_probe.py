"""Lab: cost of ONE all-reduce(sum) of the flat gradient buffer (932 504 floats) through torch.distributed on RCCL, per call, as the
update loop issues it (in-stream, between kernels).  Single rank (UPAMD_DIST_FORCE_INIT=1) measures the fixed software cost; under
torchrun with N ranks the same script measures the real collective.

    UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python tools/rccl_allreduce_probe.py
"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_urban_planning_amd import DistContext
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
torch.cuda.set_device(dev)
os.dup2(2, 1)
ctx = DistContext.from_env(device=dev)
g = torch.randn(932504, device=dev)
a = torch.randn(4096, 4096, device=dev)
def loop(n, reduce, work):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if work: b = a @ a
        if reduce: ctx.all_reduce_sum(g)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for _ in range(3): loop(10, True, True)
res = dict(world=ctx.world, backend=ctx.backend,
           allreduce_only_us=loop(200, True, False), matmul_only_us=loop(50, False, True), matmul_plus_allreduce_us=loop(50, True, True))
if ctx.rank == 0:
    sys.stderr.write('RCCL_PROBE %s\n' % res)
ctx.close()
