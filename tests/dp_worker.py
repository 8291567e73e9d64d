"""One rank of a data-parallel ``update_params`` run (launched by tests/test_gpu_dp.py, one process per rank; the
ranks share cuda:0 and talk over gloo, or over RCCL when UPAMD_DIST_BACKEND=nccl and every rank has its own GPU).

    RANK=r WORLD_SIZE=G MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_worker.py <case> <out_dir> [mode]

Goes through the documented drop-in (``HipUpdateMixin`` on a duck agent), twice, and rank 0 writes the loss log and
the updated parameters.  ``mode``: 'global' (same batch on every rank), 'bcast' (rank 0's batch broadcast as compact
records first), 'local' (each rank keeps a different half of the replay)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import dp_cases  # noqa: E402
import helpers  # noqa: E402
from duck_agent import make_duck_agent  # noqa: E402
from test_oracle_golden import CASE_B, CASE_EPOCHS, CASE_HYPER, CASE_SEED  # noqa: E402


def main():
    name, out_dir = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else 'global'
    from drl_urban_planning_amd import synth
    from drl_urban_planning_amd.dist import DistContext, broadcast_batch
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev_index = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if name == 'big_mixed':             # BASELINE model size, mixed HLG + DHM graphs (tests/dp_cases.py)
        cfg, sd, replay = dp_cases.big_mixed()
        hyper, epochs, B_glob, seed0 = dp_cases.BIG['hyper'], dp_cases.BIG['epochs'], dp_cases.BIG['B'], dp_cases.BIG['seed']
    else:
        z, sd, states = helpers.load_case(name)
        cfg = helpers.make_cfg(**helpers.CASE_MODEL[name])
        replay = synth.Replay(states, z['actions'], z['masks'], z['rewards'], z['exps'])
        hyper, epochs, B_glob, seed0 = CASE_HYPER[name], CASE_EPOCHS[name], CASE_B[name], CASE_SEED[name]
    policy_net, value_net, ac = helpers.build_product(cfg)
    ac.load_state_dict(sd)
    ac.to(dev)
    B = B_glob if mode != 'local' else B_glob // world
    agent = make_duck_agent(cfg, policy_net, value_net, ac, hyper, epochs, B)
    if mode == 'bcast':
        ctx = DistContext.from_env(device=dev)
        agent.dist_ctx = ctx
        if rank != 0:
            replay = None
        replay = broadcast_batch(ctx, replay, src=0, device=dev)
    elif mode == 'local':
        replay = dp_cases.shard(replay, rank, world)
    logs = []
    for it in range(2):
        np.random.seed(seed0 + 11 + it)
        agent.update_params(replay, it)
        up = agent._hip_updater()
        logs.append(up.last_losses.copy())
        if it == 0:
            sd1 = {k: v.detach().cpu().numpy() for k, v in ac.state_dict().items()}
    sd2 = {k: v.detach().cpu().numpy() for k, v in ac.state_dict().items()}
    up = agent._hip_updater()
    assert up.dist.world == world and up.dist.active
    if rank == 0:
        out = {'losses1': logs[0], 'losses2': logs[1], 'loss_iter': np.int64(agent.loss_iter),
               'mode': np.array(up.last_timing['dp_mode']), 'rows_per_step': np.int64(up.last_timing['rows_per_step']),
               'n_scalars': np.int64(len(agent.tb_logger.scalars)),
               'buckets': np.array(up.last_buckets if up.last_buckets else np.zeros((0, 2)), dtype=np.int64),
               'n_floats': np.int64(up.engine.n_floats)}
        out.update({'sd1/' + k: v for k, v in sd1.items()})
        out.update({'sd2/' + k: v for k, v in sd2.items()})
        np.savez(os.path.join(out_dir, 'rank0.npz'), **out)
    up.dist.barrier()
    up.dist.close()


if __name__ == '__main__':
    main()
