"""Benchmark of the hot path: PPO-update samples/s on HLG-shaped graphs (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: re-execs itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # ... or under an external launcher (the driver's form): the same line

One "step" = one PPO optimizer step on one minibatch: SGNN forward (encoder once) + clipped-surrogate /
value / entropy loss + full backward + (gradient all-reduce) + Adam, exactly what
``UrbanPlanningAgent.update_policy`` does per minibatch (urban_planning_agent.py:321-346).  The replay is
packed and resident in HBM before the timed region (value/old-log-prob pre-pass and GAE are done, as they
are once per iteration in the reference).  Workload at N=1 = BASELINE.json configs[1]: HLG-shaped graphs,
SGNN 3 layers x 256, PPO minibatch 2048; with N GPUs every rank processes 2048 rows of its own replay shard
(weak scaling, global minibatch 2048*N) and the flat gradient buffer is all-reduced once per step (RCCL).
Data is synthetic (seeded generator); weights are random-init of that architecture.

With N > 1 ranks (--dp-mode global, the default) every rank holds the same replay and draws the same permutation;
global minibatch k is the reference's ``order[k*B:(k+1)*B]`` with B = 2048*N, of which every rank processes 2048
rows (edge-balanced split), i.e. the reference's minibatch sequence sharded over the ranks.

Prints ONE JSON line on rank 0.  Extra objects: ``roofline`` (dominant kernel = the fp32-MFMA node GEMM,
timed with HIP events on the launch stream inside the timed region) and ``cpu_baseline`` (the oracle =
PyTorch-CPU port of the reference path, timed on this host's cores on a bounded sample; rank 0, N=1 only); the default
N = 1 line also carries ``ref_dims``: the reference-YAML-dims workloads hlg_ref and grid_ref (= BASELINE configs[0]) measured by
child runs of this script.  Multi-rank lines carry ``rccl_ranks_seen``, ``collective_backend``,
``allreduce_ms`` and ``allreduce_buckets``.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # BASELINE.json configs[1]: "HLG community, PPO batch 2048, SGNN 3 layers x256, 1xMI355X"
    'hlg_d256': dict(community='hlg', D=256, L=3, B=2048, T=16384, unique=1024, max_nodes=1000, max_edges=3000),
    # the reference's shipped YAML dims (hlg.yaml:21-44) -- the tiny-model regime, for orientation
    'hlg_ref': dict(community='hlg', D=16, L=2, B=256, T=8192, unique=1024, max_nodes=1000, max_edges=3000),
    # BASELINE.json configs[0]: "synthetic grid community, PPO batch 256, policy hidden 64 -- reference CPU path": the
    # reference YAML dims (grid.yaml:21-33), grid-sized graphs, land-use / road rows mixed (grid.yaml:16-18), pads 1000/3000
    'grid_ref': dict(community='grid', D=16, L=2, B=256, T=2560, unique=1024, max_nodes=1000, max_edges=3000, road_fraction=0.5),
    # BASELINE.json configs[3], per-GPU share: HLG graphs with the concept pads 1500 / 4000 (hlg_concept.yaml:27-28), 2048 rows per GPU
    'hlg_concept_d256': dict(community='hlg', D=256, L=3, B=2048, T=16384, unique=1024, max_nodes=1500, max_edges=4000),
    # BASELINE.json configs[2]
    'dhm_d256': dict(community='dhm', D=256, L=3, B=4096, T=32768, unique=1024, max_nodes=1000, max_edges=3000),
    # BASELINE.json configs[4], per-GPU share: heterogeneous HLG + DHM graphs in one minibatch (2048 rows per GPU)
    'mixed_d256': dict(community='mixed', D=256, L=3, B=2048, T=16384, unique=1024, max_nodes=1000, max_edges=3000),
}


def model_cfg(w):
    """The reference's three spec dicts (hlg.yaml:21-33) with the workload's GCN width / depth."""
    import types
    cfg = types.SimpleNamespace()
    cfg.state_encoder_specs = dict(state_encoder_hidden_size=[64, 16], gcn_node_dim=w['D'], num_gcn_layers=w['L'],
                                   num_edge_fc_layers=1, max_num_nodes=w['max_nodes'], max_num_edges=w['max_edges'],
                                   num_attention_heads=1)
    cfg.policy_specs = dict(policy_land_use_head_hidden_size=[32, 1], policy_road_head_hidden_size=[32, 1])
    cfg.value_specs = dict(value_head_hidden_size=[32, 32, 1])
    cfg.agent_specs = {}
    return cfg


def build_networks(cfg, seed=0):
    import types
    from drl_urban_planning_amd import create_sgnn_model, ActorCritic
    torch.manual_seed(seed)
    agent = types.SimpleNamespace(node_dim=23, numerical_feature_size=52, dtype=torch.float32)
    policy_net, value_net = create_sgnn_model(cfg, agent)
    return policy_net, value_net, ActorCritic(policy_net, value_net)


def algorithmic_flops_per_sample(n, e, D, L, F=23, Fn=52, S=(64, 16), H=32):
    """forward GEMM FLOPs (factorised forms), x3 for forward + backward."""
    num_enc = 2 * (Fn * S[0] + S[0] * S[1])
    node_enc = 2 * (n + 1) * F * D
    gcn = L * 2 * n * D * 2 * D
    attn = 4 * n * D + 8 * D * D
    head = e * (2 * D * H + 2 * H) + 6 * D * H
    value = 2 * ((3 * D + 19) * 32 + 32 * 32 + 32)
    return 3.0 * (num_enc + node_enc + gcn + attn + head + value)


def _cpu_row(P, orc, synth, w, Bc, threads, tight, max_steps, budget_s):
    """One timed row of the CPU baseline: `threads` torch threads, Bc rows per optimizer step."""
    torch.set_num_threads(threads)
    n_c = synth.COMMUNITY_NODES.get(w['community'], 397)          # ('mixed' -> the larger community's size)
    pads = (n_c + 1 if w['community'] != 'mixed' else 398, int(round(5.55 * (n_c if w['community'] != 'mixed' else 397))) + 8) \
        if tight else (w['max_nodes'], w['max_edges'])
    replay = synth.make_replay(Bc, w['community'], max_nodes=pads[0], max_edges=pads[1], seed=77,
                               road_fraction=w.get('road_fraction', 0.0))
    up = orc.OracleUpdater(P, mini_batch_size=Bc, num_optim_epoch=1)
    actions = torch.from_numpy(replay.actions).float()
    g = torch.Generator().manual_seed(1)
    adv, ret = torch.randn(Bc, 1, generator=g), torch.randn(Bc, 1, generator=g)
    with torch.no_grad():
        old, _ = orc.get_log_prob_entropy(P, orc.tensorfy(replay.states), actions)
    exps = torch.ones(Bc)
    t0 = time.time()
    up.step(replay.states, actions, adv, ret, old, exps)        # warm-up (and the one clipped step)
    warm = time.time() - t0
    steps = int(max(2, min(max_steps, budget_s / max(warm, 1e-3))))
    t0 = time.time()
    for _ in range(steps):
        up.step(replay.states, actions, adv, ret, old, exps)
    dt = time.time() - t0
    return dict(threads=threads, rows_per_step=Bc, steps=steps, pads=list(pads), tight_pad=bool(tight),
                samples_per_s=Bc * steps / dt, ms_per_step=1e3 * dt / steps)


def cpu_baseline(w, mode='quick'):
    """The oracle (PyTorch-CPU port of the reference's update step on padded dense batches, exactly the tensors the
    reference builds) timed on this host's cores, on a bounded sample of the same workload.

    B = 2048 does not fit a host as one autograd graph (0.24 GB per padded row at D = 256), so the CPU runs smaller
    optimizer steps; small steps are also its fastest per sample (8 rows: ~2x the rate of 32 rows on the build
    container), so both sizes are timed and the better one is reported.  Lines: 1 thread (the reference's documented OMP_NUM_THREADS=1, README.md:22-25), a thread sweep with the
    config's real pads (what the reference executes), and the same graphs padded tightly (the generous baseline; the
    results are identical).  ``value`` is the BEST padded line -- oversubscribing every core of a 128-thread host is 5-10x
    slower than 16-32 threads on these small ops, so "all cores" alone would be a strawman.  mode: 'quick' (default
    bench run: ~1 minute) | 'full' (>= 5 steps per line, every thread count) | 'off'."""
    from drl_urban_planning_amd import synth
    from oracle import sgnn_oracle as orc
    cfg = model_cfg(w)
    _, _, ac = build_networks(cfg, seed=0)
    P = orc.leaf_params(orc.split_actor_critic_state_dict(ac.state_dict()))
    ncpu = os.cpu_count() or 1
    wide = w['D'] >= 128
    full = mode == 'full'
    B_std = 32 if wide else 256
    sweep = sorted({t for t in (8, 16, 32, ncpu) if t <= ncpu} or {ncpu})
    if not full:                                    # quick: 16 and 32 threads only (where the optimum sits), fewer steps
        sweep = sorted({min(16, ncpu), min(32, ncpu)})
    B_small = 8 if wide else 64
    if mode == 'brief':     # the ref_dims child runs: the line the quick sweep has picked on every box so far (16 threads, small steps)
        rows = [_cpu_row(P, orc, synth, w, B_small, min(16, ncpu), False, 5, 5.0)]
    else:
        rows = [_cpu_row(P, orc, synth, w, B_small, 1, False, 5 if full else 2, 12.0)]
        for t in sweep:     # small steps: the fastest per sample on a CPU (the working set of a bigger batch falls out of cache)
            rows.append(_cpu_row(P, orc, synth, w, B_small, t, False, 5, 10.0 if full else 6.0))
        best_t = max(rows[1:], key=lambda r: r['samples_per_s'])['threads']
        rows.append(_cpu_row(P, orc, synth, w, B_std, best_t, False, 5, 15.0 if full else 6.0))      # and a 32-row step
    padded = [r for r in rows if r['threads'] > 1] or rows
    best = max(padded, key=lambda r: r['samples_per_s'])
    tight = _cpu_row(P, orc, synth, w, best['rows_per_step'], best['threads'], True, 5, 15.0 if full else (4.0 if mode == 'brief' else 6.0))
    rows.append(tight)
    torch.set_num_threads(min(ncpu, 32))
    return dict(value=best['samples_per_s'], unit='samples/s', cores=best['threads'], kind='port',
                sample='%d optimizer steps of %d rows, pads %d/%d, D=%d, L=%d, oracle/sgnn_oracle.py (port of the reference '
                       'path: /root/reference is absent on the GPU box), best of the (threads, rows) lines %s'
                       % (best['steps'], best['rows_per_step'], w['max_nodes'], w['max_edges'], w['D'], w['L'],
                          [(r['threads'], r['rows_per_step']) for r in padded]),
                ms_per_step=best['ms_per_step'], host_cpus=ncpu, one_thread=rows[0]['samples_per_s'] if rows[0]['threads'] == 1 else None,
                tight_pad=tight['samples_per_s'], rows=rows)


def pmc_figure(fname, args, pick):
    """(value, reason): a figure of a committed PMC summary (profiles/<fname>; the counters need their own rocprofv3 passes,
    --pmc with --kernel-trace only, so bench.py cannot measure them in its own run).  The summaries are stamped with the hash
    of the native sources they were collected on: a figure is only reported for the default workload AND when that hash is
    the one of the sources this run was built from -- otherwise null with the reason."""
    if args.workload != 'hlg_d256' or args.minibatch:
        return None, 'PMC passes are only collected for the default workload (hlg_d256, full minibatch)'
    root = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(root, 'profiles', fname)
    if not os.path.exists(path):
        return None, 'profiles/%s is absent' % fname
    with open(path) as fh:
        doc = json.load(fh)
    sys.path.insert(0, os.path.join(root, 'tools'))
    from csrc_hash import csrc_hash
    have, want = doc.get('csrc_hash'), csrc_hash(root)
    if have != want:
        return None, 'profiles/%s was collected on csrc %s, this build is %s: re-run the PMC passes (tools/pmc_*.py)' % (fname, have, want)
    v = pick(doc)
    return v, (None if v is not None else 'kernel not in profiles/%s' % fname)


def spawn_ranks(args, argv):
    """``python bench.py --gpus N`` with no launcher around it: re-exec this script under ``torch.distributed.run`` (one rank
    per GPU, rendezvous on 127.0.0.1, a free port) -- rank 0 of that run prints the JSON line on the inherited stdout.  With
    fewer visible GPUs than ranks the ranks share devices over gloo (UPAMD_DIST_BACKEND=gloo: a functional check of the rank
    logic on a one-GPU box, labelled as such in the line -- RCCL refuses two ranks on one device)."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus and 'UPAMD_DIST_BACKEND' not in env:
        sys.stderr.write('bench.py: %d ranks on %d visible GPU(s): ranks share devices over gloo (functional check, not a '
                         'scaling number)\n' % (args.gpus, ndev))
        env['UPAMD_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def route_overhead(rows_share, timeout_s=150):
    """What the data-parallel ROUTE adds to one rank's share of a strong-scaling step when there is nothing on the wire: the 256-row
    step timed by three child runs of this script -- no process group | a one-rank RCCL group with the single collective | the same
    with the bucketed all-reduce forced on (UPAMD_DIST_FORCE_INIT=1: `nccl` initialised with ONE rank, so the stream / event / work-object
    plumbing of the real backend runs and the sum is the identity).  Returns ms per step of each and the two differences."""
    import socket
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--minibatch', str(rows_share), '--steps', '96', '--warmup', '32',
            '--cpu-baseline', 'off', '--no-kernel-events', '--inclusive-pool', '--no-ref-dims', '--strong-proxy', 'off']
    out = {}
    for tag, extra in (('no_group', None), ('single_collective', {'UPAMD_GRAD_BUCKETS': '0'}), ('bucketed', {'UPAMD_GRAD_BUCKETS': 'force'})):
        env = dict(os.environ)
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'UPAMD_DIST_FORCE_INIT', 'UPAMD_GRAD_BUCKETS'):
            env.pop(k, None)
        if extra is not None:
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
            env.update(extra, UPAMD_DIST_FORCE_INIT='1', RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        try:
            res = subprocess.run(base, capture_output=True, text=True, timeout=timeout_s, env=env)
            line = [l for l in res.stdout.splitlines() if l.startswith('{')]
            if res.returncode != 0 or not line:
                return {'error': '%s: rc=%d %s' % (tag, res.returncode, res.stderr[-300:])}
            child = json.loads(line[-1])
            out['ms_' + tag] = child['ms_per_step']
            out['host_enqueue_ms_' + tag] = child.get('host_enqueue_ms_per_step')      # the launch thread's share of the step
        except Exception as exc:
            return {'error': '%s: %s: %s' % (tag, type(exc).__name__, exc)}
    out['single_collective'] = out['ms_single_collective'] - out['ms_no_group']
    out['bucketed'] = out['ms_bucketed'] - out['ms_no_group']
    out['source'] = ('MEASURED by three child runs of this script on this box (%d-row step, 96 steps after 32 warm-up): no process group | one-rank '
                     'RCCL group, single collective | bucketed all-reduce forced on; nothing is on the wire (one rank)' % rows_share)
    return out


REF_DIMS_KEYS = ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'node_steps_per_s', 'host_enqueue_ms_per_step', 'kernel_ms_per_step')


def ref_dims_line(workload, steps=400, warmup=256, timeout_s=240):
    """One of the reference-YAML-dims workloads (hlg_ref: hlg.yaml:21-44; grid_ref = BASELINE configs[0]: grid.yaml:16-33) measured by a
    child run of this script (its own process: another model, replay and engine) and reduced to the fields SURVEY 8(d) asks for
    next to every configuration: samples/s, ms/step, the fused kernel's roofline fraction, the CPU port's row, the inclusive fractions."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--workload', workload, '--steps', str(steps), '--warmup', str(warmup),
           '--cpu-baseline', 'brief', '--strong-proxy', 'off', '--no-ref-dims']
    t0 = time.time()
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in res.stdout.splitlines() if l.startswith('{')]
        if res.returncode != 0 or not line:
            return {'error': 'rc=%d: %s' % (res.returncode, res.stderr[-400:])}
        d = json.loads(line[-1])
    except Exception as exc:                       # a failed side measurement must not cost the headline its line
        return {'error': '%s: %s' % (type(exc).__name__, exc)}
    out = {k: d.get(k) for k in REF_DIMS_KEYS}
    out['workload'] = d['config']['workload']
    rl = d.get('roofline') or {}
    out['roofline'] = {k: rl.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms', 'traffic')}
    cb = d.get('cpu_baseline') or {}
    out['cpu_baseline'] = {k: cb.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample', 'ms_per_step', 'tight_pad')}
    for k in ('update_params_inclusive', 'update_params_inclusive_records'):
        if d.get(k):
            out[k] = {q: d[k].get(q) for q in ('samples_per_s', 'seconds', 'fraction_of_step_rate', 'prepare_s', 'loop_s')}
    out['wall_s'] = time.time() - t0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--workload', default='hlg_d256', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline', default='quick', choices=['quick', 'full', 'brief', 'off'],
                    help='quick: ~1 min of CPU work (default); full: the whole thread sweep, >= 5 steps per line; brief: one '
                         '16-thread line + its tight-pad twin (~15 s; what the ref_dims child runs use)')
    ap.add_argument('--no-route-overhead', action='store_true',
                    help='strong_proxy: use round 5\'s lab constants for the data-parallel route\'s own cost instead of measuring it with child runs')
    ap.add_argument('--no-ref-dims', action='store_true',
                    help='skip the ref_dims object (default run at N = 1: hlg_ref and grid_ref measured by child runs)')
    ap.add_argument('--dp-mode', default='global', choices=['global', 'local'],
                    help='global (default): every rank holds the replay, one global permutation, rank slices of each '
                         'global minibatch; local: per-rank shards shuffled locally')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--minibatch', type=int, default=0, help='override the per-GPU PPO minibatch of the workload (exploration)')
    ap.add_argument('--inclusive-unique', action='store_true', default=True,
                    help='(default) time the update_params_inclusive leg on T DISTINCT host states: the host packer reads the '
                         'full ~150 KB x T working set, as it does behind real rollouts')
    ap.add_argument('--inclusive-pool', dest='inclusive_unique', action='store_false',
                    help='update_params_inclusive on the timed region\'s own replay (a tiled pool of `unique` states: the '
                         'packer re-reads a small host working set -- an upper bound)')
    ap.add_argument('--strong-proxy', default='auto', choices=['auto', 'on', 'off'],
                    help='N = 1: also time the step on 1/8 of the minibatch (the per-GPU share of an 8-GPU strong-scaling run) and '
                         'report strong_proxy = ms(full) / (ms(share) + exposed all-reduce); auto = for the default workload only')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak (default): the minibatch per GPU is fixed; strong: the GLOBAL minibatch is fixed and split')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args, sys.argv[1:])             # does not return: this process becomes the launcher
    # The learner process of the reference runs with OMP_NUM_THREADS=1 (khrylib/rl/agents/agent.py:12).  It matters here: torch's
    # default is one OpenMP thread per core (128 on the GPU boxes), whose workers spin after every parallel region (a 64 KB host copy
    # is enough) -- inside a container with a CPU quota (16 cores on this pool) that burns the quota and the whole process is
    # throttled for the rest of the 100 ms period: 60-90 ms host stalls at random places, 1.1 instead of 0.4 ms per step at the
    # reference dims (profiles/archive/r04_lab_host_stalls.log).  cpu_baseline() sets its own thread counts for its sweep.
    torch.set_num_threads(int(os.environ.get('UPAMD_BENCH_THREADS', '1')))      # (the variable: lab A/B only)

    # stdout carries ONE JSON line and nothing else: libraries that write to file descriptor 1 behind Python's back (RCCL prints
    # its WARN / version lines there, from its own threads, and they were found spliced INTO the JSON line) are sent to stderr
    # for the rest of the process; the result goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from drl_urban_planning_amd import PPOUpdater, synth, DistContext

    w = dict(WORKLOADS[args.workload])
    if args.minibatch:
        w['B'] = args.minibatch
    if args.scaling == 'strong':
        if w['B'] % args.gpus:
            raise RuntimeError('--scaling strong needs the minibatch (%d) divisible by --gpus' % w['B'])
        w['B'] //= args.gpus
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # one process per GPU; UPAMD_DIST_BACKEND=gloo + fewer GPUs than ranks is a debugging aid only (lets the
    # multi-process path be exercised on a 1-GPU box), the real launch is one rank per GPU over RCCL
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    ctx = DistContext.from_env(backend=os.environ.get('UPAMD_DIST_BACKEND'), device=dev)
    if ctx.world != args.gpus:
        raise RuntimeError('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, ctx.world))
    rank = ctx.rank

    cfg = model_cfg(w)
    policy_net, value_net, ac = build_networks(cfg, seed=0)              # identical weights on every rank
    ac.to(dev)
    glob = args.dp_mode == 'global'
    B_step = w['B'] * ctx.world                                          # rows one optimizer step consumes over all ranks
    up = PPOUpdater(policy_net, value_net, lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=1.0, tau=0.0, clip_epsilon=0.2,
                    value_pred_coef=0.5, entropy_coef=0.01, num_optim_epoch=4,
                    mini_batch_size=B_step if glob else w['B'], dist_ctx=ctx,
                    dp_mode=args.dp_mode)
    need = args.steps + args.warmup
    if glob:
        # the SAME replay on every rank (same seed), one global permutation (same numpy seed), every rank B rows of each
        # global minibatch -- the reference's minibatch sequence, sharded
        # (two global minibatches per epoch are enough for any step count -- run() starts a new epoch whenever one is used up;
        # sizing T by the step count made every rank of an 8-GPU run generate and pack ~98 k states before the timed region)
        T = max(w['T'], 2 * B_step)
        seed_replay, seed_np = 100, 7
    else:
        T = max(w['T'], 2 * w['B'])
        seed_replay, seed_np = 100 + rank, 7 + rank
    t_gen = time.time()
    replay = synth.make_replay(T, w['community'], max_nodes=w['max_nodes'], max_edges=w['max_edges'],
                               seed=seed_replay, unique=w['unique'], road_fraction=w.get('road_fraction', 0.0))
    t_gen = time.time() - t_gen
    np.random.seed(seed_np)
    # the replay is ~10^5 small python objects: a full garbage collection that happens to fall into the timed region costs
    # 50-70 ms (seen as 2 ms 'steps' at the reference dims, where a step is 0.3 ms) -- move them out of the collector's way
    import gc
    gc.collect()
    gc.freeze()
    engine = up.attach()
    # kernel-lab A/B switches (default: the library's own choices): UPAMD_TUNE="knob=value,knob=value"
    tune = dict(kv.split('=') for kv in os.environ.get('UPAMD_TUNE', '').split(',') if kv)
    if os.environ.get('UPAMD_GEMM_NT_DMA') is not None:
        tune['gemm_nt_dma'] = os.environ['UPAMD_GEMM_NT_DMA']
    if tune:
        from drl_urban_planning_amd import native
        for knob, value in tune.items():
            native.tune(knob, int(value))
    t_prep = time.time()
    it = up.prepare(replay)
    torch.cuda.synchronize(dev)
    t_prep = time.time() - t_prep

    meta = it.packed.meta
    nodes_per_sample = float(meta[:, 0].mean())
    edges_per_sample = float(meta[:, 1].mean())

    def run(nsteps, timed):
        done = 0
        while done < nsteps:
            ep = up.make_epoch(it)
            for k in range(ep.nb):
                if done >= nsteps:
                    break
                up.step(it, ep, k)
                done += 1

    run(args.warmup, False)
    torch.cuda.synchronize(dev)
    if not args.no_kernel_events:
        engine.profile_reset()
        engine.profile(True)
    up.collective_events = [] if ctx.world > 1 else None
    ctx.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(args.steps, True)
    t_enqueued = time.perf_counter() - t0            # the host has handed every launch of the timed steps to the runtime
    torch.cuda.synchronize(dev)
    ctx.barrier()
    dt = time.perf_counter() - t0
    engine.profile(False)
    # the step's ONE collective (gradient all-reduce, P floats), HIP events on the stream the backward runs on: from the end of
    # this rank's backward to the end of the collective, i.e. wire time + the wait for the slowest rank; max over ranks of the mean
    coll = up.collective_ms() if ctx.world > 1 else []
    up.collective_events = None
    cmean = torch.tensor([sum(coll) / max(len(coll), 1)], dtype=torch.float64, device=dev)
    ctx.all_reduce_max(cmean)
    grad_bytes = int(up.grads.numel()) * 4
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    if ctx.world > 1:
        ctx.all_reduce_sum(ones)                       # every rank that takes part in the collectives adds 1
    ranks_seen, backend = int(round(float(ones.item()))), ctx.backend
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    ctx.all_reduce_max(tmax)
    dt = float(tmax.item())

    # ---- strong-scaling proxy on one GPU (north-star: >= 6x at 8 GPUs): the same resident replay stepped at 1/8 of the
    # minibatch = what each of 8 ranks computes per optimizer step of a strong-scaling run (dp 'global': B / world rows per rank)
    proxy = None
    want_proxy = args.strong_proxy == 'on' or (args.strong_proxy == 'auto' and args.workload == 'hlg_d256' and not args.minibatch)
    if ctx.world == 1 and want_proxy and w['B'] % 8 == 0:
        engine.profile(False)
        full_B = up.mini_batch_size
        n_full = max(4, min(args.steps, 12))            # both legs WITHOUT the per-kernel HIP events of the timed region above
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        run(n_full, True)
        torch.cuda.synchronize(dev)
        ms_full = 1e3 * (time.perf_counter() - t1) / n_full
        up.mini_batch_size = w['B'] // 8
        run(24, False)
        torch.cuda.synchronize(dev)
        n_share = 96
        t1 = time.perf_counter()
        run(n_share, True)
        torch.cuda.synchronize(dev)
        ms_share = 1e3 * (time.perf_counter() - t1) / n_share
        bk = engine.grad_buckets()                      # the ranges the last backward finalised, in readiness order
        up.mini_batch_size = full_B
        nfl = engine.n_floats
        tail_bytes = 4 * (bk[-1][1] - bk[-1][0]) if len(bk) > 1 else 4 * (nfl + 4)
        # the all-reduce itself cannot be measured on one GPU.  Model of an 8-rank ring over xGMI, stated so that it can be
        # replaced by a measurement: 2 (N - 1) hops of ALPHA us + 2 (N - 1) / N of the bytes over one link direction at LINK GB/s
        ALPHA_US, LINK_GBPS, NR = 2.5, 50.0, 8
        model = lambda nbytes: 1e-3 * (2 * (NR - 1) * ALPHA_US + 2.0 * (NR - 1) / NR * nbytes / (LINK_GBPS * 1e3))
        exposed = model(tail_bytes)
        proxy = {'ms_full': ms_full, 'steps_full': n_full, 'rows_full': w['B'], 'ms_share': ms_share, 'rows_share': w['B'] // 8, 'steps_share': n_share,
                 'grad_buckets': [[int(b), int(e)] for b, e in bk], 'exposed_allreduce_bytes': int(tail_bytes),
                 'exposed_allreduce_ms': exposed, 'single_collective_ms': model(4 * (nfl + 4)),
                 'allreduce_source': 'MODEL, not measured (one GPU): 8-rank ring, %.1f us per hop, %.0f GB/s per link direction; '
                                     'bucketed form = only the range that is final with the last launch of the backward is exposed'
                                     % (ALPHA_US, LINK_GBPS),
                 'value': ms_full / (ms_share + exposed), 'value_without_collective': ms_full / ms_share,
                 'value_single_collective': ms_full / (ms_share + model(4 * (nfl + 4))), 'ideal': 8.0,
                 }
        # what the data-parallel route itself adds to a 256-row step under a real RCCL process group with ONE rank (nothing on the
        # wire): measured by child runs (route_overhead); round 5's lab constants only if a child run fails, and labelled so
        ro = route_overhead(w['B'] // 8) if not args.no_route_overhead else {'error': '--no-route-overhead'}
        if 'error' in ro:
            ro = {'single_collective': 0.035, 'bucketed': 0.115, 'fallback_reason': ro['error'],
                  'source': 'CONSTANTS of profiles/r05_lab_bucket_overhead.md (one-rank RCCL group, round 5): the child runs did not complete'}
        proxy['rccl_route_overhead_ms'] = ro
        proxy['value_with_route_overhead'] = ms_full / (ms_share + max(ro['bucketed'], 0.0) + exposed)
        proxy['value_single_collective_with_route_overhead'] = ms_full / (ms_share + max(ro['single_collective'], 0.0) + model(4 * (nfl + 4)))

    kern = {}
    if not args.no_kernel_events:
        for name in ('gemm_nt_128', 'gemm_nt_128_k32', 'gemm_nt_64', 'gemm_nt_32', 'gemm_nt_128_rm', 'gemm_nt_64_rm',
                     'gemm_nt_32_rm', 'gemm_nt_generic', 'gemm_tn_128', 'gemm_tn_64', 'gemm_tn_32', 'gemm_tn_generic', 'gemm_tn_small',
                     'edge_fwd', 'edge_bwd', 'tiny_fwd', 'tiny_bwd', 'tiny_step'):
            st = engine.profile_read(name)
            if st['launches']:
                kern[name] = st
    up.detach()
    # one whole update_params call as the reference's caller sees it (urban_planning_agent.py:248-271): host packing,
    # the single upload, value / old-log-prob pre-pass, GAE, every optimizer step of every epoch, write-back
    ctx.barrier()
    replay_incl, unique_incl = replay, min(w['unique'], T)
    if args.inclusive_unique and ctx.world == 1:       # T distinct host states: the packer's host working set is the full ~150 KB x T
        # (one rank only: N ranks of one host would each generate and hold T x 150 KB -- 39 GB at 8 ranks -- for a leg that is a
        # single-GPU figure; multi-rank lines time it on the tiled pool)
        replay_incl = synth.make_replay(T, w['community'], max_nodes=w['max_nodes'], max_edges=w['max_edges'],
                                        seed=seed_replay + 50, unique=None, road_fraction=w.get('road_fraction', 0.0))
        unique_incl = T
    host_bytes = unique_incl * sum(int(np.asarray(f).nbytes) for f in replay_incl.states[0])
    np.random.seed(seed_np + 1000)
    t_incl = time.perf_counter()
    up.update_params(replay_incl, 0)
    torch.cuda.synchronize(dev)
    t_incl = time.perf_counter() - t_incl
    incl = dict(up.last_timing)
    # the same call fed with COMPACT RECORDS in one contiguous arena -- what UPAMD_ROLLOUT=server hands update_params (the workers
    # write the records while they sample; rollout.RecordBatch) instead of the reference's padded tuples
    incl_rec = None
    if ctx.world == 1:
        from drl_urban_planning_amd import packer as _packer
        recs = [_packer.compact_state(s) for s in replay_incl.states[:unique_incl]]
        sizes = np.array([r.size for r in recs], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum((sizes + 63) // 64 * 64)])
        arena = np.zeros(int(offs[-1]), dtype=np.uint8)
        for r, o in zip(recs, offs[:-1]):
            arena[int(o):int(o) + r.size] = r
        pick = np.arange(T) % len(recs)
        views = [arena[int(offs[i]):int(offs[i]) + int(sizes[i])] for i in pick]
        rec_states = _packer.RecordList(views, np.uint64(arena.ctypes.data) + offs[:-1][pick].astype(np.uint64), sizes[pick])
        rec_batch = synth.Replay(rec_states, replay_incl.actions, replay_incl.masks, replay_incl.rewards, replay_incl.exps)
        del recs
        np.random.seed(seed_np + 1000)
        torch.cuda.synchronize(dev)
        t_rec = time.perf_counter()
        up.update_params(rec_batch, 0)
        torch.cuda.synchronize(dev)
        t_rec = time.perf_counter() - t_rec
        incl_rec = dict(up.last_timing, seconds=t_rec, record_bytes=int(offs[-1]))
    ctx.barrier()
    ctx.close()

    if rank != 0:
        return
    samples = w['B'] * ctx.world * args.steps
    value = samples / dt
    out = {
        'metric': 'PPO-update samples/sec', 'value': value, 'unit': 'samples/s', 'n_gpus': ctx.world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'node_steps_per_s': value * nodes_per_sample,      # graph-nodes x steps / s: what makes HLG / DHM / mixed lines comparable
        'config': {'workload': '%s: %s-shaped graphs (n~%.0f nodes, e~%.0f edges live; pads %d/%d), SGNN %d layers x %d, '
                               'PPO minibatch %d per GPU, replay %d states resident in HBM'
                               % (args.workload, w['community'].upper(), nodes_per_sample, edges_per_sample, w['max_nodes'],
                                  w['max_edges'], w['L'], w['D'], w['B'], T),
                   'global_batch': w['B'] * ctx.world, 'parallelism': 'dp%d' % ctx.world,
                   'node_steps_per_s': value * nodes_per_sample},
        'setup_s': {'generate': t_gen, 'pack_upload_prepass_gae': t_prep},
        'host_enqueue_ms_per_step': 1e3 * t_enqueued / args.steps,      # (close to ms_per_step = the host is the bound, not the GPU)
        'update_params_inclusive': {'samples_per_s': incl['steps'] * incl['rows_per_step'] / t_incl, 'seconds': t_incl,
                                    'optimizer_steps': incl['steps'], 'rows_per_step': incl['rows_per_step'],
                                    'replay_states': T, 'unique_host_states': unique_incl,
                                    'host_working_set_bytes': host_bytes, 'prepare_s': incl['prepare'], 'loop_s': incl['loop'],
                                    'fraction_of_step_rate': (incl['steps'] * incl['rows_per_step'] / t_incl) / value,
                                    'prepare_pipeline_chunks': int(up.pipeline_chunks) or (8 if w['D'] > 32 else 2),
                                    'note': 'one update_params(batch) call from host numpy states: pack | H2D | pre-pass (pipelined '
                                            'over chunks of the replay) + GAE + all epochs + write-back; T distinct host states unless '
                                            '--inclusive-pool (then the packer re-reads a small working set: an upper bound)'},
        'dp_mode': incl.get('dp_mode'),
    }
    if incl_rec is not None:
        out['update_params_inclusive_records'] = {
            'samples_per_s': incl_rec['steps'] * incl_rec['rows_per_step'] / incl_rec['seconds'], 'seconds': incl_rec['seconds'],
            'fraction_of_step_rate': (incl_rec['steps'] * incl_rec['rows_per_step'] / incl_rec['seconds']) / value,
            'prepare_s': incl_rec['prepare'], 'loop_s': incl_rec['loop'], 'host_record_bytes': incl_rec['record_bytes'],
            'note': 'the same call on compact wire records in one arena (rollout.RecordBatch: what UPAMD_ROLLOUT=server feeds it); '
                    'the host reads ~2.7x fewer bytes than from the padded tuples'}
    if proxy is not None:
        out['strong_proxy'] = proxy
    if ctx.world > 1:
        # how long the backward's stream WAITS for the step's gradient all-reduce (HIP events on that stream): all of a single
        # collective, the exposed tail of the bucketed form (UPAMD_GRAD_BUCKETS=0 switches the buckets off)
        out['rccl_ranks_seen'] = ranks_seen      # the sum of an all-reduce of ones over the gradient's process group
        out['collective_backend'] = backend + ('' if backend == 'nccl' else ' (ranks share a GPU through host memory: a functional check of '
                                                                            'the rank logic, not a scaling number)')
        out['visible_gpus'] = torch.cuda.device_count()
        out['allreduce_buckets'] = [[int(b), int(e)] for b, e in (up.last_buckets or [])]
        out['allreduce_ms'] = float(cmean.item())
        out['allreduce_share_of_step'] = float(cmean.item()) / out['ms_per_step']
        out['allreduce_bytes'] = grad_bytes
    flops_sample = algorithmic_flops_per_sample(nodes_per_sample, edges_per_sample, w['D'], w['L'])
    out['algorithmic'] = {'flops_per_sample_step': flops_sample, 'tflops': value / ctx.world * flops_sample / 1e12,
                          'frac_of_fp32_mfma_peak': value / ctx.world * flops_sample / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    mfma_kernels = {k: v for k, v in kern.items() if v['flops'] > 0}
    if mfma_kernels:
        # FLOPs the GEMM kernels actually executed (layer 1 runs at K = 32 on the raw features, weight-gradient and
        # dgrad GEMMs counted as launched) next to the SURVEY section 8d algorithmic count above
        ex = sum(v['flops'] for v in mfma_kernels.values()) / args.steps
        out['executed'] = {'gemm_flops_per_step': ex, 'tflops': ex / (1e-3 * out['ms_per_step']) / 1e12,
                           'frac_of_fp32_mfma_peak': ex / (1e-3 * out['ms_per_step']) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                           'gemm_ms_per_step': sum(v['total_ms'] for v in mfma_kernels.values()) / args.steps}
    dom = 'gemm_nt_128' if 'gemm_nt_128' in kern else (sorted(mfma_kernels, key=lambda k: -mfma_kernels[k]['total_ms'])[0]
                                                       if mfma_kernels else None)
    if 'tiny_step' in kern:
        # the fused small-model step (csrc/tiny.hip): one launch = forward + loss + backward of the whole minibatch, one workgroup
        # per graph.  SURVEY section 8d prices this regime against HBM: compulsory bytes per sample-step = the packed inputs plus
        # the saved node embeddings (written by the forward, read by the backward), plus 16 P per step of parameter / gradient /
        # Adam traffic.  The kernel keeps the embeddings in LDS, so it moves LESS than this figure; the fraction says how far a
        # latency-bound per-graph program is from the bandwidth bound, which is the honest reading of "speed of light" here.
        st = kern['tiny_step']
        n_, e_, D, L, F, Fn = nodes_per_sample, edges_per_sample, w['D'], w['L'], 23, 52
        per_sample = n_ * F * 4 + e_ * 2 * 4 + 2 * n_ + 2 * e_ + (F + Fn + 3) * 4 + 20 + 2 * (L + 1) * n_ * D * 4
        nparam = sum(p.numel() for p in ac.parameters())
        by = w['B'] * per_sample + 16 * nparam
        ach = by / (st['total_ms'] / st['launches'] * 1e-3) / 1e9
        out['roofline'] = {'kernel': 'tiny_step', 'bound': 'hbm', 'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0,
                           'traffic': None, 'traffic_reason': 'no PMC pass over this workload', 'launches': st['launches'],
                           'avg_launch_ms': st['total_ms'] / st['launches'], 'algorithmic_bytes_per_launch': by,
                           'algorithmic_bytes_per_sample': per_sample,
                           'note': 'latency-bound: one workgroup per graph, %d workgroups on 256 CUs' % min(w['B'], 256)}
        out['kernel_ms_per_step'] = {k: v['total_ms'] / args.steps for k, v in kern.items()}
    elif dom is not None:
        st = kern[dom]
        ach = st['flops'] / (st['total_ms'] * 1e-3) / 1e12
        out['roofline'] = {'kernel': dom, 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': ach / PEAK_FP32_MFMA_TFLOPS, 'traffic': None, 'launches': st['launches'],
                           'avg_launch_ms': st['total_ms'] / st['launches'],
                           'algorithmic_flops_per_launch': st['flops'] / st['launches'],
                           'algorithmic_bytes_per_launch': st['bytes'] / st['launches']}
        # HBM bytes per launch from the PMC counters: they need their own rocprofv3 passes (FETCH_SIZE / WRITE_SIZE,
        # --kernel-trace only), so the figure is read from the committed summary of those passes over this same
        # command (tools/pmc_traffic.py -> profiles/pmc_traffic.json); null when that file is absent
        out['roofline']['traffic'], out['roofline']['traffic_reason'] = pmc_figure(
            'pmc_traffic.json', args, lambda doc: (doc.get('kernels', {}).get(dom) or {}).get('hbm_bytes_per_launch'))
        if out['roofline']['traffic'] is not None:
            out['roofline']['traffic_source'] = ('profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, '
                                                 'separate passes over bench.py, stamped with the hash of csrc/ it was taken on')
        out['kernel_ms_per_step'] = {k: v['total_ms'] / args.steps for k, v in kern.items()}
    if 'edge_fwd' in kern and 'edge_bwd' in kern:
        # the message-passing group (6 launches per step at L = 3): second-largest share of the step, no MFMA work to speak of.
        # Algorithmic HBM bytes = the slice copies it cannot avoid (per layer: P/Q + H read, H written; backward: P/Q + G read,
        # dP|dQ written; the folded first layer reads the 128-byte raw feature row instead of P/Q + H; the last layer writes /
        # reads the candidates' head inputs), measured time from HIP events; the vector-ALU share comes from the committed PMC
        # pass over this same command (tools/pmc_util.py -> profiles/pmc_util.json), null when absent
        Mn = w['B'] * nodes_per_sample
        nhe = float(meta[:, 2].mean()) * w['B']
        D, L = w['D'], w['L']
        fwd = (L - 1) * Mn * (2 * D + D + D) * 4 + Mn * (128 + D * 4) + nhe * D * 4
        bwd = (L - 1) * Mn * (2 * D + D + 2 * D) * 4 + Mn * (128 + D * 4 + 2 * D * 4) + nhe * D * 4
        ms = (kern['edge_fwd']['total_ms'] + kern['edge_bwd']['total_ms']) / args.steps
        mp = {'kernels': 'edge_fwd + edge_bwd', 'ms_per_step': ms, 'share_of_step': ms / out['ms_per_step'],
              'algorithmic_bytes_per_step': fwd + bwd, 'achieved': (fwd + bwd) / (ms * 1e-3) / 1e9, 'unit': 'GB/s',
              'peak': 8000.0, 'frac_of_hbm_peak': (fwd + bwd) / (ms * 1e-3) / 8e12, 'bound': 'valu', 'valu_busy': None}
        mp['valu_busy'], mp['valu_busy_reason'] = pmc_figure(
            'pmc_util.json', args,
            lambda doc: {k: u['valu_busy'] for k, u in doc.get('kernels', {}).items() if k.startswith('edge_')} or None)
        if mp['valu_busy'] is not None:
            mp['valu_busy_source'] = 'profiles/pmc_util.json: rocprofv3 --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE, one pass over bench.py'
        out['message_passing'] = mp
    if ctx.world == 1 and not args.no_cpu_baseline and args.cpu_baseline != 'off':
        out['cpu_baseline'] = cpu_baseline(w, args.cpu_baseline)
    if ctx.world == 1 and args.workload == 'hlg_d256' and not args.minibatch and not args.no_ref_dims:
        # "for every cfg additionally report the reference-YAML dims" -- the two small-model workloads in the same line
        # (child runs, long warm-up: sub-millisecond steps need the clock ramp)
        out['ref_dims'] = {name: ref_dims_line(name) for name in ('hlg_ref', 'grid_ref')}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + '\n').encode())
    os.close(json_fd)


if __name__ == '__main__':
    main()
