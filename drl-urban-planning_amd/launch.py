"""``python -m drl_urban_planning_amd.launch [-m MODULE | SCRIPT] [args...]`` -- run the reference's own entry point
(``urban_planning.train``, ``urban_planning.eval``) on the HIP engine WITHOUT editing the reference tree, one process or
one process per GPU:

    python -m drl_urban_planning_amd.launch -m urban_planning.train --cfg hlg --global_seed 111
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        -m drl_urban_planning_amd.launch -m urban_planning.train --cfg hlg --global_seed 111

Order of business (it matters):

1. ``configure_rank`` -- BEFORE torch touches HIP: ``urban_planning/train.py:50,54`` pins every process to
   ``cuda:<--gpu_index>`` (default 0), so under ``torch.distributed.run`` all ranks would land on one GPU.  The shim
   narrows ``HIP_VISIBLE_DEVICES`` to the LOCAL_RANK-th visible device, which makes ``cuda:0`` of every process its own
   GPU; it refuses a launch with more local ranks than devices (unless ``UPAMD_DIST_BACKEND=gloo`` says the ranks are
   meant to share, as the tests do).  ``DistContext.from_env`` re-checks the outcome through the rendezvous store.
2. ``binding.patch_reference_module`` on the imported ``urban_planning.agents.urban_planning_agent`` -- the same two
   changes INTEGRATION.md shows as a diff -- before the entry point does its ``from ... import UrbanPlanningAgent``.
3. ``runpy`` of the entry point as ``__main__`` with the remaining command line.
"""
import os
import runpy
import sys

AGENT_MODULE = 'urban_planning.agents.urban_planning_agent'


def visible_devices(environ):
    """The physical device list this process may use, as strings, or None when no mask is set and the count is unknown
    without initialising HIP."""
    for key in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = environ.get(key)
        if v is not None and v.strip() != '':
            return [x.strip() for x in v.split(',') if x.strip() != '']
    # ROCR_VISIBLE_DEVICES (the ROCr-level mask schedulers and cgroup set-ups use) filters FIRST and the HIP-level indices are
    # relative to what it leaves: with N entries there, this process has HIP devices 0 .. N-1
    v = environ.get('ROCR_VISIBLE_DEVICES')
    if v is not None and v.strip() != '':
        return [str(i) for i, x in enumerate(v.split(',')) if x.strip() != '']
    try:                                            # KFD topology: GPU nodes have a non-zero simd_count
        root = '/sys/class/kfd/kfd/topology/nodes'
        n = 0
        for node in sorted(os.listdir(root)):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                n += 1
        return [str(i) for i in range(n)] if n else None
    except (OSError, ValueError):
        return None


def configure_rank(environ=None):
    """Map LOCAL_RANK to one visible device (module docstring, step 1).  Returns the physical device id this process
    was narrowed to, or None when there is nothing to do (single process)."""
    environ = os.environ if environ is None else environ
    if 'LOCAL_RANK' not in environ:
        return None
    local_rank = int(environ['LOCAL_RANK'])
    local_world = int(environ.get('LOCAL_WORLD_SIZE', environ.get('WORLD_SIZE', '1')))
    if local_world <= 1:
        return None
    torch = sys.modules.get('torch')
    if environ is os.environ and torch is not None and torch.cuda.is_initialized():
        raise RuntimeError('configure_rank() must run before HIP is initialised (HIP_VISIBLE_DEVICES is read once)')
    devs = visible_devices(environ)
    shared_ok = environ.get('UPAMD_DIST_BACKEND') == 'gloo'
    if devs is not None and local_world > len(devs):
        if shared_ok:
            chosen = devs[local_rank % len(devs)]
        else:
            raise RuntimeError('%d local ranks but %d visible GPU(s) %s: RCCL needs one device per rank '
                               '(UPAMD_DIST_BACKEND=gloo lets test ranks share a GPU)' % (local_world, len(devs), devs))
    else:
        chosen = devs[local_rank] if devs is not None else str(local_rank)
    environ['HIP_VISIBLE_DEVICES'] = chosen
    environ.pop('CUDA_VISIBLE_DEVICES', None)       # (both set and different is an error in the HIP runtime)
    environ['UPAMD_RANK_DEVICE'] = chosen
    return chosen


def patch_reference():
    import importlib
    from . import binding
    return binding.patch_reference_module(importlib.import_module(AGENT_MODULE))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 0
    configure_rank()
    patch_reference()
    if argv[0] == '-m':
        if len(argv) < 2:
            raise SystemExit('launch: -m needs a module name')
        sys.argv = [argv[1]] + argv[2:]
        runpy.run_module(argv[1], run_name='__main__', alter_sys=True)
    else:
        sys.argv = argv
        runpy.run_path(argv[0], run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
