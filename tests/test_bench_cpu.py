"""bench.py host logic that needs no GPU: ``python bench.py --gpus N`` starts its own ranks (no external launcher), the
child-run reduction behind the ``ref_dims`` object keeps the fields SURVEY.md section 8(d) asks for."""
import argparse
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_n_without_a_launcher_reexecs_under_torch_distributed_run(monkeypatch):
    seen = {}

    def fake_exec(exe, argv, env):
        seen.update(exe=exe, argv=list(argv), env=dict(env))
        raise SystemExit(0)

    monkeypatch.setattr(os, 'execvpe', fake_exec)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('UPAMD_DIST_BACKEND', raising=False)
    args = argparse.Namespace(gpus=2)
    with pytest.raises(SystemExit):
        bench.spawn_ranks(args, ['--gpus', '2', '--steps', '3', '--warmup', '1'])
    a = seen['argv']
    assert a[0] == sys.executable and a[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in a and a[a.index('--nproc-per-node') + 1] == '2'
    assert a[a.index('--master-addr') + 1] == '127.0.0.1' and 1024 < int(a[a.index('--master-port') + 1]) < 65536
    assert a[-7] == os.path.join(ROOT, 'bench.py') and a[-6:] == ['--gpus', '2', '--steps', '3', '--warmup', '1']
    # no GPU here (or fewer than ranks): the ranks share devices over gloo, and the line says so
    assert seen['env'].get('UPAMD_DIST_BACKEND') == 'gloo'
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'


def test_ref_dims_line_reduces_a_child_bench_line(monkeypatch):
    import subprocess
    child = {'value': 1.1e6, 'unit': 'samples/s', 'ms_per_step': 0.23, 'steps': 400, 'warmup': 256, 'node_steps_per_s': 3e8,
             'host_enqueue_ms_per_step': 0.1, 'kernel_ms_per_step': {'tiny_step': 0.21}, 'config': {'workload': 'hlg_ref: ...'},
             'roofline': {'kernel': 'tiny_step', 'bound': 'hbm', 'achieved': 177.0, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.022,
                          'avg_launch_ms': 0.21, 'traffic': None, 'note': 'dropped'},
             'cpu_baseline': {'value': 396.0, 'unit': 'samples/s', 'cores': 16, 'kind': 'port', 'sample': 's', 'ms_per_step': 161.0,
                              'tight_pad': 765.0, 'rows': ['dropped']},
             'update_params_inclusive': {'samples_per_s': 5e5, 'seconds': 0.06, 'fraction_of_step_rate': 0.48, 'prepare_s': 0.03,
                                         'loop_s': 0.03, 'note': 'dropped'}}

    def fake_run(cmd, capture_output, text, timeout):
        assert cmd[1].endswith('bench.py') and '--no-ref-dims' in cmd and cmd[cmd.index('--workload') + 1] == 'hlg_ref'
        assert int(cmd[cmd.index('--warmup') + 1]) >= 200          # sub-millisecond steps need the clock ramp
        return subprocess.CompletedProcess(cmd, 0, 'noise\n' + json.dumps(child) + '\n', '')

    monkeypatch.setattr(subprocess, 'run', fake_run)
    out = bench.ref_dims_line('hlg_ref')
    assert out['value'] == 1.1e6 and out['ms_per_step'] == 0.23 and out['roofline']['frac'] == 0.022
    assert out['cpu_baseline']['value'] == 396.0 and 'rows' not in out['cpu_baseline']
    assert out['update_params_inclusive']['fraction_of_step_rate'] == 0.48 and 'note' not in out['update_params_inclusive']

    def failing(cmd, capture_output, text, timeout):
        return subprocess.CompletedProcess(cmd, 3, '', 'boom')
    monkeypatch.setattr(subprocess, 'run', failing)
    assert 'error' in bench.ref_dims_line('grid_ref')


def test_route_overhead_is_the_difference_of_three_child_runs(monkeypatch):
    import subprocess
    seen = []

    def fake_run(cmd, capture_output, text, timeout, env):
        seen.append(env)
        assert '--minibatch' in cmd and cmd[cmd.index('--minibatch') + 1] == '256' and '--no-ref-dims' in cmd
        ms = 2.20 if 'UPAMD_DIST_FORCE_INIT' not in env else (2.24 if env['UPAMD_GRAD_BUCKETS'] == '0' else 2.31)
        return subprocess.CompletedProcess(cmd, 0, json.dumps({'ms_per_step': ms}) + '\n', '')

    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setenv('WORLD_SIZE', '8')                 # a launcher's variables must not leak into the one-rank children
    out = bench.route_overhead(256)
    assert abs(out['single_collective'] - 0.04) < 1e-9 and abs(out['bucketed'] - 0.11) < 1e-9 and out['source'].startswith('MEASURED')
    assert 'WORLD_SIZE' not in seen[0] and seen[1]['WORLD_SIZE'] == '1' and seen[1]['UPAMD_DIST_FORCE_INIT'] == '1'
    assert seen[1]['UPAMD_GRAD_BUCKETS'] == '0' and seen[2]['UPAMD_GRAD_BUCKETS'] == 'force'
    assert seen[1]['MASTER_ADDR'] == '127.0.0.1' and seen[1]['MASTER_PORT'] != seen[2]['MASTER_PORT']

    def failing(cmd, capture_output, text, timeout, env):
        return subprocess.CompletedProcess(cmd, 1, '', 'no rccl')
    monkeypatch.setattr(subprocess, 'run', failing)
    assert 'error' in bench.route_overhead(256)
