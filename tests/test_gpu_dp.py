"""Data-parallel ``update_params`` on real kernels with more than one rank, without a
multi-GPU node: G processes share cuda:0 and exchange gradients over gloo.  Every rank runs the documented drop-in
(``HipUpdateMixin`` on a duck agent -> ``PPOUpdater`` picks its ``DistContext`` up from the torchrun environment).

In 'global' mode a G-rank run must reproduce the REFERENCE's single-process trajectory (golden ``upd/scalars``,
``upd_sd``, ``upd2_sd`` produced by the real reference): same global permutation, same global minibatches, each rank
``B / G`` of their rows, loss scaled by global counts, one gradient all-reduce per step."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(world, name, out_dir, mode, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), UPAMD_DIST_BACKEND='gloo', OMP_NUM_THREADS='4')
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py'), name, str(out_dir), mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (rank, out[-4000:])
    return np.load(os.path.join(str(out_dir), 'rank0.npz'))


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize('name,world,mode', [('case_a', 2, 'global'), ('case_c', 3, 'global'), ('case_a', 2, 'bcast')])
def test_multi_rank_update_reproduces_the_reference_trajectory(name, world, mode, tmp_path):
    z, sd, states = helpers.load_case(name)
    got = _launch(world, name, tmp_path, mode)
    assert str(got['mode']) == 'global' and int(got['rows_per_step']) == z['mb/adv'].shape[0]
    np.testing.assert_allclose(got['losses1'], z['upd/scalars'], rtol=1e-4, atol=2e-6)
    n1 = z['upd/scalars'].shape[0]
    np.testing.assert_allclose(got['losses2'], z['upd2/scalars'][n1:], rtol=2e-4, atol=5e-6)
    assert int(got['loss_iter']) == z['upd2/scalars'].shape[0]
    assert int(got['n_scalars']) > 0
    for k in [k[4:] for k in got.files if k.startswith('sd1/')]:
        assert _rel_l2(got['sd1/' + k], z['upd_sd/' + k]) <= 1e-4, k
        assert _rel_l2(got['sd2/' + k], z['upd2_sd/' + k]) <= 2e-4, k


def test_local_shards_run_and_agree_on_the_step_count(tmp_path):
    """'local' mode: every rank shuffles its own half of the replay; the run must complete (equal numbers of
    collectives on every rank), report the global rows per step, and move the parameters."""
    name, world = 'case_a', 2
    z, sd, states = helpers.load_case(name)
    got = _launch(world, name, tmp_path, 'local')
    assert str(got['mode']) == 'local' and int(got['rows_per_step']) == z['mb/adv'].shape[0]
    assert np.isfinite(got['losses1']).all() and got['losses1'].shape[0] > 0
    moved = max(_rel_l2(got['sd1/' + k[4:]], sd[k[4:]].numpy()) for k in got.files if k.startswith('sd1/'))
    assert moved > 0
