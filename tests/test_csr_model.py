"""Validates the CSR-form algorithm + hand-derived backward (tests/csr_model.py, the executable spec
of the HIP kernels) against the oracle's autograd on the golden cases.  CPU only."""
import numpy as np
import pytest
import torch

import csr_model
from oracle import sgnn_oracle as orc
from test_oracle_golden import load_case, CASE_B, CASE_HEADS, CASE_HYPER


@pytest.mark.parametrize('name', ['case_a', 'case_b', 'case_c'])
def test_csr_forward_backward_vs_oracle(name):
    z, sd, states = load_case(name)
    flat = orc.split_actor_critic_state_dict(sd)
    B = CASE_B[name]
    hy = CASE_HYPER[name]
    actions = z['actions'][:B]
    graphs = [csr_model.pack_state(states[b], actions[b]) for b in range(B)]
    model = csr_model.CsrModel({k: v.numpy() for k, v in flat.items()}, CASE_HEADS[name])
    value, logp, ent = model.forward(graphs)
    np.testing.assert_allclose(value, z['fwd/value'][:, 0], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(logp, z['fwd/logp'][:, 0], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent, z['fwd/entropy'][:, 0], rtol=1e-5, atol=1e-5)
    losses, dvalue, dlogp, dent = csr_model.ppo_seeds(
        value, logp, ent, z['mb/adv'][:, 0].astype(np.float64), z['mb/ret'][:, 0].astype(np.float64),
        z['mb/old_logp'][:, 0].astype(np.float64), z['exps'][:B], hy['clip_epsilon'], hy['value_pred_coef'],
        hy['entropy_coef'])
    np.testing.assert_allclose(losses, z['mb/losses'], rtol=1e-5, atol=1e-6)
    G = model.backward(dvalue, dlogp, dent)
    scale = max(np.abs(z[k]).max() for k in z.files if k.startswith('grad/'))
    for k, g in G.items():
        ref_key = 'grad/actor_net.' + k if not k.startswith('value_head.') else 'grad/value_net.' + k
        np.testing.assert_allclose(g, z[ref_key], rtol=2e-4, atol=2e-6 * max(scale, 1.0), err_msg=k)
