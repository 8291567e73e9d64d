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


def test_land_use_head_first_linear_on_the_messages_alone():
    """csrc/head.hip: with c a per-GRAPH vector the m*c half of the head input [m ; m*c] is a per-graph change of the weight,
    W_b = Wa' + Wc diag(c_b); the weight gradient splits into sum dpre^T m and its per-graph c_b-weighted version.  Both
    identities against the explicit [m ; m*c] form (float64)."""
    rng = np.random.default_rng(0)
    D, h0 = 12, 5
    sizes = [7, 0, 4, 9]                                 # candidates per graph (one graph without any)
    W = rng.standard_normal((h0, 2 * D))                # [Wa' | Wc]
    C = rng.standard_normal((len(sizes), D))
    const = rng.standard_normal((len(sizes), h0))
    m = [rng.standard_normal((k, D)) for k in sizes]
    dpre = [rng.standard_normal((k, h0)) for k in sizes]
    hid_ref = [np.tanh(np.concatenate([mb, mb * C[b]], 1) @ W.T + const[b]) for b, mb in enumerate(m)]
    dW_ref = sum(dp.T @ np.concatenate([mb, mb * C[b]], 1) for b, (mb, dp) in enumerate(zip(m, dpre)))
    dWa, dWc = np.zeros((h0, D)), np.zeros((h0, D))
    for b, (mb, dp) in enumerate(zip(m, dpre)):
        Wb = W[:, :D] + W[:, D:] * C[b]                 # head_hidden_fwd: built once per graph
        np.testing.assert_allclose(np.tanh(mb @ Wb.T + const[b]), hid_ref[b], rtol=1e-12, atol=1e-12)
        T = dp.T @ mb                                   # head_wgrad: per-graph product, accumulated plain and c-weighted
        dWa += T
        dWc += T * C[b]
    np.testing.assert_allclose(np.concatenate([dWa, dWc], 1), dW_ref, rtol=1e-12, atol=1e-12)
