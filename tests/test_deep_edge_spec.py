"""Executable specification (numpy, float64) of the per-incidence algorithm of csrc/deep_edge.hip -- edge MLPs with
K > 1 sub-layers (urban_planning/models/state_encoder.py:59-82,110-148) -- checked on the CPU against the reference's dense, padded
formulation (restated here in float64) and its autograd.  Test infrastructure only.

What is specified (and mirrored kernel by kernel): the index tables (opposite direction of every incidence by rank
matching, a candidate's incidence), the forward (gather, sub-layers, node segment sums, last-layer edge mean and candidate
messages) and the hand-derived backward (seed with the candidate term routed to the t-th matching incidence, 1 - A^2 chain,
dP | dQ gather), on graphs with a self-loop, duplicate edges, isolated nodes and candidates on dead slots.
"""
import numpy as np
import pytest
import torch

import cases
import csr_model

EPS = 1e-6


def inc_tables(g):
    """rev[k]: the incidence of the opposite direction; cand_inc[q]: an incidence of candidate q's edge (-1: dead)."""
    rp, nb, n = g['row_ptr'], g['inc_nbr'], g['n']
    rev = np.zeros(len(nb), dtype=np.int64)
    src = np.zeros(len(nb), dtype=np.int64)
    for v in range(n):
        for k in range(rp[v], rp[v + 1]):
            u = nb[k]
            rank = int((nb[rp[v]:k] == u).sum())            # k is the rank-th (v -> u) entry of v's list
            match = [j for j in range(rp[u], rp[u + 1]) if nb[j] == v]
            rev[k] = match[rank]                            # ... paired with the rank-th (u -> v) entry of u's list
            src[k] = v
    cand = -np.ones(len(g['he_src']), dtype=np.int64)
    for q in range(len(cand)):
        if g['he_live'][q]:
            s, d = g['he_src'][q], g['he_dst'][q]
            cand[q] = next(j for j in range(rp[s], rp[s + 1]) if nb[j] == d)
    return src, rev, cand


def layer_forward(g, H, Ws, bs, tables):
    """One GCN layer with K = len(Ws) sub-layers.  Returns H_out, the kept activations, S, the candidate messages."""
    src, rev, cand = tables
    D = H.shape[1]
    nb = g['inc_nbr']
    P, Q = H @ Ws[0][:, :D].T, H @ Ws[0][:, D:].T
    A = [np.tanh(P[src] + Q[nb] + bs[0])]                   # A_1[k = v -> u] = tanh(P_v + Q_u + b_0)
    for W, b in zip(Ws[1:], bs[1:]):
        A.append(np.tanh(A[-1] @ W.T + b))
    rp = g['row_ptr']
    S = np.zeros_like(H)
    for v in range(g['n']):
        for k in range(rp[v], rp[v + 1]):
            S[v] += 0.5 * (A[-1][k] + A[-1][rev[k]])
    deg = (rp[1:] - rp[:-1]).astype(np.float64)
    m = np.where(cand[:, None] >= 0, 0.5 * (A[-1][np.maximum(cand, 0)] + A[-1][rev[np.maximum(cand, 0)]]), 0.0)
    return H + S / (deg[:, None] + EPS), A, S, m


def layer_backward(g, H, Ws, A, tables, G, dhbarE, dM):
    """Gradients of sum(H_out * G) + sum(hbarE * dhbarE) + sum(m * dM) w.r.t. H and the layer's parameters."""
    src, rev, cand = tables
    D = H.shape[1]
    rp, nb, n, e = g['row_ptr'], g['inc_nbr'], g['n'], g['e']
    deg = (rp[1:] - rp[:-1]).astype(np.float64)
    dS = G / (deg[:, None] + EPS) + 0.5 * dhbarE / e        # hbarE = 1/2 sum_v S_v / e
    dA = np.zeros_like(A[-1])
    hp, hnb, hhe = g['hinc_ptr'], g['hinc_nbr'], g['hinc_he']
    for v in range(n):
        for k in range(rp[v], rp[v + 1]):
            u = nb[k]
            dm = dS[v] + dS[u]
            # the rank-th (v -> u) incidence takes the rank-th candidate of v's candidate-incidence list across (v, u)
            rank = int((nb[rp[v]:k] == u).sum())
            match = [j for j in range(hp[v], hp[v + 1]) if hnb[j] == u]
            if rank < len(match):
                dm = dm + dM[hhe[match[rank]]]
            dA[k] = 0.5 * dm
    grads = {}
    dpre = dA * (1.0 - A[-1] ** 2)
    for j in range(len(Ws) - 1, 0, -1):                     # linear_j: A_j -> A_j+1
        grads['b%d' % j] = dpre.sum(0)
        grads['W%d' % j] = dpre.T @ A[j - 1]
        dpre = (dpre @ Ws[j]) * (1.0 - A[j - 1] ** 2)
    dP, dQ = np.zeros((n, D)), np.zeros((n, D))
    for v in range(n):
        for k in range(rp[v], rp[v + 1]):
            dP[v] += dpre[k]                                # dpre_1[k = v -> u] feeds P_v ...
            dQ[v] += dpre[rev[k]]                           # ... and Q_u: Q_v collects the opposite directions
    grads['b0'] = dP.sum(0)
    grads['W0'] = np.concatenate([dP.T @ H, dQ.T @ H], axis=1)
    grads['H'] = G + dP @ Ws[0][:, :D] + dQ @ Ws[0][:, D:]
    return grads


def _dense_layer(P, K, h, edge_index, edge_mask):
    """The reference's padded dense layer in float64 (the oracle's functions are fp32-typed): gather_to_edges
    (state_encoder.py:110-130) with K sub-layers (:59-82), scatter_to_nodes (:132-148, counts from :84-108), residual
    (:194-197), masked edge mean (:179-182)."""
    D = h.size(-1)
    i0 = edge_index[:, :, 0].unsqueeze(-1).expand(-1, -1, D)
    i1 = edge_index[:, :, 1].unsqueeze(-1).expand(-1, -1, D)
    h1, h2 = torch.gather(h, 1, i0), torch.gather(h, 1, i1)

    def fc(z):
        for k in range(K):
            z = torch.tanh(torch.nn.functional.linear(z, P['shared_net.edge_fc_layers.0.linear_%d.weight' % k],
                                                      P['shared_net.edge_fc_layers.0.linear_%d.bias' % k]))
        return z
    he = (fc(torch.cat([h1, h2], -1)) + fc(torch.cat([h2, h1], -1))) / 2
    mask = edge_mask.unsqueeze(-1).expand_as(he)
    he = torch.where(mask, he, torch.zeros_like(he))
    cnt = mask.to(h.dtype)
    agg = torch.zeros_like(h).scatter_add(1, i0, he).scatter_add(1, i1, he)
    num = torch.zeros_like(h).scatter_add(1, i0, cnt).scatter_add(1, i1, cnt)
    hbar_e = (he * cnt).sum(1) / edge_mask.to(h.dtype).sum(1, keepdim=True)
    return he, h + agg / (num + EPS), hbar_e


@pytest.mark.parametrize('K', [2, 3])
def test_per_incidence_layer_matches_the_dense_reference_formulation(K):
    D, max_nodes, max_edges = 8, 24, 60
    replay = cases.quirky_replay(6, max_nodes, max_edges, seed=11, road_fraction=0.0, full_row=False, dead_candidate=True)
    rng = np.random.default_rng(3)
    Ws = [0.6 * rng.standard_normal((D, 2 * D))] + [0.8 * rng.standard_normal((D, D)) for _ in range(K - 1)]
    bs = [0.3 * rng.standard_normal(D) for _ in range(K)]
    for t, (state, action) in enumerate(zip(replay.states, replay.actions)):
        g = csr_model.pack_state(state, action)
        n, e = g['n'], g['e']
        tables = inc_tables(g)
        src, rev, cand = tables
        assert np.array_equal(rev[rev], np.arange(2 * e)) and np.array_equal(g['inc_nbr'][rev], src)      # an involution that swaps the endpoints
        H = rng.standard_normal((n, D))
        Hout, A, S, m = layer_forward(g, H, Ws, bs, tables)
        # ---- the oracle's dense padded layer on the same graph (batch of one)
        P = {}
        for k in range(K):
            P['shared_net.edge_fc_layers.0.linear_%d.weight' % k] = torch.tensor(Ws[k], requires_grad=True)
            P['shared_net.edge_fc_layers.0.linear_%d.bias' % k] = torch.tensor(bs[k], requires_grad=True)
        Hd = torch.zeros(1, max_nodes, D, dtype=torch.float64)
        Hd[0, :n] = torch.tensor(H)
        Hd.requires_grad_(True)
        ei = torch.tensor(state[2]).unsqueeze(0)
        em = torch.tensor(state[5]).unsqueeze(0)
        h_edges, h_new, hbarE = _dense_layer(P, K, Hd, ei, em)
        np.testing.assert_allclose(Hout, h_new[0, :n].detach().numpy(), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(0.5 * S.sum(0) / e, hbarE[0].detach().numpy(), rtol=1e-9, atol=1e-10)
        m_ref = h_edges[0, g['he_slot']].detach().numpy()                       # candidates = land-use-mask slots, slot order
        np.testing.assert_allclose(m, m_ref, rtol=1e-9, atol=1e-10)
        if t >= 3:
            assert (g['he_live'] == 0).any()                                    # a candidate on a dead slot: message 0
        # ---- backward
        G = rng.standard_normal((n, D))
        dhE = rng.standard_normal(D)
        dM = rng.standard_normal(m.shape) * g['he_live'][:, None]              # (he_feat_bwd masks dead candidates)
        loss = (h_new[0, :n] * torch.tensor(G)).sum() + (hbarE[0] * torch.tensor(dhE)).sum() + \
            (h_edges[0, g['he_slot']] * torch.tensor(dM)).sum()
        loss.backward()
        mine = layer_backward(g, H, Ws, A, tables, G, dhE, dM)
        np.testing.assert_allclose(mine['H'], Hd.grad[0, :n].numpy(), rtol=1e-8, atol=1e-9)
        for k in range(K):
            np.testing.assert_allclose(mine['W%d' % k], P['shared_net.edge_fc_layers.0.linear_%d.weight' % k].grad.numpy(),
                                       rtol=1e-8, atol=1e-9, err_msg='W%d' % k)
            np.testing.assert_allclose(mine['b%d' % k], P['shared_net.edge_fc_layers.0.linear_%d.bias' % k].grad.numpy(),
                                       rtol=1e-8, atol=1e-9, err_msg='b%d' % k)
