"""Executable specification (numpy, float64) of the ragged/CSR algorithm the HIP kernels implement.

Test infrastructure only.  It restates, stage by stage, what the native library does:
the python reference packer (padded state -> CSR graph), the factorised forward
(P/Q node GEMM + node-centric incidence sums, single-query attention through the collapsed
W_in*W_key products, pointer heads over candidate lists only) and the HAND-DERIVED backward.
``tests/test_csr_model.py`` checks it against the oracle's autograd, which validates the
derivations on CPU before any kernel runs; the GPU tests then compare the kernels' stage
outputs with the oracle.
"""
import numpy as np

PAD_LOGIT = np.float32(-2. ** 32 + 1)
EPS = 1e-6


# ----------------------------------------------------------------------------- packer spec

def pack_state(state, action):
    """Padded 9-field state -> CSR graph dict (mirrors csrc/packer.cpp)."""
    numerical, feat, edge_index, cur, node_mask, edge_mask, land_mask, road_mask, stage = state
    stage_id = int(np.argmax(stage))
    live = np.flatnonzero(edge_mask)
    n = 1
    nm = np.flatnonzero(node_mask)
    if nm.size:
        n = max(n, int(nm[-1]) + 1)
    if live.size:
        n = max(n, int(edge_index[live].max()) + 1)
    if stage_id == 1:
        rm = np.flatnonzero(road_mask)
        if rm.size:
            n = max(n, int(rm[-1]) + 1)
    if stage_id == 0:
        hm = np.flatnonzero(land_mask)
        # a head edge that is not a live edge carries m = 0 and touches no node (reference zeroes it)
    src = edge_index[live, 0].astype(np.int64)
    dst = edge_index[live, 1].astype(np.int64)
    e = live.size
    # head edges (stage 0 only), in padded slot order
    he_slot = np.flatnonzero(land_mask) if stage_id == 0 else np.zeros(0, dtype=np.int64)
    he_src = edge_index[he_slot, 0].astype(np.int64)
    he_dst = edge_index[he_slot, 1].astype(np.int64)
    he_live = edge_mask[he_slot].astype(np.uint8)
    if he_slot.size:
        dead = he_live == 0
        he_src[dead] = 0
        he_dst[dead] = 0
    he_of_slot = -np.ones(edge_index.shape[0], dtype=np.int64)
    he_of_slot[he_slot] = np.arange(he_slot.size)
    # incidence CSR: for live edge k=(i,j) in slot order: (i <- j), (j <- i)
    cnt = np.zeros(n, dtype=np.int64)
    np.add.at(cnt, src, 1)
    np.add.at(cnt, dst, 1)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(cnt)
    fill = row_ptr[:-1].copy()
    inc_nbr = np.zeros(2 * e, dtype=np.int64)
    inc_he = -np.ones(2 * e, dtype=np.int64)
    for k in range(e):
        i, j = src[k], dst[k]
        h = he_of_slot[live[k]]
        inc_nbr[fill[i]] = j
        inc_he[fill[i]] = h
        fill[i] += 1
        inc_nbr[fill[j]] = i
        inc_he[fill[j]] = h
        fill[j] += 1
    rn_node = np.flatnonzero(road_mask) if stage_id == 1 else np.zeros(0, dtype=np.int64)
    order = np.argsort(-cnt, kind='stable')          # edge-kernel processing order: degree descending, stable
    # candidate-incidence lists: live candidate h = (i, j): i <- (j, h), j <- (i, h)
    hcnt = np.zeros(n, dtype=np.int64)
    for q in range(he_slot.size):
        if he_live[q]:
            hcnt[he_src[q]] += 1
            hcnt[he_dst[q]] += 1
    hinc_ptr = np.zeros(n + 1, dtype=np.int64)
    hinc_ptr[1:] = np.cumsum(hcnt)
    hfill = hinc_ptr[:-1].copy()
    hinc_nbr = np.zeros(int(hinc_ptr[-1]), dtype=np.int64)
    hinc_he = np.zeros(int(hinc_ptr[-1]), dtype=np.int64)
    for q in range(he_slot.size):
        if he_live[q]:
            i, j = he_src[q], he_dst[q]
            hinc_nbr[hfill[i]], hinc_he[hfill[i]] = j, q
            hfill[i] += 1
            hinc_nbr[hfill[j]], hinc_he[hfill[j]] = i, q
            hfill[j] += 1
    act = -1
    if stage_id == 0:
        a = int(action[0])
        if 0 <= a < he_of_slot.size:
            act = int(he_of_slot[a])
    elif stage_id == 1:
        a = int(action[1])
        pos = np.flatnonzero(rn_node == a)
        act = int(pos[0]) if pos.size else -1
    return dict(n=n, e=e, stage=stage_id, X=feat[:n].astype(np.float64), nmask=node_mask[:n].copy(),
                row_ptr=row_ptr, inc_nbr=inc_nbr, inc_he=inc_he, he_src=he_src, he_dst=he_dst, he_live=he_live,
                he_slot=he_slot, rn_node=rn_node, act=act, order=order, hinc_ptr=hinc_ptr, hinc_nbr=hinc_nbr,
                hinc_he=hinc_he, numerical=numerical.astype(np.float64).ravel(),
                cur=cur.astype(np.float64), stage_vec=stage.astype(np.float64), pad_n=feat.shape[0],
                pad_e=edge_index.shape[0], n_mask=int(node_mask.sum()))


# ----------------------------------------------------------------------------- model

def _seq(P, prefix, stem='linear_'):
    keys, i = [], 0
    while prefix + stem + str(i) + '.weight' in P:
        keys.append(prefix + stem + str(i))
        i += 1
    return keys


class CsrModel:
    """Forward + manual backward over a list of packed graphs.  P: name -> float64 ndarray."""

    def __init__(self, P, num_heads=1):
        self.P = {k: np.asarray(v, dtype=np.float64) for k, v in P.items()}
        self.heads = num_heads
        self.L = 0
        while 'shared_net.edge_fc_layers.%d.linear_0.weight' % self.L in self.P:
            assert 'shared_net.edge_fc_layers.%d.linear_1.weight' % self.L not in self.P, 'K>1 unsupported'
            self.L += 1
        self.D = self.P['shared_net.node_encoder.weight'].shape[0]

    # ---- forward
    def forward(self, graphs):
        P, D, L, Hn = self.P, self.D, self.L, self.heads
        B = len(graphs)
        dh = D // Hn
        sv = dict(graphs=graphs)
        # numerical encoder
        U = [np.stack([g['numerical'] for g in graphs])]
        for key in _seq(P, 'shared_net.numerical_feature_encoder.'):
            U.append(np.tanh(U[-1] @ P[key + '.weight'].T + P[key + '.bias']))
        sv['U'] = U
        We, be = P['shared_net.node_encoder.weight'], P['shared_net.node_encoder.bias']
        Xc = np.stack([g['cur'] for g in graphs])
        C = Xc @ We.T + be
        sv['Xc'], sv['C'] = Xc, C
        # collapsed attention products
        pre = 'shared_net.'
        Win, bin_ = P[pre + 'attention_layer.in_proj_weight'], P[pre + 'attention_layer.in_proj_bias']
        Wiq, Wik, Wiv = Win[:D], Win[D:2 * D], Win[2 * D:]
        biq, biv = bin_[:D], bin_[2 * D:]
        Wq, bq = P[pre + 'attention_query_layer.weight'], P[pre + 'attention_query_layer.bias']
        Wk = P[pre + 'attention_key_layer.weight']
        Wv, bv = P[pre + 'attention_value_layer.weight'], P[pre + 'attention_value_layer.bias']
        Wo, bo = P[pre + 'attention_layer.out_proj.weight'], P[pre + 'attention_layer.out_proj.bias']
        Wkk = Wik @ Wk
        Wvv = Wiv @ Wv
        bvv = Wiv @ bv + biv
        scale = 1.0 / np.sqrt(dh)
        q0 = C @ Wq.T + bq
        q1 = (q0 @ Wiq.T + biq) * scale
        sv.update(q0=q0, q1=q1, Wkk=Wkk, Wvv=Wvv, bvv=bvv)
        per = []
        hbarV = np.zeros((B, D))
        hbarE = np.zeros((B, D))
        s_all = np.zeros((B, Hn, D))
        for b, g in enumerate(graphs):
            n = g['n']
            H = [g['X'] @ We.T + be]
            PQs, Ss = [], []
            cnt = (g['row_ptr'][1:] - g['row_ptr'][:-1]).astype(np.float64)
            for l in range(L):
                W = P['shared_net.edge_fc_layers.%d.linear_0.weight' % l]
                bl = P['shared_net.edge_fc_layers.%d.linear_0.bias' % l]
                Pm = H[-1] @ W[:, :D].T
                Qm = H[-1] @ W[:, D:].T
                S = np.zeros((n, D))
                for v in range(n):
                    for k in range(g['row_ptr'][v], g['row_ptr'][v + 1]):
                        u = g['inc_nbr'][k]
                        S[v] += 0.5 * (np.tanh(Pm[v] + Qm[u] + bl) + np.tanh(Pm[u] + Qm[v] + bl))
                H.append(H[-1] + S / (cnt[:, None] + EPS))
                PQs.append((Pm, Qm))
                Ss.append(S)
            HL = H[-1]
            hbarV[b] = HL[g['nmask']].sum(0) / g['n_mask']
            hbarE[b] = 0.5 * Ss[-1].sum(0) / g['e']
            # single-query attention over node_mask nodes
            alpha = np.zeros((Hn, n))
            for h in range(Hn):
                r = q1[b, h * dh:(h + 1) * dh] @ Wkk[h * dh:(h + 1) * dh]          # [D]
                sc = HL @ r
                sc = np.where(g['nmask'], sc, -np.inf)
                sc = sc - sc.max()
                ex = np.exp(sc)
                alpha[h] = ex / ex.sum()
                s_all[b, h] = alpha[h] @ HL
            per.append(dict(H=H, PQ=PQs, S=Ss, cnt=cnt, alpha=alpha))
        o = np.zeros((B, D))
        for h in range(Hn):
            o[:, h * dh:(h + 1) * dh] = s_all[:, h] @ Wvv[h * dh:(h + 1) * dh].T + bvv[h * dh:(h + 1) * dh]
        att = o @ Wo.T + bo
        stage_vec = np.stack([g['stage_vec'] for g in graphs])
        SV = np.concatenate([U[-1], hbarV, hbarE, att, stage_vec], axis=1)
        V = [SV]
        vkeys = _seq(P, 'value_head.')
        for i, key in enumerate(vkeys):
            zz = V[-1] @ P[key + '.weight'].T + P[key + '.bias']
            V.append(np.tanh(zz) if i < len(vkeys) - 1 else zz)
        value = V[-1][:, 0]
        sv.update(per=per, hbarV=hbarV, hbarE=hbarE, s=s_all, o=o, att=att, SV=SV, V=V)
        # pointer heads
        logp = np.zeros(B)
        ent = np.zeros(B)
        heads_sv = []
        lkeys = _seq(P, 'policy_land_use_head.', 'land_use_linear_')
        rkeys = _seq(P, 'policy_road_head.', 'road_linear_')
        bL = P['shared_net.edge_fc_layers.%d.linear_0.bias' % (L - 1)]
        for b, g in enumerate(graphs):
            hs = dict(kind=None)
            if g['stage'] == 0:
                Pm, Qm = per[b]['PQ'][-1]
                i, j = g['he_src'], g['he_dst']
                m = 0.5 * (np.tanh(Pm[i] + Qm[j] + bL) + np.tanh(Pm[j] + Qm[i] + bL))
                m = m * g['he_live'][:, None]
                c = np.broadcast_to(C[b], m.shape)
                feat = np.concatenate([m, c, m * c, m - c], axis=1)
                acts = [feat]
                for t, key in enumerate(lkeys):
                    zz = acts[-1] @ P[key + '.weight'].T
                    if key + '.bias' in P:
                        zz = zz + P[key + '.bias']
                    acts.append(np.tanh(zz) if t < len(lkeys) - 1 else zz)
                z = acts[-1][:, 0]
                hs = dict(kind='land', m=m, acts=acts, z=z, npad=g['pad_e'])
            elif g['stage'] == 1:
                Xr = per[b]['H'][-1][g['rn_node']]
                acts = [Xr]
                for t, key in enumerate(rkeys):
                    zz = acts[-1] @ P[key + '.weight'].T
                    if key + '.bias' in P:
                        zz = zz + P[key + '.bias']
                    acts.append(np.tanh(zz) if t < len(rkeys) - 1 else zz)
                z = acts[-1][:, 0]
                hs = dict(kind='road', acts=acts, z=z, npad=g['pad_n'])
            if hs['kind'] is not None:
                z = hs['z']
                if z.size == 0:        # no valid candidate: the reference's fp32 normalised logits are all 0
                    logp[b] = 0.0       # (logsumexp of N copies of -2^32+1 is absorbed in fp32)
                    ent[b] = 0.0
                    hs['p'] = z
                else:
                    mx = z.max()
                    lse = mx + np.log(np.exp(z - mx).sum())
                    p = np.exp(z - lse)
                    hs['p'], hs['lse'] = p, lse
                    ent[b] = lse - (p * z).sum()
                    logp[b] = (z[g['act']] - lse) if g['act'] >= 0 else (float(PAD_LOGIT) - lse)
                    hs['ent'] = ent[b]
            heads_sv.append(hs)
        sv['heads'] = heads_sv
        self.sv = sv
        return value, logp, ent

    # ---- backward from per-row seeds dvalue, dlogp, dent
    def backward(self, dvalue, dlogp, dent):
        P, D, L, Hn, sv = self.P, self.D, self.L, self.heads, self.sv
        graphs = sv['graphs']
        B = len(graphs)
        dh = D // Hn
        G = {k: np.zeros_like(v) for k, v in P.items()}
        # value head
        vkeys = _seq(P, 'value_head.')
        V = sv['V']
        dz = dvalue[:, None]
        for i in reversed(range(len(vkeys))):
            key = vkeys[i]
            if i < len(vkeys) - 1:
                dz = dz * (1 - V[i + 1] ** 2)
            G[key + '.weight'] += dz.T @ V[i]
            G[key + '.bias'] += dz.sum(0)
            dz = dz @ P[key + '.weight']
        dSV = dz
        S_last = sv['U'][-1].shape[1]
        dUlast = dSV[:, :S_last]
        dhbarV = dSV[:, S_last:S_last + D]
        dhbarE = dSV[:, S_last + D:S_last + 2 * D]
        datt = dSV[:, S_last + 2 * D:S_last + 3 * D]
        # numerical encoder
        nkeys = _seq(P, 'shared_net.numerical_feature_encoder.')
        U = sv['U']
        dz = dUlast
        for i in reversed(range(len(nkeys))):
            key = nkeys[i]
            dz = dz * (1 - U[i + 1] ** 2)
            G[key + '.weight'] += dz.T @ U[i]
            G[key + '.bias'] += dz.sum(0)
            dz = dz @ P[key + '.weight']
        # attention dense part
        pre = 'shared_net.'
        Win = P[pre + 'attention_layer.in_proj_weight']
        Wiq, Wik, Wiv = Win[:D], Win[D:2 * D], Win[2 * D:]
        Wq = P[pre + 'attention_query_layer.weight']
        Wk = P[pre + 'attention_key_layer.weight']
        Wv, bv = P[pre + 'attention_value_layer.weight'], P[pre + 'attention_value_layer.bias']
        Wo = P[pre + 'attention_layer.out_proj.weight']
        Wkk, Wvv = sv['Wkk'], sv['Wvv']
        scale = 1.0 / np.sqrt(dh)
        G[pre + 'attention_layer.out_proj.weight'] += datt.T @ sv['o']
        G[pre + 'attention_layer.out_proj.bias'] += datt.sum(0)
        do = datt @ Wo
        dWvv = np.zeros_like(Wvv)
        dbvv = do.sum(0)
        ds = np.zeros((B, Hn, D))
        for h in range(Hn):
            sl = slice(h * dh, (h + 1) * dh)
            dWvv[sl] += do[:, sl].T @ sv['s'][:, h]
            ds[:, h] = do[:, sl] @ Wvv[sl]
        dC = np.zeros((B, D))
        dq1 = np.zeros((B, D))
        dWkk = np.zeros_like(Wkk)
        We = P['shared_net.node_encoder.weight']
        lkeys = _seq(P, 'policy_land_use_head.', 'land_use_linear_')
        rkeys = _seq(P, 'policy_road_head.', 'road_linear_')
        for b, g in enumerate(graphs):
            n = g['n']
            pb = sv['per'][b]
            HL = pb['H'][-1]
            GL = np.zeros((n, D))
            GL[g['nmask']] += dhbarV[b] / g['n_mask']
            # attention core
            for h in range(Hn):
                sl = slice(h * dh, (h + 1) * dh)
                r = sv['q1'][b, sl] @ Wkk[sl]
                a = pb['alpha'][h]
                dalpha = HL @ ds[b, h]
                GL += a[:, None] * ds[b, h][None, :]
                dsc = a * (dalpha - (a * dalpha).sum())
                dr = dsc @ HL
                GL += dsc[:, None] * r[None, :]
                dq1[b, sl] = Wkk[sl] @ dr
                dWkk[sl] += np.outer(sv['q1'][b, sl], dr)
            # pointer head
            hs = sv['heads'][b]
            dm_he = None
            if hs['kind'] is not None and hs['z'].size > 0:
                p, z = hs['p'], hs['z']
                dzc = -dlogp[b] * p - dent[b] * p * ((z - hs['lse']) + hs['ent'])
                if g['act'] >= 0:
                    dzc[g['act']] += dlogp[b]
                keys = lkeys if hs['kind'] == 'land' else rkeys
                acts = hs['acts']
                dzz = dzc[:, None]
                for t in reversed(range(len(keys))):
                    key = keys[t]
                    if t < len(keys) - 1:
                        dzz = dzz * (1 - acts[t + 1] ** 2)
                    G[key + '.weight'] += dzz.T @ acts[t]
                    if key + '.bias' in P:
                        G[key + '.bias'] += dzz.sum(0)
                    dzz = dzz @ P[key + '.weight']
                if hs['kind'] == 'land':
                    m = hs['m']
                    c = sv['C'][b]
                    g1, g2, g3, g4 = dzz[:, :D], dzz[:, D:2 * D], dzz[:, 2 * D:3 * D], dzz[:, 3 * D:]
                    dm_he = (g1 + g3 * c + g4) * g['he_live'][:, None]
                    dC[b] += (g2 + g3 * m - g4).sum(0)
                else:
                    np.add.at(GL, g['rn_node'], dzz)
            # GCN layers, last to first
            Gl = GL
            for l in reversed(range(L)):
                W = P['shared_net.edge_fc_layers.%d.linear_0.weight' % l]
                bl = P['shared_net.edge_fc_layers.%d.linear_0.bias' % l]
                Pm, Qm = pb['PQ'][l]
                dS = Gl / (pb['cnt'][:, None] + EPS)
                if l == L - 1:
                    dS = dS + 0.5 * dhbarE[b] / g['e']
                dP = np.zeros((n, D))
                dQ = np.zeros((n, D))
                for v in range(n):
                    for k in range(g['row_ptr'][v], g['row_ptr'][v + 1]):
                        u = g['inc_nbr'][k]
                        dm = dS[v] + dS[u]
                        if l == L - 1 and dm_he is not None and g['inc_he'][k] >= 0:
                            dm = dm + dm_he[g['inc_he'][k]]
                        t1 = np.tanh(Pm[v] + Qm[u] + bl)
                        t2 = np.tanh(Pm[u] + Qm[v] + bl)
                        dP[v] += 0.5 * dm * (1 - t1 ** 2)
                        dQ[v] += 0.5 * dm * (1 - t2 ** 2)
                Hprev = pb['H'][l]
                G['shared_net.edge_fc_layers.%d.linear_0.weight' % l][:, :D] += dP.T @ Hprev
                G['shared_net.edge_fc_layers.%d.linear_0.weight' % l][:, D:] += dQ.T @ Hprev
                G['shared_net.edge_fc_layers.%d.linear_0.bias' % l] += dP.sum(0)
                Gl = Gl + dP @ W[:, :D] + dQ @ W[:, D:]
            G['shared_net.node_encoder.weight'] += Gl.T @ g['X']
            G['shared_net.node_encoder.bias'] += Gl.sum(0)
        # attention q-chain
        dpre = dq1 * scale
        gin_w = G[pre + 'attention_layer.in_proj_weight']
        gin_b = G[pre + 'attention_layer.in_proj_bias']
        gin_w[:D] += dpre.T @ sv['q0']
        gin_b[:D] += dpre.sum(0)
        dq0 = dpre @ Wiq
        G[pre + 'attention_query_layer.weight'] += dq0.T @ sv['C']
        G[pre + 'attention_query_layer.bias'] += dq0.sum(0)
        dC += dq0 @ Wq
        # collapsed products: Wkk = Wik Wk, Wvv = Wiv Wv, bvv = Wiv bv + biv (key-side biases get exactly 0)
        gin_w[D:2 * D] += dWkk @ Wk.T
        G[pre + 'attention_key_layer.weight'] += Wik.T @ dWkk
        gin_w[2 * D:] += dWvv @ Wv.T + np.outer(dbvv, bv)
        G[pre + 'attention_value_layer.weight'] += Wiv.T @ dWvv
        G[pre + 'attention_value_layer.bias'] += Wiv.T @ dbvv
        gin_b[2 * D:] += dbvv
        # current-node encoder
        G['shared_net.node_encoder.weight'] += dC.T @ sv['Xc']
        G['shared_net.node_encoder.bias'] += dC.sum(0)
        return G


def ppo_seeds(value, logp, ent, adv, ret, old_logp, exps, clip_eps, cv, ce):
    """Loss terms + per-row seeds (dL/dvalue, dL/dlogp, dL/dent) of
    loss = surr + cv * value_loss + ce * entropy_loss (urban_planning_agent.py:326-333, 363-371)."""
    B = value.shape[0]
    ind = exps != 0
    nind = max(int(ind.sum()), 1)
    vl = ((value - ret) ** 2).mean()
    ratio = np.exp(logp - old_logp)
    lo, hi = 1 - clip_eps, 1 + clip_eps
    s1 = ratio * adv
    s2 = np.clip(ratio, lo, hi) * adv
    surr = -(np.minimum(s1, s2)[ind]).sum() / nind
    el = -(ent[ind]).sum() / nind
    loss = surr + cv * vl + ce * el
    inside = (ratio >= lo) & (ratio <= hi)
    dsdr = np.where(inside, adv, np.where(s1 < s2, adv, 0.0))
    dlogp = np.where(ind, -dsdr * ratio / nind, 0.0)
    dent = np.where(ind, -ce / nind, 0.0)
    dvalue = cv * 2.0 * (value - ret) / B
    return (loss, vl, surr, el), dvalue, dlogp, dent
