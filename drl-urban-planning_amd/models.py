"""Drop-in ``nn.Module`` surface of the reference's SGNN policy / value networks.

``create_sgnn_model(cfg, agent) -> (policy_net, value_net)`` mirrors
urban_planning/models/model.py:8-19; attribute names, construction order (hence seeded
initialisation) and ``state_dict`` keys are those of urban_planning/models/state_encoder.py:13-33,
policy.py:9-43 and value.py:8-34, so reference checkpoints load both ways
(``ActorCritic(policy_net, value_net).load_state_dict``).

Two execution paths, selected by where the parameters live:

* **cuda** -- the hot path: states are packed to the ragged/CSR form and the whole network
  (forward and hand-written backward) runs in the HIP library through ``NativeEngine``; autograd
  sees one ``torch.autograd.Function``.  If the native library is missing this raises -- there is
  no fallback.
* **cpu** -- rollout workers (``Agent.sample`` moves the modules to the CPU and forks,
  khrylib/rl/agents/agent.py:75-100) evaluate single states with plain torch ops on the padded
  tensors.  This path never sees an optimizer step; ``update_params`` refuses to run on it.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native, packer
from .engine import NativeEngine

_PAD_LOGIT = -2. ** 32 + 1       # policy.py:50,59


def _mlp(sizes_in, hidden, prefix='linear_', act_prefix='tanh_', first_flatten=False, bias_after_first=True,
         last_act=True, flatten_last_if_one=False, last_flatten_prefix='flatten_'):
    seq = nn.Sequential()
    prev = sizes_in
    for i, h in enumerate(hidden):
        if i == 0 and first_flatten:
            seq.add_module('flatten_0', nn.Flatten())
        seq.add_module(prefix + str(i), nn.Linear(prev, h, bias=(i == 0 or bias_after_first)))
        if i < len(hidden) - 1 or last_act:
            seq.add_module(act_prefix + str(i), nn.Tanh())
        elif flatten_last_if_one and h == 1:
            seq.add_module(last_flatten_prefix + str(i), nn.Flatten())
        prev = h
    return seq


class SGNNStateEncoder(nn.Module):
    """Parameter container + CPU forward of the shared GNN state encoder (state_encoder.py:7-214)."""
    EPSILON = 1e-6

    def __init__(self, cfg, agent):
        super().__init__()
        self.cfg = cfg
        self.agent = agent
        D = cfg['gcn_node_dim']
        self.numerical_feature_encoder = _mlp(agent.numerical_feature_size, cfg['state_encoder_hidden_size'],
                                              first_flatten=True)
        self.node_encoder = nn.Linear(agent.node_dim, D)
        self.num_gcn_layers = cfg['num_gcn_layers']
        self.num_edge_fc_layers = cfg['num_edge_fc_layers']
        self.edge_fc_layers = nn.ModuleList()
        for _ in range(self.num_gcn_layers):
            fc = nn.Sequential()
            for k in range(self.num_edge_fc_layers):
                fc.add_module('linear_%d' % k, nn.Linear(2 * D if k == 0 else D, D))
                fc.add_module('tanh_%d' % k, nn.Tanh())
            self.edge_fc_layers.append(fc)
        self.max_num_nodes = cfg['max_num_nodes']
        self.max_num_edges = cfg['max_num_edges']
        self.attention_layer = nn.MultiheadAttention(D, cfg['num_attention_heads'])
        self.attention_query_layer = nn.Linear(D, D)
        self.attention_key_layer = nn.Linear(D, D)
        self.attention_value_layer = nn.Linear(D, D)
        self.output_policy_land_use_size = 4 * D
        self.output_policy_road_size = D
        self.output_value_size = 3 * D + cfg['state_encoder_hidden_size'][-1] + 3

    # ---- CPU rollout path (padded dense tensors, B is 1 in practice)
    @staticmethod
    def batch_data(x):
        return [torch.stack(f) for f in zip(*x)]

    def _messages(self, h, edge_index, edge_mask, fc):
        D = h.size(-1)
        src = edge_index[..., 0].unsqueeze(-1).expand(-1, -1, D)
        dst = edge_index[..., 1].unsqueeze(-1).expand(-1, -1, D)
        hs, hd = torch.gather(h, 1, src), torch.gather(h, 1, dst)
        m = 0.5 * (fc(torch.cat([hs, hd], -1)) + fc(torch.cat([hd, hs], -1)))
        return m * edge_mask.unsqueeze(-1).to(m.dtype), src, dst

    def _aggregate(self, m, src, dst, edge_mask, num_nodes):
        agg = torch.zeros(m.size(0), num_nodes, m.size(-1), dtype=m.dtype, device=m.device)
        cnt = torch.zeros_like(agg)
        ones = edge_mask.unsqueeze(-1).expand_as(m).to(m.dtype)
        for idx in (src, dst):
            agg = agg.scatter_add(1, idx, m)
            cnt = cnt.scatter_add(1, idx, ones)
        return agg / (cnt + self.EPSILON)

    def forward(self, x):
        numerical, nodes, edge_index, cur, node_mask, edge_mask, land_mask, road_mask, stage = self.batch_data(x)
        h_num = self.numerical_feature_encoder(numerical)
        h = self.node_encoder(nodes)
        c = self.node_encoder(cur.unsqueeze(1))
        m = None
        for fc in self.edge_fc_layers:
            m, src, dst = self._messages(h, edge_index, edge_mask, fc)
            h = h + self._aggregate(m, src, dst, edge_mask, nodes.size(1))
        fe, fn = edge_mask.unsqueeze(-1).to(h.dtype), node_mask.unsqueeze(-1).to(h.dtype)
        mean_e = (m * fe).sum(1) / fe.sum(1)
        mean_n = (h * fn).sum(1) / fn.sum(1)
        q = self.attention_query_layer(c).transpose(0, 1)
        k = self.attention_key_layer(h).transpose(0, 1)
        v = self.attention_value_layer(h).transpose(0, 1)
        att, _ = self.attention_layer(q, k, v, key_padding_mask=~node_mask)
        att = att.transpose(0, 1).squeeze(1)
        state_value = torch.cat([h_num, mean_n, mean_e, att, stage], dim=1)
        cc = c.expand(-1, m.size(1), -1)
        state_land = torch.cat([m, cc, m * cc, m - cc], dim=-1)
        return state_land, h, state_value, land_mask, road_mask, stage


class MLPStateEncoder(nn.Module):
    """Parameter container + CPU forward of the ``rl-mlp`` ablation encoder (state_encoder.py:217-308): numerical
    encoder + node encoder only -- no message passing, no attention.  An edge is embedded as the node encoder applied to
    the raw features of ONE endpoint (the second one if that node's type is FEASIBLE, else the first; masked edges have
    zero features, i.e. embed to the bias)."""
    NUM_TYPE_COLS = 14        # city_config.NUM_TYPES + 1 (urban_planning/envs/city_config.py:53)
    FEASIBLE = 1              # city_config.FEASIBLE     (urban_planning/envs/city_config.py:24)

    def __init__(self, cfg, agent):
        super().__init__()
        self.cfg = cfg
        self.agent = agent
        D = cfg['gcn_node_dim']
        self.numerical_feature_encoder = _mlp(agent.numerical_feature_size, cfg['state_encoder_hidden_size'],
                                              first_flatten=True)
        self.node_encoder = nn.Linear(agent.node_dim, D)
        self.max_num_nodes = cfg['max_num_nodes']
        self.max_num_edges = cfg['max_num_edges']
        self.output_policy_land_use_size = 4 * D
        self.output_policy_road_size = D
        self.output_value_size = 2 * D + cfg['state_encoder_hidden_size'][-1] + 3

    def forward(self, x):
        numerical, nodes, edge_index, cur, node_mask, edge_mask, land_mask, road_mask, stage = SGNNStateEncoder.batch_data(x)
        h_num = self.numerical_feature_encoder(numerical)
        F_ = nodes.size(-1)
        x1 = torch.gather(nodes, 1, edge_index[..., 0].unsqueeze(-1).expand(-1, -1, F_))
        x2 = torch.gather(nodes, 1, edge_index[..., 1].unsqueeze(-1).expand(-1, -1, F_))
        second = x2[..., :self.NUM_TYPE_COLS].argmax(-1) == self.FEASIBLE
        xe = torch.where(second.unsqueeze(-1), x2, x1) * edge_mask.unsqueeze(-1).to(nodes.dtype)
        h = self.node_encoder(nodes)
        m = self.node_encoder(xe)
        c = self.node_encoder(cur.unsqueeze(1))
        fe, fn = edge_mask.unsqueeze(-1).to(h.dtype), node_mask.unsqueeze(-1).to(h.dtype)
        mean_e = (m * fe).sum(1) / fe.sum(1)
        mean_n = (h * fn).sum(1) / fn.sum(1)
        state_value = torch.cat([h_num, mean_n, mean_e, stage], dim=1)
        cc = c.expand(-1, m.size(1), -1)
        state_land = torch.cat([m, cc, m * cc, m - cc], dim=-1)
        return state_land, h, state_value, land_mask, road_mask, stage


class _HipNetwork(torch.autograd.Function):
    """value / log-prob / entropy of a packed minibatch through the native engine."""

    @staticmethod
    def forward(ctx, flat_params, runner):
        value, logp, ent = runner.run_forward(flat_params)
        ctx.runner = runner
        ctx.save_for_backward(flat_params)
        return value, logp, ent

    @staticmethod
    def backward(ctx, dvalue, dlogp, dent):
        (flat_params,) = ctx.saved_tensors
        grads = ctx.runner.run_backward(flat_params, dvalue, dlogp, dent)
        return grads, None


class _Runner:
    """One packed batch bound to an engine.  A runner that will be differentiated owns a PRIVATE workspace: the
    activations of its forward stay intact until its own backward has run, whatever else is evaluated in between
    (``value_net(x1)``, then ``get_log_prob_entropy(x2, a)``, then ``loss.backward()`` -- or a no-grad
    ``select_action`` between a forward and its backward).  No-grad runners use the engine's scratch slot."""

    def __init__(self, engine, packed, sched, need_grad, private_ws=False):
        self.engine, self.packed, self.sched = engine, packed, sched
        self.mb, self.item = sched.minibatch(0)
        self.need_grad = need_grad
        # private_ws: a no-grad forward whose intermediates are READ BACK afterwards (the action heads' logits) also
        # owns its workspace -- the shared scratch slot could be overwritten by another thread's no-grad forward (the
        # action server next to the learner's pre-pass) between the forward and the read
        self.ws = engine.alloc_workspace(self.mb) if (need_grad or private_ws) else None
        self.slot = 0 if need_grad else 'nograd'
        self.backward_done = False

    def run_forward(self, flat_params):
        B, dev = self.mb.B, self.engine.device
        value = torch.empty(B, device=dev)
        logp = torch.empty(B, device=dev)
        ent = torch.empty(B, device=dev)
        self.engine.forward(self.packed, self.mb, flat_params.contiguous(), value, logp, ent, keep=self.need_grad,
                            ws=self.ws)
        return value, logp, ent

    def tensor(self, name):
        """Named intermediate of this runner's forward (candidate logits for the action heads)."""
        return self.engine.ws_tensor(self.mb, name, slot=self.slot, ws=self.ws)

    def run_backward(self, flat_params, dvalue, dlogp, dent):
        if self.backward_done:
            raise RuntimeError('this forward has already been differentiated once; the native backward consumes the '
                               'kept activations (retain_graph is not supported on the HIP path)')
        if self.ws is None:
            raise RuntimeError('backward through a forward that ran without gradient tracking (its activations were '
                               'not kept)')
        B, dev = self.mb.B, self.engine.device
        z = torch.zeros(B, device=dev)
        grads = torch.zeros_like(flat_params)
        self.engine.backward(self.packed, self.mb, flat_params.contiguous(),
                             z if dvalue is None else dvalue.contiguous(), z if dlogp is None else dlogp.contiguous(),
                             z if dent is None else dent.contiguous(), grads, ws=self.ws)
        self.backward_done = True
        self.ws = None                 # released (stream-ordered: the caching allocator reuses it after the backward)
        return grads


class _HipBackend:
    """Shared by policy_net and value_net (they share the encoder object): engine + name mapping."""

    def __init__(self, shared_net, policy_cfg, value_cfg):
        self.shared_net = shared_net
        self.policy_cfg, self.value_cfg = policy_cfg, value_cfg
        self.policy_net = None
        self.value_net = None
        self._engine = None

    def needs_mlp_fields(self):
        """Only the rl-mlp encoder reads the replay's he_sel / xbar sections; an SGNN replay is packed without them (two thirds of a
        state's fill time on the host, packer.plan_replay)."""
        return isinstance(self.shared_net, MLPStateEncoder)

    def desc(self):
        kind = native.ENCODER_MLP if isinstance(self.shared_net, MLPStateEncoder) else native.ENCODER_SGNN
        return native.make_desc(self.shared_net.cfg, self.policy_cfg, self.value_cfg, self.shared_net.agent.node_dim,
                                self.shared_net.agent.numerical_feature_size, encoder=kind)

    def engine(self, device):
        if self._engine is None or self._engine.device != torch.device(device):
            self._engine = NativeEngine(self.desc(), device)
        return self._engine

    def named_params(self):
        """flat-layout name -> nn.Parameter (de-duplicated: the encoder appears once)."""
        out = {}
        for k, v in self.shared_net.named_parameters():
            out['shared_net.' + k] = v
        for k, v in self.policy_net.policy_land_use_head.named_parameters():
            out['policy_land_use_head.' + k] = v
        for k, v in self.policy_net.policy_road_head.named_parameters():
            out['policy_road_head.' + k] = v
        for k, v in self.value_net.value_head.named_parameters():
            out['value_head.' + k] = v
        return out

    def flat_params_autograd(self, engine):
        """Flat parameter vector that autograd can differentiate back to the module parameters."""
        named = self.named_params()
        pieces, cursor = [], 0
        for name, off, rows, cols, _ in engine.table:
            if off > cursor:
                pieces.append(torch.zeros(off - cursor, device=engine.device))
            pieces.append(named[name].reshape(-1))
            cursor = off + rows * cols
        if engine.n_floats > cursor:
            pieces.append(torch.zeros(engine.n_floats - cursor, device=engine.device))
        return torch.cat(pieces)

    def run(self, x, action, private_ws=False, reuse=None):
        """x: list[B] of list[9] tensors on any device.  Returns (value, logp, ent) f32[B] on the GPU.  ``reuse``: a dict whose
        page-locked staging buffer is recycled (only for callers that synchronise with the device before their next call)."""
        device = next(self.shared_net.parameters()).device
        engine = self.engine(device)
        states = [s if packer.is_record(s) else
                  [f.detach().cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f) for f in s] for s in x]
        B = len(states)
        if action is None:
            act = np.zeros((B, 2), dtype=np.float32)
        else:
            act = action.detach().cpu().numpy().astype(np.float32).reshape(B, 2)
        pk = packer.pack_replay(states, act, self.shared_net.agent.node_dim,
                                self.shared_net.agent.numerical_feature_size, reuse=reuse, mlp_fields=self.needs_mlp_fields()).to(device)
        sched = packer.Schedule(pk, [np.arange(B)], device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.named_params().values())
        runner = _Runner(engine, pk, sched, need_grad, private_ws)
        flat = self.flat_params_autograd(engine)
        value, logp, ent = _HipNetwork.apply(flat, runner)
        return value, logp, ent, runner

    def serve_actions(self, x, mean_rows):
        """``select_action`` (policy.py:67-85) for the batched action server: one no-grad HIP forward for all rows, then per
        row the arg-max (``mean_rows[b]``) or a ``Categorical.sample`` over its OWN candidates -- the masked slots of the
        reference's padded logits have probability exactly 0 (pad constant -2^32 + 1), so a Categorical over the candidates
        alone is the same distribution and the same arg-max (first maximum in slot order).  Returns f32 [B, 2] on the host.
        Lean on purpose (the server calls it once per serving round): cached flat parameters (re-flattened when a parameter
        changed), recycled page-locked staging, no autograd graph, no [B, pad] logit tensors."""
        device = next(self.shared_net.parameters()).device
        engine = self.engine(device)
        cache = self.__dict__.setdefault('_serve', {})
        plist = cache.get('plist')
        if plist is None:              # (walking named_parameters() is 0.1 ms: once per backend, not once per serving round)
            plist = cache['plist'] = list(self.named_params().values())
        version = tuple(p._version for p in plist) + tuple(p.data_ptr() for p in plist)
        if cache.get('version') != version or cache.get('device') != device:
            named = self.named_params()
            cache['plist'] = list(named.values())
            cache['flat'] = engine.flatten(named, out=cache.get('flat') if cache.get('device') == device else None)
            cache['version'] = tuple(p._version for p in cache['plist']) + tuple(p.data_ptr() for p in cache['plist'])
            cache['device'] = device
        B = len(x)
        if getattr(x, 'addr', None) is not None:
            states = x                 # records with their address table (the action server's ring): no per-record Python below
        else:
            states = [s if packer.is_record(s) else
                      [f.detach().cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f) for f in s] for s in x]
        torch.cuda.current_stream(device).synchronize()      # (the previous round's copies have left the recycled staging buffer)
        pk = packer.pack_replay(states, np.zeros((B, 2), dtype=np.float32), self.shared_net.agent.node_dim,
                                self.shared_net.agent.numerical_feature_size, reuse=cache.setdefault('pack', {}),
                                mlp_fields=self.needs_mlp_fields()).to(device)
        sched = packer.Schedule(pk, [np.arange(B)], device)
        mb, _ = sched.minibatch(0)
        out = cache.get('rows')
        if out is None or out.shape[1] < B:
            out = cache['rows'] = torch.empty(3, max(B, 64), device=device)
        with torch.no_grad():
            engine.forward(pk, mb, cache['flat'], out[0, :B], out[1, :B], out[2, :B], keep=False, slot='serve')
            z_he = engine.ws_view(mb, 'z_he', slot='serve') if int(pk.layout.total_he) else None
            z_rn = engine.ws_view(mb, 'z_rn', slot='serve') if int(pk.layout.total_rn) else None
            if os.environ.get('UPAMD_SERVE_SELECT', 'hip') == 'torch':      # (lab A/B: round 5's route through torch ops)
                return ragged_actions(pk, x, z_he.reshape(-1) if z_he is not None else None,
                                      z_rn.reshape(-1) if z_rn is not None else None, mean_rows, device)
            # arg-max / inverse-CDF draw per row in ONE launch (upamd_select_actions); the uniforms come from the device generator
            sel = cache.get('sel')
            if sel is None or sel[0].numel() < B:
                cap = max(B, 64)
                sel = cache['sel'] = (torch.empty(cap, dtype=torch.uint8, device=device), torch.empty(cap, 2, device=device),
                                      torch.empty(cap, 2, dtype=torch.float32).pin_memory())
            g_dev, _ = packer.upload_pinned(np.ascontiguousarray(mean_rows, dtype=np.uint8), device)
            u = torch.rand(B, device=device)
            engine.select_actions(pk, mb, z_he, z_rn, g_dev, u, sel[1])
            sel[2][:B].copy_(sel[1][:B], non_blocking=True)
            torch.cuda.current_stream(device).synchronize()
            return sel[2][:B].numpy().copy()

    def pointer_logits(self, x):
        """The two pointer heads of a batch as the reference lays them out (policy.py:45-65): logits over the PADDED
        edge / node slots of the rows in that stage, masked slots = the pad constant.  Forward only (the differentiable
        route to log-prob / entropy is ``get_log_prob_entropy``).  Returns (land_logits | None, road_logits | None,
        stage f32[B, 3]) on the networks' device."""
        device = next(self.shared_net.parameters()).device
        with torch.no_grad():
            # (forward-only and the caller reads the result back before it calls again -- select_action, the action server --
            # so the pinned staging buffer of the pack can be recycled instead of page-locking a fresh one per call)
            if not hasattr(self, '_serve_cache'):
                self._serve_cache = {}
            torch.cuda.current_stream(device).synchronize()
            _, _, _, runner = self.run(x, None, private_ws=True, reuse=self._serve_cache)
            pk, mb, meta = runner.packed, runner.mb, runner.packed.meta
            B = mb.B
            stage_id = meta[:B, packer.M_STAGE]
            out = []
            for sid, zname, cnt_col, slot_name, slot_dt, pad_col, total in (
                    (0, 'z_he', packer.M_NH, 'he_slot', np.int32, packer.M_PADE, int(pk.layout.total_he)),
                    (1, 'z_rn', packer.M_NR, 'rn_node', np.uint16, packer.M_PADN, int(pk.layout.total_rn))):
                rows = np.flatnonzero(stage_id == sid)
                if rows.size == 0:
                    out.append(None)
                    continue
                # Categorical over the PADDED slots: records are trimmed, their header keeps the original pad sizes
                width = np.array([packer.record_pads(x[b])[0 if pad_col == packer.M_PADN else 1] if packer.is_record(x[b])
                                  else meta[b, pad_col] for b in rows])
                if (width != width[0]).any():
                    raise ValueError('forward(): the rows of one stage must share their pad size to form one '
                                     'Categorical (got %s); use select_action / get_log_prob_entropy for ragged '
                                     'batches' % sorted(set(width.tolist())))
                logits = torch.full((rows.size, int(width[0])), _PAD_LOGIT, dtype=torch.float32, device=device)
                cnt = meta[:B, cnt_col].astype(np.int64)
                off = np.concatenate([[0], np.cumsum(cnt)])           # candidate offsets in minibatch (= pack) order
                if cnt[rows].sum() > 0:
                    slots = pk.section(slot_name, slot_dt, max(total, 1)).astype(np.int64)
                    src = np.concatenate([np.arange(off[b], off[b + 1]) for b in rows])
                    dst_row = np.repeat(np.arange(rows.size), cnt[rows])
                    z = runner.tensor(zname).reshape(-1)
                    logits[torch.from_numpy(dst_row).to(device), torch.from_numpy(slots[src]).to(device)] = \
                        z[torch.from_numpy(src).to(device)]
                out.append(logits)
            stage = np.zeros((B, 3), dtype=np.float32)
            for b, s in enumerate(x):
                st = packer.record_stage(s) if packer.is_record(s) else s[8]
                stage[b] = st.detach().cpu().numpy() if isinstance(st, torch.Tensor) else np.asarray(st)
        return out[0], out[1], torch.from_numpy(stage).to(device)


def ragged_actions(pk, x, z_he, z_rn, mean_rows, device):
    """Per-row action of a packed batch from the RAGGED pointer-head logits (``z_he``: the land-use rows' candidates, ``z_rn``:
    the road rows', both in pack order): arg-max where ``mean_rows[b]``, else one ``Categorical.sample`` -- over the row's own
    candidates, padded to the widest row with the reference's pad constant (probability exactly 0, policy.py:50-52), then
    mapped back to the PADDED edge / node slot the reference's action indexes (policy.py:70-83).  f32 [B, 2] on the host."""
    meta = pk.meta
    B = len(x)
    stage = meta[:B, packer.M_STAGE]
    cnt = np.where(stage == 0, meta[:B, packer.M_NH], np.where(stage == 1, meta[:B, packer.M_NR], 0)).astype(np.int64)
    action = np.zeros((B, 2), dtype=np.float32)
    width = int(cnt.max()) if B else 0
    pick = np.zeros(B, dtype=np.int64)
    start = np.zeros(B, dtype=np.int64)
    if width > 0:
        # candidates of the land-use rows / road rows are consecutive in pack order in z_he / z_rn
        for sid, col in ((0, packer.M_NH), (1, packer.M_NR)):
            sel = stage == sid
            c = np.where(sel, meta[:B, col], 0).astype(np.int64)
            start[sel] = (np.cumsum(c) - c)[sel]
        col = np.arange(width)[None, :]
        valid = col < cnt[:, None]
        src = np.where(valid, start[:, None] + col, 0)
        idx, _ = packer.upload_pinned(np.stack([src, valid.astype(np.int64), np.broadcast_to((stage == 1).astype(np.int64)[:, None], src.shape)]), device)
        zeros = torch.zeros(src.shape, dtype=torch.float32, device=device)
        zl = z_he[idx[0].clamp(max=z_he.numel() - 1)] if z_he is not None and z_he.numel() else zeros
        zr = z_rn[idx[0].clamp(max=z_rn.numel() - 1)] if z_rn is not None and z_rn.numel() else zeros
        dense = torch.where(idx[1].bool(), torch.where(idx[2].bool(), zr, zl), torch.full_like(zeros, _PAD_LOGIT))
        dist = torch.distributions.Categorical(logits=dense)
        greedy = torch.from_numpy(np.ascontiguousarray(mean_rows, dtype=bool)).to(device)
        pick = torch.where(greedy, dist.probs.argmax(dim=1), dist.sample()).cpu().numpy()
    he_slot = pk.section('he_slot', np.int32, max(int(pk.layout.total_he), 1))
    rn_node = pk.section('rn_node', np.uint16, max(int(pk.layout.total_rn), 1))
    for b in range(B):
        if stage[b] not in (0, 1):
            continue
        if cnt[b] == 0:
            # no candidate at all: the reference's logits are the pad constant everywhere -- a uniform Categorical over the
            # padded slots (arg-max: slot 0).  torch's stream, as the reference's sample; never numpy's global one (the
            # update's permutations live there)
            pad = packer.record_pads(x[b])[1 - int(stage[b])] if packer.is_record(x[b]) else int(meta[b, packer.M_PADE if stage[b] == 0 else packer.M_PADN])
            action[b, int(stage[b])] = 0.0 if mean_rows[b] else float(torch.randint(max(pad, 1), (1,)).item())
            continue
        k = int(start[b] + min(int(pick[b]), int(cnt[b]) - 1))
        action[b, int(stage[b])] = float(he_slot[k] if stage[b] == 0 else rn_node[k])
    return action


def _on_gpu(module):
    return next(module.parameters()).device.type == 'cuda'


def _dense_states(x):
    """CPU path input: compact wire records (rollout.ActionClient / arenas) expanded to the padded 9-tensor form."""
    if not any(packer.is_record(s) for s in x):
        return x
    return [[torch.from_numpy(a) for a in packer.expand_state(s, padded=True)] if packer.is_record(s) else s for s in x]


class UrbanPlanningPolicy(nn.Module):
    """Two masked-softmax pointer heads over edges / nodes (policy.py:5-104)."""

    def __init__(self, cfg, agent, shared_net, backend=None):
        super().__init__()
        self.cfg = cfg
        self.agent = agent
        self.shared_net = shared_net
        self.policy_land_use_head = self._head(shared_net.output_policy_land_use_size,
                                               cfg['policy_land_use_head_hidden_size'], 'land_use')
        self.policy_road_head = self._head(shared_net.output_policy_road_size,
                                           cfg['policy_road_head_hidden_size'], 'road')
        self._backend = [backend]          # list: keep the backend out of nn.Module attribute registration

    @staticmethod
    def _head(input_size, hidden, name):
        return _mlp(input_size, hidden, prefix=name + '_linear_', act_prefix=name + '_tanh_', bias_after_first=False,
                    last_act=False, flatten_last_if_one=True, last_flatten_prefix=name + '_flatten_')

    def forward(self, x):
        """(land_use_dist | None, road_dist | None, stage) exactly as policy.py:45-65: ``Categorical`` objects over
        the padded edge / node slots of the rows in each stage.  On the GPU the logits come from the HIP forward
        (no gradient flows through this route; ``get_log_prob_entropy`` is the differentiable one)."""
        if _on_gpu(self):
            land, road, stage = self._backend[0].pointer_logits(x)
            land_dist = None if land is None else torch.distributions.Categorical(logits=land)
            road_dist = None if road is None else torch.distributions.Categorical(logits=road)
            return land_dist, road_dist, stage
        s_land, s_road, _, land_mask, road_mask, stage = self.shared_net(_dense_states(x))
        land_dist = road_dist = None
        is_land, is_road = stage[:, 0].bool(), stage[:, 1].bool()
        if is_land.any():
            z = self.policy_land_use_head(s_land[is_land])
            z = torch.where(land_mask[is_land], z, torch.full_like(z, _PAD_LOGIT))
            land_dist = torch.distributions.Categorical(logits=z)
        if is_road.any():
            z = self.policy_road_head(s_road[is_road])
            z = torch.where(road_mask[is_road], z, torch.full_like(z, _PAD_LOGIT))
            road_dist = torch.distributions.Categorical(logits=z)
        return land_dist, road_dist, stage

    def select_action(self, x, mean_action=False):
        """policy.py:67-85 on either device: ``Categorical.sample`` (torch's generator of that device) or arg-max,
        written into column 0 / 1 of the rows in the land-use / road stage."""
        land_dist, road_dist, stage = self.forward(x)
        action = torch.zeros(stage.shape[0], 2, dtype=self.agent.dtype, device=stage.device)
        for col, dist in ((0, land_dist), (1, road_dist)):
            if dist is not None:
                a = dist.probs.argmax(dim=1) if mean_action else dist.sample()
                action[stage[:, col].bool(), col] = a.to(self.agent.dtype)
        return action

    def get_log_prob_entropy(self, x, action):
        if _on_gpu(self):
            _, logp, ent, _ = self._backend[0].run(x, action)
            return logp.unsqueeze(1), ent.unsqueeze(1)
        land_dist, road_dist, stage = self.forward(x)
        B = stage.shape[0]
        log_prob = torch.zeros(B, dtype=self.agent.dtype, device=stage.device)
        entropy = torch.zeros(B, dtype=self.agent.dtype, device=stage.device)
        for col, dist in ((0, land_dist), (1, road_dist)):
            if dist is not None:
                sel = stage[:, col].bool()
                log_prob[sel] = dist.log_prob(action[sel, col])
                entropy[sel] = dist.entropy()
        return log_prob.unsqueeze(1), entropy.unsqueeze(1)


class UrbanPlanningValue(nn.Module):
    """MLP value head on the pooled graph features (value.py:4-39)."""

    def __init__(self, cfg, agent, shared_net, backend=None):
        super().__init__()
        self.cfg = cfg
        self.agent = agent
        self.shared_net = shared_net
        self.value_head = _mlp(shared_net.output_value_size, cfg['value_head_hidden_size'], last_act=False)
        self._backend = [backend]

    def forward(self, x):
        if _on_gpu(self):
            value, _, _, _ = self._backend[0].run(x, None)
            return value.unsqueeze(1)
        _, _, state_value, _, _, _ = self.shared_net(_dense_states(x))
        return self.value_head(state_value)


def create_sgnn_model(cfg, agent):
    """Drop-in for urban_planning/models/model.py:8-19.  ``cfg`` carries the three spec dicts
    (``state_encoder_specs``, ``policy_specs``, ``value_specs``); ``agent`` carries ``node_dim``,
    ``numerical_feature_size`` and ``dtype``."""
    shared_net = SGNNStateEncoder(cfg.state_encoder_specs, agent)
    backend = _HipBackend(shared_net, cfg.policy_specs, cfg.value_specs)
    policy_net = UrbanPlanningPolicy(cfg.policy_specs, agent, shared_net, backend)
    value_net = UrbanPlanningValue(cfg.value_specs, agent, shared_net, backend)
    backend.policy_net, backend.value_net = policy_net, value_net
    return policy_net, value_net


def create_mlp_model(cfg, agent):
    """Drop-in for urban_planning/models/model.py:22-33 (``--agent rl-mlp``): the same policy / value heads on the
    ``MLPStateEncoder``; on a GPU device it runs on the same HIP engine (encoder kind UPAMD_ENCODER_MLP)."""
    shared_net = MLPStateEncoder(cfg.state_encoder_specs, agent)
    backend = _HipBackend(shared_net, cfg.policy_specs, cfg.value_specs)
    policy_net = UrbanPlanningPolicy(cfg.policy_specs, agent, shared_net, backend)
    value_net = UrbanPlanningValue(cfg.value_specs, agent, shared_net, backend)
    backend.policy_net, backend.value_net = policy_net, value_net
    return policy_net, value_net


class ActorCritic(nn.Module):
    """Same container as urban_planning/models/model.py:36-47 (one optimizer / checkpoint dict)."""

    def __init__(self, actor_net, value_net):
        super().__init__()
        self.actor_net = actor_net
        self.value_net = value_net


def backend_of(net):
    """The shared HIP backend of a policy_net / value_net created by ``create_sgnn_model``."""
    return net._backend[0]
