"""ORACLE -- test infrastructure, NOT product code.

CPU (PyTorch fp32) restatement of the reference's SGNN policy/value forward + PPO update
hot path, written from the reference's behaviour (padded dense batches, same op sequence)
so that it (a) is a faithful stand-in for timing the reference's CPU path where
/root/reference is absent (the GPU box) and (b) is the checker the HIP path is compared
with.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package never does.

Pinning: ``tests/golden/*.npz`` were produced by running the REAL reference modules
(``tests/golden/make_golden.py``, import recipe in ``oracle/ref_import.py``) and
``tests/test_oracle_golden.py`` checks this restatement against them, so parity is pinned
by outputs of the reference itself (the reference ships no tests or golden vectors of its
own -- SURVEY.md section 8c).

Each function cites the reference file:line it follows (paths relative to /root/reference).
Semantics that live in torch itself (nn.MultiheadAttention, Categorical, clip_grad_norm_,
optim.Adam) are pinned by calling the installed torch, exactly as the reference does.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PAD_LOGIT = -2. ** 32 + 1      # policy.py:50,59
EPSILON = 1e-6                 # state_encoder.py:11


# ----------------------------------------------------------------------------- parameters

def split_actor_critic_state_dict(sd):
    """ActorCritic.state_dict() (model.py:36-47) -> de-duplicated flat dict.

    Keys: 'shared_net.*' (from actor_net), 'policy_land_use_head.*', 'policy_road_head.*',
    'value_head.*'.
    """
    out = {}
    for k, v in sd.items():
        if k.startswith('actor_net.'):
            out[k[len('actor_net.'):]] = v
        elif k.startswith('value_net.value_head.'):
            out[k[len('value_net.'):]] = v
    return out


def leaf_params(sd, requires_grad=True):
    return {k: v.detach().clone().float().requires_grad_(requires_grad) for k, v in sd.items()}


def _seq_keys(P, prefix, stem='linear_'):
    idx = 0
    keys = []
    while prefix + stem + str(idx) + '.weight' in P:
        keys.append(prefix + stem + str(idx))
        idx += 1
    return keys


# ----------------------------------------------------------------------------- batching

def tensorfy(np_list):
    """urban_planning_agent.py:16-20."""
    return [[torch.tensor(x) for x in y] for y in np_list]


def batch_data(x):
    """state_encoder.py:163-177 -- zip(*x) + 9 stacks."""
    return [torch.stack(f) for f in zip(*x)]


# ----------------------------------------------------------------------------- encoder

def gather_to_edges(P, layer, h_nodes, edge_index, edge_mask):
    """state_encoder.py:110-130 (K = num_edge_fc_layers sub-layers, :59-82)."""
    D = h_nodes.size(-1)
    h1 = torch.gather(h_nodes, 1, edge_index[:, :, 0].unsqueeze(-1).expand(-1, -1, D))
    h2 = torch.gather(h_nodes, 1, edge_index[:, :, 1].unsqueeze(-1).expand(-1, -1, D))

    def fc(z):
        for key in _seq_keys(P, 'shared_net.edge_fc_layers.%d.' % layer):
            z = torch.tanh(F.linear(z, P[key + '.weight'], P[key + '.bias']))
        return z
    h_edges = (fc(torch.cat([h1, h2], -1)) + fc(torch.cat([h2, h1], -1))) / 2
    mask = edge_mask.unsqueeze(-1).expand_as(h_edges)
    return torch.where(mask, h_edges, torch.zeros_like(h_edges))


def scatter_count(h_edges, indices, edge_mask, max_num_nodes):
    """state_encoder.py:84-108."""
    B, _, D = h_edges.shape
    h_nodes = torch.zeros(B, max_num_nodes, D)
    count_edge = torch.zeros_like(h_nodes)
    count = edge_mask.unsqueeze(-1).expand_as(h_edges).float()
    idx = indices.unsqueeze(-1).expand(-1, -1, D)
    h_nodes = h_nodes.scatter_add(1, idx, h_edges)
    count_edge = count_edge.scatter_add(1, idx, count)
    return h_nodes, count_edge


def scatter_to_nodes(h_edges, edge_index, edge_mask, max_num_nodes):
    """state_encoder.py:132-148."""
    h1, c1 = scatter_count(h_edges, edge_index[:, :, 0], edge_mask, max_num_nodes)
    h2, c2 = scatter_count(h_edges, edge_index[:, :, 1], edge_mask, max_num_nodes)
    return (h1 + h2) / (c1 + c2 + EPSILON)


def mean_features(h, mask):
    """state_encoder.py:179-182."""
    return (h * mask.unsqueeze(-1).float()).sum(dim=1) / mask.float().sum(dim=1, keepdim=True)


def self_attention(P, h_current_node, h_nodes, node_mask, num_heads):
    """state_encoder.py:150-161 + nn.MultiheadAttention(D, heads) (:26), seq-first layout."""
    pre = 'shared_net.'
    q = F.linear(h_current_node, P[pre + 'attention_query_layer.weight'], P[pre + 'attention_query_layer.bias'])
    k = F.linear(h_nodes, P[pre + 'attention_key_layer.weight'], P[pre + 'attention_key_layer.bias'])
    v = F.linear(h_nodes, P[pre + 'attention_value_layer.weight'], P[pre + 'attention_value_layer.bias'])
    q, k, v = q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1)
    D = q.size(-1)
    out, _ = F.multi_head_attention_forward(
        q, k, v, D, num_heads,
        P[pre + 'attention_layer.in_proj_weight'], P[pre + 'attention_layer.in_proj_bias'],
        None, None, False, 0.0,
        P[pre + 'attention_layer.out_proj.weight'], P[pre + 'attention_layer.out_proj.bias'],
        training=True, key_padding_mask=~node_mask, need_weights=True)
    return out.transpose(0, 1).squeeze(1)


MLP_TYPE_COLS = 14        # city_config.NUM_TYPES + 1 (urban_planning/envs/city_config.py:53)
MLP_FEASIBLE = 1          # city_config.FEASIBLE     (urban_planning/envs/city_config.py:24)


def is_mlp_params(P):
    """The rl-mlp encoder (MLPStateEncoder, state_encoder.py:217-236) has neither edge MLPs nor attention."""
    return 'shared_net.attention_layer.in_proj_weight' not in P


def mlp_encoder_forward(P, x, keep=None):
    """MLPStateEncoder.forward, state_encoder.py:284-308 (+ compute_edge_features :262-282)."""
    numerical, node_features, edge_index, cur, node_mask, edge_mask, land_use_mask, road_mask, stage = batch_data(x)
    E = edge_index.size(1)
    h = numerical.flatten(1)
    for key in _seq_keys(P, 'shared_net.numerical_feature_encoder.'):
        h = torch.tanh(F.linear(h, P[key + '.weight'], P[key + '.bias']))
    h_numerical = h
    We, be = P['shared_net.node_encoder.weight'], P['shared_net.node_encoder.bias']
    Fd = node_features.size(-1)
    x1 = torch.gather(node_features, 1, edge_index[:, :, 0].unsqueeze(-1).expand(-1, -1, Fd))
    x2 = torch.gather(node_features, 1, edge_index[:, :, 1].unsqueeze(-1).expand(-1, -1, Fd))
    second = torch.eq(torch.argmax(x2[:, :, :MLP_TYPE_COLS], dim=-1), MLP_FEASIBLE)
    edge_features = torch.where(second.unsqueeze(-1).expand_as(x2), x2, x1)
    edge_features = torch.where(edge_mask.unsqueeze(-1).expand_as(edge_features), edge_features,
                                torch.zeros_like(edge_features))
    h_nodes = F.linear(node_features, We, be)
    h_edges = F.linear(edge_features, We, be)
    h_cur = F.linear(cur.unsqueeze(1), We, be)
    h_edges_mean = mean_features(h_edges, edge_mask)
    h_nodes_mean = mean_features(h_nodes, node_mask)
    state_value = torch.cat([h_numerical, h_nodes_mean, h_edges_mean, stage], dim=1)
    h_cur_rep = h_cur.repeat(1, E, 1)
    state_policy_land_use = torch.cat([h_edges, h_cur_rep, h_edges * h_cur_rep, h_edges - h_cur_rep], dim=-1)
    if keep is not None:
        keep.update(h_nodes_0=h_nodes, h_edges_0=h_edges, h_numerical=h_numerical, h_cur=h_cur.squeeze(1),
                    h_edges_mean=h_edges_mean, h_nodes_mean=h_nodes_mean, state_value=state_value)
    return state_policy_land_use, h_nodes, state_value, land_use_mask, road_mask, stage


def encoder_forward(P, x, num_heads=1, keep=None):
    """SGNNStateEncoder.forward, state_encoder.py:184-214.  ``x``: list[B] of list[9] tensors.

    ``keep`` (dict) receives named intermediates for stage-by-stage parity checks.
    """
    if is_mlp_params(P):
        return mlp_encoder_forward(P, x, keep)
    numerical, node_features, edge_index, cur, node_mask, edge_mask, land_use_mask, road_mask, stage = batch_data(x)
    N = node_features.size(1)
    E = edge_index.size(1)
    h = numerical.flatten(1)
    for key in _seq_keys(P, 'shared_net.numerical_feature_encoder.'):
        h = torch.tanh(F.linear(h, P[key + '.weight'], P[key + '.bias']))
    h_numerical = h
    We, be = P['shared_net.node_encoder.weight'], P['shared_net.node_encoder.bias']
    h_nodes = F.linear(node_features, We, be)
    h_cur = F.linear(cur.unsqueeze(1), We, be)
    num_layers = 0
    while 'shared_net.edge_fc_layers.%d.linear_0.weight' % num_layers in P:
        num_layers += 1
    if keep is not None:
        keep['h_nodes_0'] = h_nodes
    for layer in range(num_layers):
        h_edges = gather_to_edges(P, layer, h_nodes, edge_index, edge_mask)
        h_new = scatter_to_nodes(h_edges, edge_index, edge_mask, N)
        h_nodes = h_nodes + h_new
        if keep is not None:
            keep['h_nodes_%d' % (layer + 1)] = h_nodes
            keep['h_edges_%d' % (layer + 1)] = h_edges
    h_edges_mean = mean_features(h_edges, edge_mask)
    h_nodes_mean = mean_features(h_nodes, node_mask)
    h_att = self_attention(P, h_cur, h_nodes, node_mask, num_heads)
    state_value = torch.cat([h_numerical, h_nodes_mean, h_edges_mean, h_att, stage], dim=1)
    h_cur_rep = h_cur.repeat(1, E, 1)
    state_policy_land_use = torch.cat([h_edges, h_cur_rep, h_edges * h_cur_rep, h_edges - h_cur_rep], dim=-1)
    state_policy_road = h_nodes
    if keep is not None:
        keep.update(h_numerical=h_numerical, h_cur=h_cur.squeeze(1), h_edges_mean=h_edges_mean,
                    h_nodes_mean=h_nodes_mean, h_att=h_att, state_value=state_value)
    return state_policy_land_use, state_policy_road, state_value, land_use_mask, road_mask, stage


# ----------------------------------------------------------------------------- heads

def _policy_head(P, prefix, name, z):
    """create_policy_head, policy.py:19-43: Linear(+bias) Tanh ... Linear(no bias) [Flatten]."""
    keys = _seq_keys(P, prefix, stem=name + '_linear_')
    for i, key in enumerate(keys):
        z = F.linear(z, P[key + '.weight'], P.get(key + '.bias'))
        if i < len(keys) - 1:
            z = torch.tanh(z)
        elif z.size(-1) == 1:
            z = z.flatten(1)
    return z


def policy_forward(P, x, num_heads=1, keep=None):
    """UrbanPlanningPolicy.forward, policy.py:45-65."""
    s_land, s_road, _, land_use_mask, road_mask, stage = encoder_forward(P, x, num_heads, keep)
    land_dist = road_dist = None
    if stage[:, 0].sum() > 0:
        sel = stage[:, 0].bool()
        logits = _policy_head(P, 'policy_land_use_head.', 'land_use', s_land[sel])
        pad = torch.ones_like(land_use_mask[sel], dtype=torch.float32) * PAD_LOGIT
        masked = torch.where(land_use_mask[sel], logits, pad)
        land_dist = torch.distributions.Categorical(logits=masked)
        if keep is not None:
            keep['land_logits'] = masked
    if stage[:, 1].sum() > 0:
        sel = stage[:, 1].bool()
        logits = _policy_head(P, 'policy_road_head.', 'road', s_road[sel])
        pad = torch.ones_like(road_mask[sel], dtype=torch.float32) * PAD_LOGIT
        masked = torch.where(road_mask[sel], logits, pad)
        road_dist = torch.distributions.Categorical(logits=masked)
        if keep is not None:
            keep['road_logits'] = masked
    return land_dist, road_dist, stage


def get_log_prob_entropy(P, x, action, num_heads=1, keep=None):
    """policy.py:87-104 -> (f32[B,1], f32[B,1])."""
    land_dist, road_dist, stage = policy_forward(P, x, num_heads, keep)
    B = stage.shape[0]
    log_prob = torch.zeros(B)
    entropy = torch.zeros(B)
    if land_dist is not None:
        sel = stage[:, 0].bool()
        log_prob[sel] = land_dist.log_prob(action[sel, 0])
        entropy[sel] = land_dist.entropy()
    if road_dist is not None:
        sel = stage[:, 1].bool()
        log_prob[sel] = road_dist.log_prob(action[sel, 1])
        entropy[sel] = road_dist.entropy()
    return log_prob.unsqueeze(1), entropy.unsqueeze(1)


def value_forward(P, x, num_heads=1, keep=None):
    """UrbanPlanningValue.forward, value.py:36-39 (head :15-34)."""
    _, _, state_value, _, _, _ = encoder_forward(P, x, num_heads, keep)
    keys = _seq_keys(P, 'value_head.')
    z = state_value
    for i, key in enumerate(keys):
        z = F.linear(z, P[key + '.weight'], P[key + '.bias'])
        if i < len(keys) - 1:
            z = torch.tanh(z)
    return z


# ----------------------------------------------------------------------------- RL math

def estimate_advantages(rewards, masks, values, gamma, tau):
    """khrylib/rl/core/common.py:5-26 (no normalisation).  1-D f32 rewards/masks, values f32[T,1]."""
    T = rewards.size(0)
    deltas = torch.zeros(T, 1)
    advantages = torch.zeros(T, 1)
    prev_value = 0
    prev_advantage = 0
    for i in reversed(range(T)):
        deltas[i] = rewards[i] + gamma * prev_value * masks[i] - values[i]
        advantages[i] = deltas[i] + gamma * tau * prev_advantage * masks[i]
        prev_value = values[i, 0]
        prev_advantage = advantages[i, 0]
    returns = values + advantages
    return advantages, returns


def ppo_losses(P, states_b, actions_b, advantages_b, returns_b, fixed_log_probs_b, exps_b, clip_epsilon,
               value_pred_coef, entropy_coef, num_heads=1):
    """One minibatch's loss terms: urban_planning_agent.py:326-333, 363-371; agent_pg.py:19-23.

    The encoder is evaluated twice (value_net then policy_net), as the reference does.
    """
    ind = exps_b.nonzero(as_tuple=False).squeeze(1)
    values_pred = value_forward(P, states_b, num_heads)
    value_loss = (values_pred - returns_b).pow(2).mean()
    log_probs, entropy = get_log_prob_entropy(P, states_b, actions_b, num_heads)
    ratio = torch.exp(log_probs[ind] - fixed_log_probs_b[ind])
    adv = advantages_b[ind]
    surr1 = ratio * adv
    surr2 = torch.clamp(ratio, 1.0 - clip_epsilon, 1.0 + clip_epsilon) * adv
    surr_loss = -torch.min(surr1, surr2).mean()
    entropy_loss = -entropy[ind].mean()
    loss = surr_loss + value_pred_coef * value_loss + entropy_coef * entropy_loss
    return loss, value_loss, surr_loss, entropy_loss


POLICY_PREFIXES = ('shared_net.', 'policy_land_use_head.', 'policy_road_head.')
VALUE_PREFIXES = ('shared_net.', 'value_head.')


class OracleUpdater:
    """update_params / update_policy restated (urban_planning_agent.py:248-361) on oracle params.

    Reproduces: value pre-pass, GAE, old-logp pre-pass, cumulative numpy-RNG permutations,
    floor(T/B) minibatches, first-step-only double gradient clipping (urban_planning_agent.py:46 +
    agent_ppo.py:43-46: the clip lists are generators, exhausted by the first call), torch Adam.
    """

    def __init__(self, P, lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=1.0, tau=0.0, clip_epsilon=0.2,
                 value_pred_coef=0.5, entropy_coef=0.01, num_optim_epoch=4, mini_batch_size=256, num_heads=1,
                 batch_stage=False, legacy_zero_grad=False):
        self.P = P
        # cfg.agent_specs['batch_stage'] (urban_planning_agent.py:314-319)
        self.batch_stage = batch_stage
        # the reference pins torch <= 1.13 (requirements.txt:3) whose Optimizer.zero_grad() zero-fills; torch >= 2.0
        # sets the gradients to None.  They differ for a head that gets no row in a minibatch (Adam skips a None
        # gradient but steps on a zero one).  False = the installed torch's semantics (what the goldens' `upd/*` hold)
        self.legacy_zero_grad = legacy_zero_grad
        self.names = list(P.keys())
        self.optimizer = torch.optim.Adam([P[k] for k in self.names], lr=lr, eps=eps, weight_decay=weight_decay)
        self.gamma, self.tau = gamma, tau
        self.clip_epsilon = clip_epsilon
        self.value_pred_coef, self.entropy_coef = value_pred_coef, entropy_coef
        self.num_optim_epoch, self.mini_batch_size = num_optim_epoch, mini_batch_size
        self.num_heads = num_heads
        # generators, like [(policy_net.parameters(), 1), (value_net.parameters(), 1)]
        self.policy_grad_clip = [
            ((P[k] for k in self.names if k.startswith(POLICY_PREFIXES)), 1),
            ((P[k] for k in self.names if k.startswith(VALUE_PREFIXES)), 1)]
        self.loss_log = []          # (loss, value_loss, surr_loss, entropy_loss) per minibatch

    def clip_policy_grad(self):
        for params, max_norm in self.policy_grad_clip:
            torch.nn.utils.clip_grad_norm_(params, max_norm)

    def step(self, states_b, actions_b, advantages_b, returns_b, fixed_log_probs_b, exps_b):
        loss, vl, sl, el = ppo_losses(self.P, tensorfy(states_b), actions_b, advantages_b, returns_b,
                                      fixed_log_probs_b, exps_b, self.clip_epsilon, self.value_pred_coef,
                                      self.entropy_coef, self.num_heads)
        self.optimizer.zero_grad(set_to_none=not self.legacy_zero_grad)
        loss.backward()
        self.clip_policy_grad()
        self.optimizer.step()
        self.loss_log.append((loss.item(), vl.item(), sl.item(), el.item()))

    def prepass(self, states, actions):
        chunk = self.mini_batch_size
        values, logps = [], []
        with torch.no_grad():
            for i in range(0, len(states), chunk):
                xs = tensorfy(states[i:i + chunk])
                values.append(value_forward(self.P, xs, self.num_heads))
                logps.append(get_log_prob_entropy(self.P, xs, actions[i:i + chunk], self.num_heads)[0])
        return torch.cat(values), torch.cat(logps)

    def update_params(self, batch, max_steps=None):
        states = batch.states
        actions = torch.from_numpy(np.asarray(batch.actions)).float()
        rewards = torch.from_numpy(np.asarray(batch.rewards)).float()
        masks = torch.from_numpy(np.asarray(batch.masks)).float()
        exps = torch.from_numpy(np.asarray(batch.exps)).float()
        values, fixed_log_probs = self.prepass(states, actions)
        advantages, returns = estimate_advantages(rewards, masks, values, self.gamma, self.tau)
        num_state = len(states)
        steps = 0
        for _ in range(self.num_optim_epoch):
            perm_np = np.arange(num_state)
            np.random.shuffle(perm_np)
            perm = torch.from_numpy(perm_np).long()
            states = [states[i] for i in perm_np]
            actions, returns, advantages, fixed_log_probs, exps = \
                actions[perm].clone(), returns[perm].clone(), advantages[perm].clone(), \
                fixed_log_probs[perm].clone(), exps[perm].clone()
            if self.batch_stage:
                # get_perm_batch_stage (urban_planning_agent.py:273-279): land-use rows first, then road rows, each in
                # the shuffled order; a row of any other stage raises IndexError there, and here
                inds = [[], []]
                for i, x in enumerate(states):
                    inds[int(np.argmax(np.asarray(x[-1])))].append(i)
                ps_np = np.array(inds[0] + inds[1])
                ps = torch.from_numpy(ps_np).long()
                states = [states[i] for i in ps_np]
                actions, returns, advantages, fixed_log_probs, exps = \
                    actions[ps].clone(), returns[ps].clone(), advantages[ps].clone(), \
                    fixed_log_probs[ps].clone(), exps[ps].clone()
            for i in range(int(math.floor(num_state / self.mini_batch_size))):
                ind = slice(i * self.mini_batch_size, min((i + 1) * self.mini_batch_size, num_state))
                self.step(states[ind], actions[ind], advantages[ind], returns[ind], fixed_log_probs[ind], exps[ind])
                steps += 1
                if max_steps is not None and steps >= max_steps:
                    return values, fixed_log_probs, advantages, returns
        return values, fixed_log_probs, advantages, returns
