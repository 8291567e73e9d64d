"""ORACLE tooling -- import the REAL reference modules (build container only).

/root/reference is present only in the build container (never on the GPU box), and its
geometry stack (geopandas/shapely/momepy/libpysal/...) is not installed, so the hot-path
modules are imported with those names stubbed (SURVEY.md section 8c recipe).  Used by
``tests/golden/make_golden.py`` to generate the committed golden vectors and by
``tests/test_reference_live.py`` (skipped where /root/reference is absent).
"""
import os
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get('UPAMD_REFERENCE_ROOT', '/root/reference')

_STUBS = ['geopandas', 'shapely', 'shapely.geometry', 'shapely.ops', 'momepy', 'libpysal', 'absl', 'absl.app',
          'absl.flags', 'setproctitle', 'pygad', 'torch.utils.tensorboard', 'osmnx']


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'urban_planning'))


def load_reference():
    """Returns a namespace with the reference's hot-path symbols."""
    if not available():
        raise RuntimeError('reference tree not found at %s' % REFERENCE_ROOT)
    for m in _STUBS:
        if m not in sys.modules:
            sys.modules[m] = MagicMock()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torch  # noqa: F401
    from urban_planning.models.model import create_sgnn_model, create_mlp_model, ActorCritic
    from khrylib.rl.core import estimate_advantages
    from khrylib.rl.agents import AgentPPO
    from urban_planning.agents.urban_planning_agent import UrbanPlanningAgent, tensorfy
    ns = types.SimpleNamespace(create_sgnn_model=create_sgnn_model, create_mlp_model=create_mlp_model, ActorCritic=ActorCritic,
                               estimate_advantages=estimate_advantages, AgentPPO=AgentPPO,
                               UrbanPlanningAgent=UrbanPlanningAgent, tensorfy=tensorfy)
    return ns


class DuckCfg:
    """The three spec dicts ``create_sgnn_model`` reads (model.py:8-19; hlg.yaml:21-33)."""

    def __init__(self, D=16, L=2, K=1, S=(64, 16), heads=1, max_nodes=1000, max_edges=3000,
                 land_head=(32, 1), road_head=(32, 1), value_head=(32, 32, 1)):
        self.state_encoder_specs = dict(state_encoder_hidden_size=list(S), gcn_node_dim=D, num_gcn_layers=L,
                                        num_edge_fc_layers=K, max_num_nodes=max_nodes, max_num_edges=max_edges,
                                        num_attention_heads=heads)
        self.policy_specs = dict(policy_land_use_head_hidden_size=list(land_head),
                                 policy_road_head_hidden_size=list(road_head))
        self.value_specs = dict(value_head_hidden_size=list(value_head))
        self.agent_specs = {}


class DuckAgent:
    """Fields the model constructors read (state_encoder.py:19,46; policy.py:50,70)."""
    node_dim = 23
    numerical_feature_size = 52

    def __init__(self):
        import torch
        self.dtype = torch.float32


class ScalarLog:
    """Stub tb_logger capturing add_scalar calls (urban_planning_agent.py:342-361)."""

    def __init__(self):
        self.scalars = []

    def add_scalar(self, tag, value, step):
        self.scalars.append((tag, float(value), int(step)))

    def flush(self):
        pass


def make_reference_agent(ref, cfg, policy_net, value_net, lr=4e-4, eps=1e-5, weight_decay=0.0, gamma=1.0,
                         tau=0.0, clip_epsilon=0.2, value_pred_coef=0.5, entropy_coef=0.01, num_optim_epoch=4,
                         mini_batch_size=256, batch_stage=False, legacy_zero_grad=False):
    """UrbanPlanningAgent without env/logger setup (object.__new__ + the fields __init__ would set,
    urban_planning_agent.py:28-47, 145-151)."""
    import torch
    ag = object.__new__(ref.UrbanPlanningAgent)
    cfg.mini_batch_size = mini_batch_size
    if batch_stage:
        cfg.agent_specs = dict(cfg.agent_specs, batch_stage=True)      # urban_planning_agent.py:314
    ag.cfg = cfg
    ag.training = True
    ag.device = torch.device('cpu')
    ag.dtype = torch.float32
    ag.loss_iter = 0
    ag.tb_logger = ScalarLog()
    ag.policy_net, ag.value_net = policy_net, value_net
    ag.actor_critic_net = ref.ActorCritic(policy_net, value_net)
    ag.optimizer = torch.optim.Adam(ag.actor_critic_net.parameters(), lr=lr, eps=eps, weight_decay=weight_decay)
    ag.update_modules = [policy_net, value_net]
    ag.sample_modules = [policy_net]
    ag.gamma, ag.tau = gamma, tau
    ag.clip_epsilon = clip_epsilon
    ag.value_pred_coef, ag.entropy_coef = value_pred_coef, entropy_coef
    ag.opt_num_epochs = num_optim_epoch
    ag.mini_batch_size = mini_batch_size
    ag.policy_grad_clip = [(policy_net.parameters(), 1), (value_net.parameters(), 1)]
    if legacy_zero_grad:
        # the reference pins torch <= 1.13 (requirements.txt:3), whose Optimizer.zero_grad() zero-fills the gradients;
        # the installed torch sets them to None.  Re-create the pinned behaviour on the reference's own optimizer
        import functools
        ag.optimizer.zero_grad = functools.partial(ag.optimizer.zero_grad, set_to_none=False)
    return ag
