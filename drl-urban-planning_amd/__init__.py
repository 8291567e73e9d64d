"""MI355X-native SGNN policy/value + PPO-update hot path (drop-in for the reference's
``urban_planning.models.model.create_sgnn_model`` and ``UrbanPlanningAgent.update_params``).

Sub-modules are imported lazily: ``synth`` (synthetic replay), ``native`` (ctypes binding of
the C-ABI library ``csrc/libupamd.so``), ``packer`` (ragged/CSR replay packer), ``models``
(nn.Module surface), ``agent`` (PPO update surface), ``dist`` (data-parallel helpers), ``binding`` (the reference-side
binding: ``bind_reference_agent``), ``launch`` (``python -m drl_urban_planning_amd.launch -m urban_planning.train ...``).
"""
__version__ = '0.1.0'


_LAZY = {
    'create_sgnn_model': ('models', 'create_sgnn_model'),
    'create_mlp_model': ('models', 'create_mlp_model'),
    'ActorCritic': ('models', 'ActorCritic'),
    'PPOUpdater': ('agent', 'PPOUpdater'),
    'HipUpdateMixin': ('agent', 'HipUpdateMixin'),
    'install': ('agent', 'install'),
    'bind_reference_agent': ('binding', 'bind_reference_agent'),
    'patch_reference_module': ('binding', 'patch_reference_module'),
    'DistContext': ('dist', 'DistContext'),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module('drl_urban_planning_amd.' + mod), attr)
    raise AttributeError(name)
