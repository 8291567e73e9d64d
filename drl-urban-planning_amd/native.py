"""ctypes binding of the C-ABI library ``csrc/libupamd.so`` (declared in ``include/upamd.h``).

The library is the product: there is NO Python/PyTorch fallback for the hot path.  ``lib()``
raises ``RuntimeError`` when the shared object is missing or does not export the ABI the header
declares; ``build()`` compiles it in-tree with hipcc for gfx950.
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# (UPAMD_LIB_PATH: kernel-lab A/B runs load another build of the same ABI, e.g. tools/lab/libupamd_base.so)
LIB_PATH = os.environ.get('UPAMD_LIB_PATH') or os.path.join(CSRC, 'libupamd.so')
ABI_VERSION = 7
MAX_MLP = 4
MAX_EDGE_FC = 4
META_STRIDE = 16
NODE_PAD = 24
E_REPLAN = -5


class ModelDesc(C.Structure):
    _fields_ = [('node_dim', C.c_int32), ('numerical_dim', C.c_int32), ('D', C.c_int32), ('L', C.c_int32),
                ('heads', C.c_int32),
                ('n_num', C.c_int32), ('num_hidden', C.c_int32 * MAX_MLP),
                ('n_land', C.c_int32), ('land_hidden', C.c_int32 * MAX_MLP),
                ('n_road', C.c_int32), ('road_hidden', C.c_int32 * MAX_MLP),
                ('n_value', C.c_int32), ('value_hidden', C.c_int32 * MAX_MLP), ('encoder', C.c_int32),
                ('edge_fc_layers', C.c_int32)]


class PackLayout(C.Structure):
    _fields_ = [('T', C.c_int64), ('total_nodes', C.c_int64), ('total_edges', C.c_int64), ('total_he', C.c_int64),
                ('total_rn', C.c_int64), ('node_dim', C.c_int32), ('numerical_dim', C.c_int32),
                ('off_meta', C.c_int64), ('off_x', C.c_int64), ('off_nmask', C.c_int64), ('off_rowptr', C.c_int64),
                ('off_inc_nbr', C.c_int64), ('off_he_src', C.c_int64),
                ('off_he_dst', C.c_int64), ('off_he_live', C.c_int64), ('off_he_slot', C.c_int64),
                ('off_rn_node', C.c_int64), ('off_numerical', C.c_int64), ('off_cur', C.c_int64),
                ('off_order', C.c_int64), ('off_hinc_ptr', C.c_int64), ('off_hinc_nbr', C.c_int64),
                ('off_hinc_he', C.c_int64), ('off_he_sel', C.c_int64), ('off_xbar', C.c_int64), ('total_bytes', C.c_int64)]


class Minibatch(C.Structure):
    _fields_ = [('B', C.c_int32), ('n_nodes', C.c_int64), ('n_he', C.c_int64), ('n_rn', C.c_int64),
                ('max_n', C.c_int32), ('max_inc', C.c_int32), ('idx_dev', C.c_void_p), ('node_off_dev', C.c_void_p),
                ('he_off_dev', C.c_void_p), ('rn_off_dev', C.c_void_p),
                ('n_inc', C.c_int64), ('inc_off_dev', C.c_void_p), ('max_cand', C.c_int32)]


# every symbol include/upamd.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    'upamd_abi_version': (C.c_int, []),
    'upamd_last_error': (C.c_char_p, []),
    'upamd_param_count': (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'upamd_param_info': (C.c_int, [C.POINTER(ModelDesc), C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'upamd_param_groups': (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'upamd_pack_plan': (C.c_int, [C.c_int64, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(PackLayout)]),
    'upamd_pack_fill': (C.c_int, [C.c_int64, _P, _P, C.POINTER(PackLayout), C.c_int32, _P]),
    'upamd_record_table': (C.c_int, [C.c_int64, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    'upamd_pack_plan_ex': (C.c_int, [C.c_int64, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(PackLayout)]),
    'upamd_pack_fill_range': (C.c_int, [C.c_int64, _P, _P, C.POINTER(PackLayout), C.c_int64, C.c_int64, C.c_int32, _P]),
    'upamd_engine_create': (C.c_int, [C.POINTER(ModelDesc), C.POINTER(_P)]),
    'upamd_engine_destroy': (None, [_P]),
    'upamd_workspace_bytes': (C.c_int, [_P, C.POINTER(Minibatch), C.c_int32, C.POINTER(C.c_int64)]),
    'upamd_forward': (C.c_int, [_P, _P, C.POINTER(PackLayout), C.POINTER(Minibatch), _P, _P, C.c_int64, _P, _P, _P,
                                C.c_int32, _P]),
    'upamd_backward': (C.c_int, [_P, _P, C.POINTER(PackLayout), C.POINTER(Minibatch), _P, _P, C.c_int64, _P, _P, _P,
                                 _P, _P]),
    'upamd_select_actions': (C.c_int, [_P, C.POINTER(PackLayout), C.POINTER(Minibatch), _P, _P, _P, _P, _P, _P]),
    'upamd_grad_buckets': (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'upamd_grad_bucket_wait': (C.c_int, [_P, _P, C.c_int32, _P]),
    'upamd_step_fused_ok': (C.c_int, [_P, C.POINTER(Minibatch)]),
    'upamd_step_fused': (C.c_int, [_P, _P, C.POINTER(PackLayout), C.POINTER(Minibatch), _P, _P, C.c_int64, _P, _P, _P, _P, _P,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P]),
    'upamd_ws_tensor': (C.c_int, [_P, C.POINTER(Minibatch), C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'upamd_ppo_loss': (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_float, _P, _P, _P, _P, _P]),
    'upamd_ppo_loss_rows': (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_float, _P, _P, _P, _P, _P, C.c_int64, _P]),
    'upamd_gae': (C.c_int, [C.c_int64, _P, _P, _P, C.c_double, C.c_double, _P, _P, _P]),
    'upamd_clip_first_step': (C.c_int, [C.POINTER(ModelDesc), _P, C.c_float, _P, _P]),
    'upamd_adam_step': (C.c_int, [C.c_int64, C.c_int64, _P, _P, _P, _P, C.c_int32, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, _P]),
    'upamd_adam_groups': (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_double, _P, _P, _P]),
    'upamd_gemm_nt': (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int64, C.c_int32, _P, C.c_int32, C.c_int64, _P, _P, _P, C.c_int64,
                                C.c_int32, C.c_int32, C.c_float, _P]),
    'upamd_gemm_nt_split_scratch_bytes': (C.c_int64, [C.c_int32, C.c_int32]),
    'upamd_gemm_nt_split': (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int32, C.c_int64, _P, _P, _P, C.c_int32, C.c_float,
                                      C.c_int32, _P, _P]),
    'upamd_gemm_tn_scratch_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int64]),
    'upamd_gemm_tn': (C.c_int, [_P, C.c_int32, C.c_int64, _P, C.c_int32, C.c_int64, C.c_int64, C.c_int32, _P, _P, _P]),
    'upamd_tune': (C.c_int, [C.c_char_p, C.c_int32]),
    'upamd_clock_probe': (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    'upamd_tiny_profile': (C.c_int, [_P]),
    'upamd_profile_enable': (C.c_int, [_P, C.c_int32]),
    'upamd_profile_read': (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'upamd_profile_reset': (C.c_int, [_P]),
}

_lib = None


def build(verbose=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ['make', '-C', CSRC, '-j', str(os.cpu_count() or 4)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('building libupamd.so failed:\n' + res.stdout[-4000:] + res.stderr[-8000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def lib():
    """The loaded library; raises loudly if it is missing or stale (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # not built yet: compile it in-tree (hipcc is part of the image); there is no Python fallback to run instead
        try:
            build()
        except Exception as exc:
            raise RuntimeError('native library %s not found and building it failed (%s). Run `python -c "import '
                               '__graft_entry__ as g; g.build()"` (or `make -C %s`). The HIP path has no Python '
                               'fallback.' % (LIB_PATH, exc, CSRC))
    # torch first: its wheel bundles its own HIP runtime; if libupamd.so were loaded before it, the process would end up
    # with /opt/rocm's runtime for this library and torch's for the tensors ("no ROCm-capable device" at first launch)
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise RuntimeError('%s does not export %s (stale build?)' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if handle.upamd_abi_version() != ABI_VERSION:
        raise RuntimeError('libupamd.so ABI version %d != binding version %d' % (handle.upamd_abi_version(), ABI_VERSION))
    _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().upamd_last_error()
        raise RuntimeError('%s failed (%d): %s' % (what or 'native call', rc, msg.decode() if msg else ''))


# ---- kernel-lab knobs: process-wide in the library (upamd_tune), per ENGINE here.
# The library reads its knobs when a call is ENQUEUED (launch configuration, which kernel, which stream), so an engine that wants
# its own settings -- a learner next to an action server on another model in one process -- gets them by having its overrides
# applied around its own native calls and the process defaults put back behind them, under one process-wide lock (only the enqueue
# is serialised; the kernels of two engines still overlap on their streams).  `tune()` sets a process default (what engines
# without an override of that knob see); `tuned(overrides)` is the bracket NativeEngine puts around its calls.
TUNE_DEFAULTS = {'gemm_nt_dma': 1, 'gemm_split': 0, 'fold_layer1': 1, 'he_fused': 1, 'side_stream': 1, 'fwd_h_hbm': 1, 'fe_half': 1,
                 'pq_exp': 1, 'bwd_nb_global': 1, 'nt_min_wgs': 128, 'tiny_fused': 1, 'tiny_threads': 1024, 'side_heads': 1,
                 'side_wgrad': 1, 'side_priority': 1, 'grad_buckets': 1, 'gemm_lds_pad': 12 * 1024, 'gemm_stagger_mode': 1,
                 'gemm_stagger_cycles': 37000}      # the library's built-in defaults (include/upamd.h, the block above upamd_tune)
_tune_lock = threading.RLock()
_tune_depth = threading.local()
_tune_process = {}            # knob -> process default set through tune()
_tune_isolation = False       # some engine of this process has (had) overrides: EVERY engine's native calls take the lock from then on
                              # (an engine without overrides must not enqueue while another engine's temporary values are applied)


def note_engine_overrides():
    global _tune_isolation
    _tune_isolation = True


def tune(name, value):
    """Set a PROCESS default of a kernel-lab knob (engines created before or after see it unless they override that knob)."""
    if name not in TUNE_DEFAULTS:
        raise KeyError('unknown kernel-lab knob %r' % (name,))
    with _tune_lock:
        check(lib().upamd_tune(name.encode(), int(value)), 'upamd_tune')
        _tune_process[name] = int(value)


class tuned:
    """``with native.tuned({'knob': value, ...}):`` -- the overrides hold for the native calls made inside the block (by this
    thread; other threads' brackets wait at the lock), the process defaults are restored on the way out.  Empty overrides cost
    nothing until some engine of the process sets an override (``note_engine_overrides``); from then on every bracket takes the lock."""

    def __init__(self, overrides):
        self.overrides = overrides

    def __enter__(self):
        self.locked = bool(self.overrides) or _tune_isolation
        if self.locked:
            _tune_lock.acquire()
        if self.overrides:
            _tune_depth.n = getattr(_tune_depth, 'n', 0) + 1
            if _tune_depth.n == 1:              # (an engine's methods call each other: only the outermost bracket applies / restores)
                L = lib()
                for name, value in self.overrides.items():
                    check(L.upamd_tune(name.encode(), int(value)), 'upamd_tune')
        return self

    def __exit__(self, *exc):
        try:
            if self.overrides:
                _tune_depth.n -= 1
                if _tune_depth.n == 0:
                    L = lib()
                    for name in self.overrides:
                        L.upamd_tune(name.encode(), int(_tune_process.get(name, TUNE_DEFAULTS[name])))
        finally:
            if self.locked:
                _tune_lock.release()
        return False


ENCODER_SGNN, ENCODER_MLP = 0, 1


def make_desc(state_encoder_specs, policy_specs, value_specs, node_dim, numerical_dim, encoder=ENCODER_SGNN):
    """Model description from the reference's three spec dicts (hlg.yaml:21-33)."""
    n_fc = int(state_encoder_specs.get('num_edge_fc_layers', 1)) if encoder == ENCODER_SGNN else 1
    if not 1 <= n_fc <= MAX_EDGE_FC:
        raise NotImplementedError('num_edge_fc_layers must be between 1 and %d (got %d)' % (MAX_EDGE_FC, n_fc))
    d = ModelDesc()
    d.edge_fc_layers = n_fc
    d.node_dim, d.numerical_dim = int(node_dim), int(numerical_dim)
    d.encoder = int(encoder)
    d.D = int(state_encoder_specs['gcn_node_dim'])
    d.L = int(state_encoder_specs['num_gcn_layers']) if encoder == ENCODER_SGNN else 0
    d.heads = int(state_encoder_specs['num_attention_heads']) if encoder == ENCODER_SGNN else 1

    def fill(n_field, arr_field, values):
        values = [int(v) for v in values]
        if len(values) > MAX_MLP:
            raise NotImplementedError('MLPs deeper than %d layers are not supported' % MAX_MLP)
        setattr(d, n_field, len(values))
        arr = getattr(d, arr_field)
        for i, v in enumerate(values):
            arr[i] = v
    fill('n_num', 'num_hidden', state_encoder_specs['state_encoder_hidden_size'])
    fill('n_land', 'land_hidden', policy_specs['policy_land_use_head_hidden_size'])
    fill('n_road', 'road_hidden', policy_specs['policy_road_head_hidden_size'])
    fill('n_value', 'value_hidden', value_specs['value_head_hidden_size'])
    return d


def param_table(desc):
    """[(name, offset, rows, cols, group)], n_floats, group ranges."""
    L = lib()
    n_floats, n_tensors = C.c_int64(), C.c_int32()
    check(L.upamd_param_count(C.byref(desc), C.byref(n_floats), C.byref(n_tensors)), 'upamd_param_count')
    out = []
    name = C.create_string_buffer(160)
    off, rows, cols, grp = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
    for i in range(n_tensors.value):
        check(L.upamd_param_info(C.byref(desc), i, name, 160, C.byref(off), C.byref(rows), C.byref(cols), C.byref(grp)),
              'upamd_param_info')
        out.append((name.value.decode(), off.value, rows.value, cols.value, grp.value))
    gb, ge = (C.c_int64 * 3)(), (C.c_int64 * 3)()
    check(L.upamd_param_groups(C.byref(desc), gb, ge), 'upamd_param_groups')
    return out, n_floats.value, [(gb[i], ge[i]) for i in range(3)]
