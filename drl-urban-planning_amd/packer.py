"""Replay packer: list of padded 9-field states -> one ragged/CSR buffer (host C++ packer in
``csrc/packer.cpp``), uploaded to HBM in a single copy.

Replaces the reference's per-minibatch ``tensorfy`` (9*B tiny H2D copies,
urban_planning/agents/urban_planning_agent.py:16-20) and ``batch_data`` (9 stacks,
urban_planning/models/state_encoder.py:163-177).  Input format = the wire format of
``ObservationExtractor.get_obs`` (urban_planning/envs/observation_extractor.py:207-228).
"""
import ctypes as C
import importlib.util
import os

import numpy as np
import torch

from . import native

_FIELD_DTYPES = [np.float32, np.float32, np.int64, np.float32, np.bool_, np.bool_, np.bool_, np.bool_, np.float32]

# meta columns (include/upamd.h)
M_N, M_E, M_NH, M_NR, M_STAGE, M_ACT, M_NMASK, M_PADN, M_PADE, M_NODE_OFF, M_EDGE_OFF, M_HE_OFF, M_RN_OFF = range(13)


class PackedReplay:
    """Host + device form of one PPO iteration's replay."""

    def __init__(self, meta, layout, host_buf):
        self.meta = meta                 # np.int32 [T, 16]
        self.layout = layout             # native.PackLayout
        self.host_buf = host_buf         # torch.uint8 [total_bytes] (pinned when CUDA is available)
        self.dev_buf = None
        self.T = int(meta.shape[0])

    def to(self, device):
        self.dev_buf = self.host_buf.to(device, non_blocking=True)
        return self

    def section(self, name, dtype, count):
        """numpy view of a section of the host buffer (tests / debugging)."""
        off = getattr(self.layout, 'off_' + name)
        raw = self.host_buf.numpy()
        return raw[off:off + count * np.dtype(dtype).itemsize].view(dtype)


_host_mod = False      # False = not looked up yet, None = unavailable


def _host_helper():
    """The optional CPython helper csrc/_upamd_host.so (pointer-table extraction in C); None when it is not built."""
    global _host_mod
    if _host_mod is False:
        _host_mod = None
        path = os.path.join(native.CSRC, '_upamd_host.so')
        if os.path.exists(path):
            try:
                spec = importlib.util.spec_from_file_location('_upamd_host', path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _host_mod = mod
            except Exception:           # glue only: the Python loop below does the same job
                _host_mod = None
    return _host_mod


def _as_array(x, dtype):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    if not isinstance(x, np.ndarray) or x.dtype != dtype or not x.flags['C_CONTIGUOUS']:
        x = np.ascontiguousarray(x, dtype=dtype)
    return x


def pack_replay(states, actions, node_dim, numerical_dim, n_threads=0, pin=None, reuse=None):
    """states: list[T] of list[9] arrays (or tensors); actions: f32[T,2] (padded-slot indices).
    ``reuse``: optional dict owned by the caller; its pinned staging buffer is recycled across iterations
    (pinning ~1 GB per PPO iteration is otherwise a measurable part of the set-up time)."""
    T = len(states)
    if T == 0:
        raise ValueError('empty replay')
    L = native.lib()
    ptrs = np.empty((9, T), dtype=np.uint64)
    pad_n = np.empty(T, dtype=np.int32)
    pad_e = np.empty(T, dtype=np.int32)
    keep = []
    first_slow = 0
    helper = _host_helper()
    if helper is not None:
        # fast path: every state whose fields already are C-contiguous arrays of the wire dtypes is handled in C
        bad = helper.addr_table(states, ptrs, pad_n, pad_e, int(node_dim))
        first_slow = T if bad < 0 else 0          # anything unusual: redo the whole table the slow, validating way
    for t, s in enumerate(states if first_slow < T else ()):
        if len(s) != 9:
            raise ValueError('state %d has %d fields, expected 9' % (t, len(s)))
        for f in range(9):
            a = s[f]
            if not (isinstance(a, np.ndarray) and a.dtype == _FIELD_DTYPES[f] and a.flags['C_CONTIGUOUS']):
                a = _as_array(a, _FIELD_DTYPES[f])
                keep.append(a)
            ptrs[f, t] = a.__array_interface__['data'][0]
        nf, ei = s[1], s[2]
        pad_n[t] = nf.shape[0]
        pad_e[t] = ei.shape[0]
        if nf.shape[-1] != node_dim:
            raise ValueError('state %d: node feature width %d != node_dim %d' % (t, nf.shape[-1], node_dim))
        if len(s[4]) != pad_n[t] or len(s[7]) != pad_n[t] or len(s[5]) != pad_e[t] or len(s[6]) != pad_e[t]:
            raise ValueError('state %d: mask lengths do not match the padded node/edge counts' % t)
    actions = _as_array(np.asarray(actions).reshape(T, 2), np.float32)
    meta = np.zeros((T, native.META_STRIDE), dtype=np.int32)
    layout = native.PackLayout()
    native.check(L.upamd_pack_plan(T, ptrs.ctypes.data, pad_n.ctypes.data, pad_e.ctypes.data, actions.ctypes.data,
                                   int(node_dim), int(numerical_dim), int(n_threads), meta.ctypes.data,
                                   C.byref(layout)), 'upamd_pack_plan')
    if pin is None:
        pin = torch.cuda.is_available()
    need = int(layout.total_bytes)
    host = None
    if reuse is not None:
        cached = reuse.get('host')
        if cached is not None and cached.numel() >= need and cached.is_pinned() == bool(pin):
            host = cached[:need]
    if host is None:
        host = torch.empty(need + (need >> 3 if reuse is not None else 0), dtype=torch.uint8, pin_memory=bool(pin))
        if reuse is not None:
            reuse['host'] = host
        host = host[:need]
    native.check(L.upamd_pack_fill(T, ptrs.ctypes.data, meta.ctypes.data, C.byref(layout), int(n_threads),
                                   host.data_ptr()), 'upamd_pack_fill')
    del keep
    return PackedReplay(meta, layout, host)


class Schedule:
    """Device-side index arrays of a sequence of minibatches (uploaded once per epoch).

    For minibatch k with rows ``rows_k`` (state ids): ``idx`` and the prefix sums of n / head-edge /
    road-node counts in minibatch order, plus the host-side totals the engine needs.
    """

    def __init__(self, packed, row_lists, device):
        meta = packed.meta
        chunks, self.items = [], []
        cursor = 0
        for rows in row_lists:
            rows = np.asarray(rows, dtype=np.int64)
            B = rows.size
            n = meta[rows, M_N].astype(np.int64)
            nh = meta[rows, M_NH].astype(np.int64)
            nr = meta[rows, M_NR].astype(np.int64)
            arr = np.zeros(B + 3 * (B + 1), dtype=np.int32)
            arr[:B] = rows
            arr[B + 1:2 * B + 1] = np.cumsum(n)
            arr[2 * B + 2:3 * B + 2] = np.cumsum(nh)
            arr[3 * B + 3:4 * B + 3] = np.cumsum(nr)
            pad = (-arr.size) % 64            # keep every minibatch's arrays 256-byte aligned
            chunks.append(np.concatenate([arr, np.zeros(pad, dtype=np.int32)]))
            self.items.append(dict(B=int(B), n_nodes=int(n.sum()), n_he=int(nh.sum()), n_rn=int(nr.sum()),
                                   max_n=int(n.max()), max_inc=int(2 * meta[rows, M_E].max()), base=cursor,
                                   n_land=int((meta[rows, M_STAGE] == 0).sum()),
                                   n_road=int((meta[rows, M_STAGE] == 1).sum())))
            cursor += arr.size + pad
        flat = np.concatenate(chunks)
        self.dev = torch.from_numpy(flat).to(device, non_blocking=False)

    def minibatch(self, k):
        it = self.items[k]
        B = it['B']
        base = self.dev.data_ptr() + 4 * it['base']
        mb = native.Minibatch()
        mb.B, mb.n_nodes, mb.n_he, mb.n_rn = B, it['n_nodes'], it['n_he'], it['n_rn']
        mb.max_n, mb.max_inc = it['max_n'], it['max_inc']
        mb.idx_dev = base
        mb.node_off_dev = base + 4 * B
        mb.he_off_dev = base + 4 * (2 * B + 1)
        mb.rn_off_dev = base + 4 * (3 * B + 2)
        return mb, it
