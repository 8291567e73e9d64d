"""Replay packer: list of padded 9-field states -> one ragged/CSR buffer (host C++ packer in
``csrc/packer.cpp``), uploaded to HBM in a single copy.

Replaces the reference's per-minibatch ``tensorfy`` (9*B tiny H2D copies,
urban_planning/agents/urban_planning_agent.py:16-20) and ``batch_data`` (9 stacks,
urban_planning/models/state_encoder.py:163-177).  Input format = the wire format of
``ObservationExtractor.get_obs`` (urban_planning/envs/observation_extractor.py:207-228).
"""
import ctypes as C
import threading
import importlib.util
import os

import numpy as np
import torch

from . import native

_FIELD_DTYPES = [np.float32, np.float32, np.int64, np.float32, np.bool_, np.bool_, np.bool_, np.bool_, np.float32]

# meta columns (include/upamd.h)
M_N, M_E, M_NH, M_NR, M_STAGE, M_ACT, M_NMASK, M_PADN, M_PADE, M_NODE_OFF, M_EDGE_OFF, M_HE_OFF, M_RN_OFF = range(13)


class NeedsExactPlan(RuntimeError):
    """``fill`` found a live edge beyond the extent a masks-only plan assumed: plan again with ``exact=True``."""


class PackedReplay:
    """Host + device form of one PPO iteration's replay.

    Every section of the packed buffer is state-major (include/upamd.h: upamd_pack_layout), so the states [t0, t1) are ONE
    contiguous byte range per section: ``fill(t0, t1)`` packs them on the host threads, ``upload(t0, t1)`` moves exactly
    those ranges to HBM asynchronously.  ``pack_replay`` fills everything at once; the PPO updater streams the replay chunk
    by chunk (fill k + 1 | H2D k | pre-pass forward k - 1, agent.PPOUpdater.prepare)."""

    def __init__(self, meta, layout, host_buf, ptrs=None, keep=None, n_threads=0):
        self.meta = meta                 # np.int32 [T, 16]
        self.layout = layout             # native.PackLayout
        self.host_buf = host_buf         # torch.uint8 [total_bytes] (pinned when CUDA is available)
        self.dev_buf = None
        self.T = int(meta.shape[0])
        self._ptrs, self._keep, self._threads = ptrs, keep, n_threads      # alive until the last state has been filled
        self.filled = 0 if ptrs is not None else self.T

    def to(self, device):
        self.dev_buf = self.host_buf.to(device, non_blocking=True)
        return self

    def fill(self, t0, t1):
        """Pack the states [t0, t1) into the host buffer (host threads; the GIL is released for the duration)."""
        if self._ptrs is None:
            raise RuntimeError('this replay is already packed')
        rc = native.lib().upamd_pack_fill_range(self.T, self._ptrs.ctypes.data, self.meta.ctypes.data, C.byref(self.layout),
                                                int(t0), int(t1), int(self._threads), self.host_buf.data_ptr())
        if rc == native.E_REPLAN:
            raise NeedsExactPlan(native.lib().upamd_last_error().decode())
        native.check(rc, 'upamd_pack_fill_range')
        self.filled = max(self.filled, int(t1)) if int(t0) <= self.filled else self.filled
        if self.filled >= self.T:
            self._ptrs = self._keep = None

    def byte_ranges(self, t0, t1):
        """[(begin, end)] byte ranges of the packed buffer that hold the states [t0, t1): one per section."""
        L, m = self.layout, self.meta
        a, b = m[t0], m[t1 - 1]
        n0, n1 = int(a[M_NODE_OFF]), int(b[M_NODE_OFF]) + int(b[M_N])
        e0, e1 = int(a[M_EDGE_OFF]), int(b[M_EDGE_OFF]) + int(b[M_E])
        h0, h1 = int(a[M_HE_OFF]), int(b[M_HE_OFF]) + int(b[M_NH])
        r0, r1 = int(a[M_RN_OFF]), int(b[M_RN_OFF]) + int(b[M_NR])
        p0, p1 = int(a[13]), int(b[13]) + int(b[M_N]) + 1
        Fn, NP, MS = int(L.numerical_dim), native.NODE_PAD, native.META_STRIDE
        spec = [(L.off_meta, 4 * MS, t0, t1), (L.off_x, 4 * NP, n0, n1), (L.off_nmask, 1, n0, n1), (L.off_rowptr, 4, p0, p1),
                (L.off_inc_nbr, 4, e0, e1), (L.off_he_src, 2, h0, h1), (L.off_he_dst, 2, h0, h1), (L.off_he_live, 1, h0, h1),
                (L.off_he_slot, 4, h0, h1), (L.off_rn_node, 2, r0, r1), (L.off_numerical, 4 * Fn, t0, t1),
                (L.off_cur, 4 * NP, t0, t1), (L.off_order, 2, n0, n1), (L.off_hinc_ptr, 4, p0, p1), (L.off_hinc_nbr, 4, h0, h1),
                (L.off_hinc_he, 4, h0, h1), (L.off_he_sel, 2, h0, h1), (L.off_xbar, 4 * NP, t0, t1)]
        return [(int(off) + w * lo, int(off) + w * hi) for off, w, lo, hi in spec if hi > lo and off >= 0]      # (off < 0: section not planned)

    def alloc_device(self, device):
        if self.dev_buf is None or self.dev_buf.numel() != self.host_buf.numel() or self.dev_buf.device != torch.device(device):
            self.dev_buf = torch.empty(self.host_buf.numel(), dtype=torch.uint8, device=device)
        return self.dev_buf

    def upload(self, t0, t1):
        """Enqueue the H2D copies of the states [t0, t1) on the CURRENT stream (page-locked source: asynchronous)."""
        for lo, hi in self.byte_ranges(t0, t1):
            self.dev_buf[lo:hi].copy_(self.host_buf[lo:hi], non_blocking=True)

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


# ---------------------------------------------------------------------------------------------------------------
# Compact wire record of one state.  The reference's rollout workers push the padded
# 9-field tuple (~148 KB at 1000 / 3000 pads) through a multiprocessing queue (khrylib/rl/agents/agent.py:92-97);
# the record keeps the same nine arrays but only up to the last non-empty node / edge row (~54 KB for an HLG state),
# in one contiguous uint8 buffer that can live in shared memory.  It is lossless: every dropped row is all-zero /
# all-False, `expand_state(record, padded=True)` gives back the exact padded tuple, and without `padded` it returns
# zero-copy views that `pack_replay` consumes directly (the engine never looks at pad rows).
_REC_MAGIC = 0x31535055          # 'UPS1'
_REC_HEADER = np.dtype([('magic', '<u4'), ('node_dim', '<u4'), ('numerical_len', '<u4'), ('cur_len', '<u4'),
                        ('stage_len', '<u4'), ('pad_n', '<u4'), ('pad_e', '<u4'), ('n_rows', '<u4'), ('e_rows', '<u4'),
                        ('edge_fill', '<i4')])       # index value of the dropped edge rows (the extractor pads with N-1)


def _align8(x):
    return (x + 7) & ~7


def _record_sections(h):
    """[(field index, dtype, shape, byte offset)] of a record with header values h, and its total size."""
    F, nr, er = int(h['node_dim']), int(h['n_rows']), int(h['e_rows'])
    spec = [(0, np.float32, (int(h['numerical_len']),)), (1, np.float32, (nr, F)), (2, np.int64, (er, 2)),
            (3, np.float32, (int(h['cur_len']),)), (4, np.bool_, (nr,)), (5, np.bool_, (er,)), (6, np.bool_, (er,)),
            (7, np.bool_, (nr,)), (8, np.float32, (int(h['stage_len']),))]
    out, off = [], _align8(_REC_HEADER.itemsize)
    for f, dt, shape in spec:
        out.append((f, dt, shape, off))
        off = _align8(off + int(np.prod(shape)) * np.dtype(dt).itemsize)
    return out, off


def _write_record(h, arrays):
    """header + the nine (already trimmed) arrays -> one contiguous uint8 record"""
    sections, total = _record_sections(h)
    rec = np.zeros(total, dtype=np.uint8)
    rec[:_REC_HEADER.itemsize] = np.frombuffer(h.tobytes(), dtype=np.uint8)
    for f, dt, shape, off in sections:
        n = int(np.prod(shape))
        if n:
            rec[off:off + n * np.dtype(dt).itemsize] = arrays[f].reshape(-1)[:n].view(np.uint8)
    return rec


def compact_state(state):
    """9-field padded state (observation_extractor.py:207-228) -> compact uint8 record (see above)."""
    if len(state) != 9:
        raise ValueError('state has %d fields, expected 9' % len(state))
    a = [_as_array(state[f], _FIELD_DTYPES[f]) for f in range(9)]
    nf, ei = a[1], a[2]
    if nf.ndim != 2 or ei.ndim != 2 or ei.shape[1] != 2:
        raise ValueError('node features must be [N, F] and edge index [E, 2]')
    if not (len(a[4]) == len(a[7]) == nf.shape[0] and len(a[5]) == len(a[6]) == ei.shape[0]):
        raise ValueError('mask lengths do not match the padded node / edge counts')
    used_n = np.flatnonzero(a[4] | a[7] | (nf != 0).any(axis=1))
    # padded edge rows repeat one index value (observation_extractor.py pads the edge list with max_num_nodes - 1):
    # trailing rows that are unmasked and hold exactly that value are dropped and re-created on expansion
    fill = int(ei[-1, 0]) if ei.shape[0] and ei[-1, 0] == ei[-1, 1] and abs(int(ei[-1, 0])) < 2 ** 31 else 0
    used_e = np.flatnonzero(a[5] | a[6] | (ei != fill).any(axis=1))
    h = np.zeros((), dtype=_REC_HEADER)
    h['magic'], h['node_dim'], h['numerical_len'] = _REC_MAGIC, nf.shape[1], a[0].size
    h['cur_len'], h['stage_len'] = a[3].size, a[8].size
    h['pad_n'], h['pad_e'], h['edge_fill'] = nf.shape[0], ei.shape[0], fill
    h['n_rows'] = int(used_n[-1]) + 1 if used_n.size else 0
    h['e_rows'] = int(used_e[-1]) + 1 if used_e.size else 0
    nr, er = int(h['n_rows']), int(h['e_rows'])
    return _write_record(h, [a[0], a[1][:nr], a[2][:er], a[3], a[4][:nr], a[5][:er], a[6][:er], a[7][:nr], a[8]])


def compact_from_arrays(numerical, node_features, edge_index, current_node, land_use_mask, road_mask, stage, pad_n, pad_e,
                        node_mask=None, edge_mask=None):
    """The compact record of an observation straight from the UNPADDED arrays the extractor holds before it pads them
    (observation_extractor.py:99-132: ``obs_nodes [n, F]``, ``edges [e, 2]``; :207-228: the masks handed to ``get_obs``)
    -- byte for byte the record ``compact_state`` makes of the padded 9-field tuple, without ever building the ~148 KB
    padded arrays.  ``pad_n`` / ``pad_e`` = the extractor's ``max_num_nodes`` / ``max_num_edges``; the node / edge masks
    default to all-True over the live rows (what ``_get_obs_graph`` emits).  Raises the extractor's own ``ValueError``
    when a limit is exceeded (:80-81, :94-95)."""
    nf = _as_array(node_features, np.float32)
    ei = _as_array(edge_index, np.int64)
    if nf.ndim != 2 or (ei.size and (ei.ndim != 2 or ei.shape[1] != 2)):
        raise ValueError('node features must be [n, F] and edge index [e, 2]')
    ei = ei.reshape(-1, 2)
    n, e = nf.shape[0], ei.shape[0]
    pad_n, pad_e = int(pad_n), int(pad_e)
    if n > pad_n:
        raise ValueError('The number of nodes exceeds the maximum limit.')
    if e > pad_e:
        raise ValueError('The number of edges exceeds the maximum limit.')

    def mask(m, rows, limit, what):
        m = np.ones(rows, dtype=np.bool_) if m is None else _as_array(m, np.bool_).reshape(-1)
        if m.size > limit:
            raise ValueError('The number of %s exceeds the maximum limit.' % what)
        return m
    nm, rm = mask(node_mask, n, pad_n, 'nodes'), mask(road_mask, n, pad_n, 'nodes')
    em, lm = mask(edge_mask, e, pad_e, 'edges'), mask(land_use_mask, e, pad_e, 'edges')
    # what compact_state would find in the padded tuple: the pad rows repeat (pad_n - 1, pad_n - 1) and all-zero features
    if e < pad_e:
        fill = pad_n - 1
    else:
        fill = int(ei[-1, 0]) if e and ei[-1, 0] == ei[-1, 1] and abs(int(ei[-1, 0])) < 2 ** 31 else 0

    def last_used(flags):
        idx = np.flatnonzero(flags)
        return int(idx[-1]) + 1 if idx.size else 0

    def grow(m, rows):                      # a mask shorter / longer than the live rows: pad with False like _pad_mask
        out = np.zeros(rows, dtype=np.bool_)
        out[:min(rows, m.size)] = m[:rows]
        return out
    rows_n = max(n, nm.size, rm.size)
    used_n = grow(nm, rows_n) | grow(rm, rows_n)
    used_n[:n] |= (nf != 0).any(axis=1)
    rows_e = max(e, em.size, lm.size)
    used_e = grow(em, rows_e) | grow(lm, rows_e)
    used_e[:e] |= (ei != fill).any(axis=1)
    nr, er = last_used(used_n), last_used(used_e)
    h = np.zeros((), dtype=_REC_HEADER)
    num = _as_array(numerical, np.float32).reshape(-1)
    cur = _as_array(current_node, np.float32).reshape(-1)
    st = _as_array(stage, np.float32).reshape(-1)
    h['magic'], h['node_dim'], h['numerical_len'] = _REC_MAGIC, nf.shape[1], num.size
    h['cur_len'], h['stage_len'] = cur.size, st.size
    h['pad_n'], h['pad_e'], h['edge_fill'] = pad_n, pad_e, fill
    h['n_rows'], h['e_rows'] = nr, er

    def rows(a, k, fill_value=0):           # the first k rows of an array with >= 0 live rows (zero / fill beyond them)
        if a.shape[0] >= k:
            return a[:k]
        out = np.full((k,) + a.shape[1:], fill_value, dtype=a.dtype)
        out[:a.shape[0]] = a
        return out
    return _write_record(h, [num, rows(nf, nr), rows(ei, er, fill), cur, grow(nm, nr), grow(em, er), grow(lm, er),
                             grow(rm, nr), st])


def record_bytes_bound(pad_n, pad_e, node_dim, numerical_len, stage_len=3):
    """Largest compact record a state with these pad sizes can produce (every padded row in use)."""
    h = np.zeros((), dtype=_REC_HEADER)
    h['node_dim'], h['numerical_len'], h['cur_len'], h['stage_len'] = node_dim, numerical_len, node_dim, stage_len
    h['n_rows'], h['e_rows'] = pad_n, pad_e
    return _record_sections(h)[1]


class RecordList(list):
    """A list of compact records that also carries their addresses and byte sizes as arrays (``addr`` uint64[T], ``size``
    int64[T]): ``plan_replay`` hands those to the C packer instead of touching the T record objects."""

    def __init__(self, records, addr=None, size=None):
        super().__init__(records)
        self.addr, self.size = addr, size


class RecordViews:
    """The records of one or more byte buffers (rollout arenas) as a read-only sequence that builds the zero-copy ``uint8`` view of
    a record only when it is ASKED for: ``plan_replay`` takes ``addr`` / ``size`` and never touches an element, so a batch of T
    records costs no per-record Python on its way into ``update_params`` (T views cost ~1 us each: 8 ms for an 8192-row batch)."""

    def __init__(self, bufs, which, offs, sizes, addr):
        self._bufs, self._which, self._offs = bufs, np.asarray(which, dtype=np.int64), np.asarray(offs, dtype=np.int64)
        self.size = np.asarray(sizes, dtype=np.int64)
        self.addr = np.asarray(addr, dtype=np.uint64)

    def __len__(self):
        return int(self.size.size)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        o = int(self._offs[i])
        return self._bufs[int(self._which[i])][o:o + int(self.size[i])]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def is_record(obj):
    return isinstance(obj, np.ndarray) and obj.dtype == np.uint8 and obj.ndim == 1 and obj.size >= _REC_HEADER.itemsize \
        and int(obj[:4].view('<u4')[0]) == _REC_MAGIC


def record_stage(record):
    """The stage one-hot (field 8, f32[stage_len]) of a record, without building views of the other eight fields."""
    w = record[:_REC_HEADER.itemsize].view('<u4')
    F, Fn, cur, st, nr, er = int(w[1]), int(w[2]), int(w[3]), int(w[4]), int(w[7]), int(w[8])
    off = _align8(_REC_HEADER.itemsize)
    for nbytes in (Fn * 4, nr * F * 4, er * 16, cur * 4, nr, er, er, nr):
        off = _align8(off + nbytes)
    return record[off:off + 4 * st].view(np.float32)


def record_pads(record):
    """(pad_n, pad_e) of the padded state a record was made from."""
    h = record[:_REC_HEADER.itemsize].view(_REC_HEADER)[0]
    return int(h['pad_n']), int(h['pad_e'])


def expand_state(record, padded=False):
    """Record -> list of the 9 arrays.  padded=False: zero-copy views with the trimmed row counts (what pack_replay
    needs); padded=True: fresh arrays of the original pad sizes, equal to the state the record was made from."""
    if not is_record(record):
        raise ValueError('not a compact state record')
    h = record[:_REC_HEADER.itemsize].view(_REC_HEADER)[0]
    sections, total = _record_sections(h)
    if record.size < total:
        raise ValueError('truncated state record (%d < %d bytes)' % (record.size, total))
    out = []
    for f, dt, shape, off in sections:
        n = int(np.prod(shape))
        v = record[off:off + n * np.dtype(dt).itemsize].view(dt).reshape(shape)
        if padded and f in (1, 2, 4, 5, 6, 7):
            rows = int(h['pad_n']) if f in (1, 4, 7) else int(h['pad_e'])
            full = np.zeros((rows,) + tuple(shape[1:]), dtype=dt)
            if f == 2:
                full[:] = int(h['edge_fill'])
            full[:shape[0]] = v
            v = full
        elif padded:
            v = v.copy()
        out.append(v)
    return out


def pack_replay(states, actions, node_dim, numerical_dim, n_threads=0, pin=None, reuse=None, mlp_fields=True):
    """states: list[T] of list[9] arrays (or tensors); actions: f32[T,2] (padded-slot indices).
    ``reuse``: optional dict owned by the caller; its pinned staging buffer is recycled across iterations
    (pinning ~1 GB per PPO iteration is otherwise a measurable part of the set-up time)."""
    try:
        pk = plan_replay(states, actions, node_dim, numerical_dim, n_threads, pin, reuse, exact=False, mlp_fields=mlp_fields)
        pk.fill(0, pk.T)
    except NeedsExactPlan:
        pk = plan_replay(states, actions, node_dim, numerical_dim, n_threads, pin, reuse, exact=True, mlp_fields=mlp_fields)
        pk.fill(0, pk.T)
    return pk


def plan_replay(states, actions, node_dim, numerical_dim, n_threads=0, pin=None, reuse=None, exact=True, mlp_fields=True):
    """The first half of ``pack_replay``: address tables, the counting pass (``upamd_pack_plan_ex``: meta table + layout) and
    the host buffer -- NOT yet filled: ``PackedReplay.fill(t0, t1)`` packs a range of states.  ``exact=False``: the counting
    pass reads the masks only and ``fill`` raises ``NeedsExactPlan`` if a live edge lies beyond the extent they give (never
    for states of the reference's extractor) -- the caller then plans again with ``exact=True``."""
    T = len(states)
    if T == 0:
        raise ValueError('empty replay')
    L = native.lib()
    ptrs = np.empty((9, T), dtype=np.uint64)
    pad_n = np.empty(T, dtype=np.int32)
    pad_e = np.empty(T, dtype=np.int32)
    keep = []
    first_slow = 0
    # compact wire records: the address table straight from the record headers in C.  A ``RecordList`` (rollout.RecordBatch
    # over shared-memory arenas) brings every record's address and size as arrays; any other all-record list costs one
    # address look-up per record here.  (Nine numpy views per record in Python took 90 us a state -- 0.7 s for an 8192-row replay.)
    addr = getattr(states, 'addr', None)
    size = getattr(states, 'size', None)
    nd = np.ndarray       # (a padded state is a list / tuple of nine arrays: only an ndarray can be a record -- the type test first)
    if addr is None and all(isinstance(s, nd) and is_record(s) for s in states):
        addr = np.fromiter((s.ctypes.data for s in states), dtype=np.uint64, count=T)
        size = np.fromiter((s.size for s in states), dtype=np.int64, count=T)
    if addr is not None:
        addr = np.ascontiguousarray(addr, dtype=np.uint64)
        size = np.ascontiguousarray(size, dtype=np.int64)
        if addr.size != T or size.size != T:
            raise ValueError('RecordList: %d addresses / %d sizes for %d states' % (addr.size, size.size, T))
        native.check(L.upamd_record_table(T, addr.ctypes.data, size.ctypes.data, int(node_dim), int(numerical_dim),
                                          ptrs.ctypes.data, pad_n.ctypes.data, pad_e.ctypes.data), 'upamd_record_table')
        first_slow = T
    elif any(isinstance(s, nd) and is_record(s) for s in states):      # records and padded tuples mixed: zero-copy views of the records
        states = [expand_state(s) if is_record(s) else s for s in states]
    helper = _host_helper() if addr is None else None
    if helper is not None:
        # fast path: every state whose fields already are C-contiguous arrays of the wire dtypes is handled in C
        try:
            bad = helper.addr_table(states, ptrs, pad_n, pad_e, int(node_dim), int(numerical_dim))
        except TypeError:                         # a stale _upamd_host.so with the older signature: glue only, skip it
            bad = 0
        first_slow = T if bad < 0 else 0          # anything unusual: redo the whole table the slow, validating way
    for t, s in enumerate(states if first_slow < T else ()):
        if len(s) != 9:
            raise ValueError('state %d has %d fields, expected 9' % (t, len(s)))
        sizes = [0] * 9
        for f in range(9):
            a = s[f]
            if not (isinstance(a, np.ndarray) and a.dtype == _FIELD_DTYPES[f] and a.flags['C_CONTIGUOUS']):
                a = _as_array(a, _FIELD_DTYPES[f])
                keep.append(a)
            ptrs[f, t] = a.__array_interface__['data'][0]
            sizes[f] = a.size
        nf, ei = s[1], s[2]
        pad_n[t] = nf.shape[0]
        pad_e[t] = ei.shape[0]
        if nf.shape[-1] != node_dim:
            raise ValueError('state %d: node feature width %d != node_dim %d' % (t, nf.shape[-1], node_dim))
        if len(s[4]) != pad_n[t] or len(s[7]) != pad_n[t] or len(s[5]) != pad_e[t] or len(s[6]) != pad_e[t]:
            raise ValueError('state %d: mask lengths do not match the padded node/edge counts' % t)
        # the C packer copies exactly numerical_dim / node_dim / 3 floats out of these three fields
        if sizes[0] != numerical_dim or sizes[3] != node_dim or sizes[8] != 3:
            raise ValueError('state %d: numerical / current-node / stage fields have %d / %d / %d entries, expected '
                             '%d / %d / 3' % (t, sizes[0], sizes[3], sizes[8], numerical_dim, node_dim))
    actions = _as_array(np.asarray(actions).reshape(T, 2), np.float32)
    meta = np.zeros((T, native.META_STRIDE), dtype=np.int32)
    layout = native.PackLayout()
    native.check(L.upamd_pack_plan_ex(T, ptrs.ctypes.data, pad_n.ctypes.data, pad_e.ctypes.data, actions.ctypes.data,
                                      int(node_dim), int(numerical_dim), int(n_threads), (1 if exact else 0) | (0 if mlp_fields else 2),
                                      meta.ctypes.data,
                                      C.byref(layout)), 'upamd_pack_plan_ex')
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
    # (`states` holds the arrays the address table points at; `keep` the converted copies: both stay alive with the plan)
    return PackedReplay(meta, layout, host, ptrs=ptrs, keep=(keep, states), n_threads=n_threads)


class _PinnedRing:
    """Per (device, dtype) ring of page-locked staging buffers for small asynchronous uploads.  A buffer is reused only after
    the copy that read it has certainly finished (an event recorded behind the copy; by the time the ring comes round it is
    long past), and grows when a bigger upload comes along."""

    def __init__(self, slots=3):
        self.slots, self.bufs, self.events, self.next = slots, {}, {}, {}
        # one ring per process, several host threads (a rollout.ActionServer thread next to the learner): slot choice, the
        # wait for the slot's previous copy, the memcpy into it and the recording of the new copy's event are ONE critical
        # section -- two threads can neither take the same slot nor lap the ring past a buffer whose copy is still in flight
        self.lock = threading.Lock()

    def upload(self, array, device):
        device = torch.device(device)
        t = torch.from_numpy(np.ascontiguousarray(array))
        if device.type != 'cuda':
            return t.to(device), t
        shape = t.shape
        with self.lock:
            dev, host = self._upload_locked(t.reshape(-1), device)
        return dev.view(shape), host.view(shape)

    def _upload_locked(self, t, device):
        key = (device.index, t.dtype)
        k = self.next.get(key, 0)
        self.next[key] = (k + 1) % self.slots
        slot = key + (k,)
        buf = self.bufs.get(slot)
        if buf is None:
            # first upload of this kind: page-lock every slot of the ring NOW (a hipHostMalloc costs milliseconds and may drain
            # the device: it belongs in the first iteration's set-up, not in a later epoch)
            for q in range(self.slots):
                self.bufs[key + (q,)] = torch.empty(max(int(t.numel() * 1.25), 1024), dtype=t.dtype).pin_memory()
            buf = self.bufs[slot]
        if slot in self.events:
            self.events[slot].synchronize()       # the slot's previous copy has read its buffer (also before replacing it)
        if buf.numel() < t.numel():
            buf = torch.empty(int(t.numel() * 1.25), dtype=t.dtype).pin_memory()
            self.bufs[slot] = buf
        host = buf[:t.numel()]
        host.numpy()[...] = t.numpy()             # (a plain memcpy: torch's copy_ is an OpenMP region above 32 K elements)
        dev = host.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[slot] = ev
        return dev, host


_ring = _PinnedRing()


def upload_pinned(array, device):
    """(device tensor, host staging view) of a numpy array, uploaded asynchronously from a recycled page-locked buffer."""
    return _ring.upload(array, device)


class Schedule:
    """Device-side index arrays of a sequence of minibatches (uploaded once per epoch).

    For minibatch k with rows ``rows_k`` (state ids): ``idx`` and the prefix sums of n / head-edge /
    road-node / incidence (2 e) counts in minibatch order, plus the host-side totals the engine needs.
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
            ninc = 2 * meta[rows, M_E].astype(np.int64)      # incidences (edge directions) per graph
            arr = np.zeros(B + 4 * (B + 1), dtype=np.int32)
            arr[:B] = rows
            arr[B + 1:2 * B + 1] = np.cumsum(n)
            arr[2 * B + 2:3 * B + 2] = np.cumsum(nh)
            arr[3 * B + 3:4 * B + 3] = np.cumsum(nr)
            arr[4 * B + 4:5 * B + 4] = np.cumsum(ninc)
            pad = (-arr.size) % 64            # keep every minibatch's arrays 256-byte aligned
            chunks.append(np.concatenate([arr, np.zeros(pad, dtype=np.int32)]))
            self.items.append(dict(B=int(B), n_nodes=int(n.sum()), n_he=int(nh.sum()), n_rn=int(nr.sum()),
                                   max_n=int(n.max()), max_inc=int(ninc.max()), n_inc=int(ninc.sum()), base=cursor,
                                   max_cand=max(1, int(np.where(meta[rows, M_STAGE] == 0, nh, np.where(meta[rows, M_STAGE] == 1, nr, 0)).max())),
                                   n_land=int((meta[rows, M_STAGE] == 0).sum()),
                                   n_road=int((meta[rows, M_STAGE] == 1).sum())))
            cursor += arr.size + pad
        flat = np.concatenate(chunks)
        # pinned + asynchronous: a pageable, blocking upload made the host wait for every kernel queued before it -- one
        # pipeline drain per epoch.  The page-locked staging buffers are RECYCLED (upload_pinned): allocating one per epoch
        # costs 1-2 ms of host time (and now and then tens of ms), which is a whole epoch of small-model optimizer steps.
        self.dev, self._host = upload_pinned(flat, device)

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
        mb.n_inc, mb.inc_off_dev = it['n_inc'], base + 4 * (4 * B + 3)
        mb.max_cand = it['max_cand']
        return mb, it
