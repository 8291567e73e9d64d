"""Thin host wrapper around the native engine: owns the flat parameter / gradient / Adam buffers
and the scratch workspace (torch tensors = device memory plumbing) and forwards every call to
the C ABI with raw pointers.  No math happens here.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import native


import contextlib

_NULL_CTX = contextlib.nullcontext()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device):
    """The caller's current HIP stream ON THE ENGINE'S DEVICE (not on the process's current device)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class NativeEngine:
    """One engine per (model description, device)."""

    def __init__(self, desc, device):
        self.lib = native.lib()
        self.desc = desc
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('the HIP engine needs a GPU device (got %s); there is no CPU fallback' % self.device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        h = C.c_void_p()
        native.check(self.lib.upamd_engine_create(C.byref(desc), C.byref(h)), 'upamd_engine_create')
        self.handle = h
        self.table, self.n_floats, self.groups = native.param_table(desc)
        self.index = {name: (off, rows, cols, grp) for name, off, rows, cols, grp in self.table}
        self.ws_slots = {}        # scratch workspaces; slot > 0 = concurrent sub-batches on side streams
        # The native entry points are not re-entrant (per-stream side contexts, profiler tables, the slot workspaces):
        # host threads that share one engine -- a rollout.ActionServer thread next to the learner -- are serialised
        # here.  EVERY method that calls into the library takes the lock (forward / backward, the PPO-math launches,
        # workspace read-backs, the profiler).  Only the ENQUEUE is under it; the kernels of two callers still overlap
        # on their streams.
        self.lock = threading.RLock()
        # this engine's own kernel-lab knobs (`set_tune`): applied around every native call of THIS engine, the process defaults
        # restored behind it (native.tuned) -- two engines in one process no longer share a setting one of them changed for itself
        self.tune_overrides = {}

    def _st(self):
        return _stream(self.device)

    def set_tune(self, name, value):
        """Kernel-lab knob `name` for THIS engine only (None: back to the process default).  Names: include/upamd.h, upamd_tune."""
        if name not in native.TUNE_DEFAULTS:
            raise KeyError('unknown kernel-lab knob %r' % (name,))
        with self.lock:
            if value is None:
                self.tune_overrides.pop(name, None)
            else:
                self.tune_overrides[name] = int(value)
                native.note_engine_overrides()

    def _on_device(self):
        """Native launches go to the device that is current in the calling thread (the library never calls
        hipSetDevice): make the engine's device current around every call."""
        if torch.cuda.current_device() == self._dev_index:
            return _NULL_CTX                   # already current (the usual case): a device guard costs two runtime calls per native call
        return torch.cuda.device(self.device)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.upamd_engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- flat parameter buffer <-> named tensors
    def new_flat(self):
        return torch.zeros(self.n_floats, dtype=torch.float32, device=self.device)

    def flatten(self, named, out=None):
        """named: dict name -> tensor (any device).  Returns the flat fp32 device buffer."""
        flat = self.new_flat() if out is None else out
        for name, off, rows, cols, _ in self.table:
            flat[off:off + rows * cols].copy_(named[name].detach().reshape(-1), non_blocking=True)
        return flat

    def unflatten(self, flat, named):
        """Copies the flat buffer back into the named tensors (in place, keeps their devices)."""
        with torch.no_grad():
            for name, off, rows, cols, _ in self.table:
                dst = named[name]
                dst.copy_(flat[off:off + rows * cols].view_as(dst))

    def views(self, flat):
        return {name: flat[off:off + rows * cols].view(rows, cols) if cols > 1 or name.endswith('weight')
                else flat[off:off + rows] for name, off, rows, cols, _ in self.table}

    # ---- workspace
    # The engine keeps every activation the backward needs in ONE caller-owned workspace buffer.  Two kinds:
    #   * slots (``slot=k``): engine-owned, reused by every call on that slot -- the PPO update loop (forward and
    #     backward of the same minibatch are always adjacent there) and no-grad forwards;
    #   * private (``ws=<tensor from alloc_workspace>``): owned by ONE forward/backward pair of the autograd surface
    #     (models._Runner), so interleaved forwards can never clobber activations a pending backward still needs.
    @property
    def ws(self):
        return self.ws_slots.get(0)

    def workspace_bytes(self, mb):
        need = C.c_int64()
        with self.lock, native.tuned(self.tune_overrides):
            native.check(self.lib.upamd_workspace_bytes(self.handle, C.byref(mb), 1, C.byref(need)), 'upamd_workspace_bytes')
        return int(need.value)

    def alloc_workspace(self, mb):
        """A private workspace for one forward/backward pair (exact size + alignment slack)."""
        return torch.empty(self.workspace_bytes(mb) + 512, dtype=torch.uint8, device=self.device)

    def ensure_workspace(self, mb, slot=0):
        need = self.workspace_bytes(mb)
        ws = self.ws_slots.get(slot)
        if ws is None or ws.numel() < need + 256:
            self.ws_slots[slot] = None
            ws = torch.empty(int(need * 1.05) + 4096, dtype=torch.uint8, device=self.device)
            self.ws_slots[slot] = ws
        return ws

    def _ws_args(self, slot=0, ws=None):
        ws = self.ws_slots[slot] if ws is None else ws
        base = ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        return C.c_void_p(aligned), C.c_int64(ws.numel() - (aligned - base)), aligned - base

    # ---- forward / backward
    def forward(self, packed, mb, flat_params, value, logp, ent, keep=True, slot=None, ws=None):
        """slot=None: slot 0 for a forward whose activations are kept, the scratch slot 'nograd' for a no-grad one, so
        that a stray no-grad forward never overwrites activations a backward on slot 0 still needs.  A caller that
        knows no backward is pending (the PPO pre-pass) names slot 0 itself and spares the second arena."""
        with self.lock, native.tuned(self.tune_overrides):
            if ws is None:
                if slot is None:
                    slot = 0 if keep else 'nograd'
                self.ensure_workspace(mb, slot)
            wsp, wsb, _ = self._ws_args(slot, ws)
            with self._on_device():
                native.check(self.lib.upamd_forward(self.handle, _ptr(packed.dev_buf), C.byref(packed.layout), C.byref(mb),
                                                    _ptr(flat_params), wsp, wsb, _ptr(value), _ptr(logp), _ptr(ent),
                                                    1 if keep else 0, self._st()), 'upamd_forward')
            self.last_forward_slot = slot if ws is None else None

    def release_slot(self, slot):
        """Give a slot's workspace back to the caching allocator (stream-ordered)."""
        with self.lock, native.tuned(self.tune_overrides):
            self.ws_slots.pop(slot, None)

    def backward(self, packed, mb, flat_params, dvalue, dlogp, dent, grads, slot=0, ws=None):
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            wsp, wsb, _ = self._ws_args(slot, ws)
            native.check(self.lib.upamd_backward(self.handle, _ptr(packed.dev_buf), C.byref(packed.layout), C.byref(mb),
                                                 _ptr(flat_params), wsp, wsb, _ptr(dvalue), _ptr(dlogp), _ptr(dent),
                                                 _ptr(grads), self._st()), 'upamd_backward')

    # ---- gradient buckets of the last backward on the caller's stream (data parallelism; include/upamd.h)
    def grad_buckets(self):
        """[(begin, end)] float ranges of the flat gradient buffer in the order the last `backward` on the current stream
        made them final (one range for the paths that finalise everything at the end)."""
        n = C.c_int32()
        b, e = (C.c_int64 * 24)(), (C.c_int64 * 24)()
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_grad_buckets(self.handle, self._st(), 24, C.byref(n), b, e), 'upamd_grad_buckets')
        return [(int(b[i]), int(e[i])) for i in range(n.value)]

    def grad_bucket_wait(self, k, waiter):
        """`waiter` (a torch.cuda.Stream) waits until bucket k of the last backward on the current stream is final"""
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_grad_bucket_wait(self.handle, self._st(), int(k), C.c_void_p(waiter.cuda_stream)),
                         'upamd_grad_bucket_wait')

    # ---- fused optimizer-step front end of small models (csrc/tiny.hip)
    def step_fused_ok(self, mb):
        """True when forward + PPO loss + backward of this minibatch run as ONE launch (gcn_node_dim <= 32, graphs that fit
        a workgroup's LDS): `step_fused` then replaces forward / ppo_loss_rows / backward."""
        with self.lock, native.tuned(self.tune_overrides):
            return bool(self.lib.upamd_step_fused_ok(self.handle, C.byref(mb)))

    def step_fused(self, packed, mb, flat_params, rows, adv, ret, old_logp, exps, clip_eps, cv, ce, inv_rows, inv_ind, value,
                   logp, ent, grads, losses, slot=0):
        """forward + loss seeds + backward; `grads` (flat, n_floats) is overwritten, `losses` (4) gets the loss scalars"""
        assert rows is None or (rows.dtype == torch.int64 and rows.is_contiguous())
        with self.lock, native.tuned(self.tune_overrides):
            self.ensure_workspace(mb, slot)
            wsp, wsb, _ = self._ws_args(slot, None)
            with self._on_device():
                native.check(self.lib.upamd_step_fused(self.handle, _ptr(packed.dev_buf), C.byref(packed.layout), C.byref(mb),
                                                       _ptr(flat_params), wsp, wsb, _ptr(rows), _ptr(adv), _ptr(ret),
                                                       _ptr(old_logp), _ptr(exps), clip_eps, cv, ce, inv_rows, inv_ind,
                                                       _ptr(value), _ptr(logp), _ptr(ent), _ptr(grads), _ptr(losses),
                                                       self._st()), 'upamd_step_fused')

    def ws_tensor(self, mb, name, slot=0, ws=None):
        """Row-major copy of a named intermediate of the last forward on that workspace (parity tests, action heads)."""
        off, rows, cols, kind = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        with self.lock, native.tuned(self.tune_overrides):
            native.check(self.lib.upamd_ws_tensor(self.handle, C.byref(mb), name.encode(), C.byref(off), C.byref(rows),
                                                  C.byref(cols), C.byref(kind)), 'upamd_ws_tensor')
            _, _, shift = self._ws_args(slot, ws)
            buf = self.ws_slots[slot] if ws is None else ws
        n = rows.value * cols.value
        raw = buf[shift + off.value: shift + off.value + 4 * n].view(torch.float32)
        if kind.value == 1:
            return raw.view(cols.value // 16, rows.value, 16).permute(1, 0, 2).reshape(rows.value, cols.value).clone()
        return raw.view(rows.value, cols.value).clone()

    def ws_view(self, mb, name, slot=0, ws=None):
        """A named ROW-MAJOR intermediate of the last forward on that workspace as a view (no copy): valid until the next
        forward on the slot."""
        off, rows, cols, kind = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        with self.lock, native.tuned(self.tune_overrides):
            native.check(self.lib.upamd_ws_tensor(self.handle, C.byref(mb), name.encode(), C.byref(off), C.byref(rows),
                                                  C.byref(cols), C.byref(kind)), 'upamd_ws_tensor')
            _, _, shift = self._ws_args(slot, ws)
            buf = self.ws_slots[slot] if ws is None else ws
        if kind.value != 0:
            raise ValueError('%s is stored panel-major: use ws_tensor' % name)
        n = rows.value * cols.value
        return buf[shift + off.value: shift + off.value + 4 * n].view(torch.float32).view(rows.value, cols.value)

    # ---- rollout inference
    def select_actions(self, packed, mb, z_he, z_rn, greedy, uniform, actions):
        """`select_action` (policy.py:67-85) of every row of the minibatch from its ragged pointer-head logits: arg-max where
        `greedy[b]` (uint8, device), else one inverse-CDF draw with `uniform[b]`; `actions` f32 [B, 2] (device) is written."""
        assert greedy.dtype == torch.uint8 and uniform.dtype == torch.float32 and actions.dtype == torch.float32
        assert greedy.numel() >= mb.B and uniform.numel() >= mb.B and actions.numel() >= 2 * mb.B and actions.is_contiguous()
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_select_actions(_ptr(packed.dev_buf), C.byref(packed.layout), C.byref(mb),
                                                       _ptr(z_he) if z_he is not None else None,
                                                       _ptr(z_rn) if z_rn is not None else None, _ptr(greedy), _ptr(uniform),
                                                       _ptr(actions), self._st()), 'upamd_select_actions')

    # ---- PPO math
    def ppo_loss(self, B, value, logp, ent, adv, ret, old_logp, exps, clip_eps, cv, ce, inv_rows, inv_ind, dvalue,
                 dlogp, dent, losses):
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_ppo_loss(B, _ptr(value), _ptr(logp), _ptr(ent), _ptr(adv), _ptr(ret),
                                                 _ptr(old_logp), _ptr(exps), clip_eps, cv, ce, inv_rows, inv_ind,
                                                 _ptr(dvalue), _ptr(dlogp), _ptr(dent), _ptr(losses), self._st()),
                         'upamd_ppo_loss')

    def ppo_loss_rows(self, B, value, logp, ent, rows, adv, ret, old_logp, exps, clip_eps, cv, ce, inv_rows, inv_ind,
                      dvalue, dlogp, dent, losses, zero=None):
        """loss of the minibatch whose replay rows are `rows` (int64, device) + zeroing of `zero` in the same launch"""
        assert rows.dtype == torch.int64 and rows.is_contiguous()
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_ppo_loss_rows(B, _ptr(value), _ptr(logp), _ptr(ent), _ptr(rows), _ptr(adv),
                                                      _ptr(ret), _ptr(old_logp), _ptr(exps), clip_eps, cv, ce, inv_rows,
                                                      inv_ind, _ptr(dvalue), _ptr(dlogp), _ptr(dent), _ptr(losses),
                                                      _ptr(zero) if zero is not None else None,
                                                      zero.numel() if zero is not None else 0, self._st()),
                         'upamd_ppo_loss_rows')

    def gae(self, rewards, masks, values, gamma, tau, adv, ret):
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_gae(rewards.numel(), _ptr(rewards), _ptr(masks), _ptr(values), float(gamma),
                                            float(tau), _ptr(adv), _ptr(ret), self._st()), 'upamd_gae')

    def clip_first_step(self, grads, max_norm, scratch):
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_clip_first_step(C.byref(self.desc), _ptr(grads), float(max_norm), _ptr(scratch),
                                                        self._st()), 'upamd_clip_first_step')

    def adam_step(self, group, params, grads, m, v, step, lr, beta1, beta2, eps, weight_decay):
        b, e = self.groups[group]
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_adam_step(b, e, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), int(step),
                                                  float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                                  self._st()), 'upamd_adam_step')

    def adam_groups(self, steps, params, grads, m, v, lr, beta1, beta2, eps, weight_decay, loss_src=None, loss_dst=None):
        """all optimizer groups in one launch; steps[g] = that group's 1-based step count, 0 = skip the group"""
        n = len(self.groups)
        b = (C.c_int64 * n)(*[g[0] for g in self.groups])
        e = (C.c_int64 * n)(*[g[1] for g in self.groups])
        st = (C.c_int32 * n)(*[int(x) for x in steps])
        with self.lock, native.tuned(self.tune_overrides), self._on_device():
            native.check(self.lib.upamd_adam_groups(n, b, e, st, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), float(lr),
                                                    float(beta1), float(beta2), float(eps), float(weight_decay),
                                                    _ptr(loss_src) if loss_src is not None else None,
                                                    _ptr(loss_dst) if loss_dst is not None else None, self._st()),
                         'upamd_adam_groups')

    # ---- profiling
    def profile(self, on):
        with self.lock, native.tuned(self.tune_overrides):
            native.check(self.lib.upamd_profile_enable(self.handle, 1 if on else 0), 'upamd_profile_enable')

    def profile_reset(self):
        with self.lock, native.tuned(self.tune_overrides):
            native.check(self.lib.upamd_profile_reset(self.handle), 'upamd_profile_reset')

    def profile_read(self, name):
        n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        with self.lock, native.tuned(self.tune_overrides):
            native.check(self.lib.upamd_profile_read(self.handle, name.encode(), C.byref(n), C.byref(ms), C.byref(fl),
                                                     C.byref(by)), 'upamd_profile_read')
        return dict(launches=n.value, total_ms=ms.value, flops=fl.value, bytes=by.value)
