"""Rollout side of the drop-in: what sits between the env workers and the HIP update.

The reference samples with ``num_threads`` forked CPU workers (khrylib/rl/agents/agent.py:75-100); every env step
evaluates the policy on ONE padded state on the CPU (urban_planning/agents/urban_planning_agent.py:49-91,
policy.py:67-85), stores the padded 9-field state (~148 KB) in a python ``Memory`` (khrylib/utils/memory.py:4-23)
and finally pickles the whole memory through a ``multiprocessing.Queue`` (agent.py:92-97).  After the update moved to
the GPU that is where an iteration's time goes, so this module provides, each usable on its own:

* ``ActionServer`` / ``ActionClient`` -- batched action serving: the networks stay on the GPU in the learner process;
  workers write their state as a compact record into a shared-memory slot, the server answers every pending request
  (B ~ number of workers) with ONE HIP forward and samples on the device (``Categorical.sample`` / arg-max, exactly the
  reference's ``select_action`` semantics).  A client has the ``select_action(x, mean_action)`` signature of
  ``policy_net``, so a worker just swaps the object it calls.  The per-iteration greedy evaluation
  (urban_planning_agent.py:402-467) becomes one more client that runs while the sampling workers do -- no
  CPU<->GPU round trip of the model, no serial eval phase.
* ``SharedArena`` / ``ArenaMemory`` -- the worker-side replay container: ``push`` has ``Memory.push``'s signature but
  appends the state as a compact record (packer.compact_state, lossless, ~2.5x smaller) to a shared-memory arena
  instead of a python list, so nothing is pickled at the end of sampling: the worker reports a row count.
* ``RecordBatch`` -- the learner-side ``TrajBatchDisc`` (urban_planning/utils/tools.py:4-16) over arenas (or plain
  memories): ``states`` are zero-copy record views that ``pack_replay`` consumes directly.

INTEGRATION.md shows the three-line patches of ``sample_worker`` / ``sample``.
"""
import multiprocessing
import multiprocessing.connection
import threading
import time
from multiprocessing import shared_memory

import numpy as np
import torch

from . import packer

_ALIGN = 64


def _align(x, a=_ALIGN):
    return (x + a - 1) // a * a


def _to_record(state):
    if packer.is_record(state):
        return state
    return packer.compact_state([f.detach().cpu().numpy() if isinstance(f, torch.Tensor) else f for f in state])


# ------------------------------------------------------------------------------------------------ action serving
# Transport (round 6): a request RING in shared memory + eventfd wake-ups, no pipes and nothing pickled.  Every client owns one
# slot: a header (request / response sequence numbers, row count, greedy flag, record sizes, status + error text), the answer
# (f32 [MAX_ROWS, 2]) and the records.  A client writes its records and header, bumps ``req_seq`` and rings the ONE doorbell
# eventfd all clients share; the server wakes once per serving round, finds the pending slots with one strided numpy compare
# (req_seq != resp_seq), answers them with one forward, writes answers + ``resp_seq`` and rings each answered client's own eventfd.
# (Round 5 moved a pickled (sizes, flag) tuple and a pickled 'ok' through a duplex pipe per request: 2 x 64 send / recv system calls
# with pickling per 64-row round in the serving thread, which was busy 72-91 % of the time doing that.)
_H_REQ, _H_RESP, _H_N, _H_MEAN, _H_STATUS, _H_ERRLEN, _H_SIZES = 0, 8, 16, 20, 24, 28, 32
_ERR_CAP = 480


def _eventfd():
    import os
    return os.eventfd(0, os.EFD_NONBLOCK)


def _efd_wait(fd, timeout):
    """True when the eventfd was signalled within `timeout` seconds (the counter is consumed)."""
    import os
    import select
    r, _, _ = select.select([fd], [], [], timeout)
    if not r:
        return False
    try:
        os.eventfd_read(fd)
    except BlockingIOError:
        return False
    return True


class ActionClient:
    """Worker-side handle: ``select_action(x, mean_action)`` like ``UrbanPlanningPolicy.select_action``."""

    MAX_ROWS = 64
    HDR = _align(_H_SIZES + 4 * MAX_ROWS + _ERR_CAP)     # header + record sizes + error text
    ACT = HDR                                            # answers: f32 [MAX_ROWS, 2]
    REC = _align(HDR + 8 * MAX_ROWS)                     # first record

    def __init__(self, shm_name, index, slot_bytes, doorbell, wake, timeout_s=120.0):
        self._shm_name, self.index, self.slot_bytes = shm_name, index, slot_bytes
        self.doorbell, self.wake = doorbell, wake        # eventfds: the server's (shared by all clients), this client's own
        self.timeout_s = timeout_s            # a dead server must not hang the sampling phase for ever
        self._shm = None
        self._seq = 0
        self._dead = None                     # set after a timeout: the request is still in flight on the shared slot
        self.type = 'discrete'

    def _slot(self):
        if self._shm is None:               # attached lazily, i.e. in the process that uses the client
            self._shm = shared_memory.SharedMemory(name=self._shm_name)
        base = self.index * self.slot_bytes
        return np.ndarray((self.slot_bytes,), dtype=np.uint8, buffer=self._shm.buf, offset=base)

    def post(self, sizes, mean_action):
        """Publish a request whose records already sit in the slot (``select_action`` does both; tests post by hand)."""
        import os
        slot = self._slot()
        slot[_H_N:_H_N + 8].view(np.uint32)[:] = (len(sizes), 1 if mean_action else 0)
        slot[_H_SIZES:_H_SIZES + 4 * len(sizes)].view(np.uint32)[:] = sizes
        req = slot[_H_REQ:_H_REQ + 8].view(np.uint64)
        self._seq = int(req[0]) + 1             # (continues the SLOT's sequence: a fresh handle on a used slot is fine)
        req[0] = self._seq                      # published last (x86: stores stay in order)
        os.eventfd_write(self.doorbell, 1)

    def select_action(self, x, mean_action=False):
        if self._dead is not None:
            raise RuntimeError('action client %d is closed: %s' % (self.index, self._dead))
        if len(x) > self.MAX_ROWS:
            raise ValueError('at most %d states per request' % self.MAX_ROWS)
        slot = self._slot()
        cursor = self.REC
        sizes = []
        for s in x:
            rec = _to_record(s)
            if cursor + rec.size > self.slot_bytes:
                raise ValueError('state records of one request exceed the %d-byte slot' % self.slot_bytes)
            slot[cursor:cursor + rec.size] = rec
            sizes.append(int(rec.size))
            cursor = _align(cursor + rec.size)
        self.post(sizes, mean_action)
        deadline = time.monotonic() + self.timeout_s
        resp = slot[_H_RESP:_H_RESP + 8].view(np.uint64)
        while int(resp[0]) != self._seq:
            left = deadline - time.monotonic()
            if left <= 0 or not _efd_wait(self.wake, left):
                if int(resp[0]) == self._seq:
                    break
                if time.monotonic() < deadline:
                    continue
                # the request stays in flight: the server may still read or answer it on the shared slot -- this client is finished
                self._dead = 'a request timed out after %.0f s and may still be in flight' % self.timeout_s
                raise TimeoutError('action server did not answer within %.0f s (is its serving thread alive?)' % self.timeout_s)
        status, err_len = (int(v) for v in slot[_H_STATUS:_H_STATUS + 8].view(np.uint32))
        if status != 0:
            text = bytes(slot[_H_SIZES + 4 * self.MAX_ROWS:_H_SIZES + 4 * self.MAX_ROWS + min(err_len, _ERR_CAP)]).decode('utf-8', 'replace')
            raise RuntimeError('action server: %s' % text)
        act = slot[self.ACT:self.ACT + 8 * len(x)].view(np.float32).reshape(len(x), 2).copy()
        return torch.from_numpy(act)

    def close(self):
        if self._shm is not None:
            self._shm.close()
            self._shm = None


class ActionServer:
    """Owns the request slots; ``serve_once`` answers everything that is pending with one batched forward."""

    def __init__(self, policy_net, num_clients, slot_bytes=1 << 20, linger_s=0.0002, mp_context=None):
        self.policy_net = policy_net
        self.num_clients, self.slot_bytes, self.linger_s = num_clients, _align(max(slot_bytes, ActionClient.REC + 4096)), linger_s
        self.shm = shared_memory.SharedMemory(create=True, size=num_clients * self.slot_bytes)
        # (a fresh POSIX shared-memory object is zero-filled by the kernel: sequence numbers start at 0 without touching the slots,
        # 64 MB for 64 clients, here)
        self.doorbell = _eventfd()                      # every client rings this one: ONE wake-up of the serving thread per round
        self.wakes = [_eventfd() for _ in range(num_clients)]
        n, sb, buf = num_clients, self.slot_bytes, self.shm.buf
        # strided views over all slots at once: the pending scan and the answers are a handful of vector operations per round
        self._seqs = np.ndarray((n, 2), dtype=np.uint64, buffer=buf, strides=(sb, 8))                  # req_seq, resp_seq
        self._ctl = np.ndarray((n, 4), dtype=np.uint32, buffer=buf, offset=_H_N, strides=(sb, 4))      # n, mean, status, err_len
        self._sizes = np.ndarray((n, ActionClient.MAX_ROWS), dtype=np.uint32, buffer=buf, offset=_H_SIZES, strides=(sb, 4))
        self._acts = np.ndarray((n, ActionClient.MAX_ROWS, 2), dtype=np.float32, buffer=buf, offset=ActionClient.ACT, strides=(sb, 8, 4))
        self._base = np.frombuffer(buf, dtype=np.uint8).ctypes.data
        self.stats = dict(batches=0, requests=0, rows=0, max_rows=0, busy_s=0.0)
        self.fast = True                # GPU modules: models._HipBackend.serve_actions instead of policy_net.forward + Categorical over pads
        self.last_error = None
        self.last_traceback = None
        self._thread = None
        self._stop = threading.Event()

    def client(self, i):
        return ActionClient(self.shm.name, i, self.slot_bytes, self.doorbell, self.wakes[i])

    def _slot(self, i):
        return np.ndarray((self.slot_bytes,), dtype=np.uint8, buffer=self.shm.buf, offset=i * self.slot_bytes)

    def _answer(self, idx, error=None):
        """publish the answers of the slots `idx` (their actions are already in place) and wake their clients"""
        import os
        if error is not None:
            raw = error.encode('utf-8', 'replace')[:_ERR_CAP]
            for i in idx:
                s = self._slot(int(i))
                s[_H_SIZES + 4 * ActionClient.MAX_ROWS:_H_SIZES + 4 * ActionClient.MAX_ROWS + len(raw)] = np.frombuffer(raw, dtype=np.uint8)
            self._ctl[idx, 2], self._ctl[idx, 3] = 1, len(raw)
        else:
            self._ctl[idx, 2] = 0
        self._seqs[idx, 1] = self._seqs[idx, 0]         # published last
        for i in idx:
            os.eventfd_write(self.wakes[int(i)], 1)

    def serve_once(self, timeout=0.05):
        """Waits up to ``timeout`` for a request, lingers ``linger_s`` for the other workers' requests to arrive, then
        answers all of them with one forward.  Returns the number of requests answered."""
        if not _efd_wait(self.doorbell, timeout):
            if not (self._seqs[:, 0] != self._seqs[:, 1]).any():      # (a ring that raced with the previous round's scan)
                return 0
        if self.linger_s > 0:
            time.sleep(self.linger_s)
        pend = np.flatnonzero(self._seqs[:, 0] != self._seqs[:, 1])
        if pend.size == 0:
            return 0
        # From here on every pending request is taken: whatever happens, its worker gets an answer
        t0 = time.perf_counter()
        cnt = self._ctl[pend, 0].astype(np.int64)
        mean = self._ctl[pend, 1] != 0
        good, states, addrs, sizes_all, owner, rows_of = [], [], [], [], [], []
        sb = self.slot_bytes
        sz1 = self._sizes[pend, 0].astype(np.int64)
        single = bool((cnt == 1).all() and (sz1 > 0).all() and (ActionClient.REC + sz1 <= sb).all())
        if single:
            # the common round -- every worker asks about ONE state (sample_worker, urban_planning_agent.py:60-61): no per-request
            # arithmetic, one slice per record
            flat = np.frombuffer(self.shm.buf, dtype=np.uint8)
            first = pend.astype(np.int64) * sb + ActionClient.REC
            states = [flat[a:a + z] for a, z in zip(first.tolist(), sz1.tolist())]
            good, rows_of = pend.tolist(), [1] * pend.size
            addrs, sizes_all, owner = [self._base + first], [sz1], [mean]
        for j, i in enumerate(pend if not single else ()):
            n, i = int(cnt[j]), int(i)
            sz = self._sizes[i, :max(min(n, ActionClient.MAX_ROWS), 0)].astype(np.int64)
            off = ActionClient.REC + np.concatenate([[0], np.cumsum((sz[:-1] + _ALIGN - 1) // _ALIGN * _ALIGN)]) if n > 0 else np.zeros(0, np.int64)
            if n <= 0 or n > ActionClient.MAX_ROWS or (sz <= 0).any() or (off + sz > self.slot_bytes).any():
                self.last_error = 'ValueError: request of client %d does not fit its slot (%d states)' % (i, n)
                self._answer(np.array([i]), self.last_error)
                continue
            slot = self._slot(i)
            good.append(i)
            rows_of.append(n)
            for o, z in zip(off, sz):
                states.append(slot[int(o):int(o) + int(z)])
            addrs.append(self._base + i * self.slot_bytes + off)
            sizes_all.append(sz)
            owner.append(np.full(n, mean[j]))
        n_req = len(pend)
        if good:
            good = np.asarray(good)
            try:
                recs = packer.RecordList(states, np.concatenate(addrs).astype(np.uint64), np.concatenate(sizes_all))
                actions = np.ascontiguousarray(self._actions(recs, np.concatenate(owner)), dtype=np.float32)
                row = 0
                if all(r == 1 for r in rows_of):
                    self._acts[good, 0, :] = actions
                else:
                    for i, r in zip(good, rows_of):
                        self._acts[int(i), :r, :] = actions[row:row + r]
                        row += r
                self._answer(good)
            except Exception as exc:                    # report to the workers instead of dying silently
                import traceback
                self.last_error = '%s: %s' % (type(exc).__name__, exc)
                self.last_traceback = traceback.format_exc()
                self._answer(good, self.last_error)
        st = self.stats
        st['batches'] += 1
        st['requests'] += n_req
        st['rows'] += len(states)
        st['max_rows'] = max(st['max_rows'], len(states))
        st['busy_s'] += time.perf_counter() - t0
        return n_req

    def _actions(self, states, mean_rows):
        """One forward for all rows; per row arg-max (mean_action) or a sample (policy.py:67-85)."""
        backend = getattr(self.policy_net, '_backend', [None])[0]
        if self.fast and backend is not None and next(self.policy_net.parameters()).device.type == 'cuda':
            return backend.serve_actions(states, mean_rows)         # the lean HIP route (models._HipBackend.serve_actions)
        with torch.no_grad():
            land_dist, road_dist, stage = self.policy_net.forward(states)
            action = torch.zeros(stage.shape[0], 2, dtype=torch.float32, device=stage.device)
            mean_t = torch.from_numpy(mean_rows).to(stage.device)
            for col, dist in ((0, land_dist), (1, road_dist)):
                if dist is None:
                    continue
                sel = stage[:, col].bool()
                greedy = dist.probs.argmax(dim=1)
                drawn = dist.sample()
                action[sel, col] = torch.where(mean_t[sel], greedy, drawn).to(torch.float32)
        return action.cpu().numpy()

    # ---- background service in the learner process
    def launch(self, worker_fn, worker_args, mp_context=None):
        """Fork the env workers FIRST, then start the serving thread: `fork()` while another thread of this process is
        inside the HIP runtime / the allocator can leave the child with a lock that nobody will release.  (The HIP
        runtime itself may be initialised before the fork, as in the reference, as long as the children never touch
        it.)  `worker_fn(client, *args)` runs in each child; returns the started processes."""
        if self._thread is not None:
            raise RuntimeError('launch() must fork the workers before the serving thread exists; call stop() first')
        ctx = mp_context or multiprocessing.get_context('fork')
        procs = [ctx.Process(target=worker_fn, args=(self.client(i),) + tuple(a)) for i, a in enumerate(worker_args)]
        for p in procs:
            p.start()
        self.start()
        return procs

    def start(self):
        if self._thread is None:
            # a serving phase starts from the parameters as they are NOW: the lean route's flattened copy is keyed on the parameters'
            # version counters, which writes through `p.data` (legacy optimizers, weight surgery) do not bump
            backend = getattr(self.policy_net, '_backend', [None])[0]
            if backend is not None and isinstance(getattr(backend, '__dict__', {}).get('_serve'), dict):
                backend.__dict__['_serve'].pop('version', None)
            self._stop.clear()
            self._thread = threading.Thread(target=self._loop, name='upamd-action-server', daemon=True)
            self._thread.start()
        return self

    def _loop(self):
        while not self._stop.is_set():
            try:
                self.serve_once(timeout=0.02)
            except Exception as exc:                # tell whoever is waiting, keep serving
                import traceback
                self.stats['errors'] = self.stats.get('errors', 0) + 1
                self.last_error = '%s: %s' % (type(exc).__name__, exc)
                self.last_traceback = traceback.format_exc()
                try:
                    pend = np.flatnonzero(self._seqs[:, 0] != self._seqs[:, 1])
                    if pend.size:
                        self._answer(pend, self.last_error)
                except Exception:
                    pass

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def close(self):
        import os
        self.stop()
        for fd in [self.doorbell] + self.wakes:
            try:
                os.close(fd)
            except OSError:
                pass
        self.wakes, self.doorbell = [], -1
        self._seqs = self._ctl = self._sizes = self._acts = None       # (views export the mapping's buffer)
        try:
            self.shm.close()
        except BufferError:
            pass
        try:
            self.shm.unlink()
        except FileNotFoundError:
            pass


# ------------------------------------------------------------------------------------------------ replay transport
class SharedArena:
    """A shared-memory arena one worker appends its rollout to: compact state records + the per-row scalars.

    Layout: int64 header [n_rows, n_bytes] | per-row table f64[cap_rows][6] = (record offset, record size, action0,
    action1, mask, reward) + exp in a second table | record bytes.  The learner reads it in place."""

    ROW = 7          # offset, size, action0, action1, mask, reward, exp

    def __init__(self, cap_rows, cap_bytes, name=None):
        self.cap_rows, self.cap_bytes = int(cap_rows), _align(int(cap_bytes))
        self._table_off = _align(16)
        self._data_off = _align(self._table_off + 8 * self.ROW * self.cap_rows)
        total = self._data_off + self.cap_bytes
        if name is None:
            self.shm = shared_memory.SharedMemory(create=True, size=total)
            self.owner = True
            self.header[:] = 0
        else:
            self.shm = shared_memory.SharedMemory(name=name)
            self.owner = False
        self._pinned = False
        self._closed = self._unlinked = False

    @property
    def name(self):
        return self.shm.name

    @property
    def header(self):
        return np.ndarray((2,), dtype=np.int64, buffer=self.shm.buf, offset=0)

    @property
    def table(self):
        return np.ndarray((self.cap_rows, self.ROW), dtype=np.float64, buffer=self.shm.buf, offset=self._table_off)

    @property
    def data(self):
        return np.ndarray((self.cap_bytes,), dtype=np.uint8, buffer=self.shm.buf, offset=self._data_off)

    def reset(self):
        self.header[:] = 0

    def pin(self):
        """Page-lock the arena in the LEARNER process (hipHostRegister): the pages the packer threads read once per iteration
        can no longer be swapped or migrated, and a device-side consumer may DMA from them directly.  Needs a GPU runtime;
        returns whether the arena is now registered.  Workers attached to the same shared memory are unaffected."""
        if self._pinned:
            return True
        if not torch.cuda.is_available():
            return False
        addr = np.frombuffer(self.shm.buf, dtype=np.uint8).ctypes.data
        rc = torch.cuda.cudart().cudaHostRegister(addr, self.shm.size, 0)
        self._pinned = int(rc) == 0
        return self._pinned

    def _unpin(self):
        if self._pinned:
            addr = np.frombuffer(self.shm.buf, dtype=np.uint8).ctypes.data
            torch.cuda.cudart().cudaHostUnregister(addr)
            self._pinned = False

    def append(self, state, action, mask, reward, exp):
        rec = _to_record(state)
        n, used = int(self.header[0]), int(self.header[1])
        if n >= self.cap_rows or used + rec.size > self.cap_bytes:
            raise MemoryError('rollout arena full (%d rows, %d bytes)' % (n, used))
        self.data[used:used + rec.size] = rec
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        self.table[n] = (used, rec.size, a[0], a[1] if a.size > 1 else 0.0, float(mask), float(reward), float(exp))
        self.header[1] = _align(used + rec.size, 8)
        self.header[0] = n + 1                      # published last

    def __len__(self):
        return int(self.header[0])

    def rows(self):
        """(states as zero-copy record views, actions f32[n,2], masks, rewards, exps) of what has been appended."""
        n = len(self)
        t = self.table[:n]
        data = self.data
        states = [data[int(o):int(o) + int(s)] for o, s in zip(t[:, 0], t[:, 1])]
        return states, t[:, 2:4].astype(np.float32), t[:, 4].copy(), t[:, 5].copy(), t[:, 6].copy()

    def record_addresses(self):
        """(addresses uint64[n], sizes int64[n]) of the appended records in this process's mapping of the arena"""
        t = self.table[:len(self)]
        base = np.frombuffer(self.shm.buf, dtype=np.uint8).ctypes.data + self._data_off
        return (np.uint64(base) + t[:, 0].astype(np.uint64)), t[:, 1].astype(np.int64)

    def close(self, unlink=None):
        """Unmap (and, for the owner, unlink) the arena.  Record views taken from ``data`` / ``rows()`` export the mapping's
        buffer: while one is alive the mapping is NOT torn down (``BufferError`` from the memoryview) -- the name is unlinked
        all the same, and the pages go when the last view does.  Returns True when the mapping is gone."""
        if self._closed:
            return True
        self._unpin()
        gone = True
        try:
            self.shm.close()
            self._closed = True
        except BufferError:                 # exported views are alive: leave the mapping to the garbage collector
            gone = False
        if (self.owner if unlink is None else unlink) and not self._unlinked:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
            self._unlinked = True
        return gone


class ArenaMemory:
    """Drop-in for khrylib.utils.memory.Memory on the worker side: ``push(state, action, mask, next_state, reward, exp)``
    appends to the worker's arena (``next_state`` is not kept: ``update_params`` never reads it)."""

    def __init__(self, arena):
        self.arena = arena

    def push(self, state, action, mask, next_state, reward, exp):
        self.arena.append(state, action, mask, reward, exp)

    def __len__(self):
        return len(self.arena)


class RecordBatch:
    """``TrajBatchDisc`` (urban_planning/utils/tools.py:4-16) over worker arenas and / or plain ``Memory`` objects, in
    worker order: ``states`` hold compact records (views into the arenas), the per-row arrays are stacked."""

    def __init__(self, memory_list, owns=()):
        """``owns``: arenas whose lifetime is THIS batch's -- the zero-copy record views in ``states`` and the raw addresses in
        ``states.addr`` point into their mappings, so they are closed (and unlinked) by ``close()`` or when the batch is
        collected, never while it can still be handed to ``update_params`` (a second ``sample()`` before the update, a batch
        kept for debugging or replay reuse must not read unmapped memory)."""
        import weakref
        states, actions, masks, rewards, exps, addrs, sizes = [], [], [], [], [], [], []
        bufs, which, offs = [], [], []          # all-arena batches: the record views are built on access (packer.RecordViews)
        lazy = True
        self._arenas = list(owns)
        self._held = {'states': None}        # shared with the finalizer: the views must go BEFORE the mappings they point into
        self._finalizer = weakref.finalize(self, _release_batch, self._held, self._arenas) if self._arenas else None
        self.closed = False
        for m in memory_list:
            arena = getattr(m, 'arena', m if isinstance(m, SharedArena) else None)
            if arena is not None:
                n = len(arena)
                t = arena.table[:n]
                s, a, mk, rw, ex = None, t[:, 2:4].astype(np.float32), t[:, 4].copy(), t[:, 5].copy(), t[:, 6].copy()
                ad, sz = arena.record_addresses()
                addrs.append(ad)
                sizes.append(sz)
                which.append(np.full(n, len(bufs), dtype=np.int64))
                offs.append(t[:, 0].astype(np.int64))
                bufs.append(arena.data)
                if not lazy:
                    s = arena.rows()[0]
            else:                                    # a khrylib Memory: rows of [state, action, mask, next_state, reward, exp]
                if lazy and bufs:                    # mixed batch: materialise what the arenas before this memory held
                    states += list(packer.RecordViews(bufs, np.concatenate(which), np.concatenate(offs), np.concatenate(sizes), np.concatenate(addrs)))
                lazy = False
                rows = m.sample()
                s = [_to_record(r[0]) for r in rows]
                a = np.stack([np.asarray(r[1], dtype=np.float32).reshape(-1)[:2] for r in rows]) if rows else np.zeros((0, 2), np.float32)
                mk = np.array([r[2] for r in rows], dtype=np.float64)
                rw = np.array([r[4] for r in rows], dtype=np.float64)
                ex = np.array([r[5] for r in rows], dtype=np.float64)
                addrs.append(np.array([r.ctypes.data for r in s], dtype=np.uint64))
                sizes.append(np.array([r.size for r in s], dtype=np.int64))
            if s is not None:
                states += s
            actions.append(a)
            masks.append(mk)
            rewards.append(rw)
            exps.append(ex)
        # (a list of record views that also carries the records' addresses: the packer never touches the T view objects)
        if lazy and bufs:
            self._held['states'] = packer.RecordViews(bufs, np.concatenate(which), np.concatenate(offs), np.concatenate(sizes), np.concatenate(addrs))
        else:
            self._held['states'] = packer.RecordList(states, np.concatenate(addrs) if addrs else None, np.concatenate(sizes) if sizes else None)
        del states, bufs
        self.actions = np.concatenate(actions) if actions else np.zeros((0, 2), np.float32)
        self.masks = np.concatenate(masks) if masks else np.zeros(0)
        self.rewards = np.concatenate(rewards) if rewards else np.zeros(0)
        self.exps = np.concatenate(exps) if exps else np.zeros(0)
        self.next_states = None

    def close(self):
        """Drop the record views and addresses, then unmap / unlink the arenas this batch owns.  A closed batch is empty: handing
        it to ``update_params`` raises ('empty replay') instead of reading freed memory."""
        self.closed = True
        if self._finalizer is not None:
            self._finalizer()              # runs _release_batch once; a later collection does nothing
        self._held['states'] = packer.RecordList([], None, None)

    @property
    def states(self):
        return self._held['states']

    def __len__(self):
        return len(self._held['states'])


def _release_batch(held, arenas):
    held['states'] = None                  # the zero-copy record views (and the raw address table) first ...
    for a in arenas:                       # ... then the mappings they pointed into
        try:
            a.close()
        except Exception:
            pass
