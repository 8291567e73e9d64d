"""SURVEY.md section 8f rows 1-3 bound into the class ``bind_reference_agent`` returns (opt-in, per process):

``UPAMD_ROLLOUT=server``
    ``sample()`` (khrylib/rl/agents/agent.py:75-100) keeps the networks on the GPU and forks ``nthreads`` env workers
    that run the reference's OWN ``sample_worker`` (urban_planning/agents/urban_planning_agent.py:49-91) unchanged --
    only two names it looks up are different in the child: ``self.policy_net`` is a ``rollout.ActionClient`` (same
    ``select_action(x, mean_action)``; the learner process answers every pending request with ONE batched HIP forward
    and samples on the device) and ``Memory()`` makes a ``rollout.ArenaMemory`` (same ``push``; the transition lands
    as a compact record in the worker's shared-memory arena).  Nothing is pickled but the worker's ``LoggerRL``; the
    batch handed to ``update_params`` is a ``rollout.RecordBatch`` whose states are views into the arenas, which
    ``packer.pack_replay`` reads in place.  In the reference worker 0 is the learner process itself; here the learner
    serves, so all ``nthreads`` workers are children: worker 0 is seeded from the learner's RNG streams (which advance
    once per ``sample()``), workers 1.. seed themselves exactly as the reference does (``seed_worker``, agent.py:66-69).
    Actions are drawn by ``Categorical.sample`` on the GPU's generator: the same distribution as the reference's CPU
    draw, not the same stream ("re-baselined").

    ``eval_agent()`` (urban_planning_agent.py:402-467) runs the reference's own body in one more forked child behind a
    client: the model never makes the CPU <-> GPU round trip of ``to_cpu`` (:406).

``UPAMD_EVAL=overlap`` (needs the server)
    the greedy evaluation episode becomes one more client of the SAME serving phase as the sampling workers, i.e. it runs
    concurrently with them on the weights sampling uses.  ``optimize_policy`` (:225-246) is untouched, so its
    ``eval_agent`` call after the update receives that log: ``eval_R_eps`` of iteration i then describes the weights the
    iteration STARTED with (one update behind the reference's number, whose T_eval it removes from the iteration).  The
    evaluated weights are kept (a CPU ``state_dict``), and ``save_checkpoint`` writes ``best.p`` from THEM, so a best
    checkpoint still pairs a reward with the weights that earned it.

Checkpoints (on by default; ``UPAMD_CKPT_OPTIMIZER=0`` writes / resumes exactly as the reference does): ``save_checkpoint`` adds ``'hip_optimizer'`` -- ``PPOUpdater.state_dict()``: Adam
moments, per-group step counts, the first-step clipping flag -- to every file the reference's ``save_checkpoint``
(:172-194) has just written, and ``load_checkpoint`` (:153-170) of a FRESH process restores it, so a resumed run
continues the Adam trajectory instead of restarting it (the reference omits optimizer state).  Files without the key load
as before; a load in the middle of a run (``freeze_land_use``, :215-222) leaves Adam alone, as the reference's does.
"""
import math
import os
import pickle
import time
import traceback

import numpy as np
import torch

from . import packer


def _child_log_path(diag_dir, kind, pid):
    return os.path.join(diag_dir, '%s%d.log' % (kind, pid))


def _rollout_child(client, agent, kind, pid, arena_spec, n, mean_action, seeds, out_q, diag_dir=None):
    """Body of one forked env worker.  The child never touches the (GPU) modules or the HIP runtime."""
    from . import rollout
    arena = None
    if diag_dir:
        # whatever ends this child, the learner can say why: stderr and faulthandler (fatal signals, and a stack dump of every
        # thread if the worker is still here after UPAMD_ROLLOUT_STUCK_S seconds) go to a per-worker file the parent quotes on failure
        try:
            import faulthandler
            fh = open(_child_log_path(diag_dir, kind, pid), 'w', buffering=1)
            os.dup2(fh.fileno(), 2)
            faulthandler.enable(file=fh, all_threads=True)
            faulthandler.dump_traceback_later(float(os.environ.get('UPAMD_ROLLOUT_STUCK_S', '90')), repeat=True, file=fh)
        except Exception:
            pass
    # One OpenMP thread, like the reference's workers (khrylib/rl/agents/agent.py:12 sets OMP_NUM_THREADS=1 before anything forks):
    # the learner's OpenMP pool does not exist in a forked child, and a parallel region entered there waits for its threads for ever
    torch.set_num_threads(1)
    try:
        agent.policy_net = client               # what sample_worker / eval_agent call select_action on
        agent.sample_modules = []               # to_test / to_cpu (agent.py:79-80, urban_planning_agent.py:404-406) see nothing
        if kind == 'sample':
            arena = rollout.SharedArena(arena_spec[1], arena_spec[2], name=arena_spec[0])
            worker = agent.sample_worker
            getattr(worker, '__func__', worker).__globals__['Memory'] = lambda: rollout.ArenaMemory(arena)
            if pid == 0 and seeds is not None:
                torch.manual_seed(int(seeds[0]))
                np.random.seed(int(seeds[1]))
            memory, logger = agent.sample_worker(pid, None, n, mean_action)
            out_q.put((kind, pid, len(memory), logger, None))
        else:
            out_q.put((kind, pid, 0, agent._upamd_eval_reference(n, mean_action), None))
    except BaseException as exc:                # report instead of dying silently: the learner is waiting on the queue
        out_q.put((kind, pid, -1, None, '%s: %s\n%s' % (type(exc).__name__, exc, traceback.format_exc())))
    finally:
        try:
            client.close()
            if arena is not None:
                arena.close(unlink=False)
        except Exception:
            pass
        # Leave WITHOUT the interpreter's teardown: this is a fork of a process that holds an initialised GPU runtime, its allocator
        # and their threads' state -- exit handlers and destructors running against that copy can crash or hang the child after its
        # work is done (and a non-zero exit code of a worker that has long reported must not be read as a failure: see _upamd_serve).
        try:
            out_q.close()
            out_q.join_thread()         # the report is in the pipe
        finally:
            os._exit(0)


class RolloutMixin:
    """``sample`` / ``eval_agent`` through ``rollout.ActionServer`` + ``rollout.SharedArena`` (module docstring)."""

    # ---- switches
    @staticmethod
    def _upamd_rollout_mode():
        mode = os.environ.get('UPAMD_ROLLOUT', 'reference')
        if mode not in ('reference', 'server'):
            raise ValueError("UPAMD_ROLLOUT must be 'reference' or 'server'")
        return mode

    @staticmethod
    def _upamd_eval_overlap():
        mode = os.environ.get('UPAMD_EVAL', 'serial')
        if mode not in ('serial', 'overlap'):
            raise ValueError("UPAMD_EVAL must be 'serial' or 'overlap'")
        return mode == 'overlap'

    def _upamd_eval_reference(self, num_samples=1, mean_action=True, visualize=False):
        """The reference's own eval_agent body (urban_planning_agent.py:402-467), past every mixin."""
        return super(RolloutMixin, self).eval_agent(num_samples, mean_action, visualize)

    def _upamd_arena_caps(self, rows):
        specs = getattr(self.cfg, 'state_encoder_specs', None) or {}
        pad_n, pad_e = int(specs.get('max_num_nodes', 1000)), int(specs.get('max_num_edges', 3000))
        rec = packer.record_bytes_bound(pad_n, pad_e, int(self.node_dim), int(self.numerical_feature_size))
        # a worker appends whole episodes until it has `rows` steps (urban_planning_agent.py:54): room for the overshoot.
        # Shared memory is committed page by page as it is written, so the worst-case byte bound costs nothing up front
        cap_rows = int(rows) + max(int(getattr(self.cfg, 'max_sequence_length', 0) or 0), 256)
        return cap_rows, cap_rows * rec

    # ---- one serving phase: fork the children, serve until every one of them has reported
    def _upamd_serve(self, jobs, arenas, seeds=None, timeout_s=None):
        """jobs: [(kind, pid, n, mean_action)], arenas: {job index: SharedArena}.  Returns the children's reports in job
        order: (kind, pid, rows, logger)."""
        from . import rollout
        import multiprocessing
        import shutil
        import tempfile
        ctx = multiprocessing.get_context('fork')
        out_q = ctx.Queue()
        diag_dir = tempfile.mkdtemp(prefix='upamd_rollout_')       # per-worker stderr / faulthandler files (quoted on failure)
        server = rollout.ActionServer(self.policy_net, len(jobs), slot_bytes=int(os.environ.get('UPAMD_ROLLOUT_SLOT', 1 << 20)),
                                      mp_context=ctx)
        args = []
        for j, (kind, pid, n, mean) in enumerate(jobs):
            a = arenas.get(j)
            args.append((self, kind, pid, (a.name, a.cap_rows, a.cap_bytes) if a is not None else None, n, mean, seeds, out_q, diag_dir))
        procs = server.launch(_rollout_child, args, ctx)       # forks FIRST, then starts the serving thread
        reports, failed = {}, None
        if timeout_s is None:
            timeout_s = float(os.environ.get('UPAMD_ROLLOUT_TIMEOUT_S', '0')) or None
        deadline = None if timeout_s is None else time.time() + timeout_s
        try:
            while len(reports) < len(jobs) and failed is None:
                try:
                    kind, pid, rows, logger, err = out_q.get(timeout=1.0)
                except Exception:               # queue.Empty: is everyone still alive, is the server still serving?
                    # (only a worker that has NOT reported counts: how a child that has delivered its rows leaves is its own business)
                    dead = [p for p, (kind, pid, _, _) in zip(procs, jobs)
                            if p.exitcode is not None and (kind, pid) not in reports]
                    if dead and out_q.empty():
                        failed = 'an env worker died (exit code %s) without reporting' % dead[0].exitcode
                    elif server.last_error and server._thread is not None and not server._thread.is_alive():
                        failed = 'the action server stopped: %s' % server.last_error
                    elif deadline is not None and time.time() > deadline:
                        failed = 'rollout timed out after %.0f s' % timeout_s
                    continue
                if err is not None:
                    failed = 'env worker %s %d failed:\n%s' % (kind, pid, err)
                    break
                reports[(kind, pid)] = (kind, pid, rows, logger)
        finally:
            if failed is not None and any(p.is_alive() for p in procs):
                for p in procs:                 # a stuck worker: have faulthandler write every thread's stack before it is ended
                    if p.is_alive():
                        try:
                            os.kill(p.pid, 6)   # SIGABRT -> faulthandler dump into the worker's file
                        except OSError:
                            pass
            for p in procs:
                p.join(timeout=5.0 if failed is None else 1.0)
                if p.is_alive():
                    p.terminate()
            self._upamd_server_stats = dict(server.stats)
            server_error, server_tb = server.last_error, server.last_traceback
            server.close()
            if failed is not None:
                failed += self._upamd_failure_report(jobs, procs, reports, diag_dir, server_error, server_tb)
            if os.environ.get('UPAMD_ROLLOUT_KEEP_DIAG') != '1':
                shutil.rmtree(diag_dir, ignore_errors=True)
        if failed is not None:
            raise RuntimeError('UPAMD_ROLLOUT=server: %s' % failed)
        return [reports[(kind, pid)] for kind, pid, _, _ in jobs]

    @staticmethod
    def _upamd_failure_report(jobs, procs, reports, diag_dir, server_error, server_tb):
        """Everything known about a failed serving phase, for the exception message: per worker whether it reported, its exit code
        (negative = the signal that killed it) and the tail of its stderr / faulthandler file; the serving thread's last error."""
        lines = ['', '---- serving phase post-mortem ----']
        for p, (kind, pid, _, _) in zip(procs, jobs):
            lines.append('worker %s %d: reported=%s exitcode=%s' % (kind, pid, (kind, pid) in reports, p.exitcode))
            try:
                with open(_child_log_path(diag_dir, kind, pid)) as fh:
                    text = '\n'.join(l for l in fh.read().splitlines() if not l.startswith('Extension modules:'))
                if text.strip():
                    # faulthandler writes the innermost frames first: keep the head, and the tail for whatever came last
                    if len(text) > 3200:
                        text = text[:2600] + '\n[...]\n' + text[-500:]
                    lines.append('  stderr / faulthandler:\n    ' + text.replace('\n', '\n    '))
            except OSError:
                pass
        lines.append('action server: last_error=%s' % server_error)
        if server_tb:
            lines.append(server_tb)
        return '\n'.join(lines)

    def _upamd_release_arenas(self):
        """Close every batch ``sample()`` has handed out that is still alive (end of training, tests).  NOT called by ``sample()``:
        a batch owns its arenas -- they are unmapped when the batch is closed or collected, never underneath a batch the caller
        still holds (a second ``sample()`` before ``update_params``, a batch kept for debugging or replay reuse)."""
        for b in list(getattr(self, '_upamd_batches', None) or []):
            b.close()

    # ---- Agent.sample (khrylib/rl/agents/agent.py:75-100)
    def sample(self, num_samples, mean_action=False, nthreads=None):
        if self._upamd_rollout_mode() != 'server':
            return super().sample(num_samples, mean_action, nthreads)
        from . import rollout
        if nthreads is None:
            nthreads = self.num_threads
        t_start = time.time()
        for m in self.sample_modules:           # to_test (:79); the modules stay where they are (no to_cpu, :80)
            m.train(False)
        thread_num_samples = int(math.floor(num_samples / nthreads))
        cap_rows, cap_bytes = self._upamd_arena_caps(thread_num_samples)
        arenas = [rollout.SharedArena(cap_rows, cap_bytes) for _ in range(nthreads)]
        if os.environ.get('UPAMD_ROLLOUT_PIN') == '1':
            for a in arenas:
                a.pin()
        # worker 0 is a child here (the learner serves): it gets its seeds from the learner's streams, which therefore advance
        # once per call -- so do the states workers 1.. derive their seeds from (seed_worker, agent.py:66-69)
        seeds = (int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), int(np.random.randint(2 ** 31 - 1)))
        jobs = [('sample', i, thread_num_samples, mean_action) for i in range(nthreads)]
        overlap = self._upamd_eval_overlap() and getattr(self, 'training', True) and not mean_action
        if overlap:
            jobs.append(('eval', nthreads, 1, True))
            self._upamd_eval_sd = {k: v.detach().to('cpu', copy=True) for k, v in self.actor_critic_net.state_dict().items()}
        try:
            with torch.no_grad():
                reports = self._upamd_serve(jobs, {i: arenas[i] for i in range(nthreads)}, seeds)
        except BaseException:
            for a in arenas:                    # no batch will ever own them
                a.close()
            raise
        if overlap:
            self._upamd_eval_ahead = reports[-1][3]
        memories = [rollout.ArenaMemory(a) for a in arenas]
        for (kind, pid, rows, _), m in zip(reports[:nthreads], memories):
            if rows != len(m):
                raise RuntimeError('env worker %d reported %d rows, its arena holds %d' % (pid, rows, len(m)))
        traj_batch = rollout.RecordBatch(memories, owns=arenas)      # the arenas live exactly as long as the batch does
        del memories
        if getattr(self, '_upamd_batches', None) is None:
            import weakref
            self._upamd_batches = weakref.WeakSet()
        self._upamd_batches.add(traj_batch)
        logger = self.logger_cls.merge([r[3] for r in reports[:nthreads]], **self.logger_kwargs)
        logger.sample_time = time.time() - t_start
        return traj_batch, logger

    # ---- UrbanPlanningAgent.eval_agent (urban_planning_agent.py:402-467)
    def eval_agent(self, num_samples=1, mean_action=True, visualize=False):
        if self._upamd_rollout_mode() != 'server' or visualize:
            return super().eval_agent(num_samples, mean_action, visualize)
        ahead = getattr(self, '_upamd_eval_ahead', None)
        if ahead is not None and num_samples == 1 and mean_action:
            self._upamd_eval_ahead = None       # the episode that ran next to this iteration's sampling workers
            return ahead
        t_start = time.time()
        for m in self.sample_modules:
            m.train(False)
        self._upamd_eval_sd = None              # evaluated == current weights
        with torch.no_grad():
            log = self._upamd_serve([('eval', 0, num_samples, mean_action)], {})[0][3]
        log.sample_time = time.time() - t_start
        return log


class CheckpointMixin:
    """Optimizer state in the reference's checkpoint files (module docstring)."""

    def _upamd_cp_path(self, checkpoint):
        cfg = self.cfg          # (:155-160)
        return '%s/iteration_%04d.p' % (cfg.model_dir, checkpoint) if isinstance(checkpoint, int) else '%s/%s.p' % (cfg.model_dir, checkpoint)

    def load_checkpoint(self, checkpoint, restore_best_rewards):
        start = super().load_checkpoint(checkpoint, restore_best_rewards)
        try:
            with open(self._upamd_cp_path(checkpoint), 'rb') as fh:
                state = pickle.load(fh).get('hip_optimizer')
        except (OSError, AttributeError):
            state = None
        if os.environ.get('UPAMD_CKPT_OPTIMIZER', '1') == '0':
            state = None                        # resume as the reference does: fresh Adam moments, first-step clip re-armed
        # the updater is built lazily (its hyper-parameters are set by AgentPPO.__init__, which runs AFTER load_checkpoint,
        # urban_planning_agent.py:38-47) and Adam's buffers live on the GPU: parked here, applied by the next update_params
        up = getattr(self, '_upamd_updater', None)
        if up is not None and up.m is not None:
            # a load in the middle of a run (freeze_land_use re-loads best.p, :215-222): the reference keeps its optimizer
            # OBJECT, i.e. Adam carries on from the current moments -- so does this
            return start
        self._upamd_pending_opt = state
        if up is not None:
            up.pending_state = state
        return start

    def _upamd_cp_written(self, iteration, best):
        """The files the reference's ``save_checkpoint`` (:172-194) writes for this call, from ITS conditions -- not guessed from
        modification times."""
        cfg, out = self.cfg, []
        interval = getattr(cfg, 'save_model_interval', 0) or 0
        if interval > 0 and (iteration + 1) % interval == 0:
            out.append('{}/iteration_{:04d}.p'.format(cfg.model_dir, iteration + 1))
        if best:
            out.append('{}/best.p'.format(cfg.model_dir))
            out.append('{}/best_reward{:.2f}_iteration_{:04d}.p'.format(cfg.model_dir, self.best_rewards, iteration + 1))
        return out

    def save_checkpoint(self, iteration):
        written = self._upamd_cp_written(iteration, bool(getattr(self, 'save_best_flag', False)))
        swap = getattr(self, '_upamd_eval_sd', None) if (getattr(self, 'save_best_flag', False) and
                                                         os.environ.get('UPAMD_EVAL') == 'overlap') else None
        if swap is None:
            super().save_checkpoint(iteration)
        else:
            # UPAMD_EVAL=overlap: the reward that set save_best_flag belongs to the weights the iteration started with.  The
            # periodic file gets the current weights, the best files the evaluated ones
            net, cfg = self.actor_critic_net, self.cfg
            self.save_best_flag = False
            super().save_checkpoint(iteration)
            current = {k: v.detach().clone() for k, v in net.state_dict().items()}
            interval, cfg.save_model_interval = cfg.save_model_interval, 0
            try:
                net.load_state_dict(swap)
                self.save_best_flag = True
                super().save_checkpoint(iteration)
            finally:
                cfg.save_model_interval = interval
                net.load_state_dict(current)
        if os.environ.get('UPAMD_CKPT_OPTIMIZER', '1') == '0':
            return                              # the reference's files as the reference writes them (a resume then restarts Adam, as its does)
        up = getattr(self, '_upamd_updater', None)
        state = up.state_dict() if (up is not None and up.m is not None) else getattr(self, '_upamd_pending_opt', None)
        if state is None:
            return
        for p in written:
            if not os.path.exists(p):
                continue                        # (a stand-in class with its own file policy: nothing to extend)
            with open(p, 'rb') as fh:
                cp = pickle.load(fh)
            if isinstance(cp, dict) and 'actor_critic_dict' in cp:
                cp['hip_optimizer'] = state
                tmp = p + '.upamd_tmp'
                with open(tmp, 'wb') as fh:     # never a truncated checkpoint: written beside it, then renamed over it
                    pickle.dump(cp, fh)
                    fh.flush()
                    os.fsync(fh.fileno())
                os.replace(tmp, p)
