"""SubBatchedPredPreyGrass: one GPU's envs as a few independent sub-batches on their own HIP streams.

Why: one launch steps every env of a batch with one co-resident wavefront per env, so all waves do
their load / movement phases together (memory pipe idle) and then their observation stores together
(memory pipe saturated), and the launch ends with its slowest CU.  Two or three sub-batches in
flight on separate streams are out of phase with each other: while one writes observations the other
moves agents, and workgroups of one kernel fill the CUs the other has already left.  Measured on
MI355X, 4096 envs: 1 stream 118 us, 2 streams 91 us per full step.

Each sub-batch is an ordinary `BatchedPredPreyGrass`; env b of the whole batch is env
b - offset[k] of sub-batch k.  Sub-batches never wait for each other -- exactly the asynchronous
vector-env pattern RL frameworks use; call `synchronize()` (or use the per-sub-batch streams) before
reading tensors from another stream.

Placement (`placement_candidates`): where the driver puts a sub-batch's observation tensors in HBM decides how fast the step's
scattered 1 KB pieces can be written -- on one MI355X, in one process, with one kernel, env groups that differ only in their
buffers step in 62-66 us or in 76-81 us, persistently, while a linear fill of the same tensors runs at the same 6.3 TB/s
(`tools/exp_placement.py`, profiles/r03/e_placement_experiments.txt; the time follows the observation tensors when new handles
and row tables are put on them).  Physical placement cannot be asked for, but it can be measured: with `placement_candidates=K`
the constructor builds up to K candidate sets of sub-batches, steps each for a few milliseconds, keeps the fastest and frees
the others (K sets of buffers exist while they are measured: 1.8 GB each at 4096 default-config envs).  Results never depend on it: the caller's `reset()` re-creates every env's state.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from .batched import BatchedPredPreyGrass
from .distributed import shard_range


class SubBatchedPredPreyGrass:
    def __init__(self, config=None, batch_size=4096, n_sub=2, device="cuda:0", seed=0, env_class=BatchedPredPreyGrass,
                 placement_candidates=1, placement_setup=None, placement_target_us=None, placement_max_candidates=12, **kw):
        """env_class: BatchedPredPreyGrass (base family) or red_queen.BatchedRedQueen (second generation).
        placement_candidates: > 1 = build that many candidate buffer sets and keep the one that steps fastest (module docstring);
        placement_setup(env): called on every candidate sub-batch before it is measured (e.g. set_walls);
        placement_target_us: if none of the candidates probes below this many microseconds per step, further candidates are drawn one
        at a time until one does or `placement_max_candidates` have been tried (some boxes hand out mostly slow placements)."""
        self.device = torch.device(device)
        self.batch_size = int(batch_size)
        self.offsets = [shard_range(self.batch_size, k, n_sub) for k in range(n_sub)]
        cuda = self.device.type == "cuda"   # (the CPU case exists only for the emulated-kernel tests)
        self.streams = [torch.cuda.Stream(device=self.device) if cuda else None for _ in self.offsets]

        def build():
            subs = [env_class(config, batch_size=hi - lo, device=device, seed=seed + lo, **kw) for lo, hi in self.offsets]
            for e in subs:   # the sub-batches run concurrently: kernel selection should look at all of them together
                e._lib.ppg_set_envs_in_flight(e._handle, self.batch_size)
                if placement_setup is not None:
                    placement_setup(e)
            return subs

        self.subs = build()
        self.placement_probe_us = None
        if cuda and int(placement_candidates) > 1:
            # all candidates exist side by side while they are measured (one that is freed first would hand its memory to the next)
            sets = [self.subs] + [build() for _ in range(int(placement_candidates) - 1)]
            self.placement_probe_us = []
            for subs in sets:
                self.subs = subs
                self._forget_handles()
                self.placement_probe_us.append(self._probe_placement())
            while placement_target_us is not None and min(self.placement_probe_us) > float(placement_target_us) \
                    and len(sets) < int(placement_max_candidates):
                sets.append(build())
                self.subs = sets[-1]
                self._forget_handles()
                self.placement_probe_us.append(self._probe_placement())
            keep = min(range(len(sets)), key=lambda k: self.placement_probe_us[k])
            self.subs = sets[keep]
            self._forget_handles()
            for k, subs in enumerate(sets):
                if k != keep:
                    for e in subs:
                        e.close()
            del sets, subs

    def _forget_handles(self):
        for name in ("_c_handles", "_c_streams", "_c_own_actions"):
            if hasattr(self, name):
                delattr(self, name)

    def _probe_placement(self, warm_steps=640, probe_steps=192):
        """Microseconds per full step of the current sub-batches on their current buffers: device reset, `warm_steps` untimed steps
        of the device-side random policy with auto-reset (the population has to grow: right after a reset an env writes a third of
        the rows it writes later, and the placements do not differ yet), then `probe_steps` timed ones (about 60 ms in all)."""
        import time
        self.reset()
        for _ in range(warm_steps):
            self.step(random_actions=True, auto_reset=True)
        self.synchronize()
        t0 = time.perf_counter()
        for _ in range(probe_steps):
            self.step(random_actions=True, auto_reset=True)
        self.synchronize()
        return (time.perf_counter() - t0) / probe_steps * 1e6

    def reset(self, seed=None):
        for k, (e, s) in enumerate(zip(self.subs, self.streams)):
            sd = None if seed is None else seed + self.offsets[k][0]
            if s is None:
                e.reset(seed=sd)
                continue
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                e.reset(seed=sd)
        return self

    def step(self, actions=None, random_actions=False, auto_reset=False):
        """One transition of every env; `actions` is a list of per-sub-batch int8 tensors (or None ->
        each sub-batch's own `.actions` tensor).  All sub-batches are launched by ONE C call
        (ppg_step_many), sub-batch k on stream k."""
        n = len(self.subs)
        if self.streams[0] is None:   # CPU (emulated kernel, tests)
            for k, e in enumerate(self.subs):
                e.step(None if actions is None else actions[k], random_actions=random_actions, auto_reset=auto_reset)
            return self
        if not hasattr(self, "_c_handles"):
            self._c_handles = (C.c_void_p * n)(*[e._handle for e in self.subs])
            self._c_streams = (C.c_void_p * n)(*[s.cuda_stream for s in self.streams])
            self._c_own_actions = (C.c_void_p * n)(*[e.actions.data_ptr() for e in self.subs])
        flags = (_abi.STEP_RANDOM_ACTIONS if random_actions else 0) | (_abi.STEP_AUTO_RESET if auto_reset else 0)
        if random_actions:
            acts = None
        elif actions is None:
            acts = self._c_own_actions
        else:
            for k, a in enumerate(actions):
                if a.dtype != torch.int8 or tuple(a.shape) != (self.subs[k].batch_size, self.subs[k].S) or \
                        not a.is_contiguous() or a.device != self.subs[k].device:
                    raise ValueError("actions[k] must be a contiguous int8 tensor [B_k, S] on the env's device")
            acts = (C.c_void_p * n)(*[a.data_ptr() for a in actions])
        # whatever the caller enqueued on the current stream (a policy writing the action tensors, edits of env_state)
        # happens before the step kernels of the sub-batch streams
        self._order_behind_current()
        lib = self.subs[0]._lib
        rc = lib.ppg_step_many(self._c_handles, n, acts, flags, self._c_streams)
        if rc != 0:
            msgs = [lib.ppg_last_error(e._handle).decode() for e in self.subs]
            raise RuntimeError(f"ppg_step_many failed ({rc}): {'; '.join(m for m in msgs if m)}")
        return self

    def _order_behind_current(self):
        """Sub-batch streams wait for the work the caller has enqueued on torch's current stream.  If that stream is idle
        (hipStreamQuery: everything given to it so far has completed -- the normal case when the actions come from the
        device-side random policy) there is nothing to wait for and no barrier packets are put into the queues: they cost
        ~3 us per launch, 10 % of a step."""
        cur = torch.cuda.current_stream(self.device)
        if cur.query():
            return
        ev = cur.record_event()
        for s in self.streams:
            s.wait_event(ev)

    def wait(self, stream=None):
        """Make `stream` (default: torch's current stream) wait for everything the sub-batch streams have been given so
        far -- call it before a consumer on that stream reads observations / rewards (no host synchronisation)."""
        if self.streams[0] is None:
            return self
        stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        for s in self.streams:
            stream.wait_stream(s)
        return self

    def rebalance(self):
        """ppg_rebalance of every sub-batch on its own stream (scheduling only; reads env_state, so it is ordered behind
        the caller's current stream like a step)."""
        if self.streams[0] is not None:
            self._order_behind_current()
        for e, s in zip(self.subs, self.streams):
            e.rebalance(stream=s)
        return self

    def rollout(self, n_steps, random_actions=True, auto_reset=False):
        """`n_steps` fused transitions per sub-batch (one launch each, on its own stream)."""
        for e, s in zip(self.subs, self.streams):
            e.rollout(n_steps, random_actions=random_actions, auto_reset=auto_reset, stream=s)
        return self

    def synchronize(self):
        for s in self.streams:
            if s is not None:
                s.synchronize()

    def locate(self, b):
        for k, (lo, hi) in enumerate(self.offsets):
            if lo <= b < hi:
                return self.subs[k], b - lo
        raise IndexError(b)
