"""Multi-GPU: one process per GPU, envs sharded by batch index, no cross-env state.

The transition needs no collective at all (envs are independent).  What a
central consumer (a learner on rank 0, a logger) may need is the *returned observation dict* of every
env; `ObservationGatherer` moves exactly that -- the observations of the rows in use plus their ids /
rewards / flags -- with ONE variable-length all-gather per tensor over RCCL (backend "nccl" on ROCm;
"gloo" in the CPU tests).  Rows are compacted first so that padding slots never cross xGMI.

Bandwidth note: float64 observations are ~2.4 KB per agent; at ~36 agents per env a
4096-env shard emits ~360 MB per step, so a synchronous gather over xGMI (7 links x ~153 GB/s per
GPU) caps the aggregate far below the compute rate -- gather only when a single consumer really
needs every observation, and prefer float32 observations on the wire.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _abi
from .batched import BatchedPredPreyGrass


def shard_range(total_envs: int, rank: int, world: int):
    """GPU r owns envs [lo, hi): contiguous, sizes differ by at most one."""
    base, rem = divmod(total_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ObservationGatherer:
    def __init__(self, env: BatchedPredPreyGrass, group=None):
        self.env = env
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        dev = env.device
        self._rows_p = torch.arange(env.pred_capacity, device=dev)[None, :]
        self._rows_q = torch.arange(env.prey_capacity, device=dev)[None, :]

    def pack_local(self):
        """Compacted view of this shard's last call: observations and per-row tables of the rows in
        use, env-major, predators and prey separately, plus the per-env row counts."""
        e = self.env
        es = e.env_state
        nP = es[:, _abi.ENV_N_PRED_ROWS:_abi.ENV_N_PRED_ROWS + 1]
        nQ = es[:, _abi.ENV_N_PREY_ROWS:_abi.ENV_N_PREY_ROWS + 1]
        mp = self._rows_p < nP
        mq = self._rows_q < nQ
        cp = e.pred_capacity
        # row indices of the slots in use; whole-row index_select copies (a boolean-mask gather of the 5-D observation
        # tensors is ~30x slower)
        ip = mp.reshape(-1).nonzero().squeeze(1)
        iq = mq.reshape(-1).nonzero().squeeze(1)

        def rows(t, idx):
            return t.reshape((t.shape[0] * t.shape[1],) + tuple(t.shape[2:])).index_select(0, idx)
        out = {
            "obs_pred": rows(e.obs_pred, ip), "obs_prey": rows(e.obs_prey, iq),
            "id_pred": rows(e.row_id[:, :cp].contiguous(), ip), "id_prey": rows(e.row_id[:, cp:].contiguous(), iq),
            "reward_pred": rows(e.row_reward[:, :cp].contiguous(), ip), "reward_prey": rows(e.row_reward[:, cp:].contiguous(), iq),
            "flags_pred": rows(e.row_flags[:, :cp].contiguous(), ip), "flags_prey": rows(e.row_flags[:, cp:].contiguous(), iq),
            "env_state": es.clone(),
        }
        return out

    def _all_gather_var(self, t: torch.Tensor, counts):
        """All-gather tensors whose first dimension differs per rank (counts known on every rank)."""
        n_max = max(counts)
        if t.shape[0] == n_max:
            pad = t.contiguous()
        else:   # (the tail is never read: the receivers slice by counts)
            pad = torch.empty((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[: t.shape[0]] = t
        out = torch.empty((self.world * n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return [out[r * n_max: r * n_max + counts[r]] for r in range(self.world)]

    def gather(self, local=None):
        """Every rank receives every shard's packed observation data (list indexed by rank).

        local: a dict from an earlier `pack_local()`.  Packing copies the rows in use out of the env's buffers, so the
        env can already run its next step while this gather moves the copies: pack on the env's stream, enqueue the next
        step, then call `gather(local)` on another stream (`bench.py`'s obs_gather_overlapped leg)."""
        if local is None:
            local = self.pack_local()
        cnt = torch.tensor([local["obs_pred"].shape[0], local["obs_prey"].shape[0], self.env.batch_size],
                           dtype=torch.int64, device=self.env.device)
        allc = torch.empty((self.world * 3,), dtype=torch.int64, device=self.env.device)
        dist.all_gather_into_tensor(allc, cnt, group=self.group)
        allc = allc.cpu().view(self.world, 3).tolist()   # the one host sync of the gather
        counts = {"pred": [c[0] for c in allc], "prey": [c[1] for c in allc], "state": [c[2] for c in allc]}
        res = {}
        for k, v in local.items():
            res[k] = self._all_gather_var(v, counts[k.rsplit("_", 1)[1]])
        self.last_bytes = sum(t.numel() * t.element_size() for k, v in res.items() for t in v)
        return res
