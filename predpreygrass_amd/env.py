"""`PredPreyGrass(config)`: the reference's environment class on top of the HIP step kernel.

Same constructor, methods, dict layouts, agent-id strings, dict ordering and error behaviour as
predpreygrass/non_evolutionary/base_environment/predpreygrass_rllib_env.py in the reference
(class PredPreyGrass, :17; reset :129; step :219; get_state_snapshot :768), so existing RLlib
training / evaluation scripts (tune_ppo_base_environment.py:72-86, random_policy.py:14-47,
evaluate_ppo_from_checkpoint_debug.py:99-302) can switch imports and run unchanged.

This is the per-environment *view*: it owns a `BatchedPredPreyGrass` of batch 1 (or attaches to
env index `index` of a shared batch) and converts the row tables into the reference's dicts.
All transition logic runs on the GPU; there is no CPU implementation behind this class.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _abi
from .batched import PREDATOR, PREY, BatchedPredPreyGrass, agent_name
from .config import resolve_config
from .placement import reference_placement

try:  # the reference subclasses RLlib's MultiAgentEnv (predpreygrass_rllib_env.py:13,17)
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _MultiAgentEnvBase  # type: ignore
except Exception:  # ray is an optional dependency; same fallback idea as the reference's
    # walls_occlusion/predpreygrass_rllib_env.py:20-38
    class _MultiAgentEnvBase:  # minimal stand-in with the attributes RLlib reads
        def __init__(self):
            pass

        def reset(self, *, seed=None, options=None):
            return None

        def close(self):
            pass

try:
    import gymnasium as _gym  # type: ignore

    def _box(shape):
        return _gym.spaces.Box(low=0.0, high=100.0, shape=shape, dtype=np.float64)

    def _discrete(n):
        return _gym.spaces.Discrete(n)
except Exception:
    class _Box:  # predpreygrass_rllib_env.py:88-89
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

        def sample(self):
            return np.random.uniform(self.low, self.high, self.shape).astype(self.dtype)

    class _Discrete:  # predpreygrass_rllib_env.py:107
        def __init__(self, n):
            self.n = int(n)

        def contains(self, x):
            return 0 <= int(x) < self.n

        def sample(self):
            return int(np.random.randint(self.n))

    def _box(shape):
        return _Box(0.0, 100.0, shape, np.float64)

    def _discrete(n):
        return _Discrete(n)


def _parse(name: str):
    kind, idx = name.rsplit("_", 1)
    if kind not in ("predator", "prey"):
        raise KeyError(name)
    return (PREDATOR if kind == "predator" else PREY), int(idx)


class PredPreyGrass(_MultiAgentEnvBase):
    def __init__(self, config=None, *, device=None, batched: BatchedPredPreyGrass | None = None, index: int = 0,
                 prey_capacity: int | None = None, _library=None):
        super().__init__()
        cfg = resolve_config(config)  # `config or config_env`, predpreygrass_rllib_env.py:20
        self.config = cfg
        for k, v in cfg.items():  # same attribute names as predpreygrass_rllib_env.py:22-61
            setattr(self, k, v)
        if batched is None:
            need = max(128, (self.n_initial_active_prey + 63) // 64 * 64)
            batched = BatchedPredPreyGrass(cfg, batch_size=1, device=device,
                                           prey_capacity=prey_capacity or min(256, need), _library=_library)
            index = 0
        self._b = batched
        self._i = int(index)
        self.possible_agents = [f"predator_{i}" for i in range(self.n_possible_predators)] + \
                               [f"prey_{j}" for j in range(self.n_possible_prey)]
        self.agents = [f"predator_{i}" for i in range(self.n_initial_active_predator)] + \
                      [f"prey_{j}" for j in range(self.n_initial_active_prey)]
        self.grass_agents = [f"grass_{k}" for k in range(self.initial_num_grass)]
        pshape = (batched.obs_pred.shape[2], self.predator_obs_range, self.predator_obs_range)
        qshape = (batched.obs_prey.shape[2], self.prey_obs_range, self.prey_obs_range)
        self._pspace, self._qspace, self._aspace = _box(pshape), _box(qshape), _discrete(9)
        self.observation_spaces = _SpaceDict(self._pspace, self._qspace, self)
        self.action_spaces = _SpaceDict(self._aspace, self._aspace, self)
        self.action_to_move_tuple = {a: (a // 3 - 1, a % 3 - 1) for a in range(9)}  # :96-106
        self.num_actions = 9
        self.grid_world_state_shape = (4, self.grid_size, self.grid_size)
        self.cumulative_rewards = {}
        self.agents_just_ate = set()
        self.current_step = 0
        self._records = []
        self._where = {}
        self._insertion_order = {}   # agent ids in the order the reference's dicts first saw them (an insertion-ordered set)
        self._tables = None

    # ------------------------------------------------------------------
    # reference API
    def reset(self, *, seed=None, options=None):
        """predpreygrass_rllib_env.py:129-217.  The initial cells are the ones the reference picks for `seed`
        (placement.reference_placement: same generator, same set order; seed=None draws OS entropy like the
        reference).  ``options={"placement": (pred_xy, prey_xy, grass_xy)}`` starts from given positions;
        ``options={"placement": "device"}`` draws them on the GPU instead (Philox keyed by `seed`)."""
        placement = (options or {}).get("placement") if isinstance(options, dict) else None
        b = self._b
        if b.batch_size != 1:
            raise RuntimeError("reset() of a view into a shared batch: reset the BatchedPredPreyGrass instead")
        if isinstance(placement, str) and placement == "device":
            b.reset(seed=0 if seed is None else int(seed))
        else:
            if placement is None:
                nP, nQ = self.n_initial_active_predator, self.n_initial_active_prey
                cells = reference_placement(self.grid_size, nP + nQ + self.initial_num_grass, seed)
                placement = (cells[:nP], cells[nP:nP + nQ], cells[nP + nQ:])   # :185-187
            p, q, g = placement
            b.set_placement(np.asarray(p).reshape(1, -1, 2), np.asarray(q).reshape(1, -1, 2), np.asarray(g).reshape(1, -1, 2))
        self.cumulative_rewards = {}
        self._insertion_order = {}
        obs = self._collect(after_reset=True)[0]
        return obs, {}

    def _stage(self, action_dict):
        """Validate an action dict like the reference would and write it into this env's row of the
        batch's action tensor.  Returns (acting slots in dict order, dict order == row order)."""
        b, i = self._b, self._i
        where = self._where   # name -> (type, row) of the agents alive after the last call (_collect)
        a = b.stage_actions(i)     # this env's row of the pinned host mirror of the action tensor
        a[:] = _abi.ACTION_NONE
        last = [-1, -1]
        slots, acts = [], []
        in_row_order = True
        truncated_call = self.current_step >= self.max_steps
        cp = b.pred_capacity
        for name, act in action_dict.items():
            tr = where.get(name)
            if tr is None:
                if truncated_call:
                    continue  # the reference returns before touching action_dict (:228-238)
                raise KeyError(name)  # dead / unknown agent: predpreygrass_rllib_env.py:246/249
            act = int(act)
            if not 0 <= act <= 8:
                raise KeyError(act)  # action_to_move_tuple[action], predpreygrass_rllib_env.py:502
            ty, row = tr
            slots.append(row if ty == PREDATOR else cp + row)
            acts.append(act)
            if row < last[ty]:
                in_row_order = False
            last[ty] = row
        if slots:
            a[slots] = acts
        return slots, in_row_order

    def _ranks(self, slots):
        """Position of each acting row within its type's action sequence (ppg_step_ordered): uint8 [S].  Needed for EVERY env of a
        launch that goes through the explicit-order kernel, also for the envs whose dict happens to be in row order."""
        b = self._b
        cp = b.pred_capacity
        ranks = np.zeros((b.S,), dtype=np.uint8)
        count = [0, 0]
        for s_ in slots:
            ty = PREDATOR if s_ < cp else PREY
            ranks[s_] = count[ty]
            count[ty] += 1
        return torch.from_numpy(ranks)

    def step(self, action_dict):
        """predpreygrass_rllib_env.py:219-473."""
        b, i = self._b, self._i
        if b.batch_size != 1:
            raise RuntimeError("step() of a view into a shared batch: use VectorPredPreyGrass.step")
        slots, in_row_order = self._stage(action_dict)
        b.upload_actions()   # one host->device copy; the step and the fetch behind it are ordered on the same stream
        if in_row_order:
            b.step()
        else:
            b.step(act_rank=self._ranks(slots)[None].to(b.device))
        return self._collect(after_reset=False)

    def close(self):
        pass

    # ------------------------------------------------------------------
    def _collect(self, after_reset, tables=None, obs=None):
        b, i = self._b, self._i
        if tables is None:   # ONE device->host copy: this env's tables and its observation rows in use (ppg_fetch)
            t, fp, fq = b.fetch(i, 1)
            op, oq = fp[0], fq[0]
        else:
            t = {k: v[i:i + 1] for k, v in tables.items()}
            op, oq = obs[0][i], obs[1][i]
        self._tables = t
        es = t["env_state"][0]
        status = int(es[_abi.ENV_STATUS])
        if status & (_abi.STATUS_PRED_OVERFLOW | _abi.STATUS_PREY_OVERFLOW):
            raise RuntimeError("agent row capacity exceeded: construct the env with a larger prey_capacity")
        if status & _abi.STATUS_FAILED_SPAWN:
            raise TypeError("no free cell for a newborn (the reference fails at predpreygrass_rllib_env.py:401-405)")
        obs, rew, term, trunc = {}, {}, {}, {}
        # one float64 copy of each species' blocks in use (the fetch buffer is reused by the next call); the dict entries are its rows
        ops = (np.array(op, dtype=np.float64), np.array(oq, dtype=np.float64))
        cp = b.pred_capacity
        es = es.tolist()
        nP, nQ, newP, newQ = es[_abi.ENV_N_PRED_ROWS], es[_abi.ENV_N_PREY_ROWS], es[_abi.ENV_N_PRED_NEW], es[_abi.ENV_N_PREY_NEW]
        ids, fls, rws, cum = (t[k][0].tolist() for k in ("row_id", "row_flags", "row_reward", "row_cumrew"))
        order, cumulative, ate, where, recs = self._insertion_order, self.cumulative_rewards, set(), {}, []
        DIED, TRUNC, ATE = _abi.ROW_DIED, _abi.ROW_TRUNC, _abi.ROW_ATE
        # the reference's dict order (BatchedPredPreyGrass.records): predator survivors, prey survivors, predator newborns, prey newborns
        for ty, lo, hi in ((PREDATOR, 0, nP - newP), (PREY, 0, nQ - newQ), (PREDATOR, nP - newP, nP), (PREY, nQ - newQ, nQ)):
            base, fmt, rows = (0, "predator_%d", ops[0]) if ty == PREDATOR else (cp, "prey_%d", ops[1])
            for row in range(lo, hi):
                s = base + row
                name, fl, r = fmt % ids[s], fls[s], rws[s]
                te, tr = bool(fl & DIED), bool(fl & TRUNC)
                recs.append((name, ty, row, r, te, tr))
                obs[name] = rows[row]
                rew[name], term[name], trunc[name] = r, te, tr
                cumulative[name] = cum[s]
                order.setdefault(name)
                if fl & ATE:
                    ate.add(name)
                if not te:
                    where[name] = (ty, row)
        fl = es[_abi.ENV_FLAGS]
        was_trunc_call = bool(fl & _abi.ENVF_TRUNC_ALL)
        self._records = recs
        self._where = where
        self.current_step = es[_abi.ENV_STEP]
        self.current_num_predators = es[_abi.ENV_N_PRED_ALIVE]
        self.current_num_prey = es[_abi.ENV_N_PREY_ALIVE]
        self._next_predator_idx = es[_abi.ENV_NEXT_PRED_ID]
        self._next_prey_idx = es[_abi.ENV_NEXT_PREY_ID]
        self.agents_just_ate = ate
        names = [r[0] for r in recs]
        # self.agents: reset leaves it in creation order (:143); every full step ends with .sort() (:468);
        # the truncation call returns before the sort but the list is already sorted by then.
        self.agents = names if (after_reset or (fl & _abi.ENVF_LIST_IS_ROW_ORDER)) else sorted(names)
        self._pending_removal = [r[0] for r in recs if r[4]]
        if after_reset:
            return obs, {}
        term["__all__"] = bool(fl & _abi.ENVF_TERM_ALL)
        trunc["__all__"] = was_trunc_call
        return obs, rew, term, trunc, {}

    def _live(self):
        b, t = self._b, self._tables
        cp = b.pred_capacity
        out = {}
        for name, ty, row, _, te, _ in self._records:
            if not te:
                out[name] = row if ty == PREDATOR else cp + row
        return out

    # viewer-facing attributes (random_policy.py:32-39, evaluate_ppo_from_checkpoint_debug.py:190-197)
    @property
    def agent_positions(self):
        live, t = self._live(), self._tables
        order = [n for n in self._insertion_order if n in live]  # dict insertion order of the reference
        return {n: (int(t["row_xy"][0][live[n]]) >> 8, int(t["row_xy"][0][live[n]]) & 255) for n in order}

    @property
    def agent_energies(self):
        live, t = self._live(), self._tables
        return {n: float(t["row_energy"][0][live[n]]) for n in self._insertion_order if n in live}

    @property
    def agent_parent(self):
        """child id -> parent id of the kickback variant (…plus_kickback/predpreygrass_rllib_env.py:86-91)."""
        live, t = self._live(), self._tables
        out = {}
        for n in self._insertion_order:
            if n in live and int(t["row_parent"][0][live[n]]) >= 0:
                out[n] = n.rsplit("_", 1)[0] + "_%d" % int(t["row_parent"][0][live[n]])
        return out

    @property
    def predator_positions(self):
        return {k: v for k, v in self.agent_positions.items() if k.startswith("predator")}

    @property
    def prey_positions(self):
        return {k: v for k, v in self.agent_positions.items() if k.startswith("prey")}

    @property
    def grass_positions(self):
        t = self._tables
        return {f"grass_{k}": (int(t["grass_xy"][0][k]) >> 8, int(t["grass_xy"][0][k]) & 255)
                for k in range(self.initial_num_grass)}

    @property
    def grass_energies(self):
        t = self._tables
        return {f"grass_{k}": float(t["grass_energy"][0][k]) for k in range(self.initial_num_grass)}

    @property
    def grid_world_state(self):
        return self._b.export_grid()[self._i].cpu().numpy()

    # seasonal variant (base_environment_seasonal/predpreygrass_rllib_env.py:224-234); 1.0 for the base env
    def _current_season_multiplier(self) -> float:
        length = int(self.config.get("season_length_steps", 0) or 0)
        if length <= 0:
            return 1.0
        phase = (self.current_step // length) % 2
        return float(self.config.get("season_high_multiplier", 1.0) if phase == 0
                     else self.config.get("season_low_multiplier", 1.0))

    def set_grass_energy(self, grass, energy):
        """White-box state surgery used by tests in the reference's style
        (base_environment_seasonal/tests/test_seasonal_grass_regrowth.py:66 assigns grass_energies[...])."""
        k = int(str(grass).rsplit("_", 1)[1])
        self._b.grass_energy[self._i, k] = float(energy)
        self._tables = self._b.host_tables(self._i)

    def _get_observation(self, agent):
        """predpreygrass_rllib_env.py:511-526 (called externally at evaluate_ppo_from_checkpoint_debug.py:182)."""
        ty, _ = _parse(agent)
        live = self._live()
        if agent not in live:
            raise KeyError(agent)
        row = live[agent] if ty == PREDATOR else live[agent] - self._b.pred_capacity
        self._b.observe()
        t = self._b.obs_pred if ty == PREDATOR else self._b.obs_prey
        return t[self._i, row].cpu().numpy().astype(np.float64)

    # snapshot / restore (predpreygrass_rllib_env.py:768-804)
    def get_state_snapshot(self):
        b, i = self._b, self._i
        snap = {
            "current_step": self.current_step,
            "agent_positions": self.agent_positions,
            "agent_energies": self.agent_energies,
            "predator_positions": self.predator_positions,
            "prey_positions": self.prey_positions,
            "grass_positions": self.grass_positions,
            "grass_energies": self.grass_energies,
            "grid_world_state": self.grid_world_state,
            "agents": list(self.agents),
            "cumulative_rewards": dict(self.cumulative_rewards),
            "current_num_predators": self.current_num_predators,
            "current_num_prey": self.current_num_prey,
            "agents_just_ate": set(self.agents_just_ate),
            "pending_removal": list(self._pending_removal),
            "next_predator_idx": self._next_predator_idx,
            "next_prey_idx": self._next_prey_idx,
            "agent_parent": self.agent_parent,
            # device state of this implementation
            "_device_state": b.export_state(i),   # ppg_export_state: the versioned POD image of include/ppg.h
            "_records": list(self._records),
            "_insertion_order": list(self._insertion_order),
        }
        return snap

    def restore_state_snapshot(self, snapshot):
        b, i = self._b, self._i
        b.import_state(snapshot["_device_state"], i)
        self.cumulative_rewards = dict(snapshot["cumulative_rewards"])
        self._insertion_order = dict.fromkeys(snapshot["_insertion_order"])
        b.observe()
        saved_agents = list(snapshot["agents"])
        self._collect(after_reset=False)
        self.agents = saved_agents
        self.cumulative_rewards = dict(snapshot["cumulative_rewards"])


class _SpaceDict(dict):
    """observation_spaces / action_spaces keyed by every possible agent id
    (predpreygrass_rllib_env.py:92-94,108) without materialising 4000 entries eagerly."""

    def __init__(self, pred_space, prey_space, env):
        super().__init__()
        self._p, self._q, self._env = pred_space, prey_space, env

    def __missing__(self, key):
        ty, idx = _parse(key)
        n = self._env.n_possible_predators if ty == PREDATOR else self._env.n_possible_prey
        if not 0 <= idx < n:
            raise KeyError(key)
        return self._p if ty == PREDATOR else self._q

    def __contains__(self, key):
        try:
            self.__missing__(key)
            return True
        except (KeyError, ValueError):
            return False

    def keys(self):
        return list(self._env.possible_agents)

    def __iter__(self):
        return iter(self._env.possible_agents)

    def __len__(self):
        return len(self._env.possible_agents)

    def items(self):
        return [(k, self[k]) for k in self._env.possible_agents]

    def values(self):
        return [self[k] for k in self._env.possible_agents]

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


class VectorPredPreyGrass:
    """`num_envs` reference-shaped environments behind ONE kernel launch per step.

    `envs[i]` is a `PredPreyGrass` view (attributes, snapshot, `_get_observation` ...) of env i of a shared
    `BatchedPredPreyGrass`; `step()` takes one action dict per env and returns one reference-style 5-tuple per
    env.  With `auto_reset=True` an env whose episode ended is reset by the next `step()` (its action dict
    is ignored for that call and the returned observations are those of the new episode, rewards 0)."""

    def __init__(self, config=None, num_envs=8, device=None, seed=0, auto_reset=False, prey_capacity=128,
                 _library=None):
        self.num_envs = int(num_envs)
        self.auto_reset = bool(auto_reset)
        self.batch = BatchedPredPreyGrass(config, batch_size=self.num_envs, device=device, seed=seed,
                                          prey_capacity=prey_capacity, _library=_library)
        self.envs = [PredPreyGrass(config, batched=self.batch, index=i) for i in range(self.num_envs)]

    def _collect_all(self, after_reset):
        b = self.batch
        tables, obs_p, obs_q = b.fetch()   # every env's tables and observation rows in use: ONE device->host copy
        obs = (obs_p, obs_q)
        out = []
        for i, e in enumerate(self.envs):
            was_reset = bool(int(tables["env_state"][i][_abi.ENV_FLAGS]) & _abi.ENVF_WAS_RESET)
            if was_reset and not after_reset:
                e.cumulative_rewards, e._insertion_order = {}, {}
                o, _ = e._collect(True, tables, obs)
                out.append((o, {a: 0.0 for a in o}, {**{a: False for a in o}, "__all__": False},
                            {**{a: False for a in o}, "__all__": False}, {"reset": True}))
            else:
                out.append(e._collect(after_reset, tables, obs))
        return out

    def reset(self, seed=None):
        """Device-side placement for every env (env i uses Philox key seed + i); returns [(obs, {}), ...]."""
        self.batch.reset(seed=seed)
        for e in self.envs:
            e.cumulative_rewards, e._insertion_order = {}, {}
        return self._collect_all(after_reset=True)

    def step(self, action_dicts):
        if len(action_dicts) != self.num_envs:
            raise ValueError("one action dict per env")
        b = self.batch
        staged = [None] * self.num_envs
        all_in_order = True
        for i, (e, ad) in enumerate(zip(self.envs, action_dicts)):
            done = bool(int(e._tables["env_state"][0][_abi.ENV_FLAGS]) & _abi.ENVF_DONE)
            if self.auto_reset and done:
                b.stage_actions(i)[:] = _abi.ACTION_NONE  # ignored: this call resets the env
                continue
            staged[i], in_order = e._stage(ad)
            all_in_order = all_in_order and in_order
        b.upload_actions()   # ONE host->device copy for all envs
        if all_in_order:
            b.step(auto_reset=self.auto_reset)
        else:   # one env out of row order sends the whole launch through ppg_step_ordered: every env needs its ranks
            ranks = torch.zeros((self.num_envs, b.S), dtype=torch.uint8)
            for i, (e, slots) in enumerate(zip(self.envs, staged)):
                if slots is not None:
                    ranks[i] = e._ranks(slots)
            b.step(auto_reset=self.auto_reset, act_rank=ranks.to(b.device))
        return self._collect_all(after_reset=False)

    def close(self):
        self.batch.close()


def env_creator(config):
    """tune_ppo_base_environment.py:72-73 / random_policy.py:6-7 of the reference."""
    return PredPreyGrass(config)
