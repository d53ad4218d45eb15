"""Second-generation PredPreyGrass ("red queen"): two agent types per species, a 5x5 action range for type 2,
distance-proportional move cost, energy caps and transfer efficiency, reproduction cooldown / chance gate /
mutation.  Host plumbing for the gen-2 kernels of libppg_hip.so (include/ppg.h: ppg_create_gen2,
ppg_step_uniforms); all environment logic runs in predpreygrass_amd/csrc/ppg_kernel.h.

"RQ:n" = line n of predpreygrass/non_evolutionary/red_queen/predpreygrass_rllib_env.py in the reference.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _abi
from .batched import BatchedPredPreyGrass, lexkey

POOLS = ("type_1_predator", "type_2_predator", "type_1_prey", "type_2_prey")  # creation order, RQ:125-133
_POSSIBLE_KEYS = ("n_possible_type_1_predators", "n_possible_type_2_predators", "n_possible_type_1_prey",
                  "n_possible_type_2_prey")  # RQ:56-59
_INITIAL_KEYS = tuple(f"n_initial_active_{p}" for p in POOLS)  # RQ:61-64

# red_queen/config/config_env_base.py:1-81 (values restated)
config_env_base = {
    "max_steps": 1000, "grid_size": 25, "num_obs_channels": 4, "predator_obs_range": 7, "prey_obs_range": 9,
    "type_1_action_range": 3, "type_2_action_range": 5,
    "reward_predator_catch_prey": {"type_1_predator": 0.0, "type_2_predator": 0.0},
    "reward_prey_eat_grass": {"type_1_prey": 0.0, "type_2_prey": 0.0},
    "reward_predator_step": {"type_1_predator": 0.0, "type_2_predator": 0.0},
    "reward_prey_step": {"type_1_prey": 0.0, "type_2_prey": 0.0},
    "penalty_prey_caught": {"type_1_prey": 0.0, "type_2_prey": 0.0},
    "reproduction_reward_predator": {"type_1_predator": 10.0, "type_2_predator": 10.0},
    "reproduction_reward_prey": {"type_1_prey": 10.0, "type_2_prey": 10.0},
    "energy_loss_per_step_predator": 0.06, "energy_loss_per_step_prey": 0.02,
    "predator_creation_energy_threshold": 12.0, "prey_creation_energy_threshold": 8.0,
    "move_energy_cost_factor": 0.01, "initial_energy_predator": 6.0, "initial_energy_prey": 3.0,
    "n_possible_type_1_predators": 2000, "n_possible_type_2_predators": 0,
    "n_possible_type_1_prey": 1600, "n_possible_type_2_prey": 1600,
    "n_initial_active_type_1_predator": 12, "n_initial_active_type_2_predator": 0,
    "n_initial_active_type_1_prey": 10, "n_initial_active_type_2_prey": 10,
    "mutation_rate_predator": 0.05, "mutation_rate_prey": 0.05,
    "initial_num_grass": 100, "initial_energy_grass": 2.0, "energy_gain_per_step_grass": 0.1,
    "verbose_engagement": False, "verbose_movement": False, "verbose_decay": False, "verbose_reproduction": False,
    "debug_mode": False,
    "max_energy_gain_per_grass": 1.5, "max_energy_gain_per_prey": 5.0, "max_energy_predator": 20.0,
    "max_energy_prey": 14.0, "max_energy_grass": 2.0,
    "reproduction_cooldown_steps": 5, "reproduction_chance_predator": 0.95, "reproduction_chance_prey": 0.95,
    "energy_transfer_efficiency": 0.9, "reproduction_energy_efficiency": 0.9,
}

# `config.get(key, default)` fallbacks of the reference (RQ:30-86 and inside step(): RQ:305,510,598-605,665-672,696,
# 700-701,754)
_IN_CODE_DEFAULTS = {
    "max_steps": 10000, "grid_size": 10, "num_obs_channels": 4, "predator_obs_range": 7, "prey_obs_range": 5,
    "n_possible_type_1_predators": 25, "n_possible_type_2_predators": 25,
    "n_possible_type_1_prey": 25, "n_possible_type_2_prey": 25,
    "n_initial_active_type_1_predator": 6, "n_initial_active_type_2_predator": 0,
    "n_initial_active_type_1_prey": 8, "n_initial_active_type_2_prey": 0,
    "initial_num_grass": 25, "type_1_action_range": 3, "type_2_action_range": 5, "reproduction_cooldown_steps": 10,
    "reward_predator_catch_prey": 0.0, "reward_prey_eat_grass": 0.0, "reward_predator_step": 0.0,
    "reward_prey_step": 0.0, "penalty_prey_caught": 0.0,
    "reproduction_reward_predator": 10.0, "reproduction_reward_prey": 10.0,
    "energy_loss_per_step_predator": 0.15, "energy_loss_per_step_prey": 0.05,
    "predator_creation_energy_threshold": 12.0, "prey_creation_energy_threshold": 8.0,
    "initial_energy_predator": 5.0, "initial_energy_prey": 3.0, "initial_energy_grass": 2.0,
    "energy_gain_per_step_grass": 0.2, "move_energy_cost_factor": 0.01,
    "max_energy_gain_per_prey": math.inf, "max_energy_gain_per_grass": math.inf,
    "max_energy_predator": math.inf, "max_energy_prey": math.inf, "max_energy_grass": math.inf,
    "energy_transfer_efficiency": 1.0, "reproduction_energy_efficiency": 1.0,
    "reproduction_chance_predator": 1.0, "reproduction_chance_prey": 1.0,
    "mutation_rate_predator": 0.1, "mutation_rate_prey": 0.1,
}


def resolve_config(config: dict) -> dict:
    """The red_queen env needs an explicit config (RQ:17-18); absent keys get the reference's in-code defaults."""
    if config is None:
        raise ValueError("Environment config must be provided explicitly.")  # RQ:17-18
    out = dict(_IN_CODE_DEFAULTS)
    out.update(config)
    if int(out["num_obs_channels"]) != 4:
        raise ValueError("num_obs_channels must be 4 (border, predator, prey, grass)")
    return out


def typed_value(raw, species: str, type_: int) -> float:
    """_get_type_specific (RQ:1099-1106): a scalar, or a dict keyed by 'type_<t>_<species>' prefixes."""
    if isinstance(raw, dict):
        name = f"type_{type_}_{species}"
        for k in raw:
            if name.startswith(k):
                return float(raw[k])
        raise KeyError(f"Type-specific key '{name}' not found")
    return float(raw)


def agent_name(pool: int, idx: int) -> str:
    return f"{POOLS[int(pool)]}_{int(idx)}"


def parse_agent(name: str) -> tuple[int, int]:
    kind, idx = name.rsplit("_", 1)
    return POOLS.index(kind), int(idx)


def make_row_id(seq, type2, idx):
    """row_id of include/ppg.h: creation number << 17 | (type - 1) << 16 | k."""
    return (np.asarray(seq, dtype=np.int64) << 17 | np.asarray(type2, dtype=np.int64) << 16 | np.asarray(idx, dtype=np.int64)).astype(np.int32)


def split_row_id(row_id):
    """-> (creation number, type - 1, k)."""
    v = np.asarray(row_id).astype(np.int64) & 0xFFFFFFFF
    return v >> 17, (v >> 16) & 1, v & 0xFFFF


class BatchedRedQueen(BatchedPredPreyGrass):
    """Tensor API of the second-generation env: same row layout as `BatchedPredPreyGrass` (predator rows, then prey
    rows; survivors in self.agents order followed by newborns), with the agent's type and creation number packed into
    ``row_id``.  Observations default to float32, the reference's dtype (RQ:352-355)."""

    def __init__(self, config, batch_size=1, device=None, obs_dtype=torch.float32, prey_capacity=128, seed=0,
                 walls=False, _library=None, obs_spread=0):
        """walls=True selects the walls_occlusion env ("WO"): observation channel 0 shows walls, wall / line-of-sight
        rules for moves and observations (config keys include_visibility_channel, respect_los_for_movement,
        mask_observation_with_visibility), per-agent move infos in `row_info`.  Walls are given with `set_walls`."""
        cfg = resolve_config(config)
        self.config = cfg
        self._obs_spread = int(obs_spread)   # (BatchedPredPreyGrass: observation tensors on spread physical pages)
        self.walls = bool(walls)
        self.vis_channel = self.walls and bool(cfg.get("include_visibility_channel", False))   # WO:104
        self.obs_channels = 5 if self.vis_channel else 4
        self.batch_size = int(batch_size)
        self._init_device(device, obs_dtype, _library)
        self.grid_size = int(cfg["grid_size"])
        self.Rp, self.Rq = int(cfg["predator_obs_range"]), int(cfg["prey_obs_range"])
        self.n_initial = [int(cfg[k]) for k in _INITIAL_KEYS]
        self.n_possible = [int(cfg[k]) for k in _POSSIBLE_KEYS]
        self.P0, self.Q0 = self.n_initial[0] + self.n_initial[1], self.n_initial[2] + self.n_initial[3]
        self.n_grass = int(cfg["initial_num_grass"])
        self.cooldown = int(cfg["reproduction_cooldown_steps"])
        self.action_ranges = (int(cfg["type_1_action_range"]), int(cfg["type_2_action_range"]))
        self._alloc_buffers(prey_capacity)

        c = _abi.PpgConfigGen2()
        c.abi_version = _abi.ABI_VERSION
        c.grid_size, c.predator_obs_range, c.prey_obs_range = self.grid_size, self.Rp, self.Rq
        c.max_steps = int(cfg["max_steps"])
        for p in range(4):
            c.n_possible[p], c.n_initial[p] = self.n_possible[p], self.n_initial[p]
        c.n_grass = self.n_grass
        c.pred_capacity, c.prey_capacity, c.grass_capacity = self.pred_capacity, self.prey_capacity, self.grass_capacity
        c.obs_dtype = self._abi_obs_dtype()
        c.type_1_action_range, c.type_2_action_range = self.action_ranges
        c.reproduction_cooldown_steps = self.cooldown
        for name in _abi.GEN2_TYPED:
            species = "predator" if name in ("reward_predator_catch_prey", "reward_predator_step",
                                             "reproduction_reward_predator") else "prey"
            for t in (1, 2):
                getattr(c, name)[t - 1] = typed_value(cfg[name], species, t)
        for name in _abi.GEN2_SCALARS:
            setattr(c, name, float(cfg[name]))
        c.walls = int(self.walls)
        c.include_visibility_channel = int(self.vis_channel)
        c.respect_los_for_movement = int(self.walls and bool(cfg.get("respect_los_for_movement", False)))          # WO:106
        c.mask_observation_with_visibility = int(self.walls and bool(cfg.get("mask_observation_with_visibility", False)))  # WO:111
        self._create_handle(c, self._lib.ppg_create_gen2)
        self.set_seeds(seed)

    # ------------------------------------------------------------------
    def set_placement(self, pred_xy, prey_xy, grass_xy, episode=0):
        """reset() with a given placement (RQ:151-195): predators / prey in creation order (type 1 then type 2),
        grass in id order; arrays [B,P0,2], [B,Q0,2], [B,n_grass,2] of (x, y)."""
        B, G = self.batch_size, self.grid_size
        pred = np.asarray(pred_xy, dtype=np.int64).reshape(B, self.P0, 2)
        prey = np.asarray(prey_xy, dtype=np.int64).reshape(B, self.Q0, 2)
        grass = np.asarray(grass_xy, dtype=np.int64).reshape(B, self.n_grass, 2)
        for a in (pred, prey, grass):
            if a.size and (a.min() < 0 or a.max() >= G):
                raise ValueError("position outside the grid")
        gcell = grass[..., 0] * G + grass[..., 1]
        for b in range(B):
            if len(np.unique(gcell[b])) != self.n_grass:
                raise ValueError("grass positions must be unique")
        S, cp = self.S, self.pred_capacity
        xy = np.zeros((B, S), dtype=np.int16)
        en = np.zeros((B, S), dtype=np.float64)
        ids = np.zeros((B, S), dtype=np.int32)
        keys = np.zeros((B, S), dtype=np.int32)
        fl = np.zeros((B, S), dtype=np.uint8)
        lastrep = np.full((B, S), -self.cooldown, dtype=np.int32)           # RQ:999
        xy[:, : self.P0] = (pred[..., 0] << 8 | pred[..., 1]).astype(np.int16)
        xy[:, cp: cp + self.Q0] = (prey[..., 0] << 8 | prey[..., 1]).astype(np.int16)
        en[:, : self.P0] = float(self.config["initial_energy_predator"])
        en[:, cp: cp + self.Q0] = float(self.config["initial_energy_prey"])
        for lo, n1, n, seq0 in ((0, self.n_initial[0], self.P0, 0), (cp, self.n_initial[2], self.Q0, self.P0)):
            i = np.arange(n)
            t2 = (i >= n1).astype(np.int64)
            idx = np.where(t2 == 1, i - n1, i)
            ids[:, lo: lo + n] = make_row_id(seq0 + i, t2, idx)
            keys[:, lo: lo + n] = (t2 * _abi.KEY_TYPE2 + lexkey(idx)).astype(np.int32)
        for b in range(B):  # grid[type, pos] = energy in creation order (RQ:168-180): the last writer owns the cell
            for lo, arr in ((0, pred[b]), (cp, prey[b])):
                owner = {}
                for i, (x, y) in enumerate(arr):
                    owner[(int(x), int(y))] = i
                for i in owner.values():
                    fl[b, lo + i] = _abi.ROW_OWNS
        es = np.zeros((B, _abi.ENV_WORDS), dtype=np.int32)
        es[:, _abi.ENV_N_PRED_ROWS] = self.P0
        es[:, _abi.ENV_N_PREY_ROWS] = self.Q0
        es[:, _abi.ENV_NEXT_PRED_ID] = self.n_initial[0]                    # RQ:129
        es[:, _abi.ENV_NEXT_PRED_ID_T2] = self.n_initial[1]
        es[:, _abi.ENV_NEXT_PREY_ID] = self.n_initial[2]
        es[:, _abi.ENV_NEXT_PREY_ID_T2] = self.n_initial[3]
        es[:, _abi.ENV_N_PRED_ALIVE] = self.P0
        es[:, _abi.ENV_N_PREY_ALIVE] = self.Q0
        es[:, _abi.ENV_FLAGS] = _abi.ENVF_WAS_RESET | _abi.ENVF_LIST_IS_ROW_ORDER
        es[:, _abi.ENV_EPISODE] = int(episode)
        gxy = np.zeros((B, self.grass_capacity), dtype=np.int16)
        ge = np.zeros((B, self.grass_capacity), dtype=np.float64)
        gxy[:, : self.n_grass] = (grass[..., 0] << 8 | grass[..., 1]).astype(np.int16)
        ge[:, : self.n_grass] = float(self.config["initial_energy_grass"])
        for t, a in ((self.row_xy, xy), (self.row_energy, en), (self.row_id, ids), (self.row_key, keys),
                     (self.row_flags, fl), (self.row_lastrep, lastrep), (self.env_state, es), (self.grass_xy, gxy),
                     (self.grass_energy, ge)):
            t.copy_(torch.from_numpy(a))
        self.row_cumrew.zero_()
        self.row_reward.zero_()
        self.observe()
        return self

    def set_walls(self, wall_xy, per_env=False, precompute_visibility=True):
        """Wall cells (walls variant): a list of (x, y) shared by all envs, or with per_env=True one such list per env.
        They stay in place across device resets until replaced; call before `set_placement` / `reset`."""
        if not self.walls:
            raise RuntimeError("set_walls needs walls=True")
        B, G = self.batch_size, self.grid_size
        if per_env and len(wall_xy) != B:
            raise ValueError("per_env=True needs one wall list per env")
        per_env = list(wall_xy) if per_env else [wall_xy] * B
        bits = np.zeros((B, self.wall_bits.shape[1]), dtype=np.uint32)
        for b, cells in enumerate(per_env):
            a = np.asarray(cells, dtype=np.int64).reshape(-1, 2)
            if a.size and (a.min() < 0 or a.max() >= G):
                raise ValueError("wall outside the grid")
            c = a[:, 0] * G + a[:, 1]
            np.bitwise_or.at(bits[b], c >> 5, (np.uint32(1) << (c & 31).astype(np.uint32)))
            n_free = G * G - len(np.unique(c))
            if n_free < self.P0 + self.Q0 + self.n_grass:   # the device reset places every entity on its own free cell
                raise ValueError(f"env {b}: {n_free} cells are free of walls but {self.P0 + self.Q0 + self.n_grass} "
                                 "entities have to be placed")
        self.wall_bits.copy_(torch.from_numpy(bits.view(np.int32)))
        if precompute_visibility:   # the per-cell line-of-sight masks the observations read from now on (ppg_walls_changed)
            self._check(self._lib.ppg_walls_changed(self._handle, self._stream()), "ppg_walls_changed")
        return self

    def step(self, actions=None, random_actions=False, auto_reset=False, act_rank=None, uniforms=None, stream=None):
        """One transition of every env (RQ:197-299).

        uniforms (optional float64 [B,U]): the values the reference's ``self.rng.random()`` would return for each env
        in this call, in draw order (RQ:701,708); ``env_state[:, ENV_DRAWS]`` tells how many were used.  Without it
        the draws come from Philox on the device."""
        if uniforms is None:
            return super().step(actions, random_actions=random_actions, auto_reset=auto_reset, act_rank=act_rank,
                                stream=stream)
        if uniforms.dtype != torch.float64 or uniforms.dim() != 2 or uniforms.shape[0] != self.batch_size or \
                uniforms.device != self.device or not uniforms.is_contiguous():
            raise ValueError("uniforms must be a contiguous float64 tensor [B,U] on the env's device")
        flags = (_abi.STEP_RANDOM_ACTIONS if random_actions else 0) | (_abi.STEP_AUTO_RESET if auto_reset else 0)
        ptr = None
        if not random_actions:
            if actions is None:
                actions = self.actions
            if actions.dtype != torch.int8 or tuple(actions.shape) != (self.batch_size, self.S) or \
                    actions.device != self.device or not actions.is_contiguous():
                raise ValueError(f"actions must be a contiguous int8 tensor [{self.batch_size},{self.S}] on {self.device}")
            ptr = C.c_void_p(actions.data_ptr())
        rank_ptr = None
        if act_rank is not None:
            if act_rank.dtype != torch.uint8 or tuple(act_rank.shape) != (self.batch_size, self.S) or \
                    act_rank.device != self.device or not act_rank.is_contiguous():
                raise ValueError("act_rank must be a contiguous uint8 tensor [B,S] on the env's device")
            rank_ptr = C.c_void_p(act_rank.data_ptr())
        self._check(self._lib.ppg_step_uniforms(self._handle, ptr, rank_ptr, C.c_void_p(uniforms.data_ptr()),
                                                int(uniforms.shape[1]), flags, self._stream(stream)), "ppg_step_uniforms")
        return self

    def rollout(self, n_steps, actions=None, random_actions=False, auto_reset=False, stream=None):
        """`n_steps` transitions in ONE launch: the fused form of the cooperative step kernel (the handle's wave plan must be
        cooperative with four waves -- the default on a full GPU, else `set_wave_plan(4, 0, 2)`; not for the walls variant).  Same
        result as `n_steps` calls of step(): actions from the device-side uniform random policy or an int8 tape [n_steps, B, S],
        reproduction uniforms from the device's Philox streams."""
        return super().rollout(n_steps, actions=actions, random_actions=random_actions, auto_reset=auto_reset, stream=stream)

    # ------------------------------------------------------------------
    def host_tables(self, b=None):
        sl = slice(None) if b is None else slice(b, b + 1)
        names = ["row_xy", "row_energy", "row_id", "row_key", "row_cumrew", "row_flags", "row_reward", "row_lastrep",
                 "row_info", "env_state", "grass_xy", "grass_energy"]
        return {n: getattr(self, n)[sl].cpu().numpy() for n in names}

    def records(self, b, tables=None):
        """The returned dicts of env b's last call as an ordered list of (name, species, row, reward, terminated,
        truncated) in the reference's dict order: self.agents order (creation order right after reset, otherwise the
        sorted id strings: type_1_predator*, type_1_prey*, type_2_predator*, type_2_prey*, RQ:270) followed by the
        newborns of the call in birth order (RQ:729)."""
        t = tables if tables is not None else self.host_tables(b)
        i = 0 if tables is None else b
        es = t["env_state"][i]
        nP, nQ = int(es[_abi.ENV_N_PRED_ROWS]), int(es[_abi.ENV_N_PREY_ROWS])
        newP, newQ = int(es[_abi.ENV_N_PRED_NEW]), int(es[_abi.ENV_N_PREY_NEW])
        flags = int(es[_abi.ENV_FLAGS])
        cp = self.pred_capacity
        creation_order = bool(flags & (_abi.ENVF_WAS_RESET | _abi.ENVF_LIST_IS_ROW_ORDER)) or \
            (int(es[_abi.ENV_STEP]) == 1 and not flags & _abi.ENVF_TRUNC_ALL)
        rows = [(0, r, r) for r in range(nP)] + [(1, r, cp + r) for r in range(nQ)]
        seq, t2, idx = split_row_id(t["row_id"][i])
        surv = [x for x in rows if not (x[1] >= (nP - newP if x[0] == 0 else nQ - newQ))]
        born = [x for x in rows if x[1] >= (nP - newP if x[0] == 0 else nQ - newQ)]
        if creation_order:
            surv.sort(key=lambda x: int(seq[x[2]]))
        else:
            surv.sort(key=lambda x: (int(t2[x[2]]), x[0], int(t["row_key"][i, x[2]])))
        born.sort(key=lambda x: int(seq[x[2]]))
        out = []
        for sp, r, s in surv + born:
            fl = int(t["row_flags"][i, s])
            out.append((agent_name(sp * 2 + int(t2[s]), idx[s]), sp, r, float(t["row_reward"][i, s]),
                        bool(fl & _abi.ROW_DIED), bool(fl & _abi.ROW_TRUNC)))
        return out


# ---------------------------------------------------------------------------------------------------------
# the reference's class on top of the kernels
# ---------------------------------------------------------------------------------------------------------

from .env import _MultiAgentEnvBase  # noqa: E402  (RLlib's MultiAgentEnv when ray is installed)
from .placement import reference_placement  # noqa: E402

try:
    import gymnasium as _gym  # type: ignore

    def _box(shape):
        return _gym.spaces.Box(low=0, high=100.0, shape=shape, dtype=np.float32)  # RQ:963,967

    def _discrete(n):
        return _gym.spaces.Discrete(n)
except Exception:
    from .env import _Box, _Discrete  # noqa: E402

    def _box(shape):
        return _Box(0, 100.0, shape, np.float32)

    def _discrete(n):
        return _Discrete(n)


class PredPreyGrass(_MultiAgentEnvBase):
    """`PredPreyGrass(config)` of red_queen/predpreygrass_rllib_env.py (class :14, reset :151, step :197): same
    constructor, dict layouts, id strings, dict ordering, float32 observations and viewer-facing attributes, with the
    transition running on the GPU.  `reset(seed=s)` places the agents where the reference does and seeds the same
    PCG64 stream for the reproduction uniforms, so an episode driven with the same actions is identical.

    The per-agent analytics the reference accumulates on the side (``unique_agent_stats``, ``death_agents_stats``,
    ``per_step_agent_data``, ``agent_ages``, offspring lists ... RQ:99-116) are kept by a host-side mirror
    (rq_analytics.AgentAnalytics) from the tables every call fetches anyway; ``analytics=False`` switches them off."""

    _walls = False                 # walls_occlusion.PredPreyGrass sets this
    _require_all_actions = True    # RQ:279 fails for a live agent without an action

    def __init__(self, config=None, *, device=None, prey_capacity: int | None = None, analytics: bool = True,
                 _check_analytics: bool = False, _library=None):
        super().__init__()
        cfg = resolve_config(config)
        self.config = config
        b = BatchedRedQueen(cfg, batch_size=1, device=device, prey_capacity=prey_capacity or 128, walls=self._walls,
                            _library=_library)
        self._b = b
        self._cfg = cfg
        for k in ("max_steps", "grid_size", "num_obs_channels", "predator_obs_range", "prey_obs_range",
                  "initial_num_grass", "initial_energy_grass", "initial_energy_predator", "initial_energy_prey"):
            setattr(self, k, cfg[k])                                               # RQ:36-78
        self.type_1_act_range, self.type_2_act_range = b.action_ranges               # RQ:85-86
        self.possible_agents = [f"{POOLS[p]}_{i}" for p in (0, 1, 2, 3) for i in range(b.n_possible[p])]  # RQ:941-955
        self._pspace = _box((b.obs_channels, b.Rp, b.Rp))
        self._qspace = _box((b.obs_channels, b.Rq, b.Rq))
        self._aspace = {1: _discrete(self.type_1_act_range ** 2), 2: _discrete(self.type_2_act_range ** 2)}  # RQ:974-985
        self.observation_spaces = {a: (self._pspace if "predator" in a else self._qspace) for a in self.possible_agents}
        self.action_spaces = {a: self._aspace[int(a[5])] for a in self.possible_agents}
        self.grass_agents = [f"grass_{k}" for k in range(b.n_grass)]
        self.rng = np.random.default_rng(cfg.get("seed", 42))                       # RQ:37
        self.agents = []
        self.cumulative_rewards = {}
        self.agents_just_ate = set()
        self.current_step = 0
        self._records = []
        self._insertion_order = []
        self._tables = None
        from .rq_analytics import AgentAnalytics
        self._an = AgentAnalytics(cfg, walls=self._walls, check=_check_analytics) if analytics else None

    # the reference's analytics attributes (RQ:99-116), served by the host-side mirror
    def _analytics_attr(self, name):
        if self._an is None:
            raise AttributeError(f"{name}: this env was built with analytics=False")
        return getattr(self._an, name)

    unique_agents = property(lambda self: self._analytics_attr("unique_agents"))
    unique_agent_stats = property(lambda self: self._analytics_attr("unique_agent_stats"))
    death_agents_stats = property(lambda self: self._analytics_attr("death_agents_stats"))
    per_step_agent_data = property(lambda self: self._analytics_attr("per_step_agent_data"))
    agent_ages = property(lambda self: self._analytics_attr("agent_ages"))
    agent_parents = property(lambda self: self._analytics_attr("agent_parents"))
    agent_offspring_counts = property(lambda self: self._analytics_attr("agent_offspring_counts"))
    agent_live_offspring_ids = property(lambda self: self._analytics_attr("agent_live_offspring_ids"))
    agent_activation_counts = property(lambda self: self._analytics_attr("agent_activation_counts"))
    death_cause_prey = property(lambda self: self._analytics_attr("death_cause_prey"))

    def _analytics_step(self, before, where, current_step, insertion_order, action_names):
        """Feed the mirror one non-truncated call: the tables before it (`before`, `where`: live name -> (species, row)) and the
        ones `_collect` just fetched."""
        b, t = self._b, self._tables
        cp = b.pred_capacity
        # (whole columns as Python lists once: NumPy scalars one by one are what such a pass costs)
        xy0, xy1 = before["row_xy"][0].tolist(), t["row_xy"][0].tolist()
        cum0, fl1 = before["row_cumrew"][0].tolist(), t["row_flags"][0].tolist()
        lr1, e1 = t["row_lastrep"][0].tolist(), t["row_energy"][0].tolist()
        slot0 = {n: cp * sp + row for n, (sp, row) in where.items()}
        slot1 = {name: cp * sp + row for name, sp, row, *_ in self._records}
        pos1 = {n: (xy1[s] >> 8 & 255, xy1[s] & 255) for n, s in slot1.items()}
        ng = b.n_grass
        pending = self._an.step(
            current_step=current_step, action_names=action_names,
            agents_in_order=[name for name, *_ in self._records if not fl1[slot1[name]] & _abi.ROW_NEWBORN],
            insertion_order=insertion_order,
            pos_before={n: (xy0[s] >> 8 & 255, xy0[s] & 255) for n, s in slot0.items()},
            pos_after=pos1,
            cum_before={n: cum0[s] for n, s in slot0.items()},
            grass_pos=[(v >> 8 & 255, v & 255) for v in before["grass_xy"][0][:ng].tolist()],
            grass_energy_before=before["grass_energy"][0][:ng].tolist(),
            terminated={name for name, _, _, _, te, _ in self._records if te},
            ate={n for n, s in slot1.items() if fl1[s] & _abi.ROW_ATE},
            newborn=[name for name, *_ in self._records if fl1[slot1[name]] & _abi.ROW_NEWBORN],
            lastrep_after={n: lr1[s] for n, s in slot1.items()},
            energy_after={n: e1[s] for n, s in slot1.items()},
            grass_energy_after=t["grass_energy"][0][:ng].tolist())
        self._an.record_step(sorted(slot1), set(pending), pos1)

    # ------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        """RQ:151-195.  ``options={"placement": (pred_xy, prey_xy, grass_xy)}`` overrides the cells."""
        b = self._b
        self.rng = np.random.default_rng(seed)                                       # RQ:91
        placement = (options or {}).get("placement") if isinstance(options, dict) else None
        if placement is None:
            cells = reference_placement(b.grid_size, b.P0 + b.Q0 + b.n_grass, seed)   # RQ:158-166
            placement = (cells[:b.P0], cells[b.P0:b.P0 + b.Q0], cells[b.P0 + b.Q0:])
        p, q, g = placement
        b.set_placement(np.asarray(p).reshape(1, -1, 2), np.asarray(q).reshape(1, -1, 2), np.asarray(g).reshape(1, -1, 2))
        self.cumulative_rewards = {}
        self._insertion_order = []
        obs = self._collect(after_reset=True)[0]
        if self._an is not None:
            self._an.reset(self.possible_agents, self.agents, self.agent_energies)
        return obs, {}

    def step(self, action_dict):
        """RQ:197-299."""
        b = self._b
        where = {name: (sp, row) for name, sp, row, _, te, _ in self._records if not te}
        before, step_before = self._tables, self.current_step
        order_before = [n for n in self._insertion_order if n in where]
        a = b.stage_actions(0)     # the pinned host mirror of the action tensor
        a[:] = _abi.ACTION_NONE
        rk = torch.zeros((b.S,), dtype=torch.uint8)
        last, count, in_order = [-1, -1], [0, 0], True
        truncated_call = self.current_step >= self.max_steps
        for name, act in action_dict.items():
            if name not in where:
                continue                                    # agents that are gone are skipped (RQ:467,521)
            act = int(act)
            if not truncated_call and not 0 <= act < self._aspace[int(name[5])].n:
                raise KeyError(act)                         # action_to_move_tuple_type_*_agents[action], RQ:323/325
            sp, row = where[name]
            s = b.pred_capacity * sp + row
            a[s], rk[s] = act, count[sp]
            count[sp] += 1
            in_order = in_order and row >= last[sp]
            last[sp] = row
        self._last_action_names = list(action_dict)
        if self._require_all_actions and not truncated_call and count[0] + count[1] != len(where):
            # the reference itself fails for a live agent without an action (KeyError in its per-step bookkeeping, RQ:279)
            missing = [n for n in where if n not in action_dict]
            raise KeyError(missing[0])
        b.upload_actions()
        # the uniforms self.rng.random() would hand out (RQ:701,708): at most two per live agent; the stream is then
        # advanced by exactly the number the step consumed
        state = self.rng.bit_generator.state
        u = torch.from_numpy(self.rng.random(2 * len(where) + 2)).reshape(1, -1).to(b.device)
        b.step(uniforms=u, act_rank=None if in_order else rk[None].to(b.device))
        out = self._collect(after_reset=False)
        self.rng.bit_generator.state = state
        self.rng.bit_generator.advance(int(self._tables["env_state"][0][_abi.ENV_DRAWS]))
        if self._an is not None and not truncated_call:
            self._analytics_step(before, where, step_before, order_before, list(action_dict))
        return out

    def close(self):
        pass

    # ------------------------------------------------------------------
    def _collect(self, after_reset):
        b = self._b
        t, fp, fq = b.fetch(0, 1)   # ONE device->host copy: the tables and the observation rows in use (ppg_fetch)
        self._tables = t
        es = t["env_state"][0]
        status = int(es[_abi.ENV_STATUS])
        if status & (_abi.STATUS_PRED_OVERFLOW | _abi.STATUS_PREY_OVERFLOW):
            raise RuntimeError("agent row capacity exceeded: construct the env with a larger prey_capacity")
        if status & _abi.STATUS_FAILED_SPAWN:
            raise TypeError("no free cell for a newborn (the reference fails at RQ:751-760)")
        recs = b.records(0, t)
        op, oq = fp[0], fq[0]
        obs, rew, term, trunc = {}, {}, {}, {}
        for name, sp, row, r, te, tr in recs:
            obs[name] = (oq if sp else op)[row].astype(np.float32)
            rew[name], term[name], trunc[name] = r, te, tr
        fl = int(es[_abi.ENV_FLAGS])
        self._records = recs
        self.current_step = int(es[_abi.ENV_STEP])
        self.active_num_predators = int(es[_abi.ENV_N_PRED_ALIVE])
        self.active_num_prey = int(es[_abi.ENV_N_PREY_ALIVE])
        self._next_idx = {("predator", 1): int(es[_abi.ENV_NEXT_PRED_ID]), ("predator", 2): int(es[_abi.ENV_NEXT_PRED_ID_T2]),
                          ("prey", 1): int(es[_abi.ENV_NEXT_PREY_ID]), ("prey", 2): int(es[_abi.ENV_NEXT_PREY_ID_T2])}
        cp = b.pred_capacity
        for name, sp, row, *_ in recs:
            self.cumulative_rewards[name] = float(t["row_cumrew"][0][cp * sp + row])
            if name not in self._insertion_order:
                self._insertion_order.append(name)
        self.agents_just_ate = {name for name, sp, row, *_ in recs if t["row_flags"][0][cp * sp + row] & _abi.ROW_ATE}
        names = [r[0] for r in recs]
        self.agents = names if (after_reset or (fl & _abi.ENVF_LIST_IS_ROW_ORDER)) else sorted(names)   # RQ:270
        self._pending_removal = [r[0] for r in recs if r[4]]
        if after_reset:
            return obs, {}
        infos = self._finish_outputs(recs, t, rew, term, trunc, bool(fl & _abi.ENVF_TRUNC_ALL))
        term["__all__"] = bool(fl & _abi.ENVF_TERM_ALL)
        trunc["__all__"] = bool(fl & _abi.ENVF_TRUNC_ALL)
        return obs, rew, term, trunc, infos

    def _finish_outputs(self, recs, tables, rew, term, trunc, truncated_call):
        return {}   # RQ:198,299: infos stays empty

    def _live(self):
        cp = self._b.pred_capacity
        return {name: cp * sp + row for name, sp, row, _, te, _ in self._records if not te}

    # viewer-facing attributes (RQ:93-98,108-109)
    @property
    def agent_positions(self):
        live, t = self._live(), self._tables
        return {n: (int(t["row_xy"][0][live[n]]) >> 8, int(t["row_xy"][0][live[n]]) & 255)
                for n in self._insertion_order if n in live}

    @property
    def predator_positions(self):
        return {k: v for k, v in self.agent_positions.items() if "predator" in k}

    @property
    def prey_positions(self):
        return {k: v for k, v in self.agent_positions.items() if "prey" in k}

    @property
    def agent_energies(self):
        live, t = self._live(), self._tables
        return {n: float(t["row_energy"][0][live[n]]) for n in self._insertion_order if n in live}

    @property
    def agent_last_reproduction(self):
        live, t = self._live(), self._tables
        return {n: int(t["row_lastrep"][0][live[n]]) for n in self._insertion_order if n in live}

    @property
    def grass_positions(self):
        t = self._tables
        return {f"grass_{k}": (int(t["grass_xy"][0][k]) >> 8, int(t["grass_xy"][0][k]) & 255) for k in range(self._b.n_grass)}

    @property
    def grass_energies(self):
        t = self._tables
        return {f"grass_{k}": float(t["grass_energy"][0][k]) for k in range(self._b.n_grass)}

    @property
    def grid_world_state(self):
        return self._b.export_grid()[0].cpu().numpy().astype(np.float32)            # RQ:137-139

    def _get_observation(self, agent):
        """RQ:345-360."""
        live = self._live()
        if agent not in live:
            raise KeyError(agent)
        sp = int("prey" in agent)
        self._b.observe()
        t = self._b.obs_prey if sp else self._b.obs_pred
        return t[0, live[agent] - self._b.pred_capacity * sp].cpu().numpy().astype(np.float32)

    def get_total_energy_by_type(self):
        """RQ:1020-1053."""
        out = {"predator": 0.0, "prey": 0.0, "grass": sum(self.grass_energies.values()),
               "type_1_predator": 0.0, "type_2_predator": 0.0, "type_1_prey": 0.0, "type_2_prey": 0.0}
        for a, e in self.agent_energies.items():
            out["predator" if "predator" in a else "prey"] += e
            out[a.rsplit("_", 1)[0]] += e
        return out

    # snapshot / restore (RQ:893-939)
    def get_state_snapshot(self):
        b = self._b
        return {
            "current_step": self.current_step, "agent_positions": self.agent_positions,
            "agent_energies": self.agent_energies, "predator_positions": self.predator_positions,
            "prey_positions": self.prey_positions, "grass_positions": self.grass_positions,
            "grass_energies": self.grass_energies, "grid_world_state": self.grid_world_state,
            "agents": list(self.agents), "cumulative_rewards": dict(self.cumulative_rewards),
            "active_num_predators": self.active_num_predators, "active_num_prey": self.active_num_prey,
            "agents_just_ate": set(self.agents_just_ate), "agent_last_reproduction": self.agent_last_reproduction,
            "pending_removal": list(self._pending_removal), "next_idx": dict(self._next_idx),
            "_device_state": b.export_state(0),   # ppg_export_state (incl. row_lastrep; walls: row_info + wall bitmap)
            "_insertion_order": list(self._insertion_order), "_rng_state": self.rng.bit_generator.state,
            **(self._an.snapshot() if self._an is not None else {}),
        }

    def restore_state_snapshot(self, snapshot):
        b = self._b
        b.import_state(snapshot["_device_state"], 0)
        self._insertion_order = list(snapshot["_insertion_order"])
        self.rng.bit_generator.state = snapshot["_rng_state"]
        b.observe()
        self._collect(after_reset=False)
        self.agents = list(snapshot["agents"])
        self.cumulative_rewards = dict(snapshot["cumulative_rewards"])
        if self._an is not None and "unique_agents" in snapshot:
            self._an.restore(snapshot)


def env_creator(config):
    """red_queen/tune_ppo_red_queen.py registers the env through a creator like this one."""
    return PredPreyGrass(config)
