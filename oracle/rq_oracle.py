"""ctypes binding for the second-generation CPU oracle (``oracle/rq_oracle.c``).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg import this module.  ``predpreygrass_amd`` never does.

`RQOracleEnv` presents the calling convention of the reference's red_queen env
(red_queen/predpreygrass_rllib_env.py:151,197: ``reset`` / ``step(action_dict)`` returning five dicts keyed
by ``type_<t>_<species>_<id>``), plus the uniform stream as an explicit argument.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "rq_oracle.c")
_HDR = os.path.join(_HERE, "rq_oracle.h")
_LIB = os.path.join(_HERE, "_build", "librq_oracle.so")

POOLS = ("type_1_predator", "type_2_predator", "type_1_prey", "type_2_prey")  # creation order, RQ:125-133
POOL_KEYS = ("type_1_predators", "type_2_predators", "type_1_prey", "type_2_prey")  # n_possible_* spelling, RQ:56-59

# Defaults the reference applies when a key is absent (RQ:30-86 and the config.get calls in step()).
DEFAULT_CONFIG = {
    "max_steps": 10000, "grid_size": 10, "num_obs_channels": 4, "predator_obs_range": 7, "prey_obs_range": 5,
    "n_possible_type_1_predators": 25, "n_possible_type_2_predators": 25,
    "n_possible_type_1_prey": 25, "n_possible_type_2_prey": 25,
    "n_initial_active_type_1_predator": 6, "n_initial_active_type_2_predator": 0,
    "n_initial_active_type_1_prey": 8, "n_initial_active_type_2_prey": 0,
    "initial_num_grass": 25, "type_1_action_range": 3, "type_2_action_range": 5,
    "reproduction_cooldown_steps": 10,
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

_TYPED_REWARDS = [  # (field, species word used in the per-type dict keys)
    ("reward_predator_catch_prey", "predator"), ("reward_prey_eat_grass", "prey"),
    ("reward_predator_step", "predator"), ("reward_prey_step", "prey"), ("penalty_prey_caught", "prey"),
    ("reproduction_reward_predator", "predator"), ("reproduction_reward_prey", "prey"),
]
_SCALARS = [
    "energy_loss_per_step_predator", "energy_loss_per_step_prey", "predator_creation_energy_threshold",
    "prey_creation_energy_threshold", "initial_energy_predator", "initial_energy_prey", "initial_energy_grass",
    "energy_gain_per_step_grass", "move_energy_cost_factor", "max_energy_gain_per_prey",
    "max_energy_gain_per_grass", "max_energy_predator", "max_energy_prey", "max_energy_grass",
    "energy_transfer_efficiency", "reproduction_energy_efficiency", "reproduction_chance_predator",
    "reproduction_chance_prey", "mutation_rate_predator", "mutation_rate_prey",
]


class _Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_steps", "grid_size", "num_obs_channels", "predator_obs_range",
                                         "prey_obs_range")] + [
        ("n_possible", C.c_int32 * 4), ("n_initial", C.c_int32 * 4), ("initial_num_grass", C.c_int32),
        ("type_1_action_range", C.c_int32), ("type_2_action_range", C.c_int32),
        ("reproduction_cooldown_steps", C.c_int32), ("pad_", C.c_int32)] + [
        (n, C.c_double * 2) for n, _ in _TYPED_REWARDS] + [(n, C.c_double) for n in _SCALARS] + [
        (n, C.c_int32) for n in ("walls", "include_visibility_channel", "respect_los_for_movement",
                                 "mask_observation_with_visibility")]


class _Record(C.Structure):
    _fields_ = [("pool", C.c_int32), ("id", C.c_int32), ("reward", C.c_double), ("terminated", C.c_int32),
                ("truncated", C.c_int32), ("obs_offset", C.c_int32), ("obs_len", C.c_int32),
                ("block_reason", C.c_int32), ("pad_", C.c_int32)]


BLOCK_REASONS = {1: "wall", 2: "occupied", 3: "corner_cut", 4: "los"}  # move_blocked_reason, WO:466-488


class _StepOut(C.Structure):
    _fields_ = [("n_records", C.c_int32), ("terminated_all", C.c_int32), ("truncated_all", C.c_int32),
                ("fallback_spawns", C.c_int32), ("failed_spawns", C.c_int32), ("draws", C.c_int32),
                ("records", C.POINTER(_Record)), ("obs", C.POINTER(C.c_float))]


def typed_value(raw, species: str, type_: int) -> float:
    """_get_type_specific (RQ:1099-1106): a scalar, or a dict keyed by 'type_<t>_<species>' prefixes."""
    if isinstance(raw, dict):
        name = f"type_{type_}_{species}"
        for k in raw:
            if name.startswith(k):
                return float(raw[k])
        raise KeyError(name)
    return float(raw)


def fill_config(config: dict) -> dict:
    cfg = dict(DEFAULT_CONFIG)
    cfg.update({k: v for k, v in (config or {}).items() if k in cfg})
    return cfg


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(os.path.getmtime(_SRC), os.path.getmtime(_HDR)):
        return _LIB
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", _LIB, _SRC, "-lm"], check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.rqo_create.restype = C.c_void_p
        L.rqo_create.argtypes = [C.POINTER(_Config)]
        L.rqo_destroy.argtypes = [C.c_void_p]
        L.rqo_set_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.rqo_set_walls.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.rqo_reset_from_placement.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_StepOut)]
        L.rqo_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                               C.POINTER(_StepOut)]
        L.rqo_observe.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.rqo_grid.restype = C.POINTER(C.c_float)
        L.rqo_grid.argtypes = [C.c_void_p]
        for name in ("rqo_current_step", "rqo_agents_len"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int32
        for name in ("rqo_num_alive", "rqo_next_id"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int32]
            getattr(L, name).restype = C.c_int32
        L.rqo_agents_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rqo_agent_alive.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.rqo_agent_get.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 7
        L.rqo_grass_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rqo_philox_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.rqo_philox_uniform.restype = C.c_double
        L.rqo_reset_philox.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(_StepOut)]
        L.rqo_random_action.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32]
        L.rqo_random_action.restype = C.c_int32
        L.rqo_rollout_random.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(_StepOut)]
        L.rqo_rollout_random.restype = C.c_int64
        _lib = L
    return _lib


def agent_name(pool: int, id_: int) -> str:
    return f"{POOLS[pool]}_{id_}"


def parse_agent(name: str) -> tuple[int, int]:
    kind, idx = name.rsplit("_", 1)
    return POOLS.index(kind), int(idx)


def philox_uniform(seed: int, episode: int, step: int, draw: int) -> float:
    return float(lib().rqo_philox_uniform(seed, episode, step, draw))


class RQOracleEnv:
    """The red_queen reference env's interface on top of the C restatement; with ``walls=True`` the walls_occlusion
    env's (static walls in observation channel 0, line-of-sight options, per-agent infos)."""

    def __init__(self, config: dict, walls: bool = False):
        cfg = fill_config(config)
        self.config = cfg
        self.walls = bool(walls)
        c = _Config()
        c.walls = int(self.walls)
        c.include_visibility_channel = int(bool((config or {}).get("include_visibility_channel", False)) and self.walls)
        c.respect_los_for_movement = int(bool((config or {}).get("respect_los_for_movement", False)) and self.walls)
        c.mask_observation_with_visibility = int(bool((config or {}).get("mask_observation_with_visibility", False)) and self.walls)
        self.channels = 4 + c.include_visibility_channel
        for n in ("max_steps", "grid_size", "num_obs_channels", "predator_obs_range", "prey_obs_range",
                  "initial_num_grass", "type_1_action_range", "type_2_action_range", "reproduction_cooldown_steps"):
            setattr(c, n, int(cfg[n]))
        for p, key in enumerate(POOL_KEYS):
            c.n_possible[p] = int(cfg[f"n_possible_{key}"])
            c.n_initial[p] = int(cfg[f"n_initial_active_{POOLS[p]}"])
        for n, species in _TYPED_REWARDS:
            for t in (1, 2):
                getattr(c, n)[t - 1] = typed_value(cfg[n], species, t)
        for n in _SCALARS:
            setattr(c, n, float(cfg[n]))
        self._L = lib()
        self._h = self._L.rqo_create(C.byref(c))
        if not self._h:
            raise ValueError("invalid oracle config")
        self.grid_size = cfg["grid_size"]
        self._out = _StepOut()
        self.last_fallback_spawns = 0
        self.last_failed_spawns = 0
        self.last_draws = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.rqo_destroy(h)

    def _obs_range(self, pool):
        return self.config["prey_obs_range"] if pool >= 2 else self.config["predator_obs_range"]

    def _records(self):
        o = self._out
        obs, rew, term, trunc = {}, {}, {}, {}
        self.last_infos = {}
        for i in range(o.n_records):
            r = o.records[i]
            name = agent_name(r.pool, r.id)
            R = self._obs_range(r.pool)
            a = np.ctypeslib.as_array(o.obs, shape=(r.obs_offset + r.obs_len,))[r.obs_offset:]
            obs[name] = a.reshape(self.channels, R, R).copy()
            if r.block_reason >= 0:   # infos of the walls env, WO:766-778
                info = {"los_rejected": int(r.block_reason == 4)}
                if r.block_reason > 0:
                    info["move_blocked_reason"] = BLOCK_REASONS[r.block_reason]
                self.last_infos[name] = info
            rew[name] = float(r.reward)
            term[name] = bool(r.terminated)
            trunc[name] = bool(r.truncated)
        self.last_fallback_spawns = int(o.fallback_spawns)
        self.last_failed_spawns = int(o.failed_spawns)
        self.last_draws = int(o.draws)
        return obs, rew, term, trunc

    # -- reference-shaped API ----------------------------------------
    def set_walls(self, wall_xy):
        w = np.ascontiguousarray(wall_xy, dtype=np.int32).reshape(-1)
        if self._L.rqo_set_walls(self._h, w.size // 2, w.ctypes.data) != 0:
            raise ValueError("wall outside the grid")

    def reset_from_placement(self, pred_xy, prey_xy, grass_xy):
        p = np.ascontiguousarray(pred_xy, dtype=np.int32).reshape(-1)
        q = np.ascontiguousarray(prey_xy, dtype=np.int32).reshape(-1)
        g = np.ascontiguousarray(grass_xy, dtype=np.int32).reshape(-1)
        rc = self._L.rqo_reset_from_placement(self._h, p.ctypes.data, q.ctypes.data, g.ctypes.data, C.byref(self._out))
        if rc != 0:
            raise ValueError(f"oracle reset failed rc={rc}")
        return self._records()[0], {}

    def reset_philox(self, seed: int, episode: int = 0):
        rc = self._L.rqo_reset_philox(self._h, seed, episode, C.byref(self._out))
        if rc != 0:
            raise ValueError(f"oracle reset failed rc={rc}")
        return self._records()[0], {}

    def set_seed(self, seed: int, episode: int = 0):
        self._L.rqo_set_seed(self._h, seed, episode)

    def step(self, action_dict, uniforms=None):
        n = len(action_dict)
        t = np.empty(n, dtype=np.int32)
        i = np.empty(n, dtype=np.int32)
        a = np.empty(n, dtype=np.int32)
        for k, (name, act) in enumerate(action_dict.items()):
            t[k], i[k] = parse_agent(name)
            a[k] = int(act)
        if uniforms is None:
            up, un = None, 0
        else:
            u = np.ascontiguousarray(uniforms, dtype=np.float64)
            up, un = u.ctypes.data, int(u.size)
        rc = self._L.rqo_step(self._h, n, t.ctypes.data, i.ctypes.data, a.ctypes.data, up, un, C.byref(self._out))
        if rc == -3:
            raise KeyError("action outside the agent's action space")
        if rc == -6:
            raise RuntimeError("uniform stream ran dry")
        if rc != 0:
            raise RuntimeError(f"oracle step failed rc={rc}")
        obs, rew, term, trunc = self._records()
        if self.walls and not self._out.truncated_all:
            # WO:370-395: the scalar dicts also carry every agent named in action_dict (defaults 0.0 / False); they are
            # built from a Python set there, so their ORDER is arbitrary in the reference -- compare them as mappings
            for name in action_dict:
                if name not in rew:
                    rew[name], term[name], trunc[name] = 0.0, False, False
        term["__all__"] = bool(self._out.terminated_all)
        trunc["__all__"] = bool(self._out.truncated_all)
        return obs, rew, term, trunc, (dict(self.last_infos) if self.walls else {})

    def rollout_random(self, seed: int, n_calls: int) -> int:
        return int(self._L.rqo_rollout_random(self._h, seed, n_calls, C.byref(self._out)))

    def last_records(self):
        o = self._out
        recs = [(o.records[i].pool, o.records[i].id, o.records[i].reward, o.records[i].terminated,
                 o.records[i].truncated) for i in range(o.n_records)]
        return recs, bool(o.terminated_all), bool(o.truncated_all)

    def infos_of_last_call(self):
        """infos dict of the walls env for the last call (also after rollout_random)."""
        o, out = self._out, {}
        for i in range(o.n_records):
            r = o.records[i]
            if r.block_reason >= 0:
                d = {"los_rejected": int(r.block_reason == 4)}
                if r.block_reason > 0:
                    d["move_blocked_reason"] = BLOCK_REASONS[r.block_reason]
                out[agent_name(r.pool, r.id)] = d
        return out

    def last_obs(self, i):
        o = self._out
        r = o.records[i]
        R = self._obs_range(r.pool)
        a = np.ctypeslib.as_array(o.obs, shape=(r.obs_offset + r.obs_len,))[r.obs_offset:]
        return a.reshape(self.channels, R, R).copy()

    # -- attributes --------------------------------------------------
    @property
    def grid_world_state(self):
        G = self.grid_size
        return np.ctypeslib.as_array(self._L.rqo_grid(self._h), shape=(4, G, G)).copy()

    @property
    def current_step(self):
        return int(self._L.rqo_current_step(self._h))

    @property
    def active_num_predators(self):
        return int(self._L.rqo_num_alive(self._h, 0))

    @property
    def active_num_prey(self):
        return int(self._L.rqo_num_alive(self._h, 1))

    @property
    def next_ids(self):
        return tuple(int(self._L.rqo_next_id(self._h, p)) for p in range(4))

    @property
    def agents(self):
        n = int(self._L.rqo_agents_len(self._h))
        t = np.empty(n, dtype=np.int32)
        i = np.empty(n, dtype=np.int32)
        self._L.rqo_agents_get(self._h, t.ctypes.data, i.ctypes.data)
        return [agent_name(int(a), int(b)) for a, b in zip(t, i)]

    def agent_state(self, name):
        t, i = parse_agent(name)
        x, y, ja, age, lr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        e, cum = C.c_double(), C.c_double()
        rc = self._L.rqo_agent_get(self._h, t, i, C.byref(x), C.byref(y), C.byref(e), C.byref(cum), C.byref(ja),
                                   C.byref(age), C.byref(lr))
        if rc != 0:
            return None
        return dict(pos=(x.value, y.value), energy=e.value, cumulative_reward=cum.value, just_ate=bool(ja.value),
                    age=age.value, last_reproduction=lr.value)

    @property
    def agent_positions(self):
        return {a: s["pos"] for a in self.agents if (s := self.agent_state(a)) is not None}

    @property
    def agent_energies(self):
        return {a: s["energy"] for a in self.agents if (s := self.agent_state(a)) is not None}

    def grass_state(self):
        n = self.config["initial_num_grass"]
        xy = np.empty(2 * n, dtype=np.int32)
        e = np.empty(n, dtype=np.float64)
        self._L.rqo_grass_get(self._h, xy.ctypes.data, e.ctypes.data)
        return xy.reshape(n, 2), e

    def _get_observation(self, name):
        t, i = parse_agent(name)
        R = self._obs_range(t)
        out = np.empty(self.channels * R * R, dtype=np.float32)
        if self._L.rqo_observe(self._h, t, i, out.ctypes.data) != 0:
            raise KeyError(name)
        return out.reshape(self.channels, R, R)
