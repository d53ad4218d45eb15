"""ctypes binding for the CPU oracle (``oracle/ppg_oracle.c``).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module.  ``predpreygrass_amd`` never does.

`OracleEnv` presents the reference's own calling convention
(``reset`` / ``step(action_dict)`` returning the five dicts with ``"__all__"``
keys; predpreygrass_rllib_env.py:129,219) so that parity tests read like tests
of the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "ppg_oracle.c")
_HDR = os.path.join(_HERE, "ppg_oracle.h")
_LIB = os.path.join(_HERE, "_build", "libppg_oracle.so")

PREDATOR, PREY = 0, 1

# Same keys / values as the reference's config_env.py:1-38 (restated, not imported).
DEFAULT_CONFIG = {
    "max_steps": 1000,
    "grid_size": 25,
    "num_obs_channels": 4,
    "predator_obs_range": 7,
    "prey_obs_range": 9,
    "reward_predator_catch_prey": 0.0,
    "reward_prey_eat_grass": 0.0,
    "reward_predator_step": 0.0,
    "reward_prey_step": 0.0,
    "penalty_prey_caught": 0.0,
    "reproduction_reward_predator": 10.0,
    "reproduction_reward_prey": 10.0,
    "energy_loss_per_step_predator": 0.15,
    "energy_loss_per_step_prey": 0.05,
    "predator_creation_energy_threshold": 12.0,
    "prey_creation_energy_threshold": 8.0,
    "n_possible_predators": 2000,
    "n_possible_prey": 2000,
    "n_initial_active_predator": 6,
    "n_initial_active_prey": 8,
    "initial_energy_predator": 5.0,
    "initial_energy_prey": 3.0,
    "initial_num_grass": 100,
    "initial_energy_grass": 2.0,
    "energy_gain_per_step_grass": 0.04,
}

_INT_FIELDS = [
    "max_steps", "grid_size", "num_obs_channels", "predator_obs_range", "prey_obs_range",
    "n_possible_predators", "n_possible_prey", "n_initial_active_predator",
    "n_initial_active_prey", "initial_num_grass",
]
_DBL_FIELDS = [
    "reward_predator_catch_prey", "reward_prey_eat_grass", "reward_predator_step",
    "reward_prey_step", "penalty_prey_caught", "reproduction_reward_predator",
    "reproduction_reward_prey", "energy_loss_per_step_predator", "energy_loss_per_step_prey",
    "predator_creation_energy_threshold", "prey_creation_energy_threshold",
    "initial_energy_predator", "initial_energy_prey", "initial_energy_grass",
    "energy_gain_per_step_grass",
]


class _Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in _INT_FIELDS] + [(n, C.c_double) for n in _DBL_FIELDS] + [
        ("season_length_steps", C.c_int32), ("season_high_multiplier", C.c_double), ("season_low_multiplier", C.c_double),
        ("reward_mode", C.c_int32), ("kickback", C.c_int32), ("kickback_reward_predator", C.c_double),
        ("kickback_reward_prey", C.c_double),
        ("n_drive", C.c_int32 * 2), ("drive_kind", (C.c_int32 * 4) * 2), ("hunger_safe_energy", C.c_double * 2),
        ("prey_opportunity_normalizer", C.c_double), ("predator_danger_normalizer", C.c_double),
        ("grass_opportunity_normalizer", C.c_double)]


DRIVE_KINDS = {"hunger_pressure": 0, "reproductive_readiness": 1, "prey_opportunity": 2, "predator_danger_pressure": 3,
               "grass_opportunity": 4}
# drive_conditioned_environment/predpreygrass_rllib_env.py:56-75
DEFAULT_PREDATOR_DRIVES = ["hunger_pressure", "reproductive_readiness", "prey_opportunity"]
DEFAULT_PREY_DRIVES = ["hunger_pressure", "reproductive_readiness", "predator_danger_pressure", "grass_opportunity"]


def fill_drive_config(c, cfg, raw, enabled):
    """The drive-channel settings of the drive-conditioned env (its :54-89) into a config struct `c`."""
    lists = (raw.get("predator_drive_channels", DEFAULT_PREDATOR_DRIVES), raw.get("prey_drive_channels", DEFAULT_PREY_DRIVES))
    on = enabled and bool(raw.get("enable_drive_channels", True))
    for t in range(2):
        c.n_drive[t] = len(lists[t]) if on else 0
        for k, name in enumerate(lists[t] if on else []):
            if name not in DRIVE_KINDS:
                raise ValueError(f"Unknown drive feature: {name!r}")
            c.drive_kind[t][k] = DRIVE_KINDS[name]
    c.hunger_safe_energy[0] = float(raw.get("predator_hunger_safe_energy", cfg["initial_energy_predator"]))
    c.hunger_safe_energy[1] = float(raw.get("prey_hunger_safe_energy", cfg["initial_energy_prey"]))
    c.prey_opportunity_normalizer = float(raw.get("prey_opportunity_normalizer", cfg["initial_energy_prey"] * 3))
    c.predator_danger_normalizer = float(raw.get("predator_danger_normalizer", cfg["initial_energy_predator"] * 2))
    c.grass_opportunity_normalizer = float(raw.get("grass_opportunity_normalizer", raw.get("initial_energy_grass", 2.0) * 5))
    return (4 + c.n_drive[0], 4 + c.n_drive[1])


class _Record(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("id", C.c_int32), ("reward", C.c_double),
        ("terminated", C.c_int32), ("truncated", C.c_int32),
        ("obs_offset", C.c_int32), ("obs_len", C.c_int32),
    ]


class _StepOut(C.Structure):
    _fields_ = [
        ("n_records", C.c_int32), ("terminated_all", C.c_int32), ("truncated_all", C.c_int32),
        ("fallback_spawns", C.c_int32), ("failed_spawns", C.c_int32),
        ("records", C.POINTER(_Record)), ("obs", C.POINTER(C.c_double)),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (scalar IEEE doubles, no FMA contraction)."""
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(
        os.path.getmtime(_SRC), os.path.getmtime(_HDR)
    ):
        return _LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", _LIB, _SRC]
    subprocess.run(cmd, check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.ppo_create.restype = C.c_void_p
        L.ppo_create.argtypes = [C.POINTER(_Config)]
        L.ppo_destroy.argtypes = [C.c_void_p]
        L.ppo_set_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.ppo_reset_from_placement.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_StepOut)]
        L.ppo_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_StepOut)]
        L.ppo_observe.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.ppo_grid.restype = C.POINTER(C.c_double)
        L.ppo_grid.argtypes = [C.c_void_p]
        for name in ("ppo_current_step", "ppo_agents_len"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int32
        for name in ("ppo_num_alive", "ppo_next_id"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int32]
            getattr(L, name).restype = C.c_int32
        L.ppo_agents_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ppo_agent_alive.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ppo_agent_get.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5
        L.ppo_grass_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ppo_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ppo_reset_philox.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(_StepOut)]
        L.ppo_random_action.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
        L.ppo_random_action.restype = C.c_int32
        L.ppo_rollout_random.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(_StepOut)]
        L.ppo_rollout_random.restype = C.c_int64
        _lib = L
    return _lib


def agent_name(type_: int, id_: int) -> str:
    return ("predator_%d" if type_ == PREDATOR else "prey_%d") % id_


def parse_agent(name: str) -> tuple[int, int]:
    kind, idx = name.rsplit("_", 1)
    return (PREDATOR if kind == "predator" else PREY), int(idx)


def philox4x32_10(ctr, key) -> np.ndarray:
    c = np.asarray(ctr, dtype=np.uint32).copy()
    k = np.asarray(key, dtype=np.uint32).copy()
    out = np.zeros(4, dtype=np.uint32)
    lib().ppo_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def random_action(seed: int, episode: int, step: int, type_: int, row: int) -> int:
    return int(lib().ppo_random_action(seed, episode, step, type_, row))


class OracleEnv:
    """The reference env's interface on top of the C restatement."""

    def __init__(self, config: dict | None = None, drive: bool | None = None):
        """drive: the drive-conditioned env (extra constant-filled observation channels); default: the config's
        `enable_drive_channels` key (absent = base env)."""
        if drive is None:
            drive = bool((config or {}).get("enable_drive_channels", False))
        cfg = dict(DEFAULT_CONFIG)
        if config:
            cfg.update({k: v for k, v in config.items() if k in cfg})
        self.config = cfg
        c = _Config()
        for n in _INT_FIELDS:
            setattr(c, n, int(cfg[n]))
        for n in _DBL_FIELDS:
            setattr(c, n, float(cfg[n]))
        # seasonal variant: only when the config carries its keys (the base env has none)
        c.season_length_steps = int((config or {}).get("season_length_steps", 0))
        c.season_high_multiplier = float((config or {}).get("season_high_multiplier", 1.0))
        c.season_low_multiplier = float((config or {}).get("season_low_multiplier", 1.0))
        c.reward_mode = {"sparse": 0, "dense_energy_delta": 1, "dense_energy_delta_plus_reproduction": 2}[
            (config or {}).get("reward_mode", "sparse")]
        kb = config or {}
        c.kickback = int("kickback_reward_predator" in kb or "kickback_reward_prey" in kb)
        c.kickback_reward_predator = float(kb.get("kickback_reward_predator", 10.0))
        c.kickback_reward_prey = float(kb.get("kickback_reward_prey", 10.0))
        self.channels = fill_drive_config(c, cfg, config or {}, drive)
        self._L = lib()
        self._h = self._L.ppo_create(C.byref(c))
        if not self._h:
            raise ValueError("invalid oracle config")
        self.grid_size = cfg["grid_size"]
        self._out = _StepOut()
        self.last_fallback_spawns = 0
        self.last_failed_spawns = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.ppo_destroy(h)

    # -- conversions -------------------------------------------------
    def _obs_range(self, type_):
        return self.config["predator_obs_range"] if type_ == PREDATOR else self.config["prey_obs_range"]

    def _records(self):
        o = self._out
        obs, rew, term, trunc = {}, {}, {}, {}
        for i in range(o.n_records):
            r = o.records[i]
            name = agent_name(r.type, r.id)
            R = self._obs_range(r.type)
            a = np.ctypeslib.as_array(o.obs, shape=(r.obs_offset + r.obs_len,))[r.obs_offset:]
            obs[name] = a.reshape(self.channels[r.type], R, R).copy()
            rew[name] = float(r.reward)
            term[name] = bool(r.terminated)
            trunc[name] = bool(r.truncated)
        self.last_fallback_spawns = int(o.fallback_spawns)
        self.last_failed_spawns = int(o.failed_spawns)
        return obs, rew, term, trunc

    # -- reference-shaped API ----------------------------------------
    def reset_from_placement(self, pred_xy, prey_xy, grass_xy):
        p = np.ascontiguousarray(pred_xy, dtype=np.int32).reshape(-1)
        q = np.ascontiguousarray(prey_xy, dtype=np.int32).reshape(-1)
        g = np.ascontiguousarray(grass_xy, dtype=np.int32).reshape(-1)
        assert p.size == 2 * self.config["n_initial_active_predator"]
        assert q.size == 2 * self.config["n_initial_active_prey"]
        assert g.size == 2 * self.config["initial_num_grass"]
        rc = self._L.ppo_reset_from_placement(self._h, p.ctypes.data, q.ctypes.data, g.ctypes.data, C.byref(self._out))
        if rc != 0:
            raise ValueError(f"oracle reset failed rc={rc}")
        return self._records()[0], {}

    def reset_philox(self, seed: int, episode: int = 0):
        rc = self._L.ppo_reset_philox(self._h, seed, episode, C.byref(self._out))
        if rc != 0:
            raise ValueError(f"oracle reset failed rc={rc}")
        return self._records()[0], {}

    def set_seed(self, seed: int, episode: int = 0):
        self._L.ppo_set_seed(self._h, seed, episode)

    def step(self, action_dict):
        n = len(action_dict)
        t = np.empty(n, dtype=np.int32)
        i = np.empty(n, dtype=np.int32)
        a = np.empty(n, dtype=np.int32)
        for k, (name, act) in enumerate(action_dict.items()):
            t[k], i[k] = parse_agent(name)
            a[k] = int(act)
        rc = self._L.ppo_step(self._h, n, t.ctypes.data, i.ctypes.data, a.ctypes.data, C.byref(self._out))
        if rc == -2:
            raise KeyError("action for an agent that is not alive")
        if rc == -3:
            raise KeyError("action outside 0..8")
        if rc != 0:
            raise RuntimeError(f"oracle step failed rc={rc}")
        obs, rew, term, trunc = self._records()
        term["__all__"] = bool(self._out.terminated_all)
        trunc["__all__"] = bool(self._out.truncated_all)
        return obs, rew, term, trunc, {}

    def rollout_random(self, seed: int, n_calls: int) -> int:
        return int(self._L.ppo_rollout_random(self._h, seed, n_calls, C.byref(self._out)))

    def last_records(self):
        """(type, id, reward, terminated, truncated) tuples + flags of the last call."""
        o = self._out
        recs = [(o.records[i].type, o.records[i].id, o.records[i].reward,
                 o.records[i].terminated, o.records[i].truncated) for i in range(o.n_records)]
        return recs, bool(o.terminated_all), bool(o.truncated_all)

    # -- attributes --------------------------------------------------
    @property
    def grid_world_state(self):
        G = self.grid_size
        return np.ctypeslib.as_array(self._L.ppo_grid(self._h), shape=(4, G, G)).copy()

    @property
    def current_step(self):
        return int(self._L.ppo_current_step(self._h))

    @property
    def current_num_predators(self):
        return int(self._L.ppo_num_alive(self._h, PREDATOR))

    @property
    def current_num_prey(self):
        return int(self._L.ppo_num_alive(self._h, PREY))

    @property
    def next_ids(self):
        return int(self._L.ppo_next_id(self._h, PREDATOR)), int(self._L.ppo_next_id(self._h, PREY))

    @property
    def agents(self):
        n = int(self._L.ppo_agents_len(self._h))
        t = np.empty(n, dtype=np.int32)
        i = np.empty(n, dtype=np.int32)
        self._L.ppo_agents_get(self._h, t.ctypes.data, i.ctypes.data)
        return [agent_name(int(a), int(b)) for a, b in zip(t, i)]

    def agent_state(self, name):
        t, i = parse_agent(name)
        x, y, ja = C.c_int32(), C.c_int32(), C.c_int32()
        e, cum = C.c_double(), C.c_double()
        rc = self._L.ppo_agent_get(self._h, t, i, C.byref(x), C.byref(y), C.byref(e), C.byref(cum), C.byref(ja))
        if rc != 0:
            return None
        return dict(pos=(x.value, y.value), energy=e.value, cumulative_reward=cum.value, just_ate=bool(ja.value))

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
        self._L.ppo_grass_get(self._h, xy.ctypes.data, e.ctypes.data)
        return xy.reshape(n, 2), e

    def _get_observation(self, name):
        t, i = parse_agent(name)
        R = self._obs_range(t)
        out = np.empty(self.channels[t] * R * R, dtype=np.float64)
        if self._L.ppo_observe(self._h, t, i, out.ctypes.data) != 0:
            raise KeyError(name)
        return out.reshape(self.channels[t], R, R)
