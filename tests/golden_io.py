"""Reader for the golden vectors in tests/golden/*.npz (written by make_golden.py)."""
from __future__ import annotations

import glob
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PREDATOR, PREY = 0, 1


def agent_name(t, i):
    return ("predator_%d" if int(t) == PREDATOR else "prey_%d") % int(i)


def case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def call_digest(grid, obs, rew, term, trunc) -> bytes:
    """Same definition as tests/golden/make_golden.py:call_digest."""
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(grid, dtype=np.float64).tobytes())
    for k, v in obs.items():
        h.update(k.encode() + np.ascontiguousarray(v, dtype=np.float64).tobytes())
    for k, v in rew.items():
        h.update(k.encode() + np.float64(v).tobytes())
    for k, v in term.items():
        h.update(k.encode() + bytes([int(bool(v))]))
    for k, v in trunc.items():
        h.update(k.encode() + bytes([int(bool(v))]))
    return h.digest()


# config keys of THIS implementation that select a base-family variant (the reference has one class per variant)
VARIANT_CONFIG = {
    "dense_rewards": {"reward_mode": "dense_energy_delta"},
    "dense_rewards_additive": {"reward_mode": "dense_energy_delta_plus_reproduction"},
    "drive_conditioned": {"enable_drive_channels": True},
}
# drive_conditioned_environment/predpreygrass_rllib_env.py:56-75
_DEFAULT_DRIVES = {"predator": 3, "prey": 4}


def obs_channels(name, cfg):
    """Channels of an agent's observation: 4, plus the drive channels of the drive-conditioned variant."""
    if not cfg.get("enable_drive_channels", False):
        return 4
    kind = "predator" if name.startswith("predator") else "prey"
    lst = cfg.get(f"{kind}_drive_channels")
    return 4 + (len(lst) if lst is not None else _DEFAULT_DRIVES[kind])


class GoldenCase:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.z = {k: z[k] for k in z.files}
        self.overrides = json.loads(str(self.z["config_json"]))
        self.variant = str(self.z["variant"]) if "variant" in self.z else "base"
        self.overrides.update(VARIANT_CONFIG.get(self.variant, {}))
        self.n_calls = len(self.z["term_all"])
        self.agents_after = json.loads(str(self.z["agents_after"]))
        self.reset_keys = json.loads(str(self.z["reset_keys"]))
        self.full_index = {int(c): k for k, c in enumerate(self.z["full_calls"])}

    def config(self, defaults):
        cfg = dict(defaults)
        cfg.update(self.overrides)
        return cfg

    @property
    def placement(self):
        return self.z["pred_xy"], self.z["prey_xy"], self.z["grass_xy"]

    def actions(self, t):
        """The action dict of call t (dict order preserved)."""
        lo, hi = self.z["act_off"][t], self.z["act_off"][t + 1]
        return {agent_name(self.z["act_type"][k], self.z["act_id"][k]): int(self.z["act_val"][k]) for k in range(lo, hi)}

    def records(self, t):
        """[(name, reward, terminated, truncated)] of call t in dict order."""
        lo, hi = self.z["rec_off"][t], self.z["rec_off"][t + 1]
        return [
            (agent_name(self.z["rec_type"][k], self.z["rec_id"][k]), float(self.z["rec_reward"][k]),
             bool(self.z["rec_term"][k]), bool(self.z["rec_trunc"][k]))
            for k in range(lo, hi)
        ]

    def flags(self, t):
        return bool(self.z["term_all"][t]), bool(self.z["trunc_all"][t])

    def digest(self, t) -> bytes:
        return self.z["digest"][t].tobytes()

    def obs_range(self, name, cfg):
        return cfg["predator_obs_range"] if name.startswith("predator") else cfg["prey_obs_range"]

    def reset_obs(self, cfg):
        out, off = {}, 0
        for k in self.reset_keys:
            R, C = self.obs_range(k, cfg), obs_channels(k, cfg)
            out[k] = self.z["reset_obs_data"][off:off + C * R * R].reshape(C, R, R)
            off += C * R * R
        return out

    def full(self, t, cfg):
        """(obs dict, grid, state list, grass energies) for a call listed in full_calls, else None."""
        k = self.full_index.get(t)
        if k is None:
            return None
        off = int(self.z["obs_off"][k])
        obs = {}
        for name, _, _, _ in self.records(t):
            R, C = self.obs_range(name, cfg), obs_channels(name, cfg)
            obs[name] = self.z["obs_data"][off:off + C * R * R].reshape(C, R, R)
            off += C * R * R
        assert off == int(self.z["obs_off"][k + 1])
        lo, hi = self.z["st_off"][k], self.z["st_off"][k + 1]
        state = {
            agent_name(self.z["st_type"][j], self.z["st_id"][j]): dict(
                pos=(int(self.z["st_x"][j]), int(self.z["st_y"][j])), energy=float(self.z["st_energy"][j]),
                cumulative_reward=float(self.z["st_cum"][j]), just_ate=bool(self.z["st_ate"][j]))
            for j in range(lo, hi)
        }
        return obs, self.z["grid_data"][k], state, self.z["grass_energy"][k]
