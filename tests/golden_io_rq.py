"""Reader for the second-generation golden vectors in tests/golden/rq/*.npz (written by make_golden_rq.py)."""
from __future__ import annotations

import glob
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rq")
WALLS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wo")   # walls_occlusion env
MOVE_REASONS = {1: "wall", 2: "occupied", 3: "corner_cut", 4: "los"}
POOLS = ("type_1_predator", "type_2_predator", "type_1_prey", "type_2_prey")


def agent_name(pool, i):
    return f"{POOLS[int(pool)]}_{int(i)}"


def case_names(walls=False):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(WALLS_DIR if walls else GOLDEN_DIR, "*.npz")))


def call_digest(grid, obs, rew, term, trunc, sort_scalars=False) -> bytes:
    """Same definition as tests/golden/make_golden_rq.py:call_digest."""
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(grid, dtype=np.float32).tobytes())
    for k, v in obs.items():
        h.update(k.encode() + np.ascontiguousarray(v, dtype=np.float32).tobytes())
    if sort_scalars:
        rew, term, trunc = ({k: d[k] for k in sorted(d)} for d in (rew, term, trunc))
    for k, v in rew.items():
        h.update(k.encode() + np.float64(v).tobytes())
    for k, v in term.items():
        h.update(k.encode() + bytes([int(bool(v))]))
    for k, v in trunc.items():
        h.update(k.encode() + bytes([int(bool(v))]))
    return h.digest()


class RQGoldenCase:
    def __init__(self, name):
        self.name = name
        self.walls = name.startswith("wo_")
        z = np.load(os.path.join(WALLS_DIR if self.walls else GOLDEN_DIR, name + ".npz"))
        self.z = {k: z[k] for k in z.files}
        self.wall_xy = self.z["wall_xy"] if self.walls else None
        self.config = json.loads(str(self.z["config_json"]))  # complete config (json's Infinity -> float inf)
        self.n_calls = len(self.z["term_all"])
        self.agents_after = json.loads(str(self.z["agents_after"]))
        self.reset_keys = json.loads(str(self.z["reset_keys"]))
        self.full_index = {int(c): k for k, c in enumerate(self.z["full_calls"])}

    @property
    def placement(self):
        return self.z["pred_xy"], self.z["prey_xy"], self.z["grass_xy"]

    def actions(self, t):
        lo, hi = self.z["act_off"][t], self.z["act_off"][t + 1]
        return {agent_name(self.z["act_pool"][k], self.z["act_id"][k]): int(self.z["act_val"][k]) for k in range(lo, hi)}

    def uniforms(self, t, extra=0):
        """The uniforms call t consumed (+ `extra` following values of the stream, to prove they are not touched)."""
        lo, hi = int(self.z["uni_off"][t]), int(self.z["uni_off"][t + 1])
        return self.z["uniforms"][lo:min(hi + extra, len(self.z["uniforms"]))], hi - lo

    def records(self, t):
        lo, hi = self.z["rec_off"][t], self.z["rec_off"][t + 1]
        return [(agent_name(self.z["rec_pool"][k], self.z["rec_id"][k]), float(self.z["rec_reward"][k]),
                 bool(self.z["rec_term"][k]), bool(self.z["rec_trunc"][k])) for k in range(lo, hi)]

    def flags(self, t):
        return bool(self.z["term_all"][t]), bool(self.z["trunc_all"][t])

    def infos(self, t):
        """infos dict of call t of the walls env (WO:766-778,393-395); {} for the red_queen env."""
        if not self.walls:
            return {}
        out = {}
        for k in range(self.z["info_off"][t], self.z["info_off"][t + 1]):
            code = int(self.z["info_reason"][k])
            d = {"los_rejected": int(code == 4)}
            if code:
                d["move_blocked_reason"] = MOVE_REASONS[code]
            out[agent_name(self.z["info_pool"][k], self.z["info_id"][k])] = d
        return out

    def extras(self, t):
        """agents that appear in the scalar dicts only (named in action_dict but gone): reward 0.0, False, False"""
        if not self.walls:
            return []
        return [agent_name(self.z["extra_pool"][k], self.z["extra_id"][k])
                for k in range(self.z["extra_off"][t], self.z["extra_off"][t + 1])]

    @property
    def channels(self):
        return 5 if (self.walls and self.config.get("include_visibility_channel")) else 4

    def digest(self, t) -> bytes:
        return self.z["digest"][t].tobytes()

    def obs_range(self, name):
        return self.config["predator_obs_range"] if "predator" in name else self.config["prey_obs_range"]

    def reset_obs(self):
        out, off = {}, 0
        C = self.channels
        for k in self.reset_keys:
            R = self.obs_range(k)
            out[k] = self.z["reset_obs_data"][off:off + C * R * R].reshape(C, R, R)
            off += C * R * R
        return out

    def full(self, t):
        """(obs dict, grid, state dict, grass energies, next_idx) for a call listed in full_calls, else None."""
        k = self.full_index.get(t)
        if k is None:
            return None
        off = int(self.z["obs_off"][k])
        obs = {}
        C = self.channels
        for name, _, _, _ in self.records(t):
            R = self.obs_range(name)
            obs[name] = self.z["obs_data"][off:off + C * R * R].reshape(C, R, R)
            off += C * R * R
        assert off == int(self.z["obs_off"][k + 1])
        lo, hi = self.z["st_off"][k], self.z["st_off"][k + 1]
        state = {
            agent_name(self.z["st_pool"][j], self.z["st_id"][j]): dict(
                pos=(int(self.z["st_x"][j]), int(self.z["st_y"][j])), energy=float(self.z["st_energy"][j]),
                cumulative_reward=float(self.z["st_cum"][j]), just_ate=bool(self.z["st_ate"][j]),
                age=int(self.z["st_age"][j]), last_reproduction=int(self.z["st_last_repro"][j]))
            for j in range(lo, hi)
        }
        return obs, self.z["grid_data"][k], state, self.z["grass_energy"][k], tuple(int(v) for v in self.z["next_idx"][k])
