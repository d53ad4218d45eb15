"""The walls_occlusion environment: the second-generation env plus static walls, line-of-sight rules and per-agent move
infos ("WO:n" = line n of predpreygrass/non_evolutionary/walls_occlusion/predpreygrass_rllib_env.py in the reference).

    from predpreygrass_amd.walls_occlusion import PredPreyGrass
    env = PredPreyGrass(config)          # same config keys: num_walls, wall_placement_mode, manual_wall_positions,
    obs, _ = env.reset(seed=3)           # include_visibility_channel, respect_los_for_movement,
                                         # mask_observation_with_visibility (WO:101-123)

The transition runs in the `ppg3_*` kernels of libppg_hip.so (predpreygrass_amd/csrc/ppg_kernel.h, WALLS); this module is
host plumbing on top of `red_queen.BatchedRedQueen(walls=True)`.
"""
from __future__ import annotations

import numpy as np

from . import _abi
from .red_queen import PredPreyGrass as _RedQueenPredPreyGrass


# walls_occlusion/config/config_env_zigzag_walls.py (values restated): the eval-style second-generation settings with two
# zigzag wall rows and every line-of-sight option switched on
config_env_zigzag_walls = {
    "max_steps": 1000, "grid_size": 25, "num_obs_channels": 4, "predator_obs_range": 7, "prey_obs_range": 9,
    "type_1_action_range": 3, "type_2_action_range": 0,
    "reward_predator_catch_prey": 0.0, "reward_prey_eat_grass": 0.0, "reward_predator_step": 0.0, "reward_prey_step": 0.0,
    "penalty_prey_caught": 0.0,
    "reproduction_reward_predator": {"type_1_predator": 10.0, "type_2_predator": 0.0},
    "reproduction_reward_prey": {"type_1_prey": 10.0, "type_2_prey": 0.0},
    "energy_loss_per_step_predator": 0.15, "energy_loss_per_step_prey": 0.05,
    "predator_creation_energy_threshold": 12.0, "prey_creation_energy_threshold": 8.0, "move_energy_cost_factor": 0.0,
    "initial_energy_predator": 5.0, "initial_energy_prey": 3.0,
    "n_possible_type_1_predators": 2000, "n_possible_type_2_predators": 0,
    "n_possible_type_1_prey": 2000, "n_possible_type_2_prey": 0,
    "n_initial_active_type_1_predator": 6, "n_initial_active_type_2_predator": 0,
    "n_initial_active_type_1_prey": 8, "n_initial_active_type_2_prey": 0,
    "mutation_rate_predator": 0.0, "mutation_rate_prey": 0.0,
    "initial_num_grass": 100, "initial_energy_grass": 2.0, "energy_gain_per_step_grass": 0.04,
    "mask_observation_with_visibility": True, "include_visibility_channel": True, "respect_los_for_movement": True,
    "max_energy_gain_per_grass": float("inf"), "max_energy_gain_per_prey": float("inf"),
    "max_energy_predator": float("inf"), "max_energy_prey": float("inf"), "max_energy_grass": 2.0,
    "reproduction_cooldown_steps": 0, "reproduction_chance_predator": 1.0, "reproduction_chance_prey": 1.0,
    "energy_transfer_efficiency": 1.0, "reproduction_energy_efficiency": 1.0,
    "wall_placement_mode": "manual",
    "manual_wall_positions": [(x, 6 + (x % 2)) for x in range(6, 18)] + [(x, 17 - (x % 2)) for x in range(6, 18)],
}


class PredPreyGrass(_RedQueenPredPreyGrass):
    """`PredPreyGrass(config)` of walls_occlusion/predpreygrass_rllib_env.py (class WO:41, reset WO:203, step WO:304).
    `reset(seed=s)` draws the walls and the initial cells from `default_rng(s)` exactly like the reference (WO:246,260),
    and that same generator then supplies the reproduction uniforms."""

    _walls = True
    _require_all_actions = False   # WO:406 tolerates live agents without an action

    def get_state_snapshot(self):
        snap = super().get_state_snapshot()
        snap["wall_positions"] = set(self.wall_positions)   # static within an episode; kept so a restore after another reset works
        snap["los_rejected_moves_total"] = self.los_rejected_moves_total
        snap["los_rejected_moves_by_type"] = dict(self.los_rejected_moves_by_type)
        return snap

    def restore_state_snapshot(self, snapshot):
        self.wall_positions = set(snapshot["wall_positions"])
        self.los_rejected_moves_total = snapshot["los_rejected_moves_total"]
        self.los_rejected_moves_by_type = dict(snapshot["los_rejected_moves_by_type"])
        self._last_action_names = []
        super().restore_state_snapshot(snapshot)

    def __init__(self, config=None, **kw):
        super().__init__(config, **kw)
        cfg = self._cfg
        self.num_walls = cfg.get("num_walls", 20)                                  # WO:118
        self.wall_placement_mode = cfg.get("wall_placement_mode", "random")        # WO:121
        self.manual_wall_positions = cfg.get("manual_wall_positions", None)        # WO:122
        self.include_visibility_channel = bool(cfg.get("include_visibility_channel", False))
        self.respect_los_for_movement = bool(cfg.get("respect_los_for_movement", False))
        self.mask_observation_with_visibility = bool(cfg.get("mask_observation_with_visibility", False))
        self.wall_positions = set()
        self.los_rejected_moves_total = 0
        self.los_rejected_moves_by_type = {"predator": 0, "prey": 0}

    def reset(self, *, seed=None, options=None):
        """WO:203-302.  ``options={"walls": [...], "placement": (pred_xy, prey_xy, grass_xy)}`` overrides the draw."""
        b = self._b
        G = b.grid_size
        self.rng = np.random.default_rng(seed)                                      # WO:135
        self.los_rejected_moves_total = 0
        self.los_rejected_moves_by_type = {"predator": 0, "prey": 0}
        opts = options if isinstance(options, dict) else {}
        max_cells = G * G
        if self.wall_placement_mode not in ("random", "manual"):
            raise ValueError("wall_placement_mode must be 'random' or 'manual'")  # WO:213-214
        walls = set()
        if "walls" in opts:
            walls = {(int(x), int(y)) for x, y in opts["walls"]}
        elif self.wall_placement_mode == "manual":                                  # WO:216-241
            for pos in (self.manual_wall_positions or []):
                try:
                    x, y = map(int, pos)
                except Exception:
                    continue
                if 0 <= x < G and 0 <= y < G:
                    walls.add((x, y))
        else:                                                                       # WO:242-250
            if self.num_walls >= max_cells:
                raise ValueError("num_walls must be less than total grid cells")
            if self.num_walls > 0:
                for idx in self.rng.choice(max_cells, size=self.num_walls, replace=False):
                    walls.add((int(idx) // G, int(idx) % G))
        self.wall_positions = walls
        placement = opts.get("placement")
        if placement is None:
            total = b.P0 + b.Q0 + b.n_grass
            free_cells = max_cells - self.num_walls                                # WO:254 (num_walls, also in manual mode)
            if total > free_cells:
                raise ValueError(f"Too many agents+grass ({total}) for free cells ({free_cells}) given {self.num_walls} "
                                 f"walls on {G}x{G} grid")
            free_indices = [i for i in range(max_cells) if (i // G, i % G) not in walls]
            cells = [(int(i) // G, int(i) % G) for i in self.rng.choice(free_indices, size=total, replace=False)] \
                if total > 0 else []                                               # WO:258-262
            placement = (cells[:b.P0], cells[b.P0:b.P0 + b.Q0], cells[b.P0 + b.Q0:])
        b.set_walls(sorted(walls))
        p, q, g = placement
        b.set_placement(np.asarray(p).reshape(1, -1, 2), np.asarray(q).reshape(1, -1, 2), np.asarray(g).reshape(1, -1, 2))
        self.cumulative_rewards = {}
        self._insertion_order = []
        self._last_action_names = []
        obs = self._collect(after_reset=True)[0]
        if self._an is not None:
            self._an.reset(self.possible_agents, self.agents, self.agent_energies)   # WO:146-165
        return obs, {}

    def _finish_outputs(self, recs, tables, rew, term, trunc, truncated_call):
        """WO:370-395: the scalar dicts also name every agent of the action dict (defaults 0.0 / False); infos carry
        `los_rejected` and `move_blocked_reason` for the agents that went through the movement phase.  (The reference
        builds these dicts from a Python set, so their order is arbitrary there; here: observation order, then the rest.)"""
        if truncated_call:
            return {}
        for a in self._last_action_names:
            if a not in rew:
                rew[a], term[a], trunc[a] = 0.0, False, False
        infos = {}
        cp = self._b.pred_capacity
        for name, sp, row, *_ in recs:
            code = int(tables["row_info"][0][cp * sp + row])
            if code:
                d = {"los_rejected": int(code - 1 == 4)}
                if code > 1:
                    d["move_blocked_reason"] = _abi.MOVE_REASONS[code - 1]
                if code - 1 == 4:                                                   # WO:770-772
                    self.los_rejected_moves_total += 1
                    self.los_rejected_moves_by_type["prey" if sp else "predator"] += 1
                infos[name] = d
        return infos


def env_creator(config):
    return PredPreyGrass(config)
