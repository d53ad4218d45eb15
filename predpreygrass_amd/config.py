"""The environment configuration: the keys of the reference's config dict with BOTH sets of defaults the reference has --
the shipped dict (predpreygrass/non_evolutionary/base_environment/config_env.py:1-38) and the in-code fallbacks its
__init__ applies to a partial dict (predpreygrass_rllib_env.py:20-61), which differ in six places."""
from __future__ import annotations

# (key, value in the shipped config_env, `config.get(key, <this>)` fallback in the env's __init__)
_KEYS = (
    # episode and geometry
    ("max_steps",                                1000,  10000),
    ("grid_size",                                  25,     10),
    ("num_obs_channels",                            4,      4),
    ("predator_obs_range",                          7,      7),
    ("prey_obs_range",                              9,      5),
    # rewards
    ("reward_predator_catch_prey",                0.0,    0.0),
    ("reward_prey_eat_grass",                     0.0,    0.0),
    ("reward_predator_step",                      0.0,    0.0),
    ("reward_prey_step",                          0.0,    0.0),
    ("penalty_prey_caught",                       0.0,    0.0),
    ("reproduction_reward_predator",             10.0,   10.0),
    ("reproduction_reward_prey",                 10.0,   10.0),
    # energy
    ("energy_loss_per_step_predator",            0.15,   0.15),
    ("energy_loss_per_step_prey",                0.05,   0.05),
    ("predator_creation_energy_threshold",       12.0,   12.0),
    ("prey_creation_energy_threshold",            8.0,    8.0),
    ("initial_energy_predator",                   5.0,    5.0),
    ("initial_energy_prey",                       3.0,    3.0),
    ("initial_energy_grass",                      2.0,    2.0),
    ("energy_gain_per_step_grass",               0.04,    0.2),
    # populations
    ("n_possible_predators",                     2000,     50),
    ("n_possible_prey",                          2000,     50),
    ("n_initial_active_predator",                   6,      6),
    ("n_initial_active_prey",                       8,      8),
    ("initial_num_grass",                         100,     25),
    # console output (accepted, unused by the kernels)
    ("verbose_engagement",                      False,  False),
    ("verbose_movement",                        False,  False),
    ("verbose_spawning",                        False,  False),
)

# the dict order of the reference file (reset() and the kernels do not depend on it; printing a config does)
_FILE_ORDER = (
    "max_steps", "grid_size", "num_obs_channels", "predator_obs_range", "prey_obs_range",
    "reward_predator_catch_prey", "reward_prey_eat_grass", "reward_predator_step", "reward_prey_step",
    "penalty_prey_caught", "reproduction_reward_predator", "reproduction_reward_prey",
    "energy_loss_per_step_predator", "energy_loss_per_step_prey", "predator_creation_energy_threshold",
    "prey_creation_energy_threshold", "n_possible_predators", "n_possible_prey", "n_initial_active_predator",
    "n_initial_active_prey", "initial_energy_predator", "initial_energy_prey", "initial_num_grass",
    "initial_energy_grass", "energy_gain_per_step_grass", "verbose_engagement", "verbose_movement",
    "verbose_spawning",
)
_SHIPPED = {k: shipped for k, shipped, _ in _KEYS}
config_env = {k: _SHIPPED[k] for k in _FILE_ORDER}
_IN_CODE_DEFAULTS = {k: fallback for k, _, fallback in _KEYS}


def resolve_config(config: dict | None) -> dict:
    """`config = config or config_env` (predpreygrass_rllib_env.py:20), then per-key
    `config.get(key, in_code_default)` (predpreygrass_rllib_env.py:22-61)."""
    base = config or config_env  # a falsy dict ({}) selects the defaults, as in the reference
    out = {k: _IN_CODE_DEFAULTS[k] for k in _FILE_ORDER}
    out.update({k: v for k, v in base.items()})
    if int(out["num_obs_channels"]) != 4:
        raise ValueError("num_obs_channels must be 4 (border, predator, prey, grass)")
    return out
